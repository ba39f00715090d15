// ukf_rts_big.hip -- fk_ukf_rts_correct_f64 at dim_x 10 .. 16: ukf_rts_kernel (kf_variants.hip; UKF.py:726-733) on the padded
// classes 12 and 16, built with -DFK_ROLLED=1 (rolled loops, scratch-resident n x n arrays).
#define FK_VARIANTS_BIG 1
#include "kf_variants.hip"
