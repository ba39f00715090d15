"""filterpy_amd -- an MI355X (gfx950) native batched-filter engine behind filterpy's API.

Only the hot path BASELINE.json names is implemented: KalmanFilter
predict/update/batch_filter/rts_smoother, the UKF sigma-point / unscented transform
arithmetic and the monte_carlo resamplers -- for one filter (drop-in) or for a bank of N
independent filters stepped in lock-step on the GPU.  All arithmetic runs in
libfilterhip.so (hand-written HIP, see filterpy_amd/csrc and include/filterhip.h).
"""
__version__ = "0.1.0"
