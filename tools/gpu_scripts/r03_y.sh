#!/bin/bash
# Round 3, lease y: the fused UKF kernels restructured around ukf_linear_step_v3 (exact instantiations, z requested at the head
# of the update half, mask byte landed in front of the stores, cooperative NumPy-order loads / stores in the smoother):
# parity of every UKF test on the exact and (FK_UKF_PADDED=1) the padded instantiations, then timing.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_scripts/r03_y.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03y
mkdir -p $O
cd $R
T="tests/test_gpu_ukf.py tests/test_gpu_ukf_dims.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf_hooks.py tests/test_gpu_tails.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py"
timeout 500 python -m pytest $T -m gpu -q -p no:cacheprovider -k "ukf or UKF or c4 or C4 or unscented" > $O/pytest_ukf.log 2>&1; echo "pytest ukf rc=$?"; tail -4 $O/pytest_ukf.log
FK_UKF_PADDED=1 timeout 500 python -m pytest $T -m gpu -q -p no:cacheprovider -k "ukf or UKF or c4 or C4 or unscented" > $O/pytest_ukf_padded.log 2>&1; echo "pytest ukf padded rc=$?"; tail -4 $O/pytest_ukf_padded.log
B="timeout 200 python tools/bench_ukf.py"
$B --dims 6x3,4x2,2x2,8x4,9x3,9x4 > $O/ukf_exact.jsonl 2> $O/ukf_exact.err; echo "exact rc=$?"
FK_UKF_PADDED=1 $B --dims 6x3,4x2 > $O/ukf_padded.jsonl 2> $O/ukf_padded.err; echo "padded rc=$?"
$B --dims 6x3 --N 1000000 --T 20 > $O/ukf_exact_1e6.jsonl 2> $O/ukf_exact_1e6.err
$B --dims 5x2,3x1 > $O/ukf_padded_dims.jsonl 2> $O/ukf_padded_dims.err
cat $O/ukf_*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('%-44s N=%-8d %7.3f ms  frac %.3f  par %.1e  %s%s' % (r['kernel'], r['N'], r['ms'], r['frac_of_8TBs'], r['parity_max_rel'], r['switches'], ' dense' if r['dense_model'] else ''))
"
tail -3 $O/*.err | cut -c1-300
