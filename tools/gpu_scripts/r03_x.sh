#!/bin/bash
# Round 3, lease x: the factor-image UKF step (ukf_linear_step_v3 / ukf_linear_rts_gain_v3) -- GPU parity of every UKF test,
# then A/B timing against round 2's step (FK_UKF_V2=1), the scalar-operand model (FK_UKF_SCALAR=1) and the straightforward
# kernel (FK_UKF_V1=1), one process per switch set (the launchers read them once).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_scripts/r03_x.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03x
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_ukf.py tests/test_gpu_ukf_dims.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf_hooks.py tests/test_gpu_tails.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -k "ukf or UKF or c4 or C4 or unscented" > $O/pytest_ukf.log 2>&1; echo "pytest ukf rc=$?"; tail -4 $O/pytest_ukf.log
B="timeout 200 python tools/bench_ukf.py"
$B --dims 6x3,4x2,2x2,8x4,9x3,9x4 > $O/ukf_v3.jsonl 2> $O/ukf_v3.err; echo "v3 rc=$?"
FK_UKF_V2=1 $B --dims 6x3 > $O/ukf_v2.jsonl 2> $O/ukf_v2.err; echo "v2 rc=$?"
FK_UKF_V1=1 $B --dims 4x2,2x2 > $O/ukf_v1.jsonl 2> $O/ukf_v1.err; echo "v1 rc=$?"
FK_UKF_SCALAR=1 $B --dims 6x3,8x4 > $O/ukf_scalar.jsonl 2> $O/ukf_scalar.err; echo "scalar rc=$?"
$B --dims 6x3 --N 1000000 --T 20 > $O/ukf_v3_1e6.jsonl 2> $O/ukf_v3_1e6.err
$B --dims 6x3 --dense > $O/ukf_v3_dense.jsonl 2> $O/ukf_v3_dense.err
cat $O/ukf_*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('%-44s N=%-8d %7.3f ms  frac %.3f  par %.1e  %s%s' % (r['kernel'], r['N'], r['ms'], r['frac_of_8TBs'], r['parity_max_rel'], r['switches'], ' dense' if r['dense_model'] else ''))
"
tail -3 $O/*.err | cut -c1-300
