#!/bin/bash
# Round 3, lease z: LDS-DMA fetch of the next filtered state in the fused UKF smoother (exact classes 2, 4, 6) -- parity of
# every UKF test with it, then A/B against FK_UKF_DMA=0 in one lease.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_scripts/r03_z.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
T="tests/test_gpu_ukf.py tests/test_gpu_ukf_dims.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf_hooks.py tests/test_gpu_tails.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py"
timeout 500 python -m pytest $T -m gpu -q -p no:cacheprovider -k "ukf or UKF or c4 or C4 or unscented" > $O/pytest_ukf.log 2>&1; echo "pytest ukf rc=$?"; tail -4 $O/pytest_ukf.log
B="timeout 200 python tools/bench_ukf.py"
$B --dims 6x3,4x2,2x2 > $O/ukf_dma.jsonl 2> $O/ukf_dma.err; echo "dma rc=$?"
FK_UKF_DMA=0 $B --dims 6x3,4x2,2x2 > $O/ukf_nodma.jsonl 2> $O/ukf_nodma.err; echo "nodma rc=$?"
$B --dims 6x3 --N 1000000 --T 20 > $O/ukf_dma_1e6.jsonl 2> $O/ukf_dma_1e6.err
$B --dims 6x3,4x2 --N 99991 --T 7 > $O/ukf_dma_ragged.jsonl 2> $O/ukf_dma_ragged.err
cat $O/ukf_*.jsonl | grep smoother | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('%-44s N=%-8d T=%-4d %7.3f ms  frac %.3f  par %.1e  %s' % (r['kernel'], r['N'], r['T'], r['ms'], r['frac_of_8TBs'], r['parity_max_rel'], r['switches']))
"
tail -3 $O/*.err | cut -c1-300
