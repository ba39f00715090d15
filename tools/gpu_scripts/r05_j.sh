#!/bin/bash
# Round 5 experiment: the scheduler strategy (-mllvm -amdgpu-sched-strategy=max-ilp) on the kernels that are bound by dependent
# fp64 chains at one or two waves per SIMD -- the fused UKF (ukf_kernels parts 5-8) and the three-lane (9,3) kernels (kf_ml) --
# A/B in one lease: the shipped library, then csrc/exp_build/libfilterhip_ilp.so in its place (same sources, those six objects
# rebuilt with the flag), timings and the parity tests of the swapped kernels.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j
mkdir -p $O
cd /tmp
run() {
    tag=$1
    timeout 300 python $R/tools/bench_ukf.py --dims 6x3,4x2,8x4,9x3 --N 100000 --T 100 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['lib']='$tag'; print(json.dumps(d))
" | tee -a $O/ukf_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['lib'], d['kernel'][:44], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
    timeout 300 python $R/tools/bench_configs.py --configs 3 --layouts soa,aos 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['lib']='$tag'; print(json.dumps(d))
" | tee -a $O/c3_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['lib'], d['kernel'][:44], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
}
run shipped
cp $R/filterpy_amd/libfilterhip.so /tmp/libfilterhip_shipped.so
cp $R/filterpy_amd/csrc/exp_build/libfilterhip_ilp.so $R/filterpy_amd/libfilterhip.so
run max-ilp
cd $R
timeout 900 python -m pytest tests/test_gpu_ukf.py tests/test_gpu_kf.py tests/test_gpu_baseline_configs.py -m gpu -q -p no:cacheprovider -k "ukf or multilane or three_lane or slab or c3 or C3 or config" > $O/tests_ilp.log 2>&1
tail -3 $O/tests_ilp.log | cut -c1-200
cp /tmp/libfilterhip_shipped.so $R/filterpy_amd/libfilterhip.so
cd /tmp
run shipped-again
