#!/bin/bash
# Round 5, the last GPU-minutes: the steady-state filter above (9,4) unrolled (kf_variants.hip) against the rolled unit it ran in
# (FK_STEADY_ROLLED=1), the padded NumPy-order kernel at one / three waves per SIMD (FK_STEADY_AOS_WAVES).
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05v
mkdir -p $O
cd $R
timeout 60 python -m pytest tests/test_gpu_variants.py tests/test_gpu_tails.py tests/test_gpu_zz_module_steadystate.py -m gpu -q -p no:cacheprovider -k "steady" 2>&1 | tail -2 | cut -c1-200 | tee $O/pytest_steady.txt
FK_STEADY_AOS_WAVES=3 timeout 40 python -m pytest tests/test_gpu_variants.py -m gpu -q -p no:cacheprovider -k "above_9_4 and aos" 2>&1 | tail -1 | cut -c1-200 | tee -a $O/pytest_steady.txt
cd /tmp
for mode in unrolled rolled aos3; do
    E=""; [ $mode = rolled ] && E="FK_STEADY_ROLLED=1"; [ $mode = aos3 ] && E="FK_STEADY_AOS_WAVES=3"
    L="soa,aos"; [ $mode = aos3 ] && L="aos"
    env $E STEADY_DIMS=16x8,14x6,12x8 timeout 40 python $R/tools/bench_configs.py --configs S --layouts $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['mode']='$mode'; print(json.dumps(d))
" | tee -a $O/steady_big.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$mode', d['kernel'][:60], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
done
