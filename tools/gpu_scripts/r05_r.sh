#!/bin/bash
# Round 5: after the per-shape fixes of r05_p / r05_q -- the whole GPU suite, every one-lane / (9,x) shape both layouts again
# (forward kernels only), the Saver histories at (9,x) on the four-lane EX kernels, and (5,4) NumPy order at one wave per SIMD
# (exp_build/libfilterhip_E.so) against the shipped two, A/B x 4.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
cd /tmp
show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['lib']='$1'; print(json.dumps(d))
" | tee -a $O/$2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['kernel'][:76], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"; }
export KF_NO_RTS=1
KF_DIMS=1x1,2x1,2x2,3x1,3x2,3x3,4x1,4x2,4x3,4x4,5x1,5x2,5x3,5x4,6x1,6x2,6x3,6x4,7x1,7x2,7x3,7x4,8x1,8x2,8x3,8x4,9x1,9x2,9x3,9x4 timeout 600 python $R/tools/bench_configs.py --configs a --layouts soa,aos 2>/dev/null | show shipped kf_dims_all.jsonl
EXTRAS_DIMS=9x1,9x2,9x4 timeout 300 python $R/tools/bench_configs.py --configs e --layouts soa,aos 2>/dev/null | grep -v generic | show shipped extras_9.jsonl
for lib in shipped E shipped E shipped E shipped E; do
    L=""; [ $lib != shipped ] && L=$R/exp_build/libfilterhip_$lib.so
    FK_LIB=$L KF_DIMS=5x4 timeout 300 python $R/tools/bench_configs.py --configs a --layouts aos 2>/dev/null | show $lib waves_ab_5x4.jsonl
done
