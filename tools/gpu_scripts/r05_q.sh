#!/bin/bash
# Round 5: the outliers of r05_p.sh after their fixes -- (3,2) / (3,3) element-major at 4 / 2 waves, (6,4) NumPy order at one wave,
# (9,1) (9,2) (9,4) on four lanes per track -- the tests of the kernels touched, then two more candidates on experimental links
# (exp_build/libfilterhip_C.so: (2,2) at 6 waves, (5,4) NumPy order at one wave; ..._D.so: (2,2) at 4 waves), A/B/A.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_zz_saver.py tests/test_gpu_tails.py tests/test_gpu_api.py -m gpu -q -x -p no:cacheprovider --durations=5 2>&1 | tail -3 | cut -c1-200
cd /tmp
show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['lib']='$1'; print(json.dumps(d))
" | tee -a $O/$2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['kernel'][:60], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"; }
export KF_NO_RTS=1
KF_DIMS=3x2,3x3,6x4,9x1,9x2,9x4 timeout 300 python $R/tools/bench_configs.py --configs a --layouts soa,aos 2>/dev/null | show shipped kf_dims_fixed.jsonl
FK_ML9=m KF_DIMS=9x1,9x2,9x4 timeout 300 python $R/tools/bench_configs.py --configs a --layouts soa,aos 2>/dev/null | show one-lane kf_dims_fixed.jsonl
EXTRAS_DIMS=9x1,9x2,9x4 timeout 300 python $R/tools/bench_configs.py --configs e --layouts soa,aos 2>/dev/null | show shipped extras_9.jsonl
for lib in shipped C D shipped C D; do
    L=""; [ $lib != shipped ] && L=$R/exp_build/libfilterhip_$lib.so
    FK_LIB=$L KF_DIMS=2x2,5x4 timeout 300 python $R/tools/bench_configs.py --configs a --layouts soa,aos 2>/dev/null | show $lib waves_ab_2x2_5x4.jsonl
done
