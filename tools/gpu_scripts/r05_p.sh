#!/bin/bash
# Round 5: kf_fast instantiations never measured (the element-major (3,2) / (3,3) ones carry 204 / 536 B of scratch under their
# six-waves launch bound): every one-lane shape both layouts, then (3,x) on two experimental links of the library
# (exp_build/libfilterhip_dim3A.so: (3,2) at 4 waves, (3,3) at 3; ...B.so: 3 and 2), A/B/A.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
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
KF_DIMS=2x2,3x1,3x2,3x3,4x1,4x3,4x4,5x1,5x3,5x4,6x1,6x4,7x1,7x2,7x3,8x1,8x2,8x3,9x1,9x2,9x4 timeout 600 python $R/tools/bench_configs.py --configs a --layouts soa,aos 2>/dev/null | show shipped kf_dims_sweep.jsonl
for lib in shipped A B shipped A B; do
    L=""; [ $lib != shipped ] && L=$R/exp_build/libfilterhip_dim3$lib.so
    FK_LIB=$L KF_DIMS=3x1,3x2,3x3 timeout 300 python $R/tools/bench_configs.py --configs a --layouts soa 2>/dev/null | show $lib dim3_waves_ab.jsonl
done
