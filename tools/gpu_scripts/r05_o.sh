#!/bin/bash
# Round 5: which instantiations of kf_fast's extras kernel serve a call without a mask better from the masked twin
# (FK_FAST_EX_MASKED=0 / 1 force either; unset = the shipped per-instantiation rule): Saver parity under the shipped rule, then A/B.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kf.py tests/test_gpu_zz_saver.py -m gpu -q -p no:cacheprovider -k "saver or Saver or histories" 2>&1 | tail -2 | cut -c1-160
cd /tmp
for kn in 0 1 0 1; do
    EXTRAS_DIMS=4x3,4x4,5x2,5x3,5x4,6x2,6x4,7x3,8x4,9x4 FK_FAST_EX_MASKED=$kn timeout 400 python $R/tools/bench_configs.py --configs e --layouts soa,aos 2>/dev/null | grep "^{" | grep "kf_fast extras" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['ex_masked']=$kn; print(json.dumps(d))
" | tee -a $O/extras_dims_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('masked-twin' if d['ex_masked'] else 'plain      ', d['kernel'][:70], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
done
timeout 300 python $R/tools/bench_configs.py --configs e --layouts soa,aos 2>/dev/null | grep "^{" | tee $O/extras_shipped.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('shipped', d['kernel'][:78], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
