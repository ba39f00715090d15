#!/bin/bash
# Round 5: kf_fast's extras (Saver histories) calls without a mask on the masked twin of the instantiation (FK_FAST_EX_MASKED=1):
# parity of the histories, then the rows of tools/bench_configs.py --configs e both ways.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
FK_FAST_EX_MASKED=1 timeout 600 python -m pytest tests/test_gpu_kf.py tests/test_gpu_zz_saver.py -m gpu -q -p no:cacheprovider -k "saver or Saver or histories" 2>&1 | tail -2 | cut -c1-160
cd /tmp
for kn in 0 1 0 1; do
    FK_FAST_EX_MASKED=$kn timeout 300 python $R/tools/bench_configs.py --configs e --layouts soa,aos 2>/dev/null | grep "^{" | grep "kf_fast extras" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['ex_masked']=$kn; print(json.dumps(d))
" | tee -a $O/extras_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('masked-twin' if d['ex_masked'] else 'shipped    ', d['kernel'][:78], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
done
