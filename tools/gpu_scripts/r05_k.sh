#!/bin/bash
# Round 5 experiment: the IMM bank classes that keep the bank in scratch with rolled loops -- (9,4) x 5, x 8 and (16,8) x 2 -- built
# UNROLLED like (9,4) x 2..4 (static scratch offsets instead of dynamically indexed arrays): parity, then the rows of
# tools/bench_configs.py --configs r.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_imm.py -m gpu -q -p no:cacheprovider > $O/tests_imm.log 2>&1
tail -3 $O/tests_imm.log | cut -c1-200
cd /tmp
timeout 900 python $R/tools/bench_configs.py --configs r --layouts soa,aos 2>$O/err.txt | grep "^{" | tee $O/imm_rows.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:70], 'ms=%.2f'%d['ms'], 'frac=%.4f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
