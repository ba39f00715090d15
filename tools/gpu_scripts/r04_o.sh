#!/bin/bash
# Round 4, lease o: the round model of the three-lane kernel -- whole rounds of waves (N = 98304 = 3 x 2048 waves x 16 tracks,
# 131072 = 4 rounds) against configs[2]'s 1e5 (3.05 rounds), as ONE launch and with the default tail filling.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04o
mkdir -p $O
cd /tmp
for n in 98304 100000 131072 65536 400000; do
  for ch in "1,1" ""; do
    FK_ML_CHUNKS=$ch timeout 200 python $R/tools/bench_configs.py --configs 3 --layouts soa,aos --N $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['chunks']='$ch' or 'default'; d['N']=$n; print(json.dumps(d))
" | tee -a $O/rounds.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['N'], d['chunks'], d['kernel'][:40], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'])
"
  done
done
