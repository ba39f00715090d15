#!/bin/bash
# Round 4, lease w: tail filling of the three-lane kernel at configs[2] -- track groups x time chunks on helper streams, which
# decomposition (FK_ML_CHUNKS=G,H) returns how much of the ~10 % the partial fourth round of waves costs.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04w
mkdir -p $O
cd /tmp
for ch in default 1,1 2,4 2,8 3,4 3,8 4,4 4,8 4,16 3,16 4,25; do
  if [ "$ch" = default ]; then unset FK_ML_CHUNKS; else export FK_ML_CHUNKS=$ch; fi
  timeout 200 python $R/tools/bench_configs.py --configs 3 --layouts soa,aos 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['chunks']='$ch'; print(json.dumps(d))
" | tee -a $O/chunks.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['chunks'], d['kernel'][:40], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'])
"
done
