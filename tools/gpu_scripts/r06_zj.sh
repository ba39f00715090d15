#!/bin/bash
# Round 6, lease zj: the four-lanes-per-filter IMM kernel, tail filling by chunked calls (FK_IMM_CHUNKS) at 6.1 rounds of workgroups
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zj
mkdir -p $O
export TMPDIR=/tmp
cd $R
for c in "" "1,1" "2,2" "2,4" "3,3" "4,4" "4,8" "7,7"; do
FK_IMM_CHUNKS=$c timeout 600 python tools/bench_configs.py --configs r --layouts soa 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if '16,8' in d['kernel'] or 'x16' in d['kernel'] or 'x8' in d['kernel']: print('chunks=$c', d['kernel'], round(d['ms'], 3))
" | tee -a $O/quad_chunks.txt
done
