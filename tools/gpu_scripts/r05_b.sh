#!/bin/bash
# Round 5, a cheap A/B the lint suggested (profiles/r04/isa_lint_kf_lanes.txt): the Kalman smoother at dim_x 13 / 14 element-major
# still runs the four-lane kernel (62 / 73 KB of code) where the eight-lane one is 29-33 KB; FK_RTS_LANES forces either.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
cd /tmp
for ln in 4 8; do
    FK_RTS_LANES=$ln timeout 300 python $R/tools/bench_configs.py --configs b --layouts soa,aos 2>>$O/err.txt | grep -i "smoother" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['rts_lanes']=$ln; print(json.dumps(d))
" | tee -a $O/rts_lanes_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('lanes', d['rts_lanes'], d['kernel'][:44], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
done
