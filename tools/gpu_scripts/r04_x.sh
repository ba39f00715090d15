#!/bin/bash
# Round 4, lease x: the persistent grid of the three-lane kernel (tickets per (time chunk, group of 64 tracks)) against the
# single launch and the multi-stream tail filling at configs[2]; IMM (9,4) x 4 unrolled.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kf.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "persistent or c3 or chunked or multilane" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
cd /tmp
run() { timeout 200 python $R/tools/bench_configs.py --configs 3 --layouts soa,aos 2>/dev/null | grep "batch_filter" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['mode']='$1'; print(json.dumps(d))
" | tee -a $O/persist.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['mode'], d['kernel'][:40], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"; }
run "persist(default H=3)"
FK_ML_PERSIST_H=4 run "persist H=4"
FK_ML_PERSIST_H=2 run "persist H=2"
FK_ML_PERSIST_H=6 run "persist H=6"

FK_ML_PERSIST=0 run "multi-stream 3x4"
FK_ML_PERSIST=0 FK_ML_CHUNKS=1,1 run "one launch"
