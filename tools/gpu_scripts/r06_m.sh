#!/bin/bash
# Round 6, lease m: per-chunk timelines of calls with few long vectors (tools/op_timeline.py)
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06m
mkdir -p $O
export TMPDIR=/tmp
cd $R
for lb in 3 0; do
FK_OP_LB=$lb timeout 300 python tools/op_timeline.py --run --shapes 1x2000000,1x8000000,8x8000000 --out $O/lb$lb > $O/timeline_lb$lb.txt 2> $O/timeline_lb$lb.err
done
cut -c1-200 $O/timeline_lb3.txt | head -150
