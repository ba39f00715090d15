#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
for i in 1 2; do timeout 300 python tools/exp_buffers.py > $O/buffers_$i.jsonl 2> $O/buffers_$i.err; tail -2 $O/buffers_$i.err | cut -c1-300; cut -c1-200 $O/buffers_$i.jsonl; done
