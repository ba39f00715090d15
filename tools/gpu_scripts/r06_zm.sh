#!/bin/bash
# Round 6, lease zm: the IMM kernels with their per-step outputs switched off one by one
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zm
mkdir -p $O
export TMPDIR=/tmp
cd $R
for d in 16x8x2 12x4x2 9x4x8; do for l in soa aos; do timeout 300 python tools/bench_imm_outputs.py --dims $d --layout $l 2>&1 | tee -a $O/imm_outputs.jsonl; done; done
