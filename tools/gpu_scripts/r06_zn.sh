#!/bin/bash
# Round 6, lease zn: imm_quad with the padded LDS model blocks: IMM suite, outputs on / off
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zn
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_imm.py -m gpu -q -x -p no:cacheprovider > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -4 $O/pytest_imm.log
for d in 16x8x2 12x4x2 16x8x8; do timeout 300 python tools/bench_imm_outputs.py --dims $d --layout soa 2>&1 | tee -a $O/imm_outputs.jsonl; done
