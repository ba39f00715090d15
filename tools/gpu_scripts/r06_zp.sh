#!/bin/bash
# Round 6, lease zp: the class (16,8) with EIGHT lanes per filter (imm_oct_*: imm_quad.hip built with FK_IQ_LPF=8) against four: IMM suite, outputs on / off
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zp
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_imm.py -m gpu -q -p no:cacheprovider > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -6 $O/pytest_imm.log
for v in 1 0; do for d in 16x8x2 16x8x8 14x6x4; do echo "== FK_IMM_OCT=$v" | tee -a $O/imm_outputs.jsonl; FK_IMM_OCT=$v timeout 300 python tools/bench_imm_outputs.py --dims $d --layout soa 2>&1 | tee -a $O/imm_outputs.jsonl; done; done
