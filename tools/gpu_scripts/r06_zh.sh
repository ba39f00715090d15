#!/bin/bash
# Round 6, lease zh: the single-phase calls (IMMEstimator.predict / .update on their own) on the one-lane-per-filter kernel: IMM suite + tails
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zh
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_imm.py tests/test_gpu_tails.py -m gpu -q -p no:cacheprovider > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -25 $O/pytest_imm.log
