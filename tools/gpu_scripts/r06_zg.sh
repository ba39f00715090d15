#!/bin/bash
# Round 6, lease zg: the extended one-lane-per-filter IMM kernel (MMAE, missing measurements, control input): the IMM suite
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zg
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_imm.py -m gpu -q -p no:cacheprovider > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -25 $O/pytest_imm.log
timeout 600 python tools/bench_configs.py --configs r --layouts soa 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['kernel'], round(d['ms'], 3))
"
