#!/bin/bash
# Round 6, lease zi: the classes (12,4) and (16,8) on four lanes per filter (imm_quad.hip): IMM suite + the IMM rows of bench_configs, A/B against the rolled general kernel
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zi
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_imm.py -m gpu -q -p no:cacheprovider > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -25 $O/pytest_imm.log
for v in 1 0; do
FK_IMM_QUAD=$v timeout 600 python tools/bench_configs.py --configs r --layouts soa,aos 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if '16,8' in d['kernel']: print('quad=$v', d['kernel'], round(d['ms'], 3))
"
done
