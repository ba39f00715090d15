#!/bin/bash
# Round 6, lease zk: after the one-lane-per-bank classes (9,4) / (16,8) left the build: the IMM, tails, API and variants suites, the IMM rows of bench_configs
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zk
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_imm.py tests/test_gpu_tails.py tests/test_gpu_api.py tests/test_gpu_zz_saver.py -m gpu -q -p no:cacheprovider > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -25 $O/pytest_imm.log
timeout 600 python tools/bench_configs.py --configs 8r --layouts soa,aos 2>/dev/null | grep "^{" > $O/imm_rows.jsonl
python -c "
import json
for l in open('$O/imm_rows.jsonl'):
    d = json.loads(l); print(d['kernel'], d.get('layout'), round(d['ms'], 3), round(d.get('frac', 0), 3))
"
