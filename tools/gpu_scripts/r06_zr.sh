#!/bin/bash
# Round 6, lease zr: the eight-lane fused UKF / smoother with the cross-quad broadcast on the VALU (DPP) instead of ds_swizzle: UKF suites, the u rows
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zr
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_ukf_mlg.py tests/test_gpu_ukf_dims.py tests/test_gpu_ukf.py -m gpu -q -p no:cacheprovider > $O/pytest_ukf.log 2>&1; echo "pytest ukf rc=$?"; tail -4 $O/pytest_ukf.log
timeout 900 python tools/bench_configs.py --configs u --layouts soa,aos 2>/dev/null | grep "^{" > $O/u_rows.jsonl
python -c "
import json
for l in open('$O/u_rows.jsonl'):
    d = json.loads(l); print(d['kernel'], round(d['ms'], 3))
"
