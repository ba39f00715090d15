#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03v
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kf.py tests/test_gpu_api.py tests/test_gpu_zz_saver.py -m gpu -q -p no:cacheprovider -k "saver or Saver or extras or lean_fast or batch_filter_goldens or tuning or tail_shapes" > $O/pytest_saver.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_saver.log | cut -c1-200 | tail -30
NX=7 NZ=1 timeout 200 python tools/debug/ex_aos_mask.py 2>&1 | cut -c1-250 | head -12
