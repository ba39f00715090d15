#!/bin/bash
# lease: fused linear UKF at dim_x 7..9, fk_ut_linear_map_f64, chunked calls of the fused kernel (bit-identity + timing)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_ukf_dims.py tests/test_gpu_ukf.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf_hooks.py tests/test_gpu_tails.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py -m gpu -q -x -p no:cacheprovider > $O/pytest_ukf.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_ukf.log
for ch in none 2,4 3,4 3,8 4,10; do
  if [ $ch = none ]; then unset FK_UKF_CHUNKS; else export FK_UKF_CHUNKS=$ch; fi
  timeout 300 python tools/bench_configs.py --configs 4 2>/dev/null | grep "fused" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); r['FK_UKF_CHUNKS'] = '$ch'; print(json.dumps(r))" >> $O/ukf_chunking.jsonl
done
unset FK_UKF_CHUNKS
cat $O/ukf_chunking.jsonl | cut -c1-400
