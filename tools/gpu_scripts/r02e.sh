#!/bin/bash
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -x -p no:cacheprovider > $O/pytest_rs.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_rs.log | cut -c1-300
timeout 200 python tools/bench_resample.py --shapes 1000x8000,125x8000,1000x2000,4000x2048,100x30000 --iters 20 2>&1 | tee $O/resample_short.jsonl
FK_RESAMPLE_SERIAL=1 timeout 200 python tools/bench_resample.py --shapes 1000x8000,125x8000 --iters 20 2>&1 | sed 's/^/serial /' | tee -a $O/resample_short.jsonl
