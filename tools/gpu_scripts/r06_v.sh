#!/bin/bash
# Round 6, lease v: workgroups per CU for calls with few long vectors (more resident chunks = fewer rounds), now that the chain is short
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06v
mkdir -p $O
export TMPDIR=/tmp
cd $R
SH=1x8000000,2x4000000,8x8000000,1x2000000,32x1000000,4x8000000
for env in "FK_OP_WAVES=5" "FK_OP_WAVES=6" "FK_OP_WAVES=7" "FK_OP_WAVES=5" "FK_OP_WAVES=7"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-110
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
