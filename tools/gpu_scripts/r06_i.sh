#!/bin/bash
# Round 6, ninth lease: the fused UKF's persistent grid (tickets over track groups x time chunks): bit-identity, then C4 timing.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_ukf.py -m gpu -q -p no:cacheprovider -x > $O/pytest_ukf.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_ukf.log
for env in "FK_UKF_PERSIST=1" "FK_UKF_PERSIST=0" "FK_UKF_PERSIST_H=5" "FK_UKF_PERSIST_H=8" "FK_UKF_PERSIST_H=10" "FK_UKF_PERSIST_H=13" "FK_UKF_PERSIST=0" "FK_UKF_PERSIST=1"; do
  echo "== $env" >> $O/c4_ab.txt
  env $env timeout 200 python tools/bench_configs.py --configs 4 2>> $O/c4.err | grep "fused" >> $O/c4_ab.txt
done
cat $O/c4_ab.txt | cut -c1-200
