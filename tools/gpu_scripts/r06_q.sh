#!/bin/bash
# Round 6, lease q: the (9) smoother on the persistent grid -- bit identity, then C3 with and without it
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06q
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_kf.py -m gpu -q -p no:cacheprovider -x -k "persistent" > $O/pytest_pers.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pers.log
for env in "FK_RTS_PERSIST=0" "FK_RTS_PERSIST=1" "FK_RTS_PERSIST=1 FK_RTS_PERSIST_H=2" "FK_RTS_PERSIST=1 FK_RTS_PERSIST_H=4" "FK_RTS_PERSIST=0" "FK_RTS_PERSIST=1" "FK_RTS_PERSIST=1 FK_ML9=m"; do
  echo "== $env" >> $O/c3_ab.txt
  env $env timeout 300 python tools/bench_configs.py --configs 3 --layouts soa,aos >> $O/c3_ab.txt 2>> $O/c3.err
done
grep "==\|rts" $O/c3_ab.txt | cut -c1-260
timeout 1200 python -m pytest tests/test_gpu_kf.py tests/test_gpu_baseline_configs.py -m gpu -q -p no:cacheprovider -x > $O/pytest_kf.log 2>&1; echo "pytest kf rc=$?"; tail -3 $O/pytest_kf.log
