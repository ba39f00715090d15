#!/bin/bash
# Round 6, eighth lease: the look-back wave rotating over the SIMDs (FK_OP_LB_ROTATE), workgroups per CU with the slow path inlined.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x -k "onepass or c5" > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH=125x8000000,1000x100000,8x8000000,1x8000000
for env in "FK_OP_LB_ROTATE=1" "FK_OP_LB_ROTATE=0" "FK_OP_WAVES=5" "FK_OP_WAVES=7" "FK_OP_V2=0" "FK_OP_LB_ROTATE=0" "FK_OP_LB_ROTATE=1"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-110
