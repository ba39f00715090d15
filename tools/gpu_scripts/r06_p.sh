#!/bin/bash
# Round 6, lease p: do six / seven workgroups per CU actually become resident?  (the same kernel with 20.2 KB of LDS instead of 22.5)
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
export TMPDIR=/tmp
cd $R
SH=125x8000000,1000x100000
for lib in "" filterpy_amd/csrc/exp_build/libfilterhip_d16.so; do
for env in "FK_OP_WAVES=5" "FK_OP_WAVES=6" "FK_OP_WAVES=7" "FK_OP_WAVES=7 FK_OP_LDS_PAD=3000" "FK_OP_WAVES=7 FK_OP_LDS_PAD=7000" "FK_OP_WAVES=5"; do
  echo "== lib=$lib $env" >> $O/rs_ab.txt
  env FK_LIB=$lib $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
done
cat $O/rs_ab.txt | cut -c1-120
