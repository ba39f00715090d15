#!/bin/bash
# Round 6, lease w: two candidate binades on the fast path (FK_OP_LB bit 4)
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06w
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH=125x8000000,1000x100000,8x8000000,1x8000000,32x1000000,2x4000000,1x2000000,4x8000000
for env in "FK_OP_LB=7" "FK_OP_LB=23" "FK_OP_LB=3" "FK_OP_LB=19" "FK_OP_LB=7" "FK_OP_LB=23"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-110
for lb in 7 23; do FK_OP_LB=$lb timeout 300 python tools/op_timeline.py --run --shapes 8x8000000,1x8000000 --out $O/lb$lb > $O/timeline_lb$lb.txt 2> $O/timeline_lb$lb.err; done
grep -h "general chunks\|quick slow\|whole call\|fast chunks" $O/timeline_lb*.txt | cut -c1-230
