#!/bin/bash
# Round 6, lease r: one clear launch + one follow-up launch around the one-pass kernel; seg_prepare barrier by barrier
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06r
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH=125x8000000,1000x100000,8x8000000,1x8000000,32x1000000,2x4000000,1x2000000
for i in 1 2; do
  echo "== run $i" >> $O/rs_ab.txt
  timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-110
timeout 300 python tools/op_timeline.py --run --shapes 1x2000000,1x8000000,1000x100000 --out $O/tl > $O/timeline.txt 2> $O/timeline.err
grep -h "general chunks\|seg_prepare\|quick slow\|whole call\|fast chunks" $O/timeline.txt | cut -c1-250
