#!/bin/bash
# Round 4, lease h: full suite (the 4 GiB bank test runs inside it now) + the one-pass resampler's look-back depth A/B on few long
# vectors (kernel durations under rocprofv3).
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04h
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --durations=6 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu_full.log | cut -c1-200
cd /tmp
RS="python $R/tools/bench_resample.py --shapes 1x8000000,8x8000000,32x8000000,125x8000000,1x1000000 --iters 10"
for deep in 0 1; do
  FK_OP_DEEP=$deep timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_deep$deep -- $RS > $O/resample_deep$deep.jsonl 2> $O/rs_deep$deep.err
  python $R/tools/kernel_trace_summary.py $O/rs_deep$deep 2>/dev/null | grep -E "onepass" | sed "s/^/deep=$deep: /" | cut -c1-200 | tee -a $O/onepass_deep_ab.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
