#!/bin/bash
# Round 4, lease i: how much of the fused UKF's step is its store path?  Kernel durations with and without the per-step histories.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04i
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for mode in "" "--no-outputs"; do
  tag=$( [ -z "$mode" ] && echo with || echo without )
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ukf_$tag -- python $R/tools/bench_ukf.py --dims 6x3,4x2 $mode > $O/ukf_$tag.jsonl 2> $O/ukf_$tag.err
  python $R/tools/kernel_trace_summary.py $O/ukf_$tag 2>/dev/null | grep -E "ukf_linear_kernel" | sed "s/^/outputs $tag: /" | cut -c1-200 | tee -a $O/ukf_store_path.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
