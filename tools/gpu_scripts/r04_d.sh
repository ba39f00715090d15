#!/bin/bash
# Round 4, lease d: does a start-up stagger of co-resident workgroups (their store bursts no longer coincide) speed the fused
# UKF up?  FK_UKF_STAGGER="div,mod,n"; kernel durations under rocprofv3.  Plus the gather-mean unroll depth A/B.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
C4="python $R/tools/bench_configs.py --configs 4 --layouts soa,aos"
for sg in "off" "256,2,1" "256,2,2" "1,2,1" "1,2,2" "1,4,1" "128,4,1"; do
  tag=$(echo $sg | tr ',' '_')
  if [ "$sg" = "off" ]; then unset FK_UKF_STAGGER; else export FK_UKF_STAGGER=$sg; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- $C4 > /dev/null 2> $O/st_$tag.err
  python $R/tools/kernel_trace_summary.py $O/st_$tag 2>/dev/null | grep -E "ukf_linear_kernel" | sed "s/^/stagger $sg: /" | cut -c1-200 | tee -a $O/stagger.txt
done
unset FK_UKF_STAGGER
cd $R
for u in 8 16; do
  FK_GATHER_MEAN_U=$u timeout 400 python tools/bench_c5.py --filters 125 --particles 8000000 > $O/bench_c5_125x8000000_u$u.json 2>/dev/null; cut -c1-520 $O/bench_c5_125x8000000_u$u.json | cut -c330-520
done
find $O -name "*kernel_trace.csv" -size +1M -delete
