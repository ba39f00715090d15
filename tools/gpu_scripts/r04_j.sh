#!/bin/bash
# Round 4, lease j: (1) why did `bench.py --force-dist` print nothing in the closing lease; (2) the fused UKF's element-major pair
# stores (SP instantiations): full suite, then kernel durations with FK_UKF_SOA_PAIRS=0 / default.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04j
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python -X faulthandler bench.py --steps 6 --warmup 2 --force-dist --no-cpu --placement none > $O/force_dist.out 2> $O/force_dist.err; echo "force-dist rc=$?"; wc -c $O/force_dist.out; tail -5 $O/force_dist.err | cut -c1-300; cut -c1-200 $O/force_dist.out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu_full.log | cut -c1-200
cd /tmp
C4="python $R/tools/bench_configs.py --configs 4 --layouts soa"
for sp in 0 1; do
  FK_UKF_SOA_PAIRS=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_sp$sp -- $C4 > /dev/null 2> $O/c4_sp$sp.err
  python $R/tools/kernel_trace_summary.py $O/c4_sp$sp 2>/dev/null | grep -E "ukf_linear_kernel" | sed "s/^/soa pairs=$sp: /" | cut -c1-200 | tee -a $O/ukf_soa_pairs.txt
  FK_UKF_SOA_PAIRS=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ukf_sp$sp -- python $R/tools/bench_ukf.py --dims 6x3,4x2,2x2 --layouts soa > /dev/null 2> $O/ukf_sp$sp.err
  python $R/tools/kernel_trace_summary.py $O/ukf_sp$sp 2>/dev/null | grep -E "ukf_linear_kernel" | sed "s/^/soa pairs=$sp (bench_ukf): /" | cut -c1-200 | tee -a $O/ukf_soa_pairs.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
