#!/bin/bash
# Round 4, lease p: every raw buffer descriptor scalar by construction (make_rsrc -> uniform_ptr; the fused UKF's element-major
# pair stores and the VAR kernels' control-input loads ran in waterfall loops): full suite, C4 with the pair stores on / off,
# the fused UKF at the other dims, the (4,2) variant rows.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu_full.log | cut -c1-220
cd /tmp
C4="python $R/tools/bench_configs.py --configs 4 --layouts soa,aos"
for sp in 1 0; do
  FK_UKF_SOA_PAIRS=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_sp$sp -- $C4 > $O/c4_sp$sp.jsonl 2> $O/c4_sp$sp.err
  python $R/tools/kernel_trace_summary.py $O/c4_sp$sp 2>/dev/null | grep -E "ukf_linear_kernel" | sed "s/^/pairs=$sp: /" | cut -c1-200 | tee -a $O/ukf_soa_pairs.txt
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ukf_stats -- python $R/tools/bench_ukf.py --dims 6x3,4x2,2x2,8x4,9x3 > $O/ukf_kernels.jsonl 2> $O/ukf_stats.err
python $R/tools/kernel_trace_summary.py $O/ukf_stats 2>/dev/null | grep -E "ukf_linear" | cut -c1-200 | tee $O/ukf_kernel_durations.txt
timeout 300 python $R/tools/bench_configs.py --configs 7 --layouts soa,aos 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['kernel'][:80], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'])
" | tee $O/variants.txt
find $O -name "*kernel_trace.csv" -size +1M -delete
