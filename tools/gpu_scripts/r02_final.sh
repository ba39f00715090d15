#!/bin/bash
# Round 2, closing lease: every kernel of DESIGN section 5 once more (after the AOS pair accesses of the sigma / steady-state
# kernels), then exactly what the driver does at round end (tools/gpu_scripts/round_check.sh).
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r02final
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3456789a --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
cd $R
grep -E "^\{" $O/prof_cfg.log > $O/configs_all.jsonl; wc -l $O/configs_all.jsonl
find $O -name "*kernel_trace.csv" -size +1M -delete
bash tools/gpu_scripts/round_check.sh
