#!/bin/bash
# Round 4, lease s: model blocks copied into registers per half-step where ONE wave per SIMD leaves LDS latency uncovered
# (kf_fast dim_x 7..9, the register-resident IMM banks): full suite, then the dims 3..8 rows, the IMM rows, the extras rows.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04s
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu_full.log | cut -c1-220
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs a8e --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
grep -E "^\{" $O/prof_cfg.log > $O/configs_a8e.jsonl; python - <<PY
import json
for l in open("$O/configs_a8e.jsonl"):
    d=json.loads(l); print(d["kernel"][:95], "ms=%.3f"%d["ms"], "frac=%.3f"%d["frac_of_8TBs"], d.get("parity_max_rel",""))
PY
grep -v "^{" $O/prof_cfg.log | grep -iE "error|assert|Traceback" | head
python $R/tools/kernel_trace_summary.py $O/prof_cfg > $O/configs_a8e_kernel_durations.txt 2>&1
for f in $(find $O/prof_cfg -name "*kernel_stats.csv"); do cp $f $O/configs_a8e_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -size +1M -delete
