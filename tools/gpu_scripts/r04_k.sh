#!/bin/bash
# Round 4, lease k: LDS operands requested one iteration ahead in the several-lanes-per-track kernels (kf_ml, rts_ml, kf_mlg,
# rts_mlg), batched tile copy-outs, kf_mlg's EX instantiations (Saver histories at dim_x >= 10 and (9,3)):
# full suite, then C3 / dims 10..16 / the extras rows under rocprofv3 stats.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu_full.log | cut -c1-220
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3be --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
grep -E "^\{" $O/prof_cfg.log > $O/configs_3be.jsonl; python - <<PY
import json
for l in open("$O/configs_3be.jsonl"):
    d=json.loads(l); print(d["kernel"][:95], "ms=%.3f"%d["ms"], "frac=%.3f"%d["frac_of_8TBs"], d.get("parity",""))
PY
python $R/tools/kernel_trace_summary.py $O/prof_cfg > $O/configs_3be_kernel_durations.txt 2>&1; cut -c1-220 $O/configs_3be_kernel_durations.txt | head -60
for f in $(find $O/prof_cfg -name "*kernel_stats.csv"); do cp $f $O/configs_3be_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -size +1M -delete
