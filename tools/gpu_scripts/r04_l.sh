#!/bin/bash
# Round 4, lease l: after the two fixes of lease k (prior-covariance stores of the VAR element-major kernel; element-major
# copy-outs predicate their stores instead of relying on an offset no 4 GiB descriptor drops): full suite, C3 / dims 10..16 /
# extras rows under rocprofv3 stats, and the SQ counters of the C3 forward kernel and smoother as ONE launch each.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04l
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu_full.log | cut -c1-220
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3be --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
grep -E "^\{" $O/prof_cfg.log > $O/configs_3be.jsonl; python - <<PY
import json
for l in open("$O/configs_3be.jsonl"):
    d=json.loads(l); print(d["kernel"][:95], "ms=%.3f"%d["ms"], "frac=%.3f"%d["frac_of_8TBs"], d.get("parity_max_rel",""))
PY
grep -v "^{" $O/prof_cfg.log | grep -iE "error|assert|Traceback" | head
python $R/tools/kernel_trace_summary.py $O/prof_cfg > $O/configs_3be_kernel_durations.txt 2>&1
for f in $(find $O/prof_cfg -name "*kernel_stats.csv"); do cp $f $O/configs_3be_kernel_stats.csv; done
C3="python $R/tools/bench_configs.py --configs 3 --layouts soa"
export FK_ML_CHUNKS=1,1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_one -- $C3 > $O/c3_one.log 2>&1
python $R/tools/kernel_trace_summary.py $O/c3_one 2>/dev/null | grep -E "kf_ml|rts_ml" | cut -c1-200
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c3_sq1 -- $C3 > /dev/null 2> $O/c3_sq1.err; echo "sq1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $O/c3_sq2 -- $C3 > /dev/null 2> $O/c3_sq2.err; echo "sq2 rc=$?"
unset FK_ML_CHUNKS
cd $R
python tools/pmc_summary.py $O/c3_sq1 $O/c3_sq2 --kernel _ml_kernel > $O/c3_sq_counters.jsonl; cut -c1-1200 $O/c3_sq_counters.jsonl
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +1M -delete
