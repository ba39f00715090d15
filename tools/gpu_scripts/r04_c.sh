#!/bin/bash
# Round 4, third lease: (1) the fused UKF with z carried from step to step and requested IN FRONT of the stores (counted vmcnt);
# (2) the interleaved covariance histories written TOGETHER (kf_fast IL instantiation) against two arrays / the pitch stores /
# the probe, several processes each; full GPU suite first.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_scripts/r04_c.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_gpu_full.log
cd /tmp
C4="python $R/tools/bench_configs.py --configs 4 --layouts soa,aos"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- $C4 > $O/c4_under_stats.jsonl 2> $O/c4_stats.err; echo "c4 stats rc=$?"
FK_UKF_PAIRED=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats_index -- $C4 > $O/c4_under_stats_index_order.jsonl 2> $O/c4_stats_index.err; echo "c4 index stats rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c4_sq1 -- $C4 > /dev/null 2> $O/c4_sq1.err; echo "sq1 rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $O/c4_sq2 -- $C4 > /dev/null 2> $O/c4_sq2.err; echo "sq2 rc=$?"
UK="python $R/tools/bench_ukf.py --dims 6x3,4x2,2x2,8x4,9x3"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ukf_stats -- $UK > $O/ukf_paired.jsonl 2> $O/ukf_stats.err; echo "ukf stats rc=$?"
cd $R
python tools/pmc_summary.py $O/c4_sq1 $O/c4_sq2 --kernel ukf_linear_kernel > $O/ukf_sq_counters.jsonl; cut -c1-900 $O/ukf_sq_counters.jsonl
python tools/kernel_trace_summary.py $O/c4_stats > $O/c4_kernel_durations.txt 2>&1; grep -E "ukf" $O/c4_kernel_durations.txt | cut -c1-220
python tools/kernel_trace_summary.py $O/c4_stats_index > $O/c4_kernel_durations_index_order.txt 2>&1; grep -E "ukf" $O/c4_kernel_durations_index_order.txt | cut -c1-220
python tools/kernel_trace_summary.py $O/ukf_stats > $O/ukf_kernel_durations.txt 2>&1; grep -E "ukf" $O/ukf_kernel_durations.txt | cut -c1-220
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for i in 1 2 3; do
  timeout 300 $BENCH --no-cpu --placement none > $O/bench_none_$i.json 2>/dev/null
  timeout 300 $BENCH --no-cpu > $O/bench_interleave_$i.json 2>/dev/null
  FK_FAST_NO_IL=1 timeout 300 $BENCH --no-cpu > $O/bench_interleave_pitch_$i.json 2>/dev/null
done
timeout 400 $BENCH --no-cpu --placement probe > $O/bench_probe.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04c"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "ms_per_step", round(d["ms_per_step"], 4),
              "frac", round(d["roofline"]["frac"], 4), {k: (v if not isinstance(v, str) else v[:24]) for k, v in d["placement"].items() if k != "grid_ms"})
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
