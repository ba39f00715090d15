#!/bin/bash
# Round 4, second lease: the pair-regrouped fused UKF with every operand of a half-step requested at its head (fk_ukf.hpp),
# measured like the first lease's build (gpurun_out/r04a/ukf_paired.jsonl) + kernel durations under rocprofv3 + SQ counters;
# the wide gather-mean kernel A/B; the full GPU suite on the second batch (IMM likelihood, residual clamp, UKF attributes).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_scripts/r04_b.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_gpu_full.log
B="python $R/tools/bench_ukf.py --dims 6x3,4x2,2x2,8x4,9x3,9x4"
timeout 300 $B > $O/ukf_paired.jsonl 2> $O/ukf_paired.err; echo "paired rc=$?"
FK_UKF_PAIRED=0 timeout 300 $B > $O/ukf_index_order.jsonl 2> $O/ukf_index_order.err; echo "index rc=$?"
timeout 200 python tools/bench_ukf.py --dims 6x3,4x2 --N 1000000 --T 20 > $O/ukf_paired_1e6.jsonl 2>> $O/ukf_paired.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04b"
for f in ("ukf_paired.jsonl", "ukf_index_order.jsonl", "ukf_paired_1e6.jsonl"):
    for l in open(O + "/" + f):
        d = json.loads(l)
        print(f[:-6], d["kernel"], d["N"], "ms %.3f frac %.3f parity %.1e" % (d["ms"], d["frac_of_8TBs"], d["parity_max_rel"]))
PY
cd /tmp
C4="python $R/tools/bench_configs.py --configs 4 --layouts soa,aos"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- $C4 > $O/c4_under_stats.jsonl 2> $O/c4_stats.err; echo "c4 stats rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c4_sq1 -- $C4 > /dev/null 2> $O/c4_sq1.err; echo "sq1 rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $O/c4_sq2 -- $C4 > /dev/null 2> $O/c4_sq2.err; echo "sq2 rc=$?"
cd $R
python tools/pmc_summary.py $O/c4_sq1 $O/c4_sq2 --kernel ukf_linear_kernel > $O/ukf_sq_counters.jsonl; cut -c1-900 $O/ukf_sq_counters.jsonl
python tools/kernel_trace_summary.py $O/c4_stats > $O/c4_kernel_durations.txt 2>&1; grep -E "ukf|sigma|ut_" $O/c4_kernel_durations.txt | cut -c1-220
# gather-mean A/B and the configs[4] step end to end
for sh in "125 8000000" "1000 8000"; do set -- $sh
  timeout 400 python tools/bench_c5.py --filters $1 --particles $2 > $O/bench_c5_$1x$2.json 2> $O/bench_c5_$1x$2.err; cut -c1-600 $O/bench_c5_$1x$2.json
  FK_GATHER_MEAN_WIDE=0 timeout 400 python tools/bench_c5.py --filters $1 --particles $2 > $O/bench_c5_$1x$2_narrow.json 2>/dev/null; cut -c1-600 $O/bench_c5_$1x$2_narrow.json
done
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
