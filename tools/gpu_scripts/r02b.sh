#!/bin/bash
# Round 2, second lease: the whole GPU suite (no -x: every failure at once), the fused-UKF model A/B, the one-pass
# resampler's build variants, the headline PMC passes (the first lease lost them to the profiler attaching to the CPU
# baseline's 256 workers) and SQ counters of the fused UKF.
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r02b2
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
cd $R
FK_PARITY_LOG=$O/parity_errors.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
for v in 1 0; do FK_UKF_LDS_MODEL=$v timeout 300 python tools/bench_configs.py --configs 4 --layouts soa,aos 2>&1 | grep "fused" | sed "s/^/lds_model=$v /" ; done | tee $O/ukf_ab.txt
timeout 600 python tools/exp_rs_variants.py --run > $O/rs_variants.log 2>&1; echo "variants rc=$?"; grep -E "time_shape|MISMATCH|ALL|checked" $O/rs_variants.log | cut -c1-700
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $BENCH > $O/bench_under_fetch.json 2> $O/prof_fetch.err; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $BENCH > $O/bench_under_write.json 2> $O/prof_write.err; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/ukf_sq -- python $R/tools/bench_configs.py --configs 4 --layouts soa > $O/ukf_sq.log 2>&1; echo "ukf sq rc=$?"
cd $R
python tools/pmc_summary.py --all $O/prof_fetch $O/prof_write $O/ukf_sq > $O/pmc_summary.txt 2>&1
grep -E "kf_fast|ukf_linear" $O/pmc_summary.txt | cut -c1-60,90-200
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
