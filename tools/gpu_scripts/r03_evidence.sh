#!/bin/bash
# Round 3 evidence in ONE lease: box state, the SAME `python bench.py --steps 20 --warmup 5` plain / under rocprofv3
# --kernel-trace --stats / under the two separate PMC passes (FETCH_SIZE, WRITE_SIZE: never combined with a trace domain), the
# resampling kernels likewise, every other kernel of DESIGN section 5 under rocprofv3 stats, the configs[4] step end to end,
# and the RCCL branch on a 1-rank group.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_scripts/r03_evidence.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ev
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_kf.py -m gpu -q -x -p no:cacheprovider -k "four_lane" > $O/pytest_four_lane.log 2>&1; echo "pytest four-lane rc=$?"; tail -3 $O/pytest_four_lane.log
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH --no-cpu > $O/bench_under_rocprof_stats.json 2> $O/prof_stats.err; echo "stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $BENCH --no-cpu > $O/bench_under_rocprof_pmc_fetch.json 2> $O/prof_fetch.err; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $BENCH --no-cpu > $O/bench_under_rocprof_pmc_write.json 2> $O/prof_write.err; echo "write rc=$?"
RS="python $R/tools/bench_resample.py --shapes 125x8000000,1000x8000,125x8000,8x8000000,1x8000000,4000x8000,1000x100000 --iters 10"
timeout 300 $RS > $O/resample_shapes.jsonl 2> $O/resample_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats -- $RS > $O/resample_under_stats.jsonl 2> $O/rs_stats.err; echo "rs stats rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rs_fetch -- $RS > /dev/null 2> $O/rs_fetch.err; echo "rs fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rs_write -- $RS > /dev/null 2> $O/rs_write.err; echo "rs write rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/rs_sq -- $RS > /dev/null 2> $O/rs_sq.err; echo "rs sq rc=$?"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3456789ab --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
cd $R
grep -E "^\{" $O/prof_cfg.log > $O/configs_all.jsonl; wc -l $O/configs_all.jsonl
for sh in "1000 8000" "125 8000" "125 8000000"; do set -- $sh; timeout 300 python tools/bench_c5.py --filters $1 --particles $2 > $O/bench_c5_$1x$2.json 2>/dev/null; cut -c1-400 $O/bench_c5_$1x$2.json; done
timeout 300 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force-dist rc=$?"; grep -E "RCCL|rccl" $O/bench_force_dist.err | head -3
timeout 300 python tools/bench_c5.py --filters 125 --particles 8000 --force-dist > $O/bench_c5_force_dist.json 2> $O/bench_c5_force_dist.err; echo "c5 force-dist rc=$?"
python tools/pmc_summary.py --all $O/prof_fetch $O/prof_write $O/rs_fetch $O/rs_write $O/rs_sq > $O/pmc_summary.txt 2>&1
python tools/pmc_reduce.py $O/prof_fetch $O/prof_write kf_fast_kernel > $O/pmc_headline.json; cat $O/pmc_headline.json
python tools/kernel_trace_summary.py $O/prof_stats $O/rs_stats > $O/kernel_durations.txt; cat $O/kernel_durations.txt | cut -c1-200
# keep what is committed small: the headline kernel's counter rows, the per-kernel stats; drop the big traces
for d in prof_fetch prof_write rs_fetch rs_write rs_sq; do for f in $(find $O/$d -name "*counter_collection.csv"); do head -1 $f > $O/${d}_fk.csv; grep "fk::" $f >> $O/${d}_fk.csv; done; done
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
