#!/bin/bash
# Round 6 closing evidence in ONE lease (the round-4 recipe): box state, the full GPU test suite, smoke, the SAME
# `python bench.py --steps 20 --warmup 5` plain / under rocprofv3 --kernel-trace --stats / under the two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: never combined with a trace domain), every other kernel of docs/MEASUREMENTS.md under rocprofv3 stats,
# the fused UKF in both summation orders, the resampling shapes, the configs[4] step end to end, the RCCL branch on a 1-rank
# group, the end-to-end cost of the API call, the C++ host example.
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/gpu_scripts/r06_final.sh'
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06fin
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
timeout 300 $BENCH --no-cpu --no-configs --placement probe > $O/bench_placement_probe.json 2>/dev/null
timeout 300 $BENCH --no-cpu --no-configs --placement none > $O/bench_placement_none.json 2>/dev/null
cd /tmp
export FK_BENCH_SKIP_PROBE=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH --no-cpu --no-configs > $O/bench_under_rocprof_stats.json 2> $O/prof_stats.err; echo "stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $BENCH --no-cpu --no-configs > $O/bench_under_rocprof_pmc_fetch.json 2> $O/prof_fetch.err; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $BENCH --no-cpu --no-configs > $O/bench_under_rocprof_pmc_write.json 2> $O/prof_write.err; echo "write rc=$?"
unset FK_BENCH_SKIP_PROBE
RS="python $R/tools/bench_resample.py --shapes 125x8000000,1000x8000,125x8000,8x8000000,1x8000000,4000x8000,1000x100000,500x8000,1000x4000,1000x2000 --iters 10"
timeout 300 $RS > $O/resample_shapes.jsonl 2> $O/resample_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats -- $RS > $O/resample_under_stats.jsonl 2> $O/rs_stats.err; echo "rs stats rc=$?"
# round 3's one-pass kernel (FK_OP_V2=0) in the same lease: the A/B of round 6's resample_onepass2_kernel
FK_OP_V2=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats_q0 -- python $R/tools/bench_resample.py --shapes 125x8000000,8x8000000,1x8000000,1000x100000,32x1000000 --iters 10 > $O/resample_under_stats_round3_onepass.jsonl 2> $O/rs_stats_q0.err
# the one-pass kernel's traffic (two separate PMC passes) and its phase clocks before (round 3's kernel) / after
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/rs_pmc_$c -- python $R/tools/bench_resample.py --shapes 125x8000000 --iters 5 > /dev/null 2> $O/rs_pmc_$c.err; done
python - <<PY > $O/onepass_pmc.json
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob("$O/rs_pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        v += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "onepass2" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c + "_KiB_mean"] = sum(v) / max(1, len(v))
    out[c + "_launches"] = len(v)
out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE_KiB_mean"] + out["WRITE_SIZE_KiB_mean"]) * 1024
out["algorithmic_bytes"] = 12.0 * 125 * 8000000
out["ratio"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes"]
print(json.dumps(out))
PY
cat $O/onepass_pmc.json
(cd $R; for v in 0 1; do FK_OP_V2=$v timeout 300 python tools/op_phase.py --run --shapes 125x8000000,8x8000000,1x8000000 --iters 3 >> $O/onepass_phase_clocks.jsonl 2>> $O/op_phase.err; done)
timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3456789abersu --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
UK="python $R/tools/bench_ukf.py --dims 6x3,4x2,2x2,8x4,9x3"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ukf_stats -- $UK > $O/ukf_kernels.jsonl 2> $O/ukf_stats.err; echo "ukf rc=$?"
FK_UKF_PAIRED=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ukf_stats_index -- $UK > $O/ukf_kernels_index_order.jsonl 2> $O/ukf_stats_index.err; echo "ukf index rc=$?"
timeout 200 python $R/tools/bench_ukf.py --dims 6x3 --N 1000000 --T 20 >> $O/ukf_kernels.jsonl 2>> $O/ukf_stats.err
cd $R
grep -E "^\{" $O/prof_cfg.log > $O/configs_all.jsonl; wc -l $O/configs_all.jsonl
# HBM traffic of the C3 / C4 kernels (configs[2], [3]): two separate PMC passes of the same command, every fk:: kernel by name
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/cfg_pmc_$c -- python $R/tools/bench_configs.py --configs 34 --layouts soa,aos > /dev/null 2> $O/cfg_pmc_$c.err; done
cd $R
python tools/pmc_configs.py $O/cfg_pmc_FETCH_SIZE $O/cfg_pmc_WRITE_SIZE "--expect=kf_ml_kernel<9, 3=14784000000" "--expect=rts_ml_kernel<9=27360000000" "--expect=ukf_linear_kernel<6, 3=3600000000" > $O/configs_traffic.jsonl 2> $O/configs_traffic.err; cut -c1-260 $O/configs_traffic.jsonl
python tools/bench_imm_outputs.py --dims 16x8x2 > $O/imm_outputs.jsonl 2>/dev/null; python tools/bench_imm_outputs.py --dims 9x4x8 >> $O/imm_outputs.jsonl 2>/dev/null
for sh in "1000 8000" "125 8000" "125 8000000"; do set -- $sh; timeout 400 python tools/bench_c5.py --filters $1 --particles $2 > $O/bench_c5_$1x$2.json 2>/dev/null; cut -c1-300 $O/bench_c5_$1x$2.json; done
timeout 300 python bench.py --steps 10 --warmup 3 --scaling strong --no-cpu --no-configs > $O/bench_strong_1rank.json 2> $O/bench_strong.err; echo "strong rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu --no-configs > $O/bench_force_dist_1rank_nccl.json 2> $O/bench_force_dist.err; echo "force-dist rc=$?"; grep -iE "rccl|nccl version" $O/bench_force_dist.err | head -2
timeout 300 python tools/bench_c5.py --filters 125 --particles 8000 --force-dist > $O/bench_c5_force_dist_1rank_nccl.json 2> $O/bench_c5_force_dist.err; echo "c5 force-dist rc=$?"
timeout 900 python tools/bench_api.py > $O/bench_api.jsonl 2> $O/bench_api.err; cut -c1-300 $O/bench_api.jsonl
timeout 120 examples/c_abi_multi_gpu 200000 50 > $O/c_abi_multi_gpu.log 2>&1; cat $O/c_abi_multi_gpu.log
python tools/pmc_reduce.py $O/prof_fetch $O/prof_write "kf_fast_kernel<4, 2, 0, false, true, false, 0, false, false, false, true>" > $O/pmc_headline.json; cat $O/pmc_headline.json
python tools/kernel_trace_summary.py --last 20 $O/prof_stats > $O/kernel_durations_bench_last20.txt; grep kf_fast $O/kernel_durations_bench_last20.txt | cut -c1-200
python tools/kernel_trace_summary.py $O/rs_stats > $O/kernel_durations.txt; cut -c1-200 $O/kernel_durations.txt
python tools/kernel_trace_summary.py $O/rs_stats_q0 > $O/kernel_durations_round3_onepass.txt
python tools/kernel_trace_summary.py $O/prof_cfg > $O/configs_all_kernel_durations.txt 2>&1
python tools/kernel_trace_summary.py $O/ukf_stats > $O/ukf_kernel_durations.txt 2>&1; python tools/kernel_trace_summary.py $O/ukf_stats_index > $O/ukf_kernel_durations_index_order.txt 2>&1
# keep what is committed small: the headline kernel's counter rows, the per-kernel stats; drop the big traces
for d in prof_fetch prof_write; do for f in $(find $O/$d -name "*counter_collection.csv"); do head -1 $f > $O/${d}_fk.csv; grep "fk::" $f >> $O/${d}_fk.csv; done; done
for f in $(find $O/prof_stats -name "*kernel_trace.csv"); do head -1 $f > $O/bench_kernel_trace_fk.csv; grep "kf_fast" $f >> $O/bench_kernel_trace_fk.csv; done
for f in $(find $O/prof_cfg -name "*kernel_stats.csv"); do cp $f $O/configs_all_kernel_stats.csv; done
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
