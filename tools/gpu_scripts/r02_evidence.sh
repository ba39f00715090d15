#!/bin/bash
# Round-2 evidence in ONE lease: full GPU tests, smoke, clocks, then the SAME
# `bench.py --steps 20 --warmup 5` invocation four times -- plain, under rocprofv3 --kernel-trace --stats, and under the two
# separate PMC passes (FETCH_SIZE / WRITE_SIZE, never combined with a trace domain) -- and the resample kernels likewise.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_scripts/r02_evidence.sh'
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r02ev
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
cd $R
rocm-smi --showclocks --showpower > $O/smi_before.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench rc=$?"; cut -c1-400 $O/bench_plain.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $BENCH > $O/bench_under_stats.json 2> $O/prof_stats.err; echo "stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $BENCH > $O/bench_under_fetch.json 2> $O/prof_fetch.err; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $BENCH > $O/bench_under_write.json 2> $O/prof_write.err; echo "write rc=$?"
RS="python $R/tools/bench_resample.py --shapes 125x8000000,1000x8000 --iters 10"
timeout 300 $RS > $O/resample_plain.jsonl 2> $O/resample_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats -- $RS > $O/resample_under_stats.jsonl 2> $O/rs_stats.err; echo "rs stats rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rs_fetch -- $RS > /dev/null 2> $O/rs_fetch.err; echo "rs fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rs_write -- $RS > /dev/null 2> $O/rs_write.err; echo "rs write rc=$?"
cd $R
rocm-smi --showclocks --showpower > $O/smi_after.txt 2>&1
python tools/pmc_summary.py --all $O/prof_fetch $O/prof_write $O/rs_fetch $O/rs_write > $O/pmc_summary.txt 2>&1
for f in $(find $O/prof_stats $O/rs_stats -name "*kernel_stats.csv"); do echo $f; cut -c1-200 $f | head -6; done
cat $O/pmc_summary.txt | grep -v "^==" | head -20
# the raw per-dispatch counter CSVs are large: keep only the per-kernel summaries + the kernel_stats
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
