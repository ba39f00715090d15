#!/bin/bash
# Round 5, fourth lease: the single-launch quick resampler (common path + unlikely tail), its durations against FK_WHOLE_QUICK=0
# on the same GPU; the end-to-end cost of the drop-in call (tools/bench_api.py); HBM traffic of the bench's dominant kernel now
# that the default placement is the interleaved array (separate FETCH_SIZE / WRITE_SIZE passes, nothing else in the command).
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider > $O/tests_1.log 2>&1
tail -3 $O/tests_1.log | cut -c1-200
cd /tmp
export TMPDIR=/tmp
RS="python $R/tools/bench_resample.py --shapes 1000x8000,125x8000,4000x8000,1000x2000,1000x4000,250x8000,500x8000 --iters 10"
for q in 1 0; do
    FK_WHOLE_QUICK=$q timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats_$q -- $RS > $O/resample_under_stats_$q.jsonl 2> $O/rs_stats_$q.err
    python $R/tools/kernel_trace_summary.py $O/rs_stats_$q > $O/kernel_durations_quick$q.txt
    echo "== FK_WHOLE_QUICK=$q"; cut -c1-200 $O/kernel_durations_quick$q.txt
    FK_WHOLE_QUICK=$q timeout 300 $RS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['filters'], d['particles'], 'ms', d['ms'], 'frac', round(d['frac_hbm'],3))
" | tee $O/resample_plain_$q.txt
done
# the literal BASELINE configs[4] step (resample + gather-mean) as bench_c5 times it
timeout 300 python $R/tools/bench_c5.py --filters 1000 --particles 8000 > $O/bench_c5_1000x8000.json 2>/dev/null; cut -c1-400 $O/bench_c5_1000x8000.json
find $O -name "*kernel_trace.csv" -size +1M -delete
timeout 900 python $R/tools/bench_api.py > $O/bench_api.jsonl 2> $O/bench_api.err; cut -c1-700 $O/bench_api.jsonl; tail -2 $O/bench_api.err
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu"
FK_BENCH_SKIP_PROBE=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $BENCH > $O/bench_under_rocprof_pmc_fetch.json 2> $O/prof_fetch.err; echo "fetch rc=$?"
FK_BENCH_SKIP_PROBE=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $BENCH > $O/bench_under_rocprof_pmc_write.json 2> $O/prof_write.err; echo "write rc=$?"
cd $R
python tools/pmc_reduce.py $O/prof_fetch $O/prof_write "kf_fast_kernel<4, 2, 0, false, true, false, 0, false, false, false, true>" > $O/pmc_headline.json; cat $O/pmc_headline.json | cut -c1-600
for d in prof_fetch prof_write; do for f in $(find $O/$d -name "*counter_collection.csv"); do head -1 $f > $O/${d}_fk.csv; grep "fk::" $f >> $O/${d}_fk.csv; done; done
find $O -name "*counter_collection.csv" -size +1M -delete
