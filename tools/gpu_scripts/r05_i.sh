#!/bin/bash
# Round 5: the quick resampler with one s_barrier where __syncthreads_or cost three (twice per filter) -- every route bit-exact,
# then kernel durations like the earlier leases; the placement test that failed on a loop variable.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider > $O/tests_1.log 2>&1
tail -3 $O/tests_1.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -p no:cacheprovider -k "placement" > $O/tests_2.log 2>&1
tail -3 $O/tests_2.log | cut -c1-200
cd /tmp
export TMPDIR=/tmp
RS="python $R/tools/bench_resample.py --shapes 1000x8000,125x8000,4000x8000,1000x2000,1000x4000,500x8000 --iters 10"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats -- $RS > $O/resample_under_stats.jsonl 2> $O/rs_stats.err
python $R/tools/kernel_trace_summary.py $O/rs_stats > $O/kernel_durations.txt; cut -c1-200 $O/kernel_durations.txt
timeout 300 $RS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['filters'], d['particles'], 'ms', d['ms'], 'frac', round(d['frac_hbm'],3))
" | tee $O/resample_plain.txt
find $O -name "*kernel_trace.csv" -size +1M -delete
timeout 200 python $R/tools/bench_ukf.py --dims 6x3 --N 100000 --T 100 > $O/ukf_6x3.jsonl 2>/dev/null; cut -c1-200 $O/ukf_6x3.jsonl
