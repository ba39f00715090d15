#!/bin/bash
# Round 5, third lease: the two-launch short-vector resampler (quick kernel + deferred filters), the smoother margin tests, the
# every-track parity test, the replay fix; C5 kernel durations measured like profiles/r04/kernel_durations.txt (tools/bench_resample.py
# under rocprofv3, 12 launches per shape) for the split (default) and the round-4 organisation (FK_WHOLE_SPLIT=0); bench.py as the
# driver runs it (placement = the API default now).
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider > $O/tests_1.log 2>&1
tail -4 $O/tests_1.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -p no:cacheprovider -k "margin or every_track" > $O/tests_2.log 2>&1
tail -4 $O/tests_2.log | cut -c1-300
grep -E "^E  " $O/tests_2.log | head -20 | cut -c1-400
cd /tmp
export TMPDIR=/tmp
RS="python $R/tools/bench_resample.py --shapes 1000x8000,125x8000,4000x8000,1000x2000,1000x4000 --iters 10"
for sp in 1 0; do
    FK_WHOLE_SPLIT=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats_$sp -- $RS > $O/resample_under_stats_$sp.jsonl 2> $O/rs_stats_$sp.err
    python $R/tools/kernel_trace_summary.py $O/rs_stats_$sp > $O/kernel_durations_split$sp.txt
    echo "== FK_WHOLE_SPLIT=$sp"; cut -c1-200 $O/kernel_durations_split$sp.txt
    FK_WHOLE_SPLIT=$sp timeout 300 $RS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['filters'], d['particles'], 'ms', d['ms'], 'frac', round(d['frac_hbm'],3))
" | tee $O/resample_plain_$sp.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["config"]["placement"][:40])
print({k: v for k, v in d["placement"].items() if k != "probe"})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
