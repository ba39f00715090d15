#!/bin/bash
# Round 5, fifth lease: FK_KF_FLAG_OUT_INTERLEAVED (all four histories in one array: one write front) -- parity, then the three
# arrangements timed in one bench.py run each way; PMC passes of the quad kernel; the C++ host example with RCCL; the Kalman
# smoother's lane organisations at dim_x 13 / 14 (r05_b); the fused UKF (6,3) for the per-lease median table.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -k "interleav or one_array or histories_in_one or cpp_host or placement or persistent" > $O/tests_1.log 2>&1
tail -4 $O/tests_1.log | cut -c1-300
grep -E "^E  " $O/tests_1.log | head -10 | cut -c1-300
for pl in interleave quad; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --placement $pl > $O/bench_$pl.json 2> $O/bench_$pl.err; echo "bench $pl rc=$?"
    python - <<PY
import json
d = json.load(open("$O/bench_$pl.json"))
print("$pl", {k: d[k] for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "parity", d["parity_max_rel_vs_oracle"])
print({k: v for k, v in d["placement"].items() if k.endswith("_ms")})
PY
done
cd /tmp
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu --placement quad"
FK_BENCH_SKIP_PROBE=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $BENCH > $O/bench_under_rocprof_pmc_fetch.json 2> $O/prof_fetch.err; echo "fetch rc=$?"
FK_BENCH_SKIP_PROBE=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $BENCH > $O/bench_under_rocprof_pmc_write.json 2> $O/prof_write.err; echo "write rc=$?"
cd $R
python tools/pmc_reduce.py $O/prof_fetch $O/prof_write "kf_fast_kernel<4, 2, 0, false, true, false, 0, false, false, false, 2>" > $O/pmc_headline_quad.json; cut -c1-400 $O/pmc_headline_quad.json
for d in prof_fetch prof_write; do for f in $(find $O/$d -name "*counter_collection.csv"); do head -1 $f > $O/${d}_fk.csv; grep "fk::" $f >> $O/${d}_fk.csv; done; done
find $O -name "*counter_collection.csv" -size +1M -delete
bash tools/gpu_scripts/r05_b.sh 2>&1 | tail -20
mkdir -p $O/r05b; cp -r $R/gpurun_out/r05b/* $O/r05b/ 2>/dev/null
cd /tmp
timeout 200 python $R/tools/bench_ukf.py --dims 6x3 --N 100000 --T 100 > $O/ukf_6x3.jsonl 2>/dev/null; cut -c1-200 $O/ukf_6x3.jsonl
