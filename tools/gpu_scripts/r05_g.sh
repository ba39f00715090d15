#!/bin/bash
# Round 5, sixth lease: the second form of FK_KF_FLAG_OUT_INTERLEAVED -- the two mean histories as the halves of ONE array next to
# the covariance pair's (two write fronts of full lines instead of three) -- parity, then bench.py with each arrangement.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -p no:cacheprovider -k "histories_in_one or interleav" > $O/tests_1.log 2>&1
tail -3 $O/tests_1.log | cut -c1-300
for pl in quad interleave; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --placement $pl > $O/bench_$pl.json 2> $O/bench_$pl.err; echo "bench $pl rc=$?"
    python - <<PY
import json
d = json.load(open("$O/bench_$pl.json"))
print("$pl", {k: d[k] for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "parity", d["parity_max_rel_vs_oracle"])
print({k: v for k, v in d["placement"].items() if k.endswith("_ms")})
PY
done
cd /tmp
timeout 200 python $R/tools/bench_ukf.py --dims 6x3 --N 100000 --T 100 > $O/ukf_6x3.jsonl 2>/dev/null; cut -c1-200 $O/ukf_6x3.jsonl
