#!/bin/bash
# Run at the end of a lease: if the default bench line of this box says the headline kernel took > 5.9 ms (the slow GPUs of
# the pool: 6.2-6.8 ms against 5.3-5.4), collect what could tell WHY -- the box state, the other layout, the other kernels.
#   bash tools/gpu_scripts/slow_box_probe.sh <out dir> <bench json of this lease>
O=$1; J=$2
R=$GRAFT_REPO_ROOT
cd $R
MS=$(python -c "import json,sys; print(json.load(open('$J'))['roofline']['kernel_ms'])" 2>/dev/null || echo 0)
echo "headline kernel on this box: $MS ms"
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
if python -c "import sys; sys.exit(0 if float('$MS') > 5.9 else 1)"; then
  echo "SLOW BOX: probing"
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --layout soa > $O/slow_bench_soa.json 2>/dev/null
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --tracks 250000 > $O/slow_bench_quarter.json 2>/dev/null
  timeout 300 python tools/bench_configs.py --configs 3a --layouts soa,aos 2>/dev/null | grep "^{" > $O/slow_configs.jsonl
  timeout 120 python tools/bench_resample.py --shapes 125x8000000,1000x8000 --iters 10 2>/dev/null | grep "^{" > $O/slow_resample.jsonl
  python - <<PY
import json
for f in ("slow_bench_soa", "slow_bench_quarter"):
    try:
        d = json.load(open("$O/" + f + ".json")); print(f, d["roofline"]["kernel_ms"], round(d["roofline"]["frac"], 3), d["under_load"].get("sclk_mhz"), d["under_load"].get("power_w"))
    except Exception as e:
        print(f, "failed", e)
for l in open("$O/slow_configs.jsonl"):
    r = json.loads(l); print(r["kernel"], round(r["ms"], 3), round(r["frac_of_8TBs"], 3))
print(open("$O/slow_resample.jsonl").read())
PY
fi
