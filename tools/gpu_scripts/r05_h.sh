#!/bin/bash
# Round 5: the new API default (placement by measurement at dim_x <= 4 / large histories) -- its tests, then bench.py as the
# driver runs it, and bench_api.py (what the first and the later calls of one shape cost end to end).
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_api.py tests/test_gpu_baseline_configs.py -m gpu -q -p no:cacheprovider -k "placement or interleav or one_array or device_outputs or bank or c2 or C2 or config" > $O/tests_1.log 2>&1
tail -3 $O/tests_1.log | cut -c1-300
grep -E "^E  " $O/tests_1.log | head -10 | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), d["roofline"]["kernel"], d["config"]["placement"][:50])
print({k: v for k, v in d["placement"].items() if k.endswith("_ms")}, d["placement"]["probe"].get("method"), d["placement"]["probe"].get("chosen_ms"))
print(d["roofline"]["traffic"], d["roofline"]["traffic_source"][:80])
PY
timeout 900 python tools/bench_api.py --N 1000000 > $O/bench_api.jsonl 2> $O/bench_api.err; cut -c1-900 $O/bench_api.jsonl
