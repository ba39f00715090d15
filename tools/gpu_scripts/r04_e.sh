#!/bin/bash
# Round 4, lease e: full GPU suite on the third batch (IMM (16,8), steady / correlated (16,8), track windows past 4 GiB,
# Bank placement="probe"), bench.py with its three arrangements, the rolled IMM classes and the big steady-state class timed.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu_full.log
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
timeout 300 $BENCH --no-cpu --placement interleave > $O/bench_interleave.json 2>/dev/null
timeout 300 $BENCH --no-cpu --placement none > $O/bench_none.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04e"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = dict(d["placement"]); pr = p.pop("probe", {}); pr.pop("grid_ms", None)
        print(os.path.basename(f), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "ms_per_step", round(d["ms_per_step"], 4),
              "frac", round(d["roofline"]["frac"], 4), {k: (v if not isinstance(v, str) else v[:20]) for k, v in p.items()}, pr)
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
timeout 600 python tools/bench_configs.py --configs rs --layouts soa,aos > $O/configs_rolled.jsonl 2> $O/configs_rolled.err; echo "rolled rc=$?"; cut -c1-230 $O/configs_rolled.jsonl
