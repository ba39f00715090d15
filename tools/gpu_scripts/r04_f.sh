#!/bin/bash
# Round 4, lease f: lease e again without the crash (the 4 GiB bank test now runs last, in its own process, no core dumps) +
# IMM with z carried across steps + the one-pass kernel's phase clocks on a few long vectors.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
cd $R
df -h /tmp $R | tail -2
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu_full.log
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
timeout 300 $BENCH --no-cpu --placement interleave > $O/bench_interleave.json 2>/dev/null
timeout 300 $BENCH --no-cpu --placement none > $O/bench_none.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04f"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = dict(d["placement"]); pr = p.pop("probe", {}); pr.pop("grid_ms", None)
        print(os.path.basename(f), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "ms_per_step", round(d["ms_per_step"], 4),
              "frac", round(d["roofline"]["frac"], 4), {k: (v if not isinstance(v, str) else v[:20]) for k, v in p.items()}, pr)
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
timeout 600 python tools/bench_configs.py --configs rs8 --layouts soa,aos > $O/configs_imm.jsonl 2> $O/configs_imm.err; echo "imm rc=$?"; python - <<'PY'
import json, os
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04f/configs_imm.jsonl"):
    d = json.loads(l); print(d["kernel"], "ms %.3f frac %.3f parity %.1e" % (d["ms"], d["frac_of_8TBs"], d.get("parity_max_rel") or 0))
PY
timeout 300 python tools/op_phase.py --run --shapes 1x8000000,8x8000000,125x8000000 --iters 5 > $O/onepass_phase_clocks.jsonl 2> $O/op_phase.err; cut -c1-900 $O/onepass_phase_clocks.jsonl
FK_TEST_BIG_BANK=1 timeout 600 python -m pytest tests/test_gpu_edges.py -m gpu -q -x -k "4_gib" -p no:cacheprovider > $O/pytest_big_bank.log 2>&1; echo "big bank rc=$?"; head -8 $O/pytest_big_bank.log | cut -c1-300; tail -3 $O/pytest_big_bank.log | cut -c1-300
