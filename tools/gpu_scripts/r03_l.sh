#!/bin/bash
# lease: the whole -m gpu suite after the round's changes, smoke, and the bench line with clocks sampled under load
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03l}
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
FK_PARITY_LOG=$O/parity_errors.jsonl timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("OUTDIR", "r03l"), "bench_default.json")))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], round(d["roofline"]["frac"], 4))
print(d["hbm_probes"], d["gpu_clocks"].get("asic_serial"), d["gpu_clocks"].get("oam_id"))
print(d["under_load"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
