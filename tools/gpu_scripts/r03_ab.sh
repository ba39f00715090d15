#!/bin/bash
# Round 3, lease ab: IMM with one logarithm per filter and reciprocal-multiplies -- the IMM / MMAE GPU tests, then the IMM rows of
# tools/bench_configs.py (configs 8), both layouts.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ab
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_imm.py tests/test_gpu_api.py tests/test_gpu_edges.py -m gpu -q -p no:cacheprovider -k "imm or IMM or mmae or MMAE" > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -3 $O/pytest_imm.log
timeout 300 python tools/bench_configs.py --configs 8 --layouts soa,aos > $O/imm.jsonl 2> $O/imm.err; echo "imm rc=$?"
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03ab/imm.jsonl")):
    r = json.loads(l)
    print("%-44s %7.3f ms  frac %.3f  par %s" % (r["kernel"], r["ms"], r["frac_of_8TBs"], r.get("parity_max_rel")))
PY
