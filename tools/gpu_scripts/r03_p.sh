#!/bin/bash
# lease: which kernel is off at dim_x >= 7 in NumPy order with extras; does the headline's time follow the relative
# placement of its arrays (one slab, views at chosen offsets)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
timeout 300 python tools/debug/ex_aos_dims.py > $O/ex_aos_dims.log 2>&1; cut -c1-420 $O/ex_aos_dims.log | tail -40
timeout 600 python tools/exp_placement.py > $O/placement.jsonl 2> $O/placement.err; tail -3 $O/placement.err; python - <<'PY'
import json, os
rows = [json.loads(l) for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03p/placement.jsonl")) if l.startswith("{")]
for r in rows:
    if "ms" in r:
        print(r["label"], r.get("deltas", ""), r["ms"])
PY
timeout 200 python tools/exp_placement.py --random 8 > $O/placement_second_process.jsonl 2>/dev/null; grep -c ms $O/placement_second_process.jsonl
