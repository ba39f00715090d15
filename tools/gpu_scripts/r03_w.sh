#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03w
mkdir -p $O
cd $R
timeout 300 python tools/exp_regions_rs.py > $O/regions_rs.jsonl 2> $O/regions_rs.err; tail -2 $O/regions_rs.err | cut -c1-200
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03w/regions_rs.jsonl")):
    r = json.loads(l)
    if "ms_by_idx_at_GiB" in r:
        print("%4d | " % r["w_at_GiB"] + " ".join("%s:%.2f" % (k, v) for k, v in r["ms_by_idx_at_GiB"].items()))
    else:
        print(r)
PY
