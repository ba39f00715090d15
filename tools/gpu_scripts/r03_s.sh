#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
timeout 600 python tools/exp_regions.py > $O/regions.jsonl 2> $O/regions.err; tail -2 $O/regions.err | cut -c1-300
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03s/regions.jsonl")):
    r = json.loads(l)
    if "ms_by_covs_p_at_GiB" in r:
        print("%4d | " % r["covs_at_GiB"] + " ".join("%s:%.2f" % (k, v) for k, v in r["ms_by_covs_p_at_GiB"].items()))
    else:
        print(json.dumps(r)[:1500])
PY
