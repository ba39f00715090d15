#!/bin/bash
# Round 6, lease s: streamed host outputs -- bit identity, then the end-to-end call (tools/bench_api.py, a process per measurement)
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -x > $O/pytest_api.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_api.log
timeout 1200 python tools/bench_api.py > $O/bench_api.jsonl 2> $O/bench_api.err; cut -c1-200 $O/bench_api.jsonl
FK_STREAM_OUTPUTS=0 timeout 1200 python tools/bench_api.py > $O/bench_api_unstreamed.jsonl 2> $O/bench_api_unstreamed.err
python - <<'PY'
import json
for f in ("bench_api.jsonl", "bench_api_unstreamed.jsonl"):
    print(f)
    for l in open("gpurun_out/r06s/" + f):
        x = json.loads(l)
        print(" ", x["config"], {k: round(x[k], 3) for k in ("sum_of_pieces_s", "api_host_outputs_s", "host_output_call_over_pieces", "d2h_GBs", "api_device_outputs_s")})
PY
