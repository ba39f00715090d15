#!/bin/bash
# Round 6, lease t: streamed host outputs with two buffer sets (the pipeline never drains); store width of element-major planes
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06t
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 tools/experiments/soa_store_width 1000000 20 40 > $O/soa_store_width.jsonl 2>&1; cat $O/soa_store_width.jsonl
timeout 300 tools/experiments/soa_store_width 1000000 20 20 >> $O/soa_store_width.jsonl 2>&1
timeout 300 tools/experiments/soa_store_width 200000 50 40 >> $O/soa_store_width.jsonl 2>&1; tail -24 $O/soa_store_width.jsonl | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -x -k "streamed or pipelined" > $O/pytest_api.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_api.log
timeout 1200 python tools/bench_api.py > $O/bench_api.jsonl 2> $O/bench_api.err
FK_STREAM_OUTPUTS=0 timeout 1200 python tools/bench_api.py > $O/bench_api_unstreamed.jsonl 2> $O/bench_api_unstreamed.err
python - <<'PY'
import json
for f in ("bench_api.jsonl", "bench_api_unstreamed.jsonl"):
    print(f)
    for l in open("gpurun_out/r06t/" + f):
        x = json.loads(l)
        print(" ", x["config"], {k: round(x[k], 3) for k in ("sum_of_pieces_s", "api_host_outputs_s", "host_output_call_over_pieces", "d2h_GBs", "api_device_outputs_s")})
PY
