#!/bin/bash
# lease: the headline kernel against how its arrays were allocated, with a translation probe; three processes
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
for i in 1 2 3; do timeout 300 python tools/exp_alloc.py > $O/alloc_$i.jsonl 2> $O/alloc_$i.err; tail -2 $O/alloc_$i.err | cut -c1-300; python - $O/alloc_$i.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(r.get("trial", r), r.get("kernel_ms"), r.get("gather_16M_ms"), r.get("stream_read_ms"), (r.get("ptrs") or {}).get("covs"))
PY
done
timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_default.json 2>/dev/null; python -c "
import json; d = json.load(open('$O/bench_default.json')); print('bench', d['roofline']['kernel_ms'])"
