#!/bin/bash
# Round 6, lease l: look-backs that wait for a crossing's carry (FK_OP_LB bit 0) and crossings that add up the sums of the binade
# they enter from (bit 1) -- the chain of a call with few long vectors.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06l
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH=125x8000000,1000x100000,8x8000000,1x8000000,32x1000000,2x4000000,1x2000000
for env in "FK_OP_LB=0" "FK_OP_LB=1" "FK_OP_LB=2" "FK_OP_LB=3" "FK_OP_LB=0" "FK_OP_LB=3"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-110
for lb in 0 3; do FK_OP_LB=$lb timeout 300 python tools/op_phase.py --run --shapes 125x8000000,8x8000000,1x8000000 --iters 3 >> $O/op_phase.jsonl 2>> $O/op_phase.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r06l/op_phase.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["shape"], d.get("env"), d["ms_per_call"], {k: int(v) for k, v in d["ticks_per_workgroup"].items()}, d["counts_per_call"], d.get("v2_slow_chunks_per_call"))
PY
