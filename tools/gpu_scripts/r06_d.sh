#!/bin/bash
# Round 6, fourth lease: the persistent one-pass kernel (a workgroup walks chunks c, c + G, ... and requests the next chunk's weights
# a chunk ahead) against one workgroup per chunk (FK_OP_PERSIST=0) and round 3's kernel (FK_OP_V2=0): correctness first, then
# timing, phase clocks, the upload / download pipeline of the host-output call.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06d
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
FK_OP_PERSIST=0 timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x -k "onepass or c5" > $O/pytest_resample_nopersist.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_resample_nopersist.log
SH=125x8000000,8x8000000,1x8000000,1000x100000,32x1000000
for env in "FK_OP_V2=1" "FK_OP_PERSIST=0" "FK_OP_V2=0" "FK_OP_GRID=1250" "FK_OP_GRID=1000" "FK_OP_V2=1" "FK_OP_PERSIST=0" "FK_OP_POLLS=128"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-80
for env in "FK_OP_V2=1" "FK_OP_PERSIST=0"; do
  env $env timeout 300 python tools/op_phase.py --run --shapes 125x8000000,8x8000000,1x8000000 --iters 3 >> $O/op_phase.jsonl 2>> $O/op_phase.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r06d/op_phase.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["shape"], d.get("env"), d["ms_per_call"], {k: int(v) for k, v in d["ticks_per_workgroup"].items()}, d["counts_per_call"], d.get("predicted_chunks_per_call"), d.get("v2_slow_chunks_per_call"))
PY
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -x -k "pipelined" > $O/pytest_api.log 2>&1; tail -2 $O/pytest_api.log
timeout 900 python tools/bench_api.py > $O/bench_api.jsonl 2> $O/bench_api.err; cut -c1-420 $O/bench_api.jsonl
