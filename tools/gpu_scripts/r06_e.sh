#!/bin/bash
# Round 6, fifth lease: workgroups per CU of the one-pass v2 kernel (FK_OP_WAVES = 5 / 6 / 7; the slow path is a call now, the
# fast path needs 70 VGPRs): correctness of every instantiation, then timing and phase clocks.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06e
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x -k "onepass or c5" > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH=125x8000000,8x8000000,1x8000000,1000x100000,32x1000000
for env in "FK_OP_WAVES=5" "FK_OP_WAVES=6" "FK_OP_WAVES=7" "FK_OP_V2=0" "FK_OP_WAVES=7 FK_OP_POLLS=128" "FK_OP_WAVES=6" "FK_OP_WAVES=7" "FK_OP_WAVES=5"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-80
for env in "FK_OP_WAVES=5" "FK_OP_WAVES=6" "FK_OP_WAVES=7"; do
  env $env timeout 300 python tools/op_phase.py --run --shapes 125x8000000,8x8000000,1x8000000 --iters 3 >> $O/op_phase.jsonl 2>> $O/op_phase.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r06e/op_phase.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["shape"], d.get("env"), d["ms_per_call"], {k: int(v) for k, v in d["ticks_per_workgroup"].items()}, d["counts_per_call"], d.get("predicted_chunks_per_call"), d.get("v2_slow_chunks_per_call"))
PY
