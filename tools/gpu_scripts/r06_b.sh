#!/bin/bash
# Round 6, second lease: the GPU suite on the library with the one-pass v2 fix (the prediction word read by ONE wave), the
# bench line, the one-pass A/B with phase clocks and the reasons chunks leave the fast path, the host-output call (pinned
# pipelined D2H) against round 5's.
#   /usr/local/graft/bin/gpurun --timeout 1700 -- 'bash tools/gpu_scripts/r06_b.sh'
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06b
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json; tail -3 $O/bench_default.err
SH=125x8000000,8x8000000,1x8000000,1000x100000,32x1000000
i=0
for env in "FK_OP_V2=0" "FK_OP_V2=1" "FK_OP_V2=1 FK_OP_PRED_BACK=0" "FK_OP_V2=1 FK_OP_POLLS=24" "FK_OP_V2=1 FK_OP_POLLS=128" "FK_OP_V2=0" "FK_OP_V2=1"; do
  i=$((i+1))
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-80
for env in "FK_OP_V2=0" "FK_OP_V2=1" "FK_OP_V2=1 FK_OP_PRED_BACK=0"; do
  env $env timeout 300 python tools/op_phase.py --run --shapes 125x8000000,8x8000000,1x8000000 --iters 3 >> $O/op_phase.jsonl 2>> $O/op_phase.err
done
python - <<'PY'
import json,os
for l in open(os.environ.get("O", "gpurun_out/r06b") + "/op_phase.jsonl") if False else open("gpurun_out/r06b/op_phase.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["shape"], d.get("env"), d["ms_per_call"], {k: int(v) for k, v in d["ticks_per_workgroup"].items()}, d["counts_per_call"], d.get("predicted_chunks_per_call"), d.get("v2_slow_chunks_per_call"))
PY
timeout 900 python tools/bench_api.py > $O/bench_api.jsonl 2> $O/bench_api.err; cut -c1-700 $O/bench_api.jsonl
FK_D2H_PIPE=0 timeout 900 python tools/bench_api.py --N 100000 > $O/bench_api_nopipe.jsonl 2>> $O/bench_api.err; cut -c1-700 $O/bench_api_nopipe.jsonl
