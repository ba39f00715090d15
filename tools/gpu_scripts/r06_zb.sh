#!/bin/bash
# Round 6, lease z: the one-lane-per-filter IMM kernel (imm_lanes.hip): the IMM suite, then the (9,4) class with and without it
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zb
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_imm.py -m gpu -q -p no:cacheprovider -x > $O/pytest_imm.log 2>&1; echo "pytest imm rc=$?"; tail -15 $O/pytest_imm.log
for mode in 1; do
  echo "== FK_IMM_LANES=$mode" >> $O/imm_ab.txt
  FK_IMM_LANES=$mode timeout 600 python tools/bench_configs.py --configs r --layouts soa,aos >> $O/imm_ab.txt 2>> $O/imm.err
done
python - <<'PY'
import json
cur = None
for l in open("gpurun_out/r06zb/imm_ab.txt"):
    if l.startswith("=="):
        cur = l.strip()
        continue
    if l.startswith("{"):
        d = json.loads(l)
        print(cur, d["kernel"], round(d["ms"], 3), d.get("parity_max_rel"), d.get("mu_max_abs"))
PY
tail -5 $O/imm.err
