#!/bin/bash
# Round 6, lease zc: imm_lanes (9,4): streamed predict / update against the register copies of the model; the small classes on lanes
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zc
mkdir -p $O
export TMPDIR=/tmp
cd $R
for rep in 1 2; do
for lib in "" filterpy_amd/csrc/exp_build/libfilterhip_il_p0.so filterpy_amd/csrc/exp_build/libfilterhip_il_u0.so; do
  echo "== lib=$lib" >> $O/il_ab.txt
  FK_LIB=$lib timeout 600 python tools/bench_configs.py --configs r --layouts soa >> $O/il_ab.txt 2>> $O/il.err
done
done
for mode in 1 2; do
  echo "== FK_IMM_LANES=$mode" >> $O/il_small.txt
  FK_IMM_LANES=$mode timeout 600 python tools/bench_configs.py --configs 8 --layouts soa,aos >> $O/il_small.txt 2>> $O/il.err
done
python - <<'PY'
import json
for f in ("il_ab.txt", "il_small.txt"):
    cur = None
    for l in open("gpurun_out/r06zc/" + f):
        if l.startswith("=="):
            cur = l.strip()
            continue
        if l.startswith("{"):
            d = json.loads(l)
            if "(16,8)" in d["kernel"]: continue
            print(cur, d["kernel"], round(d["ms"], 3), d.get("parity_max_rel"))
PY
tail -3 $O/il.err
