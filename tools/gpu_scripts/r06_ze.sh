#!/bin/bash
# Round 6, lease ze: the (9) smoother with F x and x + K dx distributed over the quad against the build before; imm_lanes (9,4) with
# both register copies of the model (p0u0) against the streamed build and the streamed-predict-only build (u0)
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06ze
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_kf.py tests/test_gpu_baseline_configs.py -m gpu -q -p no:cacheprovider -x > $O/pytest_kf.log 2>&1; echo "pytest kf rc=$?"; tail -3 $O/pytest_kf.log
for i in 1 2 3; do
for lib in filterpy_amd/csrc/exp_build/libfilterhip_oldrts.so ""; do
  echo "== lib=$lib" >> $O/c3_ab.txt
  FK_LIB=$lib timeout 300 python tools/bench_configs.py --configs 3 --layouts soa,aos >> $O/c3_ab.txt 2>> $O/c3.err
done
done
for rep in 1 2; do
for lib in "" filterpy_amd/csrc/exp_build/libfilterhip_il_p0u0.so filterpy_amd/csrc/exp_build/libfilterhip_il_u0.so; do
  echo "== lib=$lib" >> $O/il_ab.txt
  FK_LIB=$lib timeout 600 python tools/bench_configs.py --configs r --layouts soa >> $O/il_ab.txt 2>> $O/il.err
done
done
python - <<'PY'
import json
for f, key in (("c3_ab.txt", "rts"), ("il_ab.txt", "(9,")):
    cur = None
    for l in open("gpurun_out/r06ze/" + f):
        if l.startswith("=="):
            cur = l.strip().replace("filterpy_amd/csrc/exp_build/libfilterhip_", "")
            continue
        if l.startswith("{"):
            d = json.loads(l)
            if key in d["kernel"]:
                print(cur, d["kernel"], round(d["ms"], 4))
PY
