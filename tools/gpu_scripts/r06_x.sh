#!/bin/bash
# Round 6, lease x: the three-lane (9,3) filter with x' = F x distributed over the quad (27 instead of 81 fma per lane) against the build before
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06x
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_kf.py tests/test_gpu_baseline_configs.py -m gpu -q -p no:cacheprovider -x > $O/pytest_kf.log 2>&1; echo "pytest kf rc=$?"; tail -3 $O/pytest_kf.log
for i in 1 2 3; do
for lib in filterpy_amd/csrc/exp_build/libfilterhip_oldml.so ""; do
  echo "== lib=$lib" >> $O/c3_ab.txt
  FK_LIB=$lib timeout 300 python tools/bench_configs.py --configs 3 --layouts soa,aos >> $O/c3_ab.txt 2>> $O/c3.err
done
done
python - <<'PY'
import json
cur = None
for l in open("gpurun_out/r06x/c3_ab.txt"):
    if l.startswith("=="):
        cur = "old" if "oldml" in l else "new"
        continue
    if l.startswith("{"):
        d = json.loads(l)
        if "batch_filter" in d["kernel"]:
            print(cur, d["kernel"], round(d["ms"], 4))
PY
