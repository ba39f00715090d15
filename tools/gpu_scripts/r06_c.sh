#!/bin/bash
# Round 6, third lease: cache-warming prefetch A/B of the one-pass v2 kernel (FK_OP_PF), its correctness under the prefetch,
# and where the host-output call of (9,3) spends 4 s beyond its pieces.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c
mkdir -p $O
export TMPDIR=/tmp
cd $R
SH=125x8000000,8x8000000,1x8000000,1000x100000,32x1000000
for env in "FK_OP_V2=1" "FK_OP_PF=1280" "FK_OP_PF=640" "FK_OP_PF=2560" "FK_OP_PF=1280 FK_OP_POLLS=128" "FK_OP_V2=0" "FK_OP_PF=1280" "FK_OP_V2=1"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-80
FK_OP_PF=1280 timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -x -k "onepass or c5" > $O/pytest_pf.log 2>&1; tail -2 $O/pytest_pf.log
for env in "FK_OP_PF=1280" "FK_OP_V2=1"; do
  env $env timeout 300 python tools/op_phase.py --run --shapes 125x8000000 --iters 3 >> $O/op_phase.jsonl 2>> $O/op_phase.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r06c/op_phase.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["shape"], d.get("env"), d["ms_per_call"], {k: int(v) for k, v in d["ticks_per_workgroup"].items()}, d.get("v2_slow_chunks_per_call"))
PY
cd /tmp
for env in "FK_OP_V2=1" "FK_OP_PF=1280"; do
  tag=$(echo $env | tr '= ' '__')
  env $env timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -- python $R/tools/bench_resample.py --shapes 125x8000000 --iters 5 > /dev/null 2> $O/pmc_$tag.err
  python - <<PY
import csv,glob
for f in glob.glob("$O/pmc_fetch_$tag/**/*counter_collection.csv", recursive=True):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "onepass" in r["Kernel_Name"]]
    print("$tag FETCH_SIZE mean KiB x2 = %.1f MB over %d launches" % (2*sum(v)/max(1,len(v))*1024/1e6, len(v)))
PY
done
cd $R
python - <<'PY' > $O/api_profile.txt 2>&1
import cProfile, pstats, time, sys, os
import numpy as np, torch
sys.path.insert(0, ".")
from filterpy_amd.kalman import KalmanFilterBank
n, m, N, T = 9, 3, 100000, 100
rs = np.random.RandomState(5)
zs = rs.standard_normal((T, N, m))
F = np.eye(n) + np.diag(np.full(n - 3, 0.1), 3)
def bank():
    b = KalmanFilterBank(n, m, N, layout="aos")
    b.x, b.P, b.F, b.Q, b.H, b.R = np.zeros((N, n)), np.tile(100.0 * np.eye(n), (N, 1, 1)), F, 0.01 * np.eye(n), np.eye(m, n), 0.5 * np.eye(m)
    return b
for rep in range(3):
    b = bank()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if rep == 1:
        pr = cProfile.Profile(); pr.enable()
    r = b.batch_filter(zs)
    torch.cuda.synchronize()
    if rep == 1:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
    print("rep", rep, "api_host_outputs_s", time.perf_counter() - t0, flush=True)
    del r
PY
cat $O/api_profile.txt | grep -v "^$" | head -60 | cut -c1-200
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*.db" -delete
