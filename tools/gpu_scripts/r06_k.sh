#!/bin/bash
# Round 6: where do the 4 s go that KalmanFilterBank.batch_filter at (9,3) N = 1e6 (144 GB of histories) spends beyond its pieces?
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06k
mkdir -p $O
cd $R
python - <<'PY' > $O/api_profile_9x3_1e6.txt 2>&1
import cProfile, pstats, time, sys, os
import numpy as np, torch
sys.path.insert(0, ".")
from filterpy_amd.kalman import KalmanFilterBank
n, m, N, T = 9, 3, 1000000, 100
rs = np.random.RandomState(5)
zs = rs.standard_normal((T, N, m))
F = np.eye(n) + np.diag(np.full(n - 3, 0.1), 3)
x0, P0 = np.zeros((N, n)), np.tile(100.0 * np.eye(n), (N, 1, 1))
def bank():
    b = KalmanFilterBank(n, m, N, layout="aos")
    b.x, b.P, b.F, b.Q, b.H, b.R = x0.copy(), P0.copy(), F, 0.01 * np.eye(n), np.eye(m, n), 0.5 * np.eye(m)
    return b
for rep in range(3):
    b = bank()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    r = b.batch_filter(zs)
    torch.cuda.synchronize()
    pr.disable()
    print("rep", rep, "api_host_outputs_s", time.perf_counter() - t0, flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    t1 = time.perf_counter()
    del r
    print("free took", time.perf_counter() - t1, flush=True)
PY
grep -E "^rep|free took|cumtime|batch_filter|to_host|download|from_records|to_records|upload|acquire|isnan|sum|asarray|copy" $O/api_profile_9x3_1e6.txt | cut -c1-160 | head -60
free -g | head -2
