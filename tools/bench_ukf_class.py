#!/usr/bin/env python3
"""UnscentedKalmanFilter (bank) batch_filter + rts_smoother with matrix models at dim_x >= 10: the fused several-lane launch
(csrc/ukf_mlg.hip, the default) against the per-step building blocks the class used before round 5 (FK_UKF_MLG=0: sigma_kernel ->
linear_map -> ut_kernel -> ... resident in HBM, ~8 launches per step).  The library reads FK_UKF_MLG once per process, so each
setting runs in a child interpreter; one JSON line per (dims, path).

    python tools/bench_ukf_class.py [--dims 10x2,12x3,14x4,16x4] [--N 50000] [--T 50]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
from filterpy_amd import _engine as E
n, m, N, T = %(n)d, %(m)d, %(N)d, %(T)d
rs = np.random.RandomState(n * 10 + m)
F = np.eye(n) + 0.05 * rs.randn(n, n)
F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
H, Q, R = rs.randn(m, n), 0.01 * np.eye(n), 0.5 * np.eye(m)
pts = MerweScaledSigmaPoints(n, .5, 2., 3. - n)
zs = list(rs.randn(T, N, m))
def make():
    u = UnscentedKalmanFilter(n, m, dt=1.0, hx=H, fx=F, points=pts, n_tracks=N, layout="aos")
    u.x, u.P, u.Q, u.R = rs.randn(N, n), np.tile(5.0 * np.eye(n), (N, 1, 1)), Q, R
    return u
fused = bool(E.ukf_linear_supported(n, m, True))
out = {"dims": [n, m], "N": N, "T": T, "path": "fused (ukf_mlg)" if fused else "building blocks (FK_UKF_MLG=0)"}
for rep in range(2):                       # the second pass is the measurement (allocator and code warm)
    u = make()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu, cov = u.batch_filter(zs)
    torch.cuda.synchronize(); out["batch_filter_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    xs, Ps, Ks = u.rts_smoother(mu, cov)
    torch.cuda.synchronize(); out["rts_smoother_s"] = time.perf_counter() - t0
out["check"] = [float(mu[-1, N - 1, 0]), float(cov[-1, N - 1, 0, 0]), float(Ps[0, N - 1, 0, 0])]
print(json.dumps(out))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="10x2,12x3,14x4,16x4")
    ap.add_argument("--N", type=int, default=20000)
    ap.add_argument("--T", type=int, default=30)
    a = ap.parse_args()
    for d in a.dims.split(","):
        n, m = (int(v) for v in d.split("x"))
        res = []
        for sw in ("1", "0"):
            env = dict(os.environ, FK_UKF_MLG=sw)
            r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, n=n, m=m, N=a.N, T=a.T)], capture_output=True, text=True, env=env)
            if r.returncode != 0:
                print(json.dumps({"dims": [n, m], "FK_UKF_MLG": sw, "error": r.stderr[-400:]}), flush=True)
                continue
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            res.append(rec)
            print(json.dumps(rec), flush=True)
        if len(res) == 2:
            # host <-> device copies of the (T, N, n, n) histories are in both; what differs is the arithmetic path
            same = max(abs(p - q) / max(1e-300, abs(q)) for p, q in zip(res[0]["check"], res[1]["check"]))
            print(json.dumps({"dims": [n, m], "fused_over_blocks_batch_filter": res[1]["batch_filter_s"] / res[0]["batch_filter_s"],
                              "fused_over_blocks_rts": res[1]["rts_smoother_s"] / res[0]["rts_smoother_s"], "paths_agree_rel": same}), flush=True)


if __name__ == "__main__":
    main()
