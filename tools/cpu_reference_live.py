#!/usr/bin/env python3
"""The LIVE reference (rlabbe/filterpy imported from /root/reference: build container only) and the in-repo NumPy port
(oracle/kf_oracle.py: what bench.py's `cpu_baseline` times on the GPU box, where the reference does not exist) on the SAME
host cores, same inputs, same run -- VERDICT r4 missing 5: `cpu_baseline.kind = "port"` is honest only if the port's speed
relative to the reference is on file.

    python tools/cpu_reference_live.py [--seconds 6] > profiles/r05/cpu_reference_live.json

Workload: BASELINE configs[1]'s per-track work -- KalmanFilter.batch_filter, dim_x = 4, dim_z = 2, 100 steps per track --
single-threaded processes, 1 and all host cores; also configs[0] (2,1) x 1000 steps and configs[2]'s (9,3)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"


def model(n, m):
    if (n, m) == (4, 2):
        from bench import c2_model
        return c2_model()
    rs = np.random.RandomState(10 * n + m)
    F = np.eye(n) + np.diag(np.full(n - 1, 0.1), 1)
    return F, 0.01 * np.eye(n), np.eye(m, n), 0.5 * np.eye(m)


def worker(args):
    which, n, m, T, seconds, seed = args
    rs = np.random.RandomState(seed)
    F, Q, H, R = model(n, m)
    zs = rs.randn(T, m)
    zl = list(zs)
    if which == "reference":
        sys.path.insert(0, "/root/reference")
        from filterpy.kalman import KalmanFilter

        def run():
            kf = KalmanFilter(dim_x=n, dim_z=m)
            kf.x, kf.P, kf.F, kf.Q, kf.H, kf.R = np.zeros(n), 100.0 * np.eye(n), F, Q, H, R
            return kf.batch_filter(zl)
    else:
        from oracle import kf_oracle

        def run():
            return kf_oracle.kf_batch_filter(np.zeros(n), 100.0 * np.eye(n), zl, F, Q, H, R)
    run()
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run()
        done += 1
    return done * T, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    a = ap.parse_args()
    cores = len(os.sched_getaffinity(0))
    out = {"host_cpus": cores, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
           "numpy": np.__version__, "reference": "rlabbe/filterpy v1.4.5 imported from /root/reference (unmodified)",
           "port": "oracle/kf_oracle.py::kf_batch_filter (the function bench.py's cpu_baseline times)", "rows": []}
    ctx = mp.get_context("spawn")
    for (n, m, T) in ((4, 2, 100), (2, 1, 1000), (9, 3, 100)):
        for procs in (1, cores):
            row = {"dim_x": n, "dim_z": m, "T": T, "procs": procs}
            for which in ("reference", "port"):
                with ctx.Pool(procs) as pool:
                    t0 = time.perf_counter()
                    res = pool.map(worker, [(which, n, m, T, a.seconds, 100 + i) for i in range(procs)])
                    wall = time.perf_counter() - t0
                row[which + "_track_steps_per_s"] = sum(r[0] for r in res) / max(r[1] for r in res)
            row["port_over_reference"] = row["port_track_steps_per_s"] / row["reference_track_steps_per_s"]
            out["rows"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
