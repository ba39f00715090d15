"""(7,1) NumPy order with a mask: where do kf_fast's extras instantiation and the generic kernel part?"""
import os
import sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_kf import _run_ex


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


n, m = int(os.environ.get("NX", 7)), int(os.environ.get("NZ", 1))
rs = np.random.RandomState(100 * n + m)
N, T = 333, 6
A = rs.randn(N, n, n)
x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
zs = rs.randn(T, N, m) * 2
F = np.eye(n) + 0.05 * rs.randn(n, n)
B = rs.randn(n, n)
Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
H = rs.randn(m, n)
R = 0.5 * np.eye(m)
masks = {"all ones": np.ones((T, N), dtype=bool), "random": rs.rand(T, N) > 0.25, "one track missing at t=2": np.ones((T, N), dtype=bool)}
masks["one track missing at t=2"][2, 5] = False
masks["random"][0] = True
ALL = ("y", "K", "S", "SI", "log_likelihood", "mahalanobis")
for mname, mask in masks.items():
    for keys in (ALL, ("y",), ("K",), ("S",), ("log_likelihood",)):
        fast, hf = _run_ex(x0, P0, zs, F, Q, H, R, "aos", mask=mask, keys=keys)
        os.environ["FK_NO_FAST_EX"] = "1"
        gen, hg = _run_ex(x0, P0, zs, F, Q, H, R, "aos", mask=mask, keys=keys)
        del os.environ["FK_NO_FAST_EX"]
        names = ("means", "covs", "means_p", "covs_p")
        line = [f"{nm} {rel(fast[i], gen[i]):.1e}" for i, nm in enumerate(names)] + [f"{k} {rel(hf[k], hg[k]):.1e}" for k in keys]
        print(mname, "| extras", ",".join(keys), "|", "  ".join(line), flush=True)
        d = np.abs(fast[2] - gen[2]).reshape(T, N, -1).max(axis=2)       # prior means
        bad = np.argwhere(d > 1e-9)
        if len(bad):
            ts, tr = bad[:, 0], bad[:, 1]
            print("    prior means differ: first t", ts.min(), "tracks", sorted(set(tr.tolist()))[:12], "count", len(bad), "of", T * N)
        d = np.abs(fast[0] - gen[0]).reshape(T, N, -1).max(axis=2)
        bad = np.argwhere(d > 1e-9)
        if len(bad):
            ts, tr = bad[:, 0], bad[:, 1]
            print("    posterior means differ: first t", ts.min(), "tracks", sorted(set(tr.tolist()))[:12], "count", len(bad), "of", T * N)
