#!/usr/bin/env python3
"""IMMEstimator / MMAEFilterBank with MISSING measurements, from the LIVE reference -> tests/golden/imm_missing.npz.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_imm_missing_golden.py

What the reference does with update(None) (IMM.py:171-186, mmae.py:160-212, kalman_filter.py:511-520, :1203-1226):
every filter's update(None) leaves x, P alone but sets y = 0 and clears its cached likelihood, so `f.likelihood` is
re-evaluated as the density of a ZERO residual under the S of that filter's last real update (S = 0 before any: the
density evaluates to 0 and is floored at float_info.min) -- and the mode probabilities ARE re-weighted
with those numbers, mixed and re-estimated.  Frozen here as data, including a missing FIRST measurement."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("FILTERPY_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
from filterpy.kalman import IMMEstimator, MMAEFilterBank  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_goldens import make_kf, spd, stable_F  # noqa: E402

CASES = [(2, 1, 2), (4, 2, 3), (6, 3, 2), (3, 2, 2)]
MISSING = (0, 3, 4, 11, 19)


def main():
    d = {"cases": np.array(CASES), "missing": np.array(MISSING)}
    T = 20
    for (n, m, nm) in CASES:
        rs = np.random.RandomState(8800 + 11 * n + 3 * m + nm)
        Fs = [stable_F(rs, n) for _ in range(nm)]
        Qs = [spd(rs, n, 0.05 * (j + 1)) for j in range(nm)]
        H = rs.randn(m, n)
        Rs = [spd(rs, m, 0.5) for _ in range(nm)]
        xs0 = [rs.randn(n) for _ in range(nm)]
        Ps0 = [spd(rs, n, 3.0) for _ in range(nm)]
        mu0 = rs.rand(nm) + 0.2
        Mt = rs.rand(nm, nm) + np.eye(nm) * 3
        Mt /= Mt.sum(axis=1, keepdims=True)
        zs = rs.randn(T, m) * 2
        p = f"n{n}m{m}k{nm}_"
        d.update({p + "Fs": np.array(Fs), p + "Qs": np.array(Qs), p + "H": H, p + "Rs": np.array(Rs),
                  p + "xs0": np.array(xs0), p + "Ps0": np.array(Ps0), p + "mu0": mu0, p + "M": Mt, p + "zs": zs})
        for kind in ("imm", "mmae"):
            filters = [make_kf(n, m, xs0[j], Ps0[j], Fs[j], Qs[j], H.copy(), Rs[j]) for j in range(nm)]
            if kind == "imm":
                est = IMMEstimator(filters, mu0, Mt)
            else:
                est = MMAEFilterBank(filters, list(mu0 / mu0.sum()), dim_x=n, H=H)
            X, P, MU, L = [], [], [], []
            for t in range(T):
                est.predict()
                est.update(None if t in MISSING else zs[t])
                X.append(np.array(est.x, dtype=float).reshape(n).copy())
                P.append(np.array(est.P, dtype=float).copy())
                MU.append(np.array(est.mu if kind == "imm" else est.p, dtype=float).copy())
                if kind == "imm":
                    L.append(est.likelihood.copy())
            q = p + kind + "_"
            d.update({q + "x": np.array(X), q + "P": np.array(P), q + "mu": np.array(MU),
                      q + "xs_final": np.array([f.x for f in filters]), q + "Ps_final": np.array([f.P for f in filters])})
            if kind == "imm":
                d[q + "L"] = np.array(L)
    np.savez_compressed(os.path.join(OUT, "imm_missing.npz"), **d)
    print("wrote imm_missing.npz", len(d), "arrays")


if __name__ == "__main__":
    main()
