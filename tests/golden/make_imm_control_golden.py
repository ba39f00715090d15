#!/usr/bin/env python3
"""IMMEstimator.predict(u) / MMAEFilterBank.predict(u) from the LIVE reference -> tests/golden/imm_control.npz
(every filter's predict(u): x = F x + B u with its own B, kalman_filter.py:472-475; IMM.py:214-216, mmae.py:153-154).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_imm_control_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("FILTERPY_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
from filterpy.kalman import IMMEstimator, MMAEFilterBank  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_goldens import make_kf, spd, stable_F  # noqa: E402

CASES = [(2, 1, 2, 1), (4, 2, 3, 2), (6, 3, 2, 3)]          # dim_x, dim_z, filters, dim_u


def main():
    d = {"cases": np.array(CASES)}
    T = 12
    for (n, m, nm, nu) in CASES:
        rs = np.random.RandomState(8900 + 11 * n + 3 * m + nm)
        Fs = [stable_F(rs, n) for _ in range(nm)]
        Qs = [spd(rs, n, 0.05 * (j + 1)) for j in range(nm)]
        H = rs.randn(m, n)
        Rs = [spd(rs, m, 0.5) for _ in range(nm)]
        Bs = [rs.randn(n, nu) for _ in range(nm)]
        xs0 = [rs.randn(n) for _ in range(nm)]
        Ps0 = [spd(rs, n, 3.0) for _ in range(nm)]
        mu0 = rs.rand(nm) + 0.2
        Mt = rs.rand(nm, nm) + np.eye(nm) * 3
        Mt /= Mt.sum(axis=1, keepdims=True)
        zs, us = rs.randn(T, m) * 2, rs.randn(T, nu)
        p = f"n{n}m{m}k{nm}_"
        d.update({p + "Fs": np.array(Fs), p + "Qs": np.array(Qs), p + "H": H, p + "Rs": np.array(Rs), p + "Bs": np.array(Bs),
                  p + "xs0": np.array(xs0), p + "Ps0": np.array(Ps0), p + "mu0": mu0, p + "M": Mt, p + "zs": zs, p + "us": us})
        for kind in ("imm", "mmae"):
            filters = []
            for j in range(nm):
                f = make_kf(n, m, xs0[j], Ps0[j], Fs[j], Qs[j], H.copy(), Rs[j], dim_u=nu, B=Bs[j])
                filters.append(f)
            est = IMMEstimator(filters, mu0, Mt) if kind == "imm" else MMAEFilterBank(filters, list(mu0 / mu0.sum()), dim_x=n, H=H)
            X, P, MU = [], [], []
            for t in range(T):
                est.predict(us[t])
                est.update(zs[t])
                X.append(np.array(est.x, dtype=float).reshape(n).copy())
                P.append(np.array(est.P, dtype=float).copy())
                MU.append(np.array(est.mu if kind == "imm" else est.p, dtype=float).copy())
            q = p + kind + "_"
            d.update({q + "x": np.array(X), q + "P": np.array(P), q + "mu": np.array(MU)})
    np.savez_compressed(os.path.join(OUT, "imm_control.npz"), **d)
    print("wrote imm_control.npz", len(d), "arrays")


if __name__ == "__main__":
    main()
