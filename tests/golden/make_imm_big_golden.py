#!/usr/bin/env python3
"""Goldens for banks the round-2 IMM kernel refused (VERDICT r2 missing 3): up to eight filters, dim_x up to 9, dim_z up to 4,
from the LIVE reference (filterpy/kalman/IMM.py:160-249, mmae.py:140-212).  Same layout as imm.npz / mmae.npz.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_imm_big_golden.py
"""
import os
import sys

import numpy as np

REF = os.environ.get("FILTERPY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from filterpy.kalman import KalmanFilter, IMMEstimator, MMAEFilterBank  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
IMM_CASES = [(9, 3, 4), (4, 2, 8), (7, 4, 5), (6, 3, 6), (9, 4, 2), (2, 1, 7), (8, 2, 3)]
MMAE_CASES = [(9, 3, 4), (4, 2, 8), (7, 4, 3)]
T = 20
OUT_NAME = "imm_big.npz"          # (make_imm_banks16_golden.py runs this file's main() with its own cases and name)


def spd(rs, n, scale=1.0):
    A = rs.randn(n, n)
    return scale * (A @ A.T / n + 0.5 * np.eye(n))


def stable_F(rs, n):
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    return F / max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))


def make_kf(n, m, x0, P0, F, Q, H, R):
    kf = KalmanFilter(dim_x=n, dim_z=m)
    kf.x, kf.P, kf.F, kf.Q, kf.H, kf.R = x0.copy(), P0.copy(), F.copy(), Q.copy(), H.copy(), R.copy()
    return kf


def model(rs, n, m, nm):
    Fs = [stable_F(rs, n) for _ in range(nm)]
    Qs = [spd(rs, n, 0.05 * (j + 1)) for j in range(nm)]
    H = rs.randn(m, n)
    Hs = [H.copy() for _ in range(nm)]
    Rs = [spd(rs, m, 0.5) for _ in range(nm)]
    xs0 = [rs.randn(n) for _ in range(nm)]
    Ps0 = [spd(rs, n, 3.0) for _ in range(nm)]
    return Fs, Qs, Hs, Rs, xs0, Ps0


def main():
    d = {"imm_cases": np.array(IMM_CASES), "mmae_cases": np.array(MMAE_CASES)}
    for (n, m, nm) in IMM_CASES:
        rs = np.random.RandomState(18000 + 11 * n + 3 * m + nm)
        Fs, Qs, Hs, Rs, xs0, Ps0 = model(rs, n, m, nm)
        mu0 = rs.rand(nm) + 0.2
        mu0 /= mu0.sum()
        Mt = rs.rand(nm, nm) + np.eye(nm) * 3
        Mt /= Mt.sum(axis=1, keepdims=True)
        zs = rs.randn(T, m) * 2
        filters = [make_kf(n, m, xs0[j], Ps0[j], Fs[j], Qs[j], Hs[j], Rs[j]) for j in range(nm)]
        imm = IMMEstimator(filters, mu0.copy(), Mt)
        X, P, MU, XP, PP, L = [], [], [], [], [], []
        for t in range(T):
            imm.predict()
            XP.append(imm.x.copy()); PP.append(imm.P.copy())
            imm.update(zs[t])
            X.append(imm.x.copy()); P.append(imm.P.copy()); MU.append(imm.mu.copy()); L.append(imm.likelihood.copy())
        p = f"imm_n{n}m{m}k{nm}_"
        d.update({p + "Fs": np.array(Fs), p + "Qs": np.array(Qs), p + "Hs": np.array(Hs), p + "Rs": np.array(Rs),
                  p + "xs0": np.array(xs0), p + "Ps0": np.array(Ps0), p + "mu0": mu0, p + "M": Mt, p + "zs": zs,
                  p + "x": np.array(X), p + "P": np.array(P), p + "mu": np.array(MU), p + "xp": np.array(XP),
                  p + "Pp": np.array(PP), p + "L": np.array(L),
                  p + "xs_final": np.array([f.x for f in filters]), p + "Ps_final": np.array([f.P for f in filters])})
    for (n, m, nm) in MMAE_CASES:
        rs = np.random.RandomState(19000 + 11 * n + 3 * m + nm)
        Fs, Qs, Hs, Rs, xs0, Ps0 = model(rs, n, m, nm)
        p0 = rs.rand(nm) + 0.2
        p0 /= p0.sum()
        zs = rs.randn(T, m) * 2
        filters = [make_kf(n, m, xs0[j], Ps0[j], Fs[j], Qs[j], Hs[j], Rs[j]) for j in range(nm)]
        bank = MMAEFilterBank(filters, p0.copy(), dim_x=n)
        X, P, PR, L = [], [], [], []
        for t in range(T):
            bank.predict()
            bank.update(zs[t])
            X.append(bank.x.copy()); P.append(bank.P.copy()); PR.append(bank.p.copy())
            L.append(np.array([f.likelihood for f in filters]))
        q = f"mmae_n{n}m{m}k{nm}_"
        d.update({q + "Fs": np.array(Fs), q + "Qs": np.array(Qs), q + "Hs": np.array(Hs), q + "Rs": np.array(Rs),
                  q + "xs0": np.array(xs0), q + "Ps0": np.array(Ps0), q + "p0": p0, q + "zs": zs,
                  q + "x": np.array(X), q + "P": np.array(P), q + "p": np.array(PR), q + "L": np.array(L),
                  q + "xs_final": np.array([f.x for f in filters]), q + "Ps_final": np.array([f.P for f in filters])})
    np.savez_compressed(os.path.join(OUT, OUT_NAME), **d)
    print("wrote " + OUT_NAME + ":", len(d), "arrays")


if __name__ == "__main__":
    main()
