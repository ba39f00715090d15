#!/usr/bin/env python3
"""KalmanFilter.batch_filter with EVERYTHING at once -- Fs / Qs / Hs / Rs / Bs lists, us, update_first, missing
measurements -- and rts_smoother with Fs / Qs lists, at the state sizes the several-lanes-per-track kernels serve
((9,3), (12,2), (16,4)), from the LIVE reference -> tests/golden/kf_combo.npz.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_kf_combo_golden.py

The single-option goldens (kf_dims: uf / ctrl / miss; kf_models: per-step lists) pin each argument of
kalman_filter.py:826-993 on its own; this file pins their combination, which is what the VAR instantiations of
kf_ml.hip / kf_mlg.hip run as one kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("FILTERPY_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_goldens import make_kf, spd, stable_F  # noqa: E402

CASES = [(9, 3, 2), (12, 2, 3), (16, 4, 1)]
MISSING = (2, 7, 8)
T = 14


def main():
    d = {"cases": np.array(CASES), "missing": np.array(MISSING)}
    for (n, m, nu) in CASES:
        rs = np.random.RandomState(1234 + 7 * n + m)
        Fs = [stable_F(rs, n) for _ in range(T)]
        Qs = [spd(rs, n, 0.05) for _ in range(T)]
        Hs = [rs.randn(m, n) for _ in range(T)]
        Rs = [spd(rs, m, 0.5) for _ in range(T)]
        Bs = [rs.randn(n, nu) for _ in range(T)]
        us = [rs.randn(nu) for _ in range(T)]
        x0, P0 = rs.randn(n), spd(rs, n, 3.0)
        zs = [None if t in MISSING else rs.randn(m) * 2 for t in range(T)]
        p = f"n{n}m{m}_"
        d.update({p + "Fs": np.array(Fs), p + "Qs": np.array(Qs), p + "Hs": np.array(Hs), p + "Rs": np.array(Rs),
                  p + "Bs": np.array(Bs), p + "us": np.array(us), p + "x0": x0, p + "P0": P0,
                  p + "zs": np.array([np.full(m, np.nan) if z is None else z for z in zs])})
        # column-vector state so that the zs list may hold None for any m, and an object array because np.size(zs, 0) on
        # a ragged list fails under NumPy >= 1.24 (the recipe of make_goldens.py's `miss` variant)
        zl = np.empty(T, dtype=object)
        for t in range(T):
            zl[t] = None if zs[t] is None else zs[t].reshape(m, 1)
        ul = [u.reshape(nu, 1) for u in us]
        for uf in (False, True):
            kf = make_kf(n, m, x0.reshape(n, 1), P0, Fs[0], Qs[0], Hs[0], Rs[0], dim_u=nu, B=Bs[0])
            mu, cov, mup, covp = kf.batch_filter(zl, Fs=Fs, Qs=Qs, Hs=Hs, Rs=Rs, Bs=Bs, us=ul, update_first=uf)
            mu, mup = mu[..., 0], mup[..., 0]
            q = p + ("uf_" if uf else "pu_")
            d.update({q + "mu": mu, q + "cov": cov, q + "mup": mup, q + "covp": covp, q + "xfinal": kf.x[:, 0].copy(), q + "Pfinal": kf.P.copy()})
            if not uf:
                xs, Ps, Ks, Pps = kf.rts_smoother(mu, cov, Fs=Fs, Qs=Qs)
                d.update({p + "rts_x": xs, p + "rts_P": Ps, p + "rts_K": Ks, p + "rts_Pp": Pps})
    np.savez_compressed(os.path.join(OUT, "kf_combo.npz"), **d)
    print("wrote kf_combo.npz", len(d), "arrays")


if __name__ == "__main__":
    main()
