#!/usr/bin/env python3
"""KalmanFilter with a non-default `inv` (kalman_filter.py:363, 434, 541: `kf.inv = np.linalg.pinv`) and
rts_smoother(inv=np.linalg.pinv) (:995, 1069) on the LIVE reference, where the default inverse cannot be used:

  * update / batch_filter: dim_z measurements through identical rows of H with R = 0 -- S = (h P h') * ones is singular;
  * rts_smoother: F a rank-one projector and Q = 0 -- every predicted covariance Pp is singular.

Freezes the inputs and the reference's outputs (every attribute after every call of the update sequence; the four histories
of batch_filter; x, P, K, Pp of the smoother) for several (dim_x, dim_z), both state shapes.  VERDICT r5 missing 2: the
drop-in used to store `inv` and ignore it.

    PYTHONPATH=/root/reference MPLBACKEND=Agg python tests/golden/make_kf_inv_golden.py
writes tests/golden/kf_inv.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from filterpy.kalman import KalmanFilter  # noqa: E402

CASES = [(2, 2, 2), (4, 2, 2), (4, 3, 1), (6, 3, 2), (9, 4, 2), (12, 3, 1), (16, 8, 2)]     # (dim_x, dim_z, x.ndim)
ATTRS = ("x", "P", "K", "y", "S", "SI")


def spd(rs, k, scale=1.0):
    a = rs.randn(k, k)
    return scale * (a @ a.T / k + 0.5 * np.eye(k))


def main():
    out = {"cases": np.array(CASES)}
    for ci, (n, m, nd) in enumerate(CASES):
        rs = np.random.RandomState(100 + ci)
        F = np.eye(n) + 0.1 * np.triu(rs.randn(n, n), 1)
        Q, P0 = spd(rs, n, 0.05), spd(rs, n, 2.0)
        H = np.tile(rs.randn(n), (m, 1))
        R = np.zeros((m, m))
        T = 8
        zs = rs.randn(T, m, 1) if nd == 2 else rs.randn(T, m)

        def make():
            kf = KalmanFilter(n, m)
            kf.x = np.zeros((n, 1)) if nd == 2 else np.zeros(n)
            kf.P, kf.F, kf.Q, kf.H, kf.R = P0.copy(), F.copy(), Q.copy(), H.copy(), R.copy()
            kf.inv = np.linalg.pinv
            return kf
        pre = f"c{ci}_"
        for k, v in dict(F=F, Q=Q, P0=P0, H=H, R=R, zs=zs).items():
            out[pre + k] = v
        # (a) predict / update, call by call
        kf = make()
        hist = {a: [] for a in ATTRS}
        for z in zs:
            kf.predict()
            kf.update(z)
            for a in ATTRS:
                hist[a].append(np.array(getattr(kf, a), dtype=float))
        for a in ATTRS:
            out[pre + "step_" + a] = np.stack(hist[a])
        # (b) batch_filter, both orders
        for uf in (0, 1):
            kf = make()
            res = kf.batch_filter(list(zs), update_first=bool(uf))
            for name, arr in zip(("means", "covs", "means_p", "covs_p"), res):
                out[pre + f"batch{uf}_" + name] = arr
            out[pre + f"batch{uf}_xf"], out[pre + f"batch{uf}_Pf"] = np.array(kf.x), np.array(kf.P)
        # (c) rts_smoother(inv=pinv) where every Pp is singular
        v = rs.randn(n, 1)
        Fp = v @ v.T / float((v.T @ v).item())
        Xs = rs.randn(T, n, 1) if nd == 2 else rs.randn(T, n)
        Ps = np.stack([spd(rs, n) for _ in range(T)])
        Fs = [Fp * (1.0 + 0.1 * k) for k in range(T)]
        kf = KalmanFilter(n, m)
        kf.F, kf.Q = Fp.copy(), np.zeros((n, n))
        r0 = kf.rts_smoother(Xs, Ps, inv=np.linalg.pinv)
        r1 = kf.rts_smoother(Xs, Ps, Fs=Fs, Qs=[np.zeros((n, n))] * T, inv=np.linalg.pinv)
        out[pre + "rts_F"], out[pre + "rts_Fs"], out[pre + "rts_Xs"], out[pre + "rts_Ps"] = Fp, np.stack(Fs), Xs, Ps
        for tag, r in (("rts0_", r0), ("rts1_", r1)):
            for name, arr in zip(("x", "P", "K", "Pp"), r):
                out[pre + tag + name] = arr
    np.savez_compressed(os.path.join(HERE, "kf_inv.npz"), **out)
    print("wrote kf_inv.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
