#!/usr/bin/env python3
"""Goldens for SCALAR Q / R attributes from the LIVE reference (filterpy 1.4.5) -> tests/golden/kf_scalar_attr.npz.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_scalar_attr_golden.py

What the reference does with them (probed here, frozen as data):
  * update(z) with R = None takes the attribute raw (kalman_filter.py:522-523): S = dot(H, PHT) + r adds r to EVERY
    element of S (:540) and dot(dot(K, r), K.T) is r K K' (:556);
  * predict() with Q = None takes the attribute raw: + q on every element of P (:478);
  * batch_filter() hands the attributes over as kwargs (Qs = [self.Q] * n, Rs = [self.R] * n, :944-947), and a scalar
    KWARG is eye * value (:467-468, :524-525) -- so inside batch_filter the same scalar means q I / r I;
  * module-level update(x, P, z, R, H) uses R as given in both places (:1477, :1497).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("FILTERPY_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
from filterpy.kalman import KalmanFilter  # noqa: E402
import filterpy.kalman.kalman_filter as kfmod  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_goldens import spd, stable_F  # noqa: E402


def main():
    d = {}
    for (n, m) in [(2, 2), (4, 2), (6, 3), (9, 3), (3, 1)]:
        rs = np.random.RandomState(7100 + 13 * n + m)
        F, H, P0, x0 = stable_F(rs, n), rs.randn(m, n), spd(rs, n, 5.0), rs.randn(n)
        q, r = 0.37, 0.81
        T = 12
        zs = rs.randn(T, m)
        p = f"n{n}m{m}_"
        d.update({p + "F": F, p + "H": H, p + "P0": P0, p + "x0": x0, p + "zs": zs, p + "q": np.float64(q), p + "r": np.float64(r)})

        def make():
            kf = KalmanFilter(dim_x=n, dim_z=m)
            kf.x, kf.P, kf.F, kf.H = x0.copy(), P0.copy(), F.copy(), H.copy()
            kf.Q, kf.R = q, r
            return kf
        # single steps, attributes raw
        kf = make()
        kf.predict()
        d[p + "step_Pp"] = kf.P.copy()
        kf.update(zs[0])
        d.update({p + "step_x": kf.x.copy(), p + "step_P": kf.P.copy(), p + "step_y": kf.y.copy(), p + "step_K": kf.K.copy(),
                  p + "step_S": kf.S.copy(), p + "step_SI": kf.SI.copy()})
        # the epoch loop written out by a user: predict(); update(z) -- raw attributes every epoch
        kf = make()
        xs, Ps = [], []
        for z in zs:
            kf.predict()
            kf.update(z)
            xs.append(kf.x.copy())
            Ps.append(kf.P.copy())
        d[p + "loop_x"], d[p + "loop_P"] = np.array(xs), np.array(Ps)
        # batch_filter: the same attributes arrive as kwargs -> q I, r I
        kf = make()
        mu, cov, mup, covp = kf.batch_filter(list(zs))
        d.update({p + "bf_mu": mu, p + "bf_cov": cov, p + "bf_mup": mup, p + "bf_covp": covp})
        # module-level update with a scalar R
        kf = make()
        kf.predict()
        x2, P2, y2, K2, S2, ll2 = kfmod.update(kf.x.copy(), kf.P.copy(), zs[0], r, H, return_all=True)
        d.update({p + "mod_x": x2, p + "mod_P": P2, p + "mod_y": y2, p + "mod_K": K2, p + "mod_S": S2})
    np.savez_compressed(os.path.join(OUT, "kf_scalar_attr.npz"), **d)
    print("wrote kf_scalar_attr.npz", len(d), "arrays")


if __name__ == "__main__":
    main()
