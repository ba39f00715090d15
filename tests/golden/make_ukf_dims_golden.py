#!/usr/bin/env python3
"""Golden vectors for the UKF path at dim_x = 7 .. 16 from the LIVE reference (rlabbe/filterpy v1.4.5;
sigma_points.py:124-177, unscented_transform.py:99-128, UKF.py:364-504, :524-632, :634-739).

Round 2 shipped the sigma / UT / cross-variance / correct / rts kernels for the padded classes 8, 12 and 16 with GPU
parity only at n <= 6.  This file pins those classes: per (n, m) one random linear model, one
step with every intermediate (sigma points, both unscented transforms' outputs, K, S, y), a T-step batch_filter and the
smoother over it.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_ukf_dims_golden.py
"""
import os
import sys

import numpy as np

REF = os.environ.get("FILTERPY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from filterpy.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints, unscented_transform  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# (n, m, alpha, beta, kappa): the sizes VERDICT r2 names plus (11,1), (14,6), (15,7) so that every dim 7..16 and every
# dim_z 1..8 occurs once
CASES = [(7, 3, .5, 2., -4.), (8, 4, .3, 2., 0.), (9, 3, .5, 2., -6.), (10, 2, 1., 2., 0.), (11, 1, .5, 2., 1.),
         (12, 4, .5, 2., -9.), (13, 5, .7, 2., 0.), (14, 6, .5, 2., 0.), (15, 7, .5, 2., 3. - 15), (16, 8, .5, 2., 0.)]
T = 12


def spd(rs, n, scale=1.0):
    A = rs.randn(n, n)
    return scale * (A @ A.T / n + 0.5 * np.eye(n))


def stable_F(rs, n):
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    return F / max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))


def main():
    d = {"cases": np.array(CASES)}
    for ci, (n, m, alpha, beta, kappa) in enumerate(CASES):
        rs = np.random.RandomState(7000 + ci)
        pts = MerweScaledSigmaPoints(n, alpha, beta, kappa)
        F, H = stable_F(rs, n), rs.randn(m, n)
        Q, R, P0, x0 = spd(rs, n, 0.01), spd(rs, m, 0.5), spd(rs, n, 5.0), rs.randn(n)
        zs = rs.randn(T, m) * 3
        p = f"c{ci}_"
        sig = pts.sigma_points(x0, P0)
        ux, uP = unscented_transform(sig, pts.Wm, pts.Wc, Q)
        d.update({p + "F": F, p + "H": H, p + "Q": Q, p + "R": R, p + "P0": P0, p + "x0": x0, p + "zs": zs,
                  p + "Wm": pts.Wm, p + "Wc": pts.Wc, p + "sigmas": sig, p + "ut_x": ux, p + "ut_P": uP})
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=lambda x: H @ x, fx=lambda x, dt: F @ x, points=pts)
        ukf.x, ukf.P, ukf.Q, ukf.R = x0.copy(), P0.copy(), Q.copy(), R.copy()
        ukf.predict()
        d.update({p + "s1_xp": ukf.x.copy(), p + "s1_Pp": ukf.P.copy(), p + "s1_sigmas_f": ukf.sigmas_f.copy()})
        ukf.update(zs[0])
        Pxz = ukf.cross_variance(ukf.x_prior, np.dot(pts.Wm, ukf.sigmas_h), ukf.sigmas_f, ukf.sigmas_h)
        d.update({p + "s1_x": ukf.x.copy(), p + "s1_P": ukf.P.copy(), p + "s1_K": ukf.K.copy(), p + "s1_S": ukf.S.copy(),
                  p + "s1_y": ukf.y.copy(), p + "s1_sigmas_h": ukf.sigmas_h.copy(), p + "s1_Pxz": Pxz})
        ukf.x, ukf.P = x0.copy(), P0.copy()
        mu, cov = ukf.batch_filter(list(zs) if m > 1 else [np.array([z[0]]) for z in zs])
        xs, Ps, Ks = ukf.rts_smoother(mu, cov)
        d.update({p + "mu": mu, p + "cov": cov, p + "rts_x": xs, p + "rts_P": Ps, p + "rts_K": Ks})
    np.savez_compressed(os.path.join(OUT, "ukf_dims.npz"), **d)
    print("wrote ukf_dims.npz:", len(d), "arrays")


if __name__ == "__main__":
    main()
