#!/usr/bin/env python3
"""UnscentedKalmanFilter with every constructor hook set (sqrt_fn, x_mean_fn, z_mean_fn, residual_x, residual_z,
state_add; MerweScaledSigmaPoints sqrt_method, subtract), from the LIVE reference -> tests/golden/ukf_hooks.npz.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_ukf_hooks_golden.py

The problem (tests/ukf_hook_model.py) tracks a heading across the +-pi wrap from range / bearing measurements; without
the hooks the filter's plain means and differences of angles are wrong by 2 pi, so the frozen numbers really exercise
UKF.py:400-411 (x_mean, residual_x), :462-481 (z_mean, residual_z, state_add), :493-504, :714-737 and
unscented_transform.py:105-106, :120-123, sigma_points.py:106-116, :170-175."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("FILTERPY_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
from filterpy.kalman import MerweScaledSigmaPoints, UnscentedKalmanFilter, unscented_transform  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
import ukf_hook_model as hm  # noqa: E402


def main():
    sc = hm.scenario()
    N, T = sc["x0"].shape[0], sc["zs"].shape[0]
    d = {k: np.asarray(v) for k, v in sc.items()}
    mus, covs, xs, Ps, Ks = [], [], [], [], []
    for i in range(N):
        pts = MerweScaledSigmaPoints(3, sc["alpha"], sc["beta"], sc["kappa"], sqrt_method=hm.sqrt_lower_t,
                                     subtract=hm.sigma_subtract)
        kf = UnscentedKalmanFilter(3, 2, sc["dt"], hm.hx, hm.fx, pts, sqrt_fn=hm.sqrt_lower_t, x_mean_fn=hm.x_mean,
                                   z_mean_fn=hm.z_mean, residual_x=hm.residual_x, residual_z=hm.residual_z,
                                   state_add=hm.state_add)
        kf.x, kf.P, kf.Q, kf.R = sc["x0"][i].copy(), sc["P0"][i].copy(), sc["Q"], sc["R"]
        if i == 0:
            d["Wm"], d["Wc"] = pts.Wm.copy(), pts.Wc.copy()
            sig = pts.sigma_points(kf.x, kf.P)
            d["sigmas0"] = sig.copy()
            sf = np.array([hm.fx(s, sc["dt"]) for s in sig])
            ux, uP = unscented_transform(sf, pts.Wm, pts.Wc, sc["Q"], hm.x_mean, hm.residual_x)
            d["ut_x"], d["ut_P"] = ux, uP
            # mean hook only: the fast branch with a caller-supplied mean (unscented_transform.py:116-118)
            ux2, uP2 = unscented_transform(sf, pts.Wm, pts.Wc, sc["Q"], hm.x_mean, None)
            d["ut_meanonly_x"], d["ut_meanonly_P"] = ux2, uP2
        zs = [sc["zs"][t, i] for t in range(T)]
        if i == 1:
            zs[4] = None                                  # a missing measurement on one track (UKF.py:440-444)
        mu, cov = kf.batch_filter(zs)
        mus.append(mu)
        covs.append(cov)
        a, b, c = kf.rts_smoother(mu, cov)
        xs.append(a)
        Ps.append(b)
        Ks.append(c)
    d.update(mu=np.stack(mus, 1), cov=np.stack(covs, 1), rts_x=np.stack(xs, 1), rts_P=np.stack(Ps, 1), rts_K=np.stack(Ks, 1))
    # the same filter WITHOUT hooks goes wrong at the wrap -- recorded so the test can show the hooks matter
    pts = MerweScaledSigmaPoints(3, sc["alpha"], sc["beta"], sc["kappa"])
    kf = UnscentedKalmanFilter(3, 2, sc["dt"], hm.hx, hm.fx, pts)
    kf.x, kf.P, kf.Q, kf.R = sc["x0"][0].copy(), sc["P0"][0].copy(), sc["Q"], sc["R"]
    try:
        d["mu_nohooks_track0"] = kf.batch_filter([sc["zs"][t, 0] for t in range(T)])[0]
    except np.linalg.LinAlgError:
        d["mu_nohooks_track0"] = np.full((T, 3), np.nan)
    np.savez_compressed(os.path.join(OUT, "ukf_hooks.npz"), **d)
    print("wrote ukf_hooks.npz:", {k: v.shape for k, v in d.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
