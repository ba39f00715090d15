#!/usr/bin/env python3
"""Margins of tests/test_gpu_variants.py::test_ukf_rts_smoother_goldens, case by case (which case, which output, how far)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ukf_tol, golden, rel_err_rows  # noqa: E402
from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints  # noqa: E402

g = golden("ukf_merwe")
for lin in (True, False):
    for ci, c in enumerate(g["cases"]):
        n, m, alpha, beta, kappa = int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4])
        if n > 9:
            continue
        p = f"c{ci}_"
        F, H = g[p + "F"], g[p + "H"]
        pts = MerweScaledSigmaPoints(n, alpha, beta, kappa)
        if lin:
            ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=H, fx=F, points=pts)
        else:
            ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=lambda x: H @ x, fx=lambda x, dt: F @ x, points=pts)
        ukf.Q, ukf.R = g[p + "Q"].copy(), g[p + "R"].copy()
        xs, Ps, Ks = ukf.rts_smoother(g[p + "mu"], g[p + "cov"])
        e = [rel_err_rows(xs, g[p + "rts_x"]), rel_err_rows(Ps, g[p + "rts_P"]), rel_err_rows(Ks[:-1], g[p + "rts_K"][:-1])]
        t = [ukf_tol(ci, k) for k in ("rts_x", "rts_P", "rts_K")]
        cond = max(np.linalg.cond(P) for P in g[p + "cov"])
        print("linear=%s case %d (n=%d m=%d alpha=%g): x %.2e/%.1e  P %.2e/%.1e  K %.2e/%.1e  %s  max cond(P)=%.1e" % (
            lin, ci, n, m, alpha, e[0], t[0], e[1], t[1], e[2], t[2], "FAIL" if any(a >= b for a, b in zip(e, t)) else "ok", cond))
