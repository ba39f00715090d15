#!/usr/bin/env python3
"""The smallest GPU check of the four-lane UKF kernels (csrc/ukf_mlg.hip): one bank through fk_ukf_linear_batch_f64 and
fk_ukf_linear_rts_f64 at (12,3) and (16,4), two tracks each against the oracle.  FK_UKF_MLG=1 must be set.  Seconds."""
import os
import sys
import time

t0 = time.time()
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from filterpy_amd import _engine as E   # noqa: E402
from oracle import ukf_oracle          # noqa: E402

print("imports %.1fs" % (time.time() - t0), flush=True)
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))  # noqa: E731


def spd(rs, n, s, batch=()):
    A = rs.randn(*batch, n, n)
    return s * (A @ np.swapaxes(A, -1, -2) / n + 0.5 * np.eye(n))


for n, m, layout in ((12, 3, "soa"), (16, 4, "aos"), (10, 2, "aos"), (13, 5, "soa")):
    rs = np.random.RandomState(n)
    N, T, alpha, beta, kappa = 70, 5, .5, 2., 3. - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), spd(rs, n, 0.05), spd(rs, m, 0.5)
    x0, P0, zs = rs.randn(N, n), spd(rs, n, 2.0, (N,)), rs.randn(T, N, m)
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    sc = alpha ** 2 * (n + kappa)
    dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.full((N,), -1, dtype=torch.int32, device=dx.device)
    dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]
    try:
        E.ukf_linear_batch(n, m, N, T, layout, sc, *dd, E.to_records(zs, layout, 1), dx, dP, means=means, covs=covs, status=st, paired=True)
        torch.cuda.synchronize()
        mu, cov = E.from_records(means, layout, 1, (n,)), E.from_records(covs, layout, 1, (n, n))
        errs = []
        for trk in (0, N - 1):
            rmu, rcov = ukf_oracle.ukf_batch_filter(x0[trk], P0[trk], list(zs[:, trk]), lambda x, dt: F @ x, lambda x: H @ x, 1.0, Q, R, alpha, beta, kappa)
            errs.append(max(rel(mu[:, trk], rmu), rel(cov[:, trk], rcov)))
        print("filter   (%d,%d) %s status %s err %s" % (n, m, layout, sorted(set(st.cpu().numpy().tolist())), ["%.1e" % e for e in errs]), flush=True)
        xs, ps, Ks = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout), E.alloc_records((T,), N, n * n, layout)
        st.fill_(-1)
        E.ukf_linear_rts(n, N, T, layout, sc, dd[0], dd[2], dd[4], dd[5], means, covs, xs, ps, Ks, st, paired=True)
        torch.cuda.synchronize()
        gx, gp, gk = E.from_records(xs, layout, 1, (n,)), E.from_records(ps, layout, 1, (n, n)), E.from_records(Ks, layout, 1, (n, n))
        errs = []
        for trk in (0, N - 1):
            rx, rP, rK = ukf_oracle.ukf_rts_smoother(mu[:, trk], cov[:, trk], lambda x, dt: F @ x, 1.0, Q, alpha, beta, kappa)
            errs.append(max(rel(gx[:, trk], rx), rel(gp[:, trk], rP), rel(gk[:-1, trk], rK[:-1])))
        print("smoother (%d)   %s status %s err %s" % (n, layout, sorted(set(st.cpu().numpy().tolist())), ["%.1e" % e for e in errs]), flush=True)
    except Exception as exc:      # noqa: BLE001
        print("FAILED (%d,%d) %s: %r" % (n, m, layout, exc), flush=True)
print("total %.1fs" % (time.time() - t0), flush=True)
