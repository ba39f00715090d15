#!/usr/bin/env python3
"""The fused linear-model UKF on ONE MI355X: `fk_ukf_linear_batch_f64` (UKF.py:364-491 over T steps, one launch) and
`fk_ukf_linear_rts_f64` (UKF.py:634-739, the whole backward pass) at several (dim_x, dim_z), both layouts.

One JSON line per (kernel, dims, layout): time, track-steps/s, fraction of 8 TB/s on algorithmic bytes, parity against the
oracle on one track.  The launchers read their A/B switch (FK_UKF_PADDED) once per process, so an
A/B comparison is two invocations of this script in one lease; the line carries the switches it ran under.

    python tools/bench_ukf.py --dims 6x3,4x2,8x4 --N 100000 --T 100
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_configs import timeit, rel, PEAK  # noqa: E402


def model(n, m, dt=0.1):
    """constant-velocity pairs, position measurements: a sparse F like BASELINE configs[3]'s, any dims"""
    F = np.eye(n)
    h = n // 2
    for i in range(h):
        F[i, i + h] = dt
    H = np.zeros((m, n))
    for i in range(m):
        H[i, i % n] = 1.0
    return F, H, 0.01 * np.eye(n), 0.5 * np.eye(m)


NO_OUTPUTS = False      # --no-outputs: the per-step histories are not stored (zero-record descriptors: stores issued and dropped)


def run(n, m, N, T, layout, dense):
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    alpha, beta, kappa = .1, 2., 3. - n
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    F, H, Q, R = model(n, m)
    if dense:
        r = np.random.default_rng(1)
        F = F + 0.01 * r.standard_normal((n, n))
        H = H + 0.01 * r.standard_normal((m, n))
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn((N, n) if layout == "aos" else (n, N), generator=g, device=dev, dtype=torch.float64)
    P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    x, P = x0.clone(), P0.clone()
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]
    sw = {k: os.environ[k] for k in ("FK_UKF_PADDED", "FK_UKF_DMA", "FK_UKF_CHUNKS", "FK_UKF_RTS_CHUNKS", "FK_UKF_PAIRED", "FK_UKF_MLG") if k in os.environ}
    paired = E.pair_weights(Wm, Wc, n)      # looked at once, outside the timed calls (Merwe's weights: True)

    def fwd():
        x.copy_(x0)
        P.copy_(P0)
        E.ukf_linear_batch(n, m, N, T, layout, lam + n, *dd, z, x, P, means=None if NO_OUTPUTS else means,
                           covs=None if NO_OUTPUTS else covs, status=st, paired=paired)
    ms = timeit(fwd)
    if NO_OUTPUTS:
        print(json.dumps(dict(kernel=f"fused linear UKF ({n},{m}) {layout} WITHOUT per-step outputs", N=N, T=T, ms=ms)), flush=True)
        return
    assert not st.any()
    trk = N - 1                    # the bank's last track: the one a tail-handling mistake would hit
    zs_h = (z[:, trk] if layout == "aos" else z[:, :, trk]).cpu().numpy()
    x0h = (x0[trk] if layout == "aos" else x0[:, trk]).cpu().numpy()
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0h, 10 * np.eye(n), list(zs_h), lambda s, d: F @ s, lambda s: H @ s,
                                                  0.1, Q, R, alpha, beta, kappa)
    mu = E.from_records(means, layout, 1, (n,))[:, trk]
    cov = E.from_records(covs, layout, 1, (n, n))[:, trk]
    par = max(rel(mu, mu_ref), rel(cov.reshape(T, -1), cov_ref.reshape(T, -1)))
    b = 8 * (m + n + n * n)
    gbs = N * T * b / (ms * 1e-3) / 1e9
    print(json.dumps(dict(kernel=f"fused linear UKF ({n},{m}) {layout}", N=N, T=T, dense_model=dense, ms=ms,
                          track_steps_per_s=N * T / (ms * 1e-3), alg_bytes_per_unit=b, frac_of_8TBs=gbs / PEAK,
                          parity_max_rel=par, switches=sw)), flush=True)
    if not E.ukf_linear_rts_supported(n, paired):
        return
    xs, Ps = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    Ks = E.alloc_records((T,), N, n * n, layout)
    ms = timeit(lambda: E.ukf_linear_rts(n, N, T, layout, lam + n, dd[0], dd[2], dd[4], dd[5], means, covs, xs, Ps, K=Ks,
                                         status=st, paired=paired))
    assert not st.any()
    xr, Pr, Kr = ukf_oracle.ukf_rts_smoother(mu_ref, cov_ref, lambda s, d: F @ s, 0.1, Q, alpha, beta, kappa)
    got_x = E.from_records(xs, layout, 1, (n,))[:, trk]
    got_P = E.from_records(Ps, layout, 1, (n, n))[:, trk]
    par = max(rel(got_x, xr), rel(got_P.reshape(T, -1), Pr.reshape(T, -1)))
    b = 8 * (2 * n + 3 * n * n)            # reads Xs, Ps; writes xs, Ps, K
    gbs = N * T * b / (ms * 1e-3) / 1e9
    print(json.dumps(dict(kernel=f"fused linear UKF smoother n={n} {layout}", N=N, T=T, dense_model=dense, ms=ms,
                          track_steps_per_s=N * T / (ms * 1e-3), alg_bytes_per_unit=b, frac_of_8TBs=gbs / PEAK,
                          parity_max_rel=par, switches=sw)), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="6x3")
    ap.add_argument("--layouts", default="soa,aos")
    ap.add_argument("--N", type=int, default=100_000)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--dense", action="store_true", help="dense F and H instead of the constant-velocity pattern")
    ap.add_argument("--no-outputs", action="store_true", help="forward kernel only, per-step histories not stored (how much of the step is the store path?)")
    a = ap.parse_args()
    NO_OUTPUTS = a.no_outputs
    for d in a.dims.split(","):
        n, m = (int(v) for v in d.split("x"))
        for lay in a.layouts.split(","):
            run(n, m, a.N, a.T, lay, a.dense)
