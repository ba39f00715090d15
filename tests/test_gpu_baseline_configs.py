"""BASELINE.json configs[2] and configs[3] exactly as SURVEY §8(d) writes them, at their full sizes, through the C ABI:

C3  1e5 tracks, dim_x = 9, dim_z = 3, T = 100, dt = 0.1, constant acceleration in 3-D: F = blockdiag(F2, F2, F2),
    F2 = [[1, dt, dt^2/2], [0, 1, dt], [0, 0, 1]], H = position selector (rows e0, e3, e6),
    Q = blockdiag of Q_discrete_white_noise(3, dt, .01), R = .25 I, P0 = 10 I, x0 = 0;
    batch_filter forward then rts_smoother (class convention, F[k+1]); parity on 512 sampled tracks.
C4  1e5 tracks, n = 6 (CV-3D, dt = .1, F = I + dt shift), m = 3 (positions), MerweScaledSigmaPoints(6, .1, 2., -3.)
    (13 points, Wm0 ~ -199), Q = .01 I, R = .5 I, P0 = 10 I, x0 = randn; sigma_points, unscented_transform and the fused
    linear UKF over T = 100; parity on 256 tracks against UKF.py's arithmetic (the oracle, pinned to the live reference by
    tests/test_oracle_ukf.py) with fx = F x, hx = H x.

The bar is the stated one: 1e-10 normwise per vector / matrix (conftest.rel_err_rows)."""
import numpy as np
import pytest

from conftest import rel_err_rows
from oracle import kf_oracle, ukf_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-10


def ca3d_model(dt=0.1):
    F2 = np.array([[1, dt, dt * dt / 2], [0, 1, dt], [0, 0, 1.]])
    F = np.kron(np.eye(3), F2)
    H = np.zeros((3, 9))
    H[0, 0] = H[1, 3] = H[2, 6] = 1.0
    # Q_discrete_white_noise(3, dt, var=.01)  (filterpy/common/discretization.py:133-136)
    q = np.array([[.25 * dt ** 4, .5 * dt ** 3, .5 * dt ** 2], [.5 * dt ** 3, dt ** 2, dt], [.5 * dt ** 2, dt, 1.]]) * 0.01
    return F, np.kron(np.eye(3), q), H, 0.25 * np.eye(3)


def _sample(t, layout, idx):
    """device records [T][N][E] (aos) / [T][E][N] (soa) -> host (T, len(idx), E) of the sampled tracks only"""
    import torch
    ix = torch.as_tensor(idx, device=t.device)
    s = t.index_select(-2 if layout == "aos" else -1, ix)
    if layout == "soa":
        s = s.transpose(-1, -2)
    return s.contiguous().cpu().numpy()


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_c3_constant_acceleration_3d_full_size(layout):
    import torch
    from filterpy_amd import _engine as E
    n, m, N, T = 9, 3, 100_000, 100
    F, Q, H, R = ca3d_model()
    rs = np.random.RandomState(33)
    # truth: random initial position / velocity / acceleration propagated by F; z = H x + N(0, R)
    xt = rs.randn(N, n) * np.tile([10., 1., .1], 3)
    zs = np.empty((T, N, m))
    for t in range(T):
        xt = xt @ F.T
        zs[t] = xt @ H.T + 0.5 * rs.randn(N, m)
    x0, P0 = np.zeros((N, n)), np.tile(10.0 * np.eye(n), (N, 1, 1))
    dz = E.to_records(zs, layout, 1)
    dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
    outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
            E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
    for o in outs:
        o.fill_(float("nan"))
    st = torch.zeros(N, dtype=torch.int32, device=dx.device)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    dF, dQ, dH, dR = (E.dev(M) for M in (F, Q, H, R))
    E.kf_batch_filter(desc, dF, dQ, dH, dR, dz, dx, dP, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
    so = [E.alloc_records((T,), N, n, layout)] + [E.alloc_records((T,), N, n * n, layout) for _ in range(3)]
    for o in so:
        o.fill_(float("nan"))
    E.kf_rts(desc, dF, dQ, outs[0], outs[1], so[0], so[1], so[2], so[3], convention=0, status=st)
    torch.cuda.synchronize()
    assert not st.any()
    sample = np.unique(np.concatenate([[0, 1, 63, 64, 65, N - 2, N - 1], rs.randint(0, N, 505)]))[:512]
    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=sample)
    got = [_sample(o, layout, sample) for o in outs]
    for k in range(4):
        assert rel_err_rows(got[k].reshape(T * len(sample), -1), ref[k].reshape(T * len(sample), -1)) < TOL, ("forward", k)
    # the smoother is held to the oracle's smoother RUN ON THE ORACLE'S OWN filter output (what the reference would return)
    sm = kf_oracle.rts_smoother_tracks(ref[0], ref[1], F, Q, tracks=range(len(sample)))
    gs = [_sample(o, layout, sample) for o in so]
    for k in range(4):
        a, b = gs[k], sm[k]
        if k >= 2:                      # K and Pp have no entry for the last epoch (kalman_filter.py:1058-1072: zeros)
            a, b = a[:-1], b[:-1]
        assert rel_err_rows(a.reshape(a.shape[0] * a.shape[1], -1), b.reshape(a.shape[0] * a.shape[1], -1)) < TOL, ("rts", k)
    # every track finite (no NaN fill left anywhere in the four forward outputs and the smoothed x, P)
    for o in outs + so[:2]:
        assert bool(torch.isfinite(o).all())


def c4_model(dt=0.1):
    n, m = 6, 3
    F = np.eye(n)
    for i in range(3):
        F[i, i + 3] = dt
    H = np.zeros((m, n))
    H[0, 0] = H[1, 1] = H[2, 2] = 1.0
    return F, H, 0.01 * np.eye(n), 0.5 * np.eye(m)


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_c4_merwe_ukf_full_size(layout):
    import torch
    from filterpy_amd import _engine as E
    n, m, k, N, T, dt = 6, 3, 13, 100_000, 100, 0.1
    alpha, beta, kappa = .1, 2., -3.
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    assert abs(Wm[0] + 199.0) < 1e-9
    F, H, Q, R = c4_model(dt)
    rs = np.random.RandomState(44)
    x0 = rs.randn(N, n)
    P0 = np.tile(10.0 * np.eye(n), (N, 1, 1))
    xt = x0 + rs.randn(N, n)
    zs = np.empty((T, N, m))
    for t in range(T):
        xt = xt @ F.T
        zs[t] = xt @ H.T + np.sqrt(.5) * rs.randn(N, m)
    sample = np.unique(np.concatenate([[0, 63, 64, N - 1], rs.randint(0, N, 260)]))[:256]

    # sigma_points and unscented_transform standalone on the bank's initial state
    dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
    sig = E.alloc_records((), N, k * n, layout)
    xo, Po = E.alloc_records((), N, n, layout), E.alloc_records((), N, n * n, layout)
    st = torch.zeros(N, dtype=torch.int32, device=dx.device)
    E.ut_sigma_points(n, N, layout, lam + n, dx, dP, sig, status=st)
    E.ut_transform(n, k, N, layout, sig, E.dev(Wm), E.dev(Wc), E.dev(Q), xo, Po)
    torch.cuda.synchronize()
    assert not st.any()
    gs, gx, gP = (_sample(t[None], layout, sample)[0] for t in (sig, xo, Po))
    for j, i in enumerate(sample[:64]):
        s_ref = ukf_oracle.merwe_sigma_points(x0[i], P0[i], alpha, kappa)
        assert rel_err_rows(gs[j].reshape(k, n), s_ref) < 1e-12
        x_ref, P_ref = ukf_oracle.unscented_transform(s_ref, Wm, Wc, Q)
        assert rel_err_rows(gx[j][None], x_ref[None]) < TOL and rel_err_rows(gP[j].reshape(1, -1), P_ref.reshape(1, -1)) < TOL

    # the fused linear UKF over T steps
    dz = E.to_records(zs, layout, 1)
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    means.fill_(float("nan"))
    covs.fill_(float("nan"))
    E.ukf_linear_batch(n, m, N, T, layout, lam + n, E.dev(F), E.dev(H), E.dev(Q), E.dev(R), E.dev(Wm), E.dev(Wc),
                       dz, dx, dP, means=means, covs=covs, status=st)
    torch.cuda.synchronize()
    assert not st.any()
    assert bool(torch.isfinite(means).all()) and bool(torch.isfinite(covs).all())
    mu, cov = _sample(means, layout, sample), _sample(covs, layout, sample)
    for j, i in enumerate(sample):
        mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0[i], P0[i], list(zs[:, i]), lambda s, d: F @ s, lambda s: H @ s,
                                                      dt, Q, R, alpha, beta, kappa)
        assert rel_err_rows(mu[:, j], mu_ref) < TOL, i
        assert rel_err_rows(cov[:, j].reshape(T, -1), cov_ref.reshape(T, -1)) < TOL, i
