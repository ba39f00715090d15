"""-m gpu: banks smaller than a wave, partial first waves and one-past-a-workgroup sizes through the
kernels that are not covered by test_gpu_kf.py::test_tail_shapes_kf_and_rts -- IMM / MMAE, steady state,
update_correlated, sigma points / unscented transform / cross variance, fused linear UKF.  Every track is
compared with the oracle."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows

pytestmark = pytest.mark.gpu
TOL = 1e-10
SIZES = (1, 2, 44, 65, 257)


def spd(rs, n, scale=1.0, batch=()):
    A = rs.randn(*batch, n, n)
    return scale * (A @ np.swapaxes(A, -1, -2) / n + 0.5 * np.eye(n))


def stable_F(rs, n):
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    return F / max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m,nm", [(2, 1, 2), (4, 2, 3), (6, 3, 2), (3, 2, 2)])
def test_imm_tails(n, m, nm, layout):
    from oracle import imm_oracle
    from test_gpu_imm import run_imm
    rs = np.random.RandomState(n * 100 + nm)
    Fs = np.array([stable_F(rs, n) for _ in range(nm)])
    Qs = np.array([spd(rs, n, 0.05 * (j + 1)) for j in range(nm)])
    Hs = np.array([rs.randn(m, n)] * nm)
    Rs = np.array([spd(rs, m, 0.5) for _ in range(nm)])
    M = rs.rand(nm, nm) + 2 * np.eye(nm)
    M /= M.sum(axis=1, keepdims=True)
    T = 5
    for N in SIZES:
        xs0, Ps0 = rs.randn(N, nm, n), spd(rs, n, 2.0, (N, nm))
        mu0 = rs.rand(N, nm) + 0.1
        mu0 /= mu0.sum(axis=1, keepdims=True)
        zs = rs.randn(T, N, m) * 2
        r = run_imm(xs0, Ps0, mu0, M, zs, Fs, Qs, Hs, Rs, layout)
        for trk in range(N):
            x, P, mu, xp, Pp, L = imm_oracle.imm_batch(xs0[trk], Ps0[trk], mu0[trk], M, zs[:, trk], Fs, Qs, Hs, Rs)
            assert rel_err_rows(r["x_out"][:, trk], x) < TOL and rel_err_rows(r["P_out"][:, trk], P) < TOL, (N, trk)
            assert np.allclose(r["mu_out"][:, trk], mu, rtol=1e-10, atol=1e-14), (N, trk)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (9, 3), (3, 2)])
def test_steadystate_and_correlated_tails(n, m, layout):
    from filterpy_amd.kalman import KalmanFilterBank
    from oracle import kf_oracle
    rs = np.random.RandomState(n * 10 + m)
    F, H, R = stable_F(rs, n), rs.randn(m, n), spd(rs, m, 0.5)
    K, Mx = 0.2 * rs.randn(n, m), 0.1 * rs.randn(n, m)
    T = 4
    for N in SIZES:
        bank = KalmanFilterBank(n, m, N, layout=layout)
        x0, P0 = rs.randn(N, n), spd(rs, n, 2.0, (N,))
        zs = rs.randn(T, N, m)
        bank.x, bank.F, bank.H, bank.K = x0.copy(), F, H, K
        means, means_p, y = bank.steadystate_filter(zs)
        for trk in range(N):
            rx, rxp, ry = kf_oracle.steadystate_filter(x0[trk], list(zs[:, trk]), F, H, K)
            assert rel_err_rows(means[:, trk], rx) < TOL and rel_err_rows(means_p[:, trk], rxp) < TOL, (N, trk)
        bank.x, bank.P, bank.R, bank.M = x0.copy(), P0.copy(), R, Mx
        bank.update_correlated(zs[0])
        for trk in range(N):
            x, P, *_ = kf_oracle.update_correlated(x0[trk], P0[trk], zs[0, trk], R, H, Mx)
            assert rel_err_rows(bank.x[trk], x) < TOL and rel_err_rows(bank.P[trk], P) < TOL, (N, trk)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3), (3, 2)])
def test_ukf_building_blocks_tails(n, m, layout):
    from filterpy_amd.kalman import MerweScaledSigmaPoints, UnscentedKalmanFilter, unscented_transform
    from oracle import ukf_oracle
    rs = np.random.RandomState(n * 7 + m)
    alpha, beta, kappa = 0.5, 2.0, 3.0 - n
    pts = MerweScaledSigmaPoints(n, alpha, beta, kappa)
    F, H = stable_F(rs, n), rs.randn(m, n)
    Q, R = spd(rs, n, 0.05), spd(rs, m, 0.5)
    T = 4
    for N in SIZES:
        x0, P0 = rs.randn(N, n), spd(rs, n, 2.0, (N,))
        sig = pts.sigma_points(x0, P0)
        ux, uP = unscented_transform(sig, pts.Wm, pts.Wc, Q, layout=layout)
        for trk in range(N):
            s_ref = ukf_oracle.merwe_sigma_points(x0[trk], P0[trk], alpha, kappa)
            assert rel_err_rows(sig[trk], s_ref) < 1e-12, (N, trk)
            rx, rP = ukf_oracle.unscented_transform(s_ref, pts.Wm, pts.Wc, Q)
            assert rel_err_rows(ux[trk][None], rx[None]) < TOL and rel_err_rows(uP[trk], rP) < TOL, (N, trk)
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=H, fx=F, points=pts, n_tracks=N, layout=layout)
        ukf.x, ukf.P, ukf.Q, ukf.R = x0.copy(), P0.copy(), Q, R
        zs = rs.randn(T, N, m)
        mu, cov = ukf.batch_filter(zs)
        for trk in range(min(N, 70)):
            rmu, rcov = ukf_oracle.ukf_batch_filter(x0[trk], P0[trk], zs[:, trk], lambda x, dt: F @ x, lambda x: H @ x,
                                                    1.0, Q, R, alpha, beta, kappa)
            assert rel_err_rows(mu[:, trk], rmu) < 1e-10 and rel_err_rows(cov[:, trk], rcov) < 1e-10, (N, trk)
