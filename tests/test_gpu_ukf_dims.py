"""-m gpu: the UKF kernels at dim_x = 7 .. 16, dim_z = 1 .. 8 -- the padded classes 8 / 12 / 16 of sigma_kernel,
ut_kernel, cross_kernel, ukf_correct_kernel, ukf_rts_kernel and the fused linear kernels where they exist -- through the
C ABI against (a) goldens frozen from the live reference (tests/golden/make_ukf_dims_golden.py: sigma_points.py:124-177,
unscented_transform.py:99-128, UKF.py:364-504, :524-632, :634-739) and (b) the oracle on banks of different tracks with
ragged sizes.  VERDICT r2 weak 1: these classes shipped in round 2 with GPU parity only at n <= 6."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows

pytestmark = pytest.mark.gpu
TOL = 1e-10
SIZES = (1, 65, 257)


def _cases():
    g = golden("ukf_dims")
    return [(ci, int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4])) for ci, c in enumerate(g["cases"])]


CASES = _cases()
IDS = [f"{c[1]}x{c[2]}" for c in CASES]


def spd(rs, n, scale=1.0, batch=()):
    A = rs.randn(*batch, n, n)
    return scale * (A @ np.swapaxes(A, -1, -2) / n + 0.5 * np.eye(n))


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_sigma_ut_cross_blocks(case, layout):
    """fk_ut_sigma_points_f64 / fk_ut_transform_f64 / fk_ut_cross_variance_f64: track 0 carries the golden's inputs, the
    others random ones; every track against the oracle, track 0 against the live reference's arrays."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    ci, n, m, alpha, beta, kappa = case
    g, p = golden("ukf_dims"), f"c{ci}_"
    lam = alpha ** 2 * (n + kappa) - n
    k = 2 * n + 1
    Wm, Wc, Q = g[p + "Wm"], g[p + "Wc"], g[p + "Q"]
    rs = np.random.RandomState(100 * n + m)
    for N in SIZES:
        x0, P0 = rs.randn(N, n), spd(rs, n, 3.0, (N,))
        x0[0], P0[0] = g[p + "x0"], g[p + "P0"]
        dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
        sig = E.alloc_records((), N, k * n, layout)
        sig.fill_(float("nan"))
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ut_sigma_points(n, N, layout, lam + n, dx, dP, sig, st)
        torch.cuda.synchronize()
        assert not st.any()
        got = E.from_records(sig, layout, 0, (k, n))
        assert rel_err_rows(got[0], g[p + "sigmas"]) < 1e-12
        refs = np.array([ukf_oracle.merwe_sigma_points(x0[t], P0[t], alpha, kappa) for t in range(N)])
        assert rel_err_rows(got.reshape(N * k, n), refs.reshape(N * k, n)) < 1e-12, N
        # unscented transform of the oracle's sigma points (+ Q)
        xo, Po = E.alloc_records((), N, n, layout), E.alloc_records((), N, n * n, layout)
        xo.fill_(float("nan")), Po.fill_(float("nan"))
        E.ut_transform(n, k, N, layout, E.to_records(refs, layout, 0), E.dev(Wm), E.dev(Wc), E.dev(Q), xo, Po)
        torch.cuda.synchronize()
        gx, gP = E.from_records(xo, layout, 0, (n,)), E.from_records(Po, layout, 0, (n, n))
        assert rel_err_rows(gx[:1], g[p + "ut_x"][None]) < TOL and rel_err_rows(gP[:1], g[p + "ut_P"][None]) < TOL
        for t in range(N):
            rx, rP = ukf_oracle.unscented_transform(refs[t], Wm, Wc, Q)
            assert rel_err_rows(gx[t][None], rx[None]) < TOL and rel_err_rows(gP[t][None], rP[None]) < TOL, (N, t)
        # the measurement-space transform: k points of dimension m (k != 2m+1: the general ut_kernel), no noise
        sh = rs.randn(N, k, m)
        zo, So = E.alloc_records((), N, m, layout), E.alloc_records((), N, m * m, layout)
        E.ut_transform(m, k, N, layout, E.to_records(sh, layout, 0), E.dev(Wm), E.dev(Wc), None, zo, So)
        # cross variance (UKF.py:493-504)
        sf, xm, zm = rs.randn(N, k, n), rs.randn(N, n), rs.randn(N, m)
        out = E.alloc_records((), N, n * m, layout)
        out.fill_(float("nan"))
        E.ut_cross_variance(n, m, k, N, layout, E.to_records(xm, layout, 0), E.to_records(zm, layout, 0),
                            E.to_records(sf, layout, 0), E.to_records(sh, layout, 0), E.dev(Wc), out)
        torch.cuda.synchronize()
        gz, gS = E.from_records(zo, layout, 0, (m,)), E.from_records(So, layout, 0, (m, m))
        gPxz = E.from_records(out, layout, 0, (n, m))
        for t in range(N):
            rz, rS = ukf_oracle.unscented_transform(sh[t], Wm, Wc)
            assert rel_err_rows(gz[t][None], rz[None]) < TOL and rel_err_rows(gS[t][None], rS[None]) < TOL, (N, t)
            ref = ukf_oracle.cross_variance(xm[t], zm[t], sf[t], sh[t], Wc)
            assert rel_err_rows(gPxz[t][None], ref[None]) < 1e-12, (N, t)
    # the reference's own cross variance of step 1
    sf1, sh1 = g[p + "s1_sigmas_f"], g[p + "s1_sigmas_h"]
    out = E.alloc_records((), 3, n * m, layout)
    t3 = lambda a: np.tile(a, (3,) + (1,) * np.ndim(a))
    E.ut_cross_variance(n, m, k, 3, layout, E.to_records(t3(g[p + "s1_xp"]), layout, 0),
                        E.to_records(t3(np.dot(Wm, sh1)), layout, 0), E.to_records(t3(sf1), layout, 0),
                        E.to_records(t3(sh1), layout, 0), E.dev(Wc), out)
    assert rel_err_rows(E.from_records(out, layout, 0, (n, m))[2][None], g[p + "s1_Pxz"][None]) < 1e-12


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_correct_and_rts_correct_blocks(case, layout):
    """fk_ukf_correct_f64 (UKF.py:470-481: K = Pxz inv(S); x += K (z - zp); P -= K S K') and fk_ukf_rts_correct_f64
    (UKF.py:732-737: K = Pxb inv(Pb); x += K (x[k+1] - xb); P += K (P[k+1] - Pb) K') on ragged banks."""
    import torch
    from filterpy_amd import _engine as E
    ci, n, m, alpha, beta, kappa = case
    rs = np.random.RandomState(200 * n + m)
    for N in SIZES:
        Pxz, zp, S, z = rs.randn(N, n, m), rs.randn(N, m), spd(rs, m, 2.0, (N,)), rs.randn(N, m)
        x, P = rs.randn(N, n), spd(rs, n, 4.0, (N,))
        dx, dP, dK = E.to_records(x, layout, 0), E.to_records(P, layout, 0), E.alloc_records((), N, n * m, layout)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ukf_correct(n, m, N, layout, E.to_records(Pxz, layout, 0), E.to_records(zp, layout, 0), E.to_records(S, layout, 0),
                      E.to_records(z, layout, 0), dx, dP, dK, st)
        torch.cuda.synchronize()
        assert not st.any()
        gx, gP, gK = E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n)), E.from_records(dK, layout, 0, (n, m))
        for t in range(N):
            K = np.dot(Pxz[t], np.linalg.inv(S[t]))
            rx = x[t] + np.dot(K, z[t] - zp[t])
            rP = P[t] - np.dot(K, np.dot(S[t], K.T))
            assert rel_err_rows(gK[t][None], K[None]) < TOL and rel_err_rows(gx[t][None], rx[None]) < TOL, (N, t)
            assert rel_err_rows(gP[t][None], rP[None]) < TOL, (N, t)
        Pxb, xb, Pb = rs.randn(N, n, n), rs.randn(N, n), spd(rs, n, 2.0, (N,))
        xn, Pn = rs.randn(N, n), spd(rs, n, 2.0, (N,))
        dx, dP, dK = E.to_records(x, layout, 0), E.to_records(P, layout, 0), E.alloc_records((), N, n * n, layout)
        st.zero_()
        E.ukf_rts_correct(n, N, layout, E.to_records(Pxb, layout, 0), E.to_records(xb, layout, 0), E.to_records(Pb, layout, 0),
                          E.to_records(xn, layout, 0), E.to_records(Pn, layout, 0), dx, dP, dK, st)
        torch.cuda.synchronize()
        assert not st.any()
        gx, gP, gK = E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n)), E.from_records(dK, layout, 0, (n, n))
        for t in range(N):
            K = np.dot(Pxb[t], np.linalg.inv(Pb[t]))
            rx = x[t] + np.dot(K, xn[t] - xb[t])
            rP = P[t] + np.dot(K, Pn[t] - Pb[t]).dot(K.T)
            assert rel_err_rows(gK[t][None], K[None]) < TOL and rel_err_rows(gx[t][None], rx[None]) < TOL, (N, t)
            assert rel_err_rows(gP[t][None], rP[None]) < TOL, (N, t)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("mode", ["matrix", "vectorized", "device"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_filter_and_smoother_vs_live_reference(case, mode, layout):
    """UnscentedKalmanFilter.predict / update / batch_filter / rts_smoother with a linear model handed over as matrices
    (the fused kernels where they exist), as vectorised NumPy callables and as device callables (the split path) --
    replicated over a 70-track bank, against what the live reference returned for one filter."""
    import torch
    from filterpy_amd import _engine as E
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    ci, n, m, alpha, beta, kappa = case
    g, p = golden("ukf_dims"), f"c{ci}_"
    N = 70
    F, H = g[p + "F"], g[p + "H"]
    kw = {}
    if mode == "matrix":
        fx, hx = F, H
    elif mode == "vectorized":
        fx, hx = (lambda s, dt: s @ F.T), (lambda s: s @ H.T)
        kw = dict(vectorized=True)
    else:
        Fd, Hd = E.dev(F), E.dev(H)
        fx, hx = (lambda s, dt: torch.matmul(s, Fd.T)), (lambda s: torch.matmul(s, Hd.T))
        kw = dict(device_callables=True)
    ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=hx, fx=fx, points=MerweScaledSigmaPoints(n, alpha, beta, kappa),
                                n_tracks=N, layout=layout, **kw)
    tile = lambda a: np.tile(a, (N,) + (1,) * np.ndim(a))
    ukf.x, ukf.P, ukf.Q, ukf.R = tile(g[p + "x0"]), tile(g[p + "P0"]), g[p + "Q"].copy(), g[p + "R"].copy()
    zs = np.tile(g[p + "zs"][:, None, :], (1, N, 1))
    T = zs.shape[0]
    # one explicit step with the intermediates
    ukf.predict()
    for trk in (0, 63, 64, N - 1):
        assert rel_err_rows(ukf.x[trk][None], g[p + "s1_xp"][None]) < TOL and rel_err_rows(ukf.P[trk][None], g[p + "s1_Pp"][None]) < TOL
        assert rel_err_rows(ukf.sigmas_f[trk], g[p + "s1_sigmas_f"]) < TOL
    ukf.update(zs[0])
    for trk in (0, 63, 64, N - 1):
        for got, key in ((ukf.x, "s1_x"), (ukf.P, "s1_P"), (ukf.K, "s1_K"), (ukf.S, "s1_S"), (ukf.y, "s1_y")):
            assert rel_err_rows(np.asarray(got)[trk][None], g[p + key][None]) < TOL, (key, trk)
    ukf.x, ukf.P = tile(g[p + "x0"]), tile(g[p + "P0"])
    mu, cov = ukf.batch_filter(zs if mode == "device" else list(zs))
    assert mu.shape == (T, N, n) and cov.shape == (T, N, n, n)
    for trk in (0, 63, 64, N - 1):
        assert rel_err_rows(mu[:, trk], g[p + "mu"]) < TOL and rel_err_rows(cov[:, trk], g[p + "cov"]) < TOL, trk
    assert rel_err_rows(ukf.x[[0, N - 1]], np.tile(g[p + "mu"][-1], (2, 1))) < TOL
    Xs, Ps = np.tile(g[p + "mu"][:, None], (1, N, 1)), np.tile(g[p + "cov"][:, None], (1, N, 1, 1))
    xs, ps, Ks = ukf.rts_smoother(Xs, Ps)
    for trk in (0, 64, N - 1):
        assert rel_err_rows(xs[:, trk], g[p + "rts_x"]) < TOL and rel_err_rows(ps[:, trk], g[p + "rts_P"]) < TOL, trk
        assert rel_err_rows(Ks[:-1, trk], g[p + "rts_K"][:-1]) < TOL, trk


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(9, 3), (16, 4), (12, 8), (7, 1)])
def test_linear_bank_of_different_tracks_vs_oracle(n, m, layout):
    """every track its own state and measurements, ragged bank sizes, a missing measurement: matrix fx / hx through
    batch_filter + rts_smoother against the oracle's per-filter loop (UKF.py:623-632, :714-739)."""
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    from oracle import ukf_oracle
    rs = np.random.RandomState(31 * n + m)
    alpha, beta, kappa = 0.5, 2.0, 3.0 - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), spd(rs, n, 0.05), spd(rs, m, 0.5)
    T = 6
    for N in (1, 65, 130):
        x0, P0 = rs.randn(N, n), spd(rs, n, 2.0, (N,))
        zs = rs.randn(T, N, m)
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=H, fx=F, points=MerweScaledSigmaPoints(n, alpha, beta, kappa),
                                    n_tracks=N, layout=layout)
        ukf.x, ukf.P, ukf.Q, ukf.R = x0.copy(), P0.copy(), Q, R
        zl = list(zs)
        zl[3] = None
        mu, cov = ukf.batch_filter(zl)
        xs, ps, Ks = ukf.rts_smoother(mu, cov)
        for trk in sorted({0, N // 2, N - 1}):
            zt = [None if z is None else z[trk] for z in zl]
            rmu, rcov = ukf_oracle.ukf_batch_filter(x0[trk], P0[trk], zt, lambda x, dt: F @ x, lambda x: H @ x, 1.0, Q, R,
                                                    alpha, beta, kappa)
            assert rel_err_rows(mu[:, trk], rmu) < TOL and rel_err_rows(cov[:, trk], rcov) < TOL, (N, trk)
            rxs, rps, rKs = ukf_oracle.ukf_rts_smoother(rmu, rcov, lambda x, dt: F @ x, 1.0, Q, alpha, beta, kappa)
            assert rel_err_rows(xs[:, trk], rxs) < TOL and rel_err_rows(ps[:, trk], rps) < TOL, (N, trk)
            assert rel_err_rows(Ks[:-1, trk], rKs[:-1]) < TOL, (N, trk)


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_linear_map_kernel_vs_numpy(layout):
    """fk_ut_linear_map_f64 (a linear fx / hx given as a matrix, where the fused kernels do not reach): out[i] = M in[i] for
    every sigma point of every track, all dims 1..16, ragged banks -- against numpy's matrix product of the same arrays"""
    import torch
    from filterpy_amd import _engine as E
    rs = np.random.RandomState(11)
    for n_in, n_out in ((1, 1), (2, 1), (4, 2), (6, 3), (7, 7), (9, 4), (12, 12), (16, 8), (16, 16), (3, 16)):
        k = 2 * n_in + 1
        M = rs.randn(n_out, n_in)
        for N in SIZES:
            sig = rs.randn(N, k, n_in)
            out = E.alloc_records((), N, k * n_out, layout)
            out.fill_(float("nan"))
            E.ut_linear_map(n_in, n_out, k, N, layout, E.dev(M), E.to_records(sig, layout, 0), out)
            torch.cuda.synchronize()
            got = E.from_records(out, layout, 0, (k, n_out))
            ref = sig @ M.T
            assert rel_err_rows(got.reshape(N * k, n_out), ref.reshape(N * k, n_out)) < 1e-13, (n_in, n_out, N)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(6, 3), (8, 4), (9, 3)])
def test_fused_ukf_chunked_call_is_bit_identical(n, m, layout, monkeypatch):
    """FK_UKF_CHUNKS="G,H" cuts fk_ukf_linear_batch_f64 into track groups x time chunks on helper streams (fk_chunks.hpp,
    ukf_chunked_call), the state handed over through x / P in place: outputs, final state and status must be bit-identical
    to the single launch, incl. a ragged bank, a missing measurement and a track that turns non-positive-definite"""
    import torch
    from filterpy_amd import _engine as E
    rs = np.random.RandomState(5 * n + m)
    N, T = 1000 + 37, 23
    alpha, beta, kappa = 0.5, 2.0, 3.0 - n
    from oracle import ukf_oracle
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    lam = alpha ** 2 * (n + kappa) - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), spd(rs, n, 0.05), spd(rs, m, 0.5)
    x0, P0 = rs.randn(N, n), spd(rs, n, 2.0, (N,))
    P0[77] = -np.eye(n)                                   # not positive definite: status bit, garbage that must not differ
    zs = rs.randn(T, N, m)
    mask = np.ones((T, N), dtype=np.uint8)
    mask[5] = 0
    res = {}
    for tag, env in (("one", None), ("3x4", "3,4"), ("2x7", "2,7"), ("4x23", "4,23")):
        if env:
            monkeypatch.setenv("FK_UKF_CHUNKS", env)
        else:
            monkeypatch.delenv("FK_UKF_CHUNKS", raising=False)
        dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
        means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ukf_linear_batch(n, m, N, T, layout, lam + n, E.dev(F), E.dev(H), E.dev(Q), E.dev(R), E.dev(Wm), E.dev(Wc),
                           E.to_records(zs, layout, 1), dx, dP, mask=torch.as_tensor(mask, device=dx.device), means=means,
                           covs=covs, status=st)
        torch.cuda.synchronize()
        res[tag] = [t.cpu().numpy().copy() for t in (means, covs, dx, dP, st)]
    assert res["one"][4][77] != 0 and not res["one"][4][:77].any()
    for tag in ("3x4", "2x7", "4x23"):
        for a, b in zip(res["one"], res[tag]):
            assert np.array_equal(a, b, equal_nan=True), tag


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,N", [(6, 1037), (6, 1038), (4, 1038), (8, 515), (9, 514), (7, 515), (5, 1038), (3, 1037)])
def test_fused_ukf_smoother_chunked_call_is_bit_identical(n, N, layout, monkeypatch):
    """FK_UKF_RTS_CHUNKS="G,H" cuts fk_ukf_linear_rts_f64 into track groups x backward time windows on helper streams
    (fk_chunks.hpp, ukf_rts_chunked_call); a window's top step is read back from the smoothed outputs of the piece before it:
    xs, ps, Ks and the status must be bit-identical to the single launch -- odd and even banks (register fetch / LDS-DMA
    fetch), a ragged last workgroup, a track whose covariance is not positive definite"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    rs = np.random.RandomState(7 * n + N % 5)
    T = 23
    alpha, beta, kappa = 0.5, 2.0, 3.0 - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    lam = alpha ** 2 * (n + kappa) - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    Q = spd(rs, n, 0.05)
    mu, cov = rs.randn(T, N, n), spd(rs, n, 2.0, (T, N))
    cov[:, 77] = -np.eye(n)                                # not positive definite: status bit, garbage that must not differ
    res = {}
    for tag, env in (("one", "1,1"), ("3x4", "3,4"), ("2x7", "2,7"), ("4x22", "4,22"), ("default", None)):
        if env:
            monkeypatch.setenv("FK_UKF_RTS_CHUNKS", env)
        else:
            monkeypatch.delenv("FK_UKF_RTS_CHUNKS", raising=False)
        Xs, Ps = E.to_records(mu, layout, 1), E.to_records(cov, layout, 1)
        xs, ps = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
        Ks = E.alloc_records((T,), N, n * n, layout)
        for t in (xs, ps, Ks):
            t.fill_(float("nan"))
        st = torch.zeros(N, dtype=torch.int32, device=Xs.device)
        E.ukf_linear_rts(n, N, T, layout, lam + n, E.dev(F), E.dev(Q), E.dev(Wm), E.dev(Wc), Xs, Ps, xs, ps, K=Ks, status=st)
        torch.cuda.synchronize()
        res[tag] = [t.cpu().numpy().copy() for t in (xs, ps, Ks, st)]
    assert res["one"][3][77] != 0 and not res["one"][3][:77].any()
    assert not np.isnan(res["one"][0][:, :77] if layout == "aos" else res["one"][0][..., :77]).any()
    for tag in ("3x4", "2x7", "4x22", "default"):
        for a, b in zip(res["one"], res[tag]):
            assert np.array_equal(a, b, equal_nan=True), tag
