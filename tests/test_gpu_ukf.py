"""-m gpu parity tests: sigma-point / unscented-transform kernels and the fused linear UKF,
through the C ABI, against goldens frozen from the live reference (UKF.py, sigma_points.py,
unscented_transform.py)."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows, ukf_tol

pytestmark = pytest.mark.gpu
TOL = 1e-10
N = 200


def _cases():
    g = golden("ukf_merwe")
    return [(ci, int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4])) for ci, c in enumerate(g["cases"])]


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_sigma_points_and_ut(layout):
    import torch
    from filterpy_amd import _engine as E
    from gpu_util import tile_tracks
    g = golden("ukf_merwe")
    for ci, n, m, alpha, beta, kappa in _cases():
        p = f"c{ci}_"
        lam = alpha ** 2 * (n + kappa) - n
        k = 2 * n + 1
        dx, dP = E.to_records(tile_tracks(g[p + "x0"], N), layout, 0), E.to_records(tile_tracks(g[p + "P0"], N), layout, 0)
        sig = E.alloc_records((), N, k * n, layout)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ut_sigma_points(n, N, layout, lam + n, dx, dP, sig, st)
        torch.cuda.synchronize()
        assert not st.any()
        got = E.from_records(sig, layout, 0, (k, n))
        for trk in (0, 64, N - 1):
            assert rel_err_rows(got[trk], g[p + "sigmas"]) < 1e-12, (ci, trk)
        # unscented transform of those sigma points with noise Q
        xo, Po = E.alloc_records((), N, n, layout), E.alloc_records((), N, n * n, layout)
        E.ut_transform(n, k, N, layout, sig, E.dev(g[p + "Wm"]), E.dev(g[p + "Wc"]), E.dev(g[p + "Q"]), xo, Po)
        torch.cuda.synchronize()
        gx, gP = E.from_records(xo, layout, 0, (n,)), E.from_records(Po, layout, 0, (n, n))
        # UT cancels badly by design (Wm0 ~ -199 at n=6): compare against the reference's own result
        assert rel_err_rows(gx[[0, N - 1]], np.tile(g[p + "ut_x"], (2, 1))) < 1e-10, ci
        assert rel_err_rows(gP[[0, N - 1]], np.tile(g[p + "ut_P"], (2, 1, 1))) < 1e-10, ci


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_julier_sigma_points(layout):
    import torch
    from filterpy_amd import _engine as E
    from gpu_util import tile_tracks
    g = golden("ukf_merwe")
    n, kappa = 4, 0.5
    dx, dP = E.to_records(tile_tracks(g["jul_x0"], N), layout, 0), E.to_records(tile_tracks(g["jul_P0"], N), layout, 0)
    sig = E.alloc_records((), N, (2 * n + 1) * n, layout)
    E.ut_sigma_points(n, N, layout, n + kappa, dx, dP, sig)
    torch.cuda.synchronize()
    got = E.from_records(sig, layout, 0, (2 * n + 1, n))
    assert rel_err_rows(got[7], g["jul_sigmas"]) < 1e-12


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_cross_variance_vs_oracle(layout):
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    rs = np.random.RandomState(5)
    n, m, k = 6, 3, 13
    sf, sh = rs.randn(N, k, n), rs.randn(N, k, m)
    x, z, Wc = rs.randn(N, n), rs.randn(N, m), rs.randn(k)
    out = E.alloc_records((), N, n * m, layout)
    E.ut_cross_variance(n, m, k, N, layout, E.to_records(x, layout, 0), E.to_records(z, layout, 0),
                        E.to_records(sf, layout, 0), E.to_records(sh, layout, 0), E.dev(Wc), out)
    torch.cuda.synchronize()
    got = E.from_records(out, layout, 0, (n, m))
    for trk in (0, 63, 64, N - 1):
        ref = ukf_oracle.cross_variance(x[trk], z[trk], sf[trk], sh[trk], Wc)
        assert rel_err_rows(got[trk][None], ref[None]) < 1e-12


@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_fused_linear_ukf_goldens(layout, paired):
    """UnscentedKalmanFilter.batch_filter with fx = F x, hx = H x (UKF.py:524-632).  paired: the sums regrouped over the +-
    pairs of sigma points (FK_UKF_FLAG_PAIR_WEIGHTS; what the Python API asks for with Merwe's / Julier's weights) and the
    reference's index-order sums -- both held to the live-reference golden at the same bar."""
    import torch
    from filterpy_amd import _engine as E
    from gpu_util import tile_tracks
    g = golden("ukf_merwe")
    for ci, n, m, alpha, beta, kappa in _cases():
        p = f"c{ci}_"
        lam = alpha ** 2 * (n + kappa) - n
        zs = g[p + "zs"]
        T = zs.shape[0]
        dx, dP = E.to_records(tile_tracks(g[p + "x0"], N), layout, 0), E.to_records(tile_tracks(g[p + "P0"], N), layout, 0)
        dz = E.to_records(tile_tracks(zs, N, 1), layout, 1)
        means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ukf_linear_batch(n, m, N, T, layout, lam + n, E.dev(g[p + "F"]), E.dev(g[p + "H"]), E.dev(g[p + "Q"]),
                           E.dev(g[p + "R"]), E.dev(g[p + "Wm"]), E.dev(g[p + "Wc"]), dz, dx, dP,
                           means=means, covs=covs, status=st, paired=paired)
        torch.cuda.synchronize()
        assert not st.any(), ci
        assert E.pair_weights(g[p + "Wm"], g[p + "Wc"], n)
        mu, cov = E.from_records(means, layout, 1, (n,)), E.from_records(covs, layout, 1, (n, n))
        # 1e-10 everywhere except where the reference's OWN result moves more than that under one-ulp input
        # perturbations (alpha = 1e-3, Wm0 ~ -1e6: mu 2.3e-9 in the reference; tests/golden/ukf_conditioning.json)
        for trk in (0, 64, N - 1):
            assert rel_err_rows(mu[:, trk], g[p + "mu"]) < ukf_tol(ci, "mu"), (ci, trk)
            assert rel_err_rows(cov[:, trk], g[p + "cov"]) < ukf_tol(ci, "cov"), (ci, trk)


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_pair_weights_flag_is_checked_and_other_weights_keep_the_index_order(layout):
    """FK_UKF_FLAG_PAIR_WEIGHTS is the caller's assertion that the weights are equal within every +- pair; the kernels check it
    (FK_STATUS_BAD_WEIGHTS -> ValueError).  A weight set that is NOT symmetric (legal for unscented_transform.py:104-126: any
    Wm / Wc) is detected by _engine.pair_weights, runs the index-order sums and matches the oracle's transform chain."""
    import torch
    from filterpy_amd import _engine as E
    from gpu_util import tile_tracks
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ukf_oracle
    g = golden("ukf_merwe")
    ci, n, m, alpha, beta, kappa = [c for c in _cases() if c[1] == 4][-1]
    p = f"c{ci}_"
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = g[p + "Wm"].copy(), g[p + "Wc"].copy()
    Wm[2] *= 1.25
    Wm[n + 2] -= Wm[2] - g[p + "Wm"][2]            # still sums to the same total
    Wc[1] *= 0.5
    assert not E.pair_weights(Wm, Wc, n)
    zs = g[p + "zs"][:6]
    T = zs.shape[0]
    nb = 70

    def run(paired, with_smoother=False):
        dx, dP = E.to_records(tile_tracks(g[p + "x0"], nb), layout, 0), E.to_records(tile_tracks(g[p + "P0"], nb), layout, 0)
        dz = E.to_records(tile_tracks(zs, nb, 1), layout, 1)
        means, covs = E.alloc_records((T,), nb, n, layout), E.alloc_records((T,), nb, n * n, layout)
        st = torch.zeros(nb, dtype=torch.int32, device=dx.device)
        E.ukf_linear_batch(n, m, nb, T, layout, lam + n, E.dev(g[p + "F"]), E.dev(g[p + "H"]), E.dev(g[p + "Q"]),
                           E.dev(g[p + "R"]), E.dev(Wm), E.dev(Wc), dz, dx, dP, means=means, covs=covs, status=st, paired=paired)
        torch.cuda.synchronize()
        if with_smoother:
            xs, ps = E.alloc_records((T,), nb, n, layout), E.alloc_records((T,), nb, n * n, layout)
            st2 = torch.zeros(nb, dtype=torch.int32, device=dx.device)
            E.ukf_linear_rts(n, nb, T, layout, lam + n, E.dev(g[p + "F"]), E.dev(g[p + "Q"]), E.dev(Wm), E.dev(Wc), means, covs,
                             xs, ps, status=st2, paired=paired)
            torch.cuda.synchronize()
            return st, st2
        return st, E.from_records(means, layout, 1, (n,)), E.from_records(covs, layout, 1, (n, n))

    st, st2 = run(True, with_smoother=True)
    assert (st.cpu().numpy() & 16).all() and (st2.cpu().numpy() & 16).all()
    with pytest.raises(ValueError):
        E.raise_on_status(st, "fused UKF")
    st, mu, cov = run(None)
    assert not st.any()
    # the oracle's predict / update (UKF.py:400-411, 462-481) with these weights
    F, H, Q, R = g[p + "F"], g[p + "H"], g[p + "Q"], g[p + "R"]
    x, P = g[p + "x0"].copy(), g[p + "P0"].copy()
    for t in range(T):
        x, P, sf = ukf_oracle.ukf_predict(x, P, lambda s_, d_: F @ s_, 0.1, Q, Wm, Wc, alpha, kappa)
        x, P, _, _, _ = ukf_oracle.ukf_update(x, P, sf, zs[t], lambda s_: H @ s_, R, Wm, Wc)
        for trk in (0, nb - 1):
            assert rel_err_rows(mu[t, trk][None], x[None]) < TOL and rel_err_rows(cov[t, trk][None], P[None]) < TOL, t


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("nb", [1, 2, 3, 63, 64, 65, 130, 257, 513])
def test_fused_linear_ukf_smoother_goldens_every_bank_size(layout, nb, paired):
    """fk_ukf_linear_rts_f64 (UKF.py:634-739) on the reference's own filter output, banks of every shape of tail: one track,
    odd and even counts, one short of / one past a wave and a workgroup.  First, last and a middle track of every bank
    against the live-reference golden -- the LDS-DMA fetch of the exact classes reads 16-byte units, and an odd track
    count makes the last unit of the last element row straddle the end of the array (found by
    test_gpu_variants.py::test_ukf_rts_smoother_goldens on a single filter: such banks take the register fetch)."""
    import torch
    from filterpy_amd import _engine as E
    from gpu_util import tile_tracks
    g = golden("ukf_merwe")
    for ci, n, m, alpha, beta, kappa in _cases():
        if not E.ukf_linear_rts_supported(n):
            continue
        p = f"c{ci}_"
        lam = alpha ** 2 * (n + kappa) - n
        mu, cov = g[p + "mu"], g[p + "cov"]
        T = mu.shape[0]
        Xs = E.to_records(tile_tracks(mu, nb, 1), layout, 1)
        Ps = E.to_records(tile_tracks(cov, nb, 1), layout, 1)
        xs, ps = E.alloc_records((T,), nb, n, layout), E.alloc_records((T,), nb, n * n, layout)
        Ks = E.alloc_records((T,), nb, n * n, layout)
        st = torch.zeros(nb, dtype=torch.int32, device=Xs.device)
        E.ukf_linear_rts(n, nb, T, layout, lam + n, E.dev(g[p + "F"]), E.dev(g[p + "Q"]), E.dev(g[p + "Wm"]),
                         E.dev(g[p + "Wc"]), Xs, Ps, xs, ps, K=Ks, status=st, paired=paired)
        torch.cuda.synchronize()
        assert not st.any(), (ci, nb)
        hx, hP = E.from_records(xs, layout, 1, (n,)), E.from_records(ps, layout, 1, (n, n))
        hK = E.from_records(Ks, layout, 1, (n, n))
        for trk in sorted({0, nb // 2, nb - 1}):
            assert rel_err_rows(hx[:, trk], g[p + "rts_x"]) < ukf_tol(ci, "rts_x"), (ci, nb, trk)
            assert rel_err_rows(hP[:, trk], g[p + "rts_P"]) < ukf_tol(ci, "rts_P"), (ci, nb, trk)
            assert rel_err_rows(hK[:-1, trk], g[p + "rts_K"][:-1]) < ukf_tol(ci, "rts_K"), (ci, nb, trk)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_fused_linear_ukf_smoother_on_arrays_that_are_only_8_byte_aligned(layout):
    """The LDS-DMA fetch of fk_ukf_linear_rts_f64 moves 16-byte units; arrays that start 8 bytes into an allocation (a view of a
    larger tensor) must take the register fetch and give the same bits."""
    import torch
    from filterpy_amd import _engine as E
    from gpu_util import tile_tracks
    g = golden("ukf_merwe")
    ci, n, m, alpha, beta, kappa = [c for c in _cases() if c[1] == 6][0]
    p = f"c{ci}_"
    lam = alpha ** 2 * (n + kappa) - n
    nb = 130
    mu, cov = g[p + "mu"], g[p + "cov"]
    T = mu.shape[0]
    Xs0 = E.to_records(tile_tracks(mu, nb, 1), layout, 1)
    Ps0 = E.to_records(tile_tracks(cov, nb, 1), layout, 1)
    res = []
    for shift in (0, 1):
        def place(t):
            buf = torch.empty(t.numel() + 2, dtype=t.dtype, device=t.device)
            v = buf[shift:shift + t.numel()].view(t.shape)
            v.copy_(t)
            assert v.data_ptr() % 16 == 8 * shift
            return v
        Xs, Ps = place(Xs0), place(Ps0)
        xs, ps = E.alloc_records((T,), nb, n, layout), E.alloc_records((T,), nb, n * n, layout)
        Ks = E.alloc_records((T,), nb, n * n, layout)
        st = torch.zeros(nb, dtype=torch.int32, device=Xs.device)
        E.ukf_linear_rts(n, nb, T, layout, lam + n, E.dev(g[p + "F"]), E.dev(g[p + "Q"]), E.dev(g[p + "Wm"]),
                         E.dev(g[p + "Wc"]), Xs, Ps, xs, ps, K=Ks, status=st)
        torch.cuda.synchronize()
        assert not st.any()
        res.append([t.cpu().numpy().copy() for t in (xs, ps, Ks)])
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    hx = E.from_records(torch.as_tensor(res[1][0]), layout, 1, (n,))
    assert rel_err_rows(hx[:, nb - 1], g[p + "rts_x"]) < ukf_tol(ci, "rts_x")


@pytest.mark.parametrize("mask", [False, True])
@pytest.mark.parametrize("n,m", [(2, 2), (4, 2), (6, 3)])
def test_element_major_pair_stores_are_bit_identical(n, m, mask, monkeypatch):
    """Round 4: in the element-major layout the fused UKF's per-step outputs leave as 16-byte stores of two element rows each
    (wave_store_soa_pairs, the SP instantiations; the last partial workgroup runs in the same launch, its missing track pairs
    dropped by an out-of-range offset).  Only the store instructions differ: every output equals the 8-byte-store kernel (FK_UKF_SOA_PAIRS=0) bit for bit,
    bank sizes with and without a partial workgroup, odd banks (which take the plain kernel), missing measurements."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    rs = np.random.RandomState(11 * n + m)
    T = 7
    alpha, beta, kappa = .1, 2., 3. - n
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    F = np.eye(n) + 0.05 * np.triu(rs.randn(n, n), 1)
    H = np.eye(m, n) + 0.1 * rs.randn(m, n)
    Q, R = 0.01 * np.eye(n), 0.5 * np.eye(m)
    for N in (256, 778, 1024 + 130, 511):
        x0, P0 = rs.randn(N, n), np.tile(5.0 * np.eye(n), (N, 1, 1))
        zs = rs.randn(T, N, m)
        mk = (rs.rand(T, N) > 0.25).astype(np.uint8) if mask else None
        res = {}
        for tag, env in (("pairs", None), ("plain", "0")):
            if env is None:
                monkeypatch.delenv("FK_UKF_SOA_PAIRS", raising=False)
            else:
                monkeypatch.setenv("FK_UKF_SOA_PAIRS", env)
            dx, dP = E.to_records(x0, "soa", 0), E.to_records(P0, "soa", 0)
            means, covs = E.alloc_records((T,), N, n, "soa"), E.alloc_records((T,), N, n * n, "soa")
            means.fill_(float("nan"))
            covs.fill_(float("nan"))
            st = torch.zeros(N, dtype=torch.int32, device=dx.device)
            E.ukf_linear_batch(n, m, N, T, "soa", lam + n, E.dev(F), E.dev(H), E.dev(Q), E.dev(R), E.dev(Wm), E.dev(Wc),
                               E.to_records(zs, "soa", 1), dx, dP, mask=None if mk is None else torch.as_tensor(mk, device=dx.device),
                               means=means, covs=covs, status=st, paired=True)
            torch.cuda.synchronize()
            assert not st.any()
            res[tag] = [t.cpu().numpy() for t in (means, covs, dx, dP)]
        for a, b in zip(res["pairs"], res["plain"]):
            assert np.array_equal(a, b), (n, m, N, mask)
        assert np.all(np.isfinite(res["pairs"][0])) and np.all(np.isfinite(res["pairs"][1]))


@pytest.mark.parametrize("mask", [False, True])
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_persistent_grid_of_the_fused_ukf_is_bit_identical(layout, mask, monkeypatch):
    """Round 6 (FK_UKF_PERSIST=1; measured slower than the single launch and therefore off by default -- the finding is in
    csrc/ukf_kernels.hip): a whole-bank (6,3) call with pair weights, more workgroups than CUs and at least 32 steps on a persistent
    grid drawing tickets -- track groups x time chunks, the state handed from chunk to chunk through an element-major block
    (csrc/ukf_kernels.hip, PERS).  Same arithmetic per track: every output, the final state and the status equal the single
    launch's (FK_UKF_PERSIST=0) bit for bit -- ragged bank (a partial last workgroup; odd bank: the 8-byte stores), forced
    chunk counts incl. one that does not divide T, missing measurements, a non-SPD track whose status bit must survive."""
    import torch
    from filterpy_amd import _engine as E, _abi
    from oracle import ukf_oracle
    n, m, T = 6, 3, 41
    rs = np.random.RandomState(77)
    alpha, beta, kappa = .1, 2., 3. - n
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    F = np.eye(n) + 0.05 * np.triu(rs.randn(n, n), 1)
    H = np.eye(m, n) + 0.1 * rs.randn(m, n)
    Q, R = 0.01 * np.eye(n), 0.5 * np.eye(m)
    for N in (70_000, 66_001):
        x0, P0 = rs.randn(N, n), np.tile(5.0 * np.eye(n), (N, 1, 1))
        P0[N // 3] = -np.eye(n)                      # not positive definite: FK_STATUS_NOT_PD on that track, every variant
        zs = rs.randn(T, N, m)
        mk = (rs.rand(T, N) > 0.25).astype(np.uint8) if mask else None
        res = {}
        for tag, env in (("single", {"FK_UKF_PERSIST": "0"}), ("tickets", {"FK_UKF_PERSIST": "1"}), ("h7", {"FK_UKF_PERSIST": "1", "FK_UKF_PERSIST_H": "7"}),
                         ("h2", {"FK_UKF_PERSIST": "1", "FK_UKF_PERSIST_H": "2"})):
            for k in ("FK_UKF_PERSIST", "FK_UKF_PERSIST_H"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
            means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
            means.fill_(float("nan"))
            covs.fill_(float("nan"))
            st = torch.zeros(N, dtype=torch.int32, device=dx.device)
            E.ukf_linear_batch(n, m, N, T, layout, lam + n, E.dev(F), E.dev(H), E.dev(Q), E.dev(R), E.dev(Wm), E.dev(Wc),
                               E.to_records(zs, layout, 1), dx, dP, mask=None if mk is None else torch.as_tensor(mk, device=dx.device),
                               means=means, covs=covs, status=st, paired=True)
            torch.cuda.synchronize()
            res[tag] = [t.cpu().numpy() for t in (means, covs, dx, dP, st)]
        sth = res["single"][4]
        assert sth[N // 3] & _abi.FK_STATUS_NOT_PD and np.count_nonzero(sth) == 1
        for tag in ("tickets", "h7", "h2"):
            for a, b in zip(res["single"], res[tag]):
                assert np.array_equal(a, b, equal_nan=True), (layout, mask, N, tag)
