"""-m gpu: the fused linear UKF on four / eight lanes per track (csrc/ukf_mlg.hip: filter at dim_x 10..16 with dim_z 1..8,
smoother at 7..16) through the C ABI against the oracle's per-filter loop (UKF.py:364-491, 524-632, 634-739) and the live-reference
goldens.  The kernel's arithmetic (csrc/fk_ukf_quad.hpp) is held against the oracle on the host by tests/test_hostcheck_ukf_quad.py.
Round 5's first lease ran this file with the kernels switched on (profiles/r05/ukf_mlg/: 181 cases green); they are the default now."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden, rel_err_rows

pytestmark = [pytest.mark.gpu]
TOL = 1e-10


def spd(rs, n, scale=1.0, batch=()):
    A = rs.randn(*batch, n, n)
    return scale * (A @ np.swapaxes(A, -1, -2) / n + 0.5 * np.eye(n))


def _bank(n, m, N, T, layout, seed, mask_every=0, alpha=.5):
    """runs fk_ukf_linear_batch_f64 on a bank of different tracks; returns everything the checks need"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    rs = np.random.RandomState(seed)
    beta, kappa = 2.0, 3.0 - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), spd(rs, n, 0.05), spd(rs, m, 0.5)
    x0, P0 = rs.randn(N, n), spd(rs, n, 2.0, (N,))
    zs = rs.randn(T, N, m)
    mask = np.ones((T, N), dtype=np.uint8)
    if mask_every:
        mask[rs.rand(T, N) < 1.0 / mask_every] = 0
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    scale = alpha ** 2 * (n + kappa)
    dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.full((N,), -1, dtype=torch.int32, device=dx.device)
    zz = np.where(mask[..., None] != 0, zs, np.nan)             # a masked measurement is never used
    E.ukf_linear_batch(n, m, N, T, layout, scale, E.dev(F), E.dev(H), E.dev(Q), E.dev(R), E.dev(Wm), E.dev(Wc),
                       E.to_records(zz, layout, 1), dx, dP, mask=None if mask.all() else torch.as_tensor(mask, device=dx.device),
                       means=means, covs=covs, status=st, paired=True)
    mu, cov = E.from_records(means, layout, 1, (n,)), E.from_records(covs, layout, 1, (n, n))
    xe, Pe = E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n))

    def ref(trk):
        zt = [zs[t, trk] if mask[t, trk] else None for t in range(T)]
        return ukf_oracle.ukf_batch_filter(x0[trk], P0[trk], zt, lambda x, dt: F @ x, lambda x: H @ x, 1.0, Q, R, alpha, beta, kappa)
    return mu, cov, xe, Pe, st.cpu().numpy(), ref


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(n, m) for n in range(10, 17) for m in range(1, 9)])
def test_every_instantiation_vs_oracle(n, m, layout):
    """every (dim_x, dim_z) the file instantiates, both layouts: a bank that ends inside a wave (a quad-duplicated tail), a few
    missing measurements; each checked track against the oracle's loop at 1e-10."""
    N, T = 150, 7
    mu, cov, xe, Pe, st, ref = _bank(n, m, N, T, layout, 100 * n + m, mask_every=9)
    assert not st.any(), st[st != 0]
    for trk in (0, 1, 15, 16, 63, 64, 143, 144, N - 1):
        rmu, rcov = ref(trk)
        assert rel_err_rows(mu[:, trk], rmu) < TOL and rel_err_rows(cov[:, trk], rcov) < TOL, trk
    assert np.array_equal(xe, mu[-1]) and np.array_equal(Pe, cov[-1])
    assert rel_err_rows(cov.reshape(-1, n, n), np.swapaxes(cov, -1, -2).reshape(-1, n, n)) < 1e-13


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 63, 65, 1000, 4099])
def test_bank_sizes(N, layout):
    """one track, an odd tail (the element-major copy-out's 8-byte unit), whole and partial waves and workgroups"""
    n, m, T = 13, 3, 5
    mu, cov, xe, Pe, st, ref = _bank(n, m, N, T, layout, 7 + N)
    assert not st.any()
    for trk in sorted({0, N // 3, N // 2, N - 2 if N > 1 else 0, N - 1}):
        rmu, rcov = ref(trk)
        assert rel_err_rows(mu[:, trk], rmu) < TOL and rel_err_rows(cov[:, trk], rcov) < TOL, (N, trk)
    assert np.all(np.isfinite(mu)) and np.all(np.isfinite(cov))


def test_all_tracks_of_a_bank():
    """every track of a 600-track bank (sampling tracks would miss a lane mapping that is wrong for some quads only)"""
    n, m, N, T = 16, 4, 600, 3
    for layout in ("soa", "aos"):
        mu, cov, xe, Pe, st, ref = _bank(n, m, N, T, layout, 99)
        assert not st.any()
        for trk in range(N):
            rmu, rcov = ref(trk)
            assert rel_err_rows(mu[:, trk], rmu) < TOL and rel_err_rows(cov[:, trk], rcov) < TOL, (layout, trk)


def test_small_alpha():
    """Merwe's alpha = 1e-3 (weights of 1e6 that cancel, Wm[0] ~ -1e6): the reference's own outputs move by ~2e-9 per step under a
    re-ordering of its sums there (tests/golden/ukf_conditioning.json); the bar of tests/test_hostcheck_ukf.py for that set"""
    n, m, N, T = 12, 2, 40, 6
    mu, cov, xe, Pe, st, ref = _bank(n, m, N, T, "soa", 5, alpha=1e-3)
    assert not st.any()
    tol = 1e-6
    for trk in (0, 17, N - 1):
        rmu, rcov = ref(trk)
        assert rel_err_rows(mu[:, trk], rmu) < tol and rel_err_rows(cov[:, trk], rcov) < tol, trk


def test_status_bits():
    """a covariance that is not positive definite -> FK_STATUS_NOT_PD on that track only; weights that are not equal within a
    pair although the caller's flag says so -> FK_STATUS_BAD_WEIGHTS everywhere"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    n, m, N, T = 11, 2, 70, 2
    rs = np.random.RandomState(3)
    F, H, Q, R = np.eye(n), rs.randn(m, n), 0.01 * np.eye(n), np.eye(m)
    x0, P0 = rs.randn(N, n), spd(rs, n, 2.0, (N,))
    P0[33, 4, 4] = -1.0
    Wm, Wc = ukf_oracle.merwe_weights(n, .5, 2., 0.)
    for bad_w in (False, True):
        wc = Wc.copy()
        if bad_w:
            wc[3] *= 1.5
        dx, dP = E.to_records(x0, "soa", 0), E.to_records(P0, "soa", 0)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ukf_linear_batch(n, m, N, T, "soa", .25 * n, E.dev(F), E.dev(H), E.dev(Q), E.dev(R), E.dev(Wm), E.dev(wc),
                           E.to_records(rs.randn(T, N, m), "soa", 1), dx, dP, status=st, paired=True)
        s = st.cpu().numpy()
        if bad_w:
            assert np.all(s & 16)
        else:
            assert s[33] & 1 and not np.delete(s, 33).any()


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_python_api_routes_matrix_models_here(layout):
    """UnscentedKalmanFilter.batch_filter with matrix fx / hx at dim_x >= 10 against the live-reference goldens (ukf_dims.npz),
    and the by-products the reference leaves on the filter after the last epoch"""
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    g = golden("ukf_dims")
    done = 0
    for ci, c in enumerate(g["cases"]):
        n, m, alpha, beta, kappa = int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4])
        if n < 10:
            continue
        p, N = f"c{ci}_", 37
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=g[p + "H"], fx=g[p + "F"], points=MerweScaledSigmaPoints(n, alpha, beta, kappa),
                                    n_tracks=N, layout=layout)
        tile = lambda a: np.tile(a, (N,) + (1,) * np.ndim(a))  # noqa: E731
        ukf.x, ukf.P, ukf.Q, ukf.R = tile(g[p + "x0"]), tile(g[p + "P0"]), g[p + "Q"].copy(), g[p + "R"].copy()
        zs = np.tile(g[p + "zs"][:, None, :], (1, N, 1))
        mu, cov = ukf.batch_filter(list(zs))
        tol = TOL
        for trk in (0, 16, N - 1):
            assert rel_err_rows(mu[:, trk], g[p + "mu"]) < tol and rel_err_rows(cov[:, trk], g[p + "cov"]) < tol, (n, m, trk)
        assert np.array_equal(ukf.x, mu[-1]) and np.array_equal(ukf.P, cov[-1])
        done += 1
    assert done >= 7


# ------------------------------------------------------------------------------------------------ the smoother
def _smooth(n, N, T, layout, seed, in_place=False, with_K=True):
    """fk_ukf_linear_rts_f64 at dim_x 10..16 on the oracle's filter output of N different tracks"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    rs = np.random.RandomState(seed)
    m, alpha, beta, kappa = 2, .5, 2.0, 3.0 - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), spd(rs, n, 0.05), spd(rs, m, 0.5)
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mus, covs = np.empty((T, N, n)), np.empty((T, N, n, n))
    # a few distinct filter histories, dealt out over the bank (the oracle's loop is slow), each scaled per track
    base = []
    for b in range(min(N, 5)):
        base.append(ukf_oracle.ukf_batch_filter(rs.randn(n), spd(rs, n, 2.0), list(rs.randn(T, m)), lambda x, dt: F @ x,
                                                lambda x: H @ x, 1.0, Q, R, alpha, beta, kappa))
    pick = rs.randint(0, len(base), size=N)
    for i in range(N):
        mus[:, i], covs[:, i] = base[pick[i]][0], base[pick[i]][1]
    dX, dP = E.to_records(mus, layout, 1), E.to_records(covs, layout, 1)
    oxs = dX if in_place else E.alloc_records((T,), N, n, layout)
    ops = dP if in_place else E.alloc_records((T,), N, n * n, layout)
    oK = E.alloc_records((T,), N, n * n, layout) if with_K else None
    st = torch.full((N,), -1, dtype=torch.int32, device=dX.device)
    E.ukf_linear_rts(n, N, T, layout, alpha ** 2 * (n + kappa), E.dev(F), E.dev(Q), E.dev(Wm), E.dev(Wc), dX, dP, oxs, ops, oK, st, paired=True)
    assert not st.cpu().numpy().any()
    xs, ps = E.from_records(oxs, layout, 1, (n,)), E.from_records(ops, layout, 1, (n, n))
    Ks = E.from_records(oK, layout, 1, (n, n)) if with_K else None
    refs = [ukf_oracle.ukf_rts_smoother(b[0], b[1], lambda x, dt: F @ x, 1.0, Q, alpha, beta, kappa) for b in base]
    return xs, ps, Ks, pick, refs, mus, covs


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n", range(10, 17))
def test_smoother_every_dim_vs_oracle(n, layout):
    """UKF.rts_smoother (UKF.py:634-739) at every dim_x of the file, a bank that ends inside a wave: EVERY track against the
    oracle's backward pass of its history at 1e-10; the last step is the filter's own output, its gain zero."""
    N, T = 150, 8
    xs, ps, Ks, pick, refs, mus, covs = _smooth(n, N, T, layout, 40 + n)
    for trk in range(N):
        rx, rP, rK = refs[pick[trk]]
        assert rel_err_rows(xs[:, trk], rx) < TOL and rel_err_rows(ps[:, trk], rP) < TOL, trk
        assert rel_err_rows(Ks[:-1, trk], rK[:-1]) < TOL, trk
    assert np.array_equal(xs[-1], mus[-1]) and np.array_equal(ps[-1], covs[-1]) and not Ks[-1].any()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("N", [1, 2, 15, 17, 65, 1000])
def test_smoother_bank_sizes_and_without_gain(N, layout):
    """ragged banks; K == NULL (its copy-out is issued against a descriptor of zero tracks)"""
    n, T = 14, 6
    a = _smooth(n, N, T, layout, 3 + N)
    b = _smooth(n, N, T, layout, 3 + N, with_K=False)
    for trk in sorted({0, N // 2, N - 1}):
        rx, rP, rK = a[4][a[3][trk]]
        assert rel_err_rows(a[0][:, trk], rx) < TOL and rel_err_rows(a[1][:, trk], rP) < TOL, (N, trk)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_python_api_smoother_routes_here(layout):
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    g = golden("ukf_dims")
    done = 0
    for ci, c in enumerate(g["cases"]):
        n, m, alpha, beta, kappa = int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4])
        if n < 10:
            continue
        p, N = f"c{ci}_", 21
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=g[p + "H"], fx=g[p + "F"], points=MerweScaledSigmaPoints(n, alpha, beta, kappa),
                                    n_tracks=N, layout=layout)
        ukf.Q = g[p + "Q"].copy()
        Xs, Ps = np.tile(g[p + "mu"][:, None], (1, N, 1)), np.tile(g[p + "cov"][:, None], (1, N, 1, 1))
        xs, ps, Ks = ukf.rts_smoother(Xs, Ps)
        for trk in (0, 16, N - 1):
            assert rel_err_rows(xs[:, trk], g[p + "rts_x"]) < TOL and rel_err_rows(ps[:, trk], g[p + "rts_P"]) < TOL, (n, trk)
            assert rel_err_rows(Ks[:-1, trk], g[p + "rts_K"][:-1]) < TOL, (n, trk)
        done += 1
    assert done >= 5


# ------------------------------------------------------------------- dim_x 7..9 on the four-lane kernels (A/B switch)
# The smoother at 7..9 runs on them by default (pair-weight callers: measured faster than the one-lane classes, profiles/r05/ukf_mlg/);
# the FILTER at 7..9 stays with the one-lane classes (faster there) and reaches the several-lane kernels only through the A/B
# knob FK_UKF_MLG_MIN_NX=7, which the library reads once per process: those cases run in an interpreter of their own.
_AB = os.environ.get("FK_UKF_MLG_MIN_NX") == "7"


def test_small_dims_filter_on_the_four_lane_kernels_in_its_own_process():
    if _AB:
        return                  # (this IS the inner run)
    env = dict(os.environ, FK_UKF_MLG_MIN_NX="7")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-p", "no:cacheprovider",
                        "-k", "small_dims_filter_vs_oracle"], capture_output=True, text=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "12 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(7, 1), (7, 3), (8, 2), (8, 4), (9, 3), (9, 4)])
def _small_dims_filter_vs_oracle(n, m, layout):
    N, T = 200, 7
    mu, cov, xe, Pe, st, ref = _bank(n, m, N, T, layout, 100 * n + m, mask_every=9)
    assert not st.any()
    for trk in (0, 15, 16, 63, 64, 191, 192, N - 1):
        rmu, rcov = ref(trk)
        assert rel_err_rows(mu[:, trk], rmu) < TOL and rel_err_rows(cov[:, trk], rcov) < TOL, trk
    assert np.array_equal(xe, mu[-1]) and np.array_equal(Pe, cov[-1])


if _AB:            # collected only in the inner run: no skipped entries in the driver's report
    test_small_dims_filter_vs_oracle = _small_dims_filter_vs_oracle


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n", [7, 8, 9])
def test_small_dims_smoother_vs_oracle(n, layout):
    N, T = 200, 8
    xs, ps, Ks, pick, refs, mus, covs = _smooth(n, N, T, layout, 40 + n)
    for trk in range(N):
        rx, rP, rK = refs[pick[trk]]
        assert rel_err_rows(xs[:, trk], rx) < TOL and rel_err_rows(ps[:, trk], rP) < TOL, trk
        assert rel_err_rows(Ks[:-1, trk], rK[:-1]) < TOL, trk
