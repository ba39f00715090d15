"""The arithmetic of the four-lanes-per-track fused linear UKF (filterpy_amd/csrc/fk_ukf_quad.hpp: ukf_quad_step_v4 -- the very
function ukf_mlg.hip runs, there with DPP quad exchanges) compiled for the host with the four lanes of a quad as four fibers
in lockstep (tests/hostcheck/hostcheck_quad.cpp), held against the oracle's UKF (oracle/ukf_oracle.py, pinned to the reference by
tests/test_oracle_ukf.py) with fx = F x, hx = H x, and against the one-lane step it distributes (ukf_linear_step_v4)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, _build

sys.path.insert(0, ROOT)
from oracle import ukf_oracle  # noqa: E402

HC = os.path.join(ROOT, "tests", "hostcheck")


@pytest.fixture(scope="module")
def quad_lib():
    so, src = os.path.join(HC, "libhostcheck_quad.so"), os.path.join(HC, "hostcheck_quad.cpp")
    deps = [src] + [os.path.join(ROOT, "filterpy_amd", "csrc", h) for h in ("fk_ukf_quad.hpp", "fk_ukf.hpp", "fk_math.hpp", "fk_math_sym.hpp")]
    _build(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-w", "-o", so, src], so, deps)
    return ctypes.CDLL(so)


def _run(lib, entry, n, m, F, H, Q, R, Wm, Wc, scale, zs, mask, x0, P0, want_status=0):
    T = zs.shape[0]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    F, H, Q, R, Wm, Wc, zs = map(c, (F, H, Q, R, Wm, Wc, zs))
    x, P = c(x0).copy(), c(P0).copy()
    means, covs = np.full((T, n), np.nan), np.full((T, n, n), np.nan)
    mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    st = getattr(lib, entry)(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_long(T), p(F), p(H), p(Q), p(R), p(Wm), p(Wc),
                              ctypes.c_double(scale), p(zs), p(mk), p(x), p(P), p(means), p(covs))
    assert st == want_status, st
    return means, covs, x, P


def _model(n, m, seed):
    r = np.random.default_rng(seed)
    F = np.eye(n) + 0.05 * np.triu(r.standard_normal((n, n)), 1) + 0.01 * np.tril(r.standard_normal((n, n)), -1)
    H = np.eye(m, n) + 0.1 * r.standard_normal((m, n))
    A = r.standard_normal((n, n))
    Q = 0.01 * np.eye(n) + 0.002 * A @ A.T
    B = r.standard_normal((m, m))
    R = 0.5 * np.eye(m) + 0.05 * B @ B.T
    C = r.standard_normal((n, n))
    return r, F, H, Q, R, r.standard_normal(n), 10.0 * np.eye(n) + 0.3 * C @ C.T


rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))  # noqa: E731

DIMS = [(4, 2), (5, 2), (7, 3), (8, 4), (9, 3)] + [(n, m) for n in range(10, 17) for m in range(1, 9)]


@pytest.mark.parametrize("n,m", DIMS)
@pytest.mark.parametrize("abk", [(.1, 2., None), (1e-3, 2., 0.), (1., 2., .1)])
def test_quad_step_matches_the_oracle(quad_lib, n, m, abk):
    """UKF.batch_filter (UKF.py:364-491, 524-632) through the oracle against the distributed step: rows of P on four lanes, the
    factor's rows broadcast, every dim_x 10..16 x dim_z 1..4 (and small dims where slots are duplicates or the quad is not
    full); a full initial covariance, a non-triangular F."""
    alpha, beta, kappa = abk
    kappa = 3. - n if kappa is None else kappa
    r, F, H, Q, R, x0, P0 = _model(n, m, n * 10 + m)
    T = 25
    zs = r.standard_normal((T, m))
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R,
                                                  alpha, beta, kappa)
    mu, cov, xf, Pf = _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, Wm, Wc, lam + n, zs, None, x0, P0)
    # Merwe's cancelling weights (Wm0 ~ -1e6 at alpha = 1e-3) amplify rounding: the bars of tests/test_hostcheck_ukf.py
    tol = 1e-9 if alpha > 1e-2 else 1e-6
    assert rel(mu, mu_ref) < tol and rel(cov, cov_ref) < tol
    assert np.array_equal(xf, mu[-1]) and np.array_equal(Pf, cov[-1])
    # a row and its mirror image come from different lanes: equal to a rounding of the sums, not bit for bit
    assert rel(cov, np.swapaxes(cov, 1, 2)) < (1e-13 if alpha > 1e-2 else 1e-8)


@pytest.mark.parametrize("n,m", [(4, 2), (5, 2), (7, 3), (8, 4), (9, 3)])
def test_quad_step_agrees_with_the_one_lane_step_it_distributes(quad_lib, n, m):
    """Same sums in the same order (fk_ukf.hpp, ukf_linear_step_v4); the differences: a row's own elements feed the factor
    instead of the upper triangle's, and the mirrored elements of P are computed twice."""
    one = ctypes.CDLL(os.path.join(HC, "libhostcheck.so"))
    r, F, H, Q, R, x0, P0 = _model(n, m, 77 + n)
    zs = r.standard_normal((30, m))
    alpha, beta, kappa = .3, 2., 3. - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    scale = alpha ** 2 * (n + kappa)
    a = _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, Wm, Wc, scale, zs, None, x0, P0)
    b = _run(one, "hc_ukf_linear_v4", n, m, F, H, Q, R, Wm, Wc, scale, zs, None, x0, P0)
    assert rel(a[0], b[0]) < 1e-13 and rel(a[1], b[1]) < 1e-13


@pytest.mark.parametrize("n,m", [(10, 2), (13, 3), (16, 4), (12, 5), (15, 8)])
def test_quad_step_missing_measurements_skip_the_update(quad_lib, n, m):
    """update(None) (UKF.py:462-466): the step runs its update half on z = 0 with the gain selected to zero -- x and P must come
    out as the prior, bit for bit what the next predict sees in the reference's flow."""
    r, F, H, Q, R, x0, P0 = _model(n, m, 5 + n)
    T = 12
    zs = r.standard_normal((T, m))
    mask = (np.arange(T) % 3 != 1).astype(np.uint8)
    alpha, beta, kappa = .1, 2., -1.
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    zl = [z if k else None for z, k in zip(zs, mask)]
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0, P0, zl, lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R,
                                                  alpha, beta, kappa)
    zs_bad = zs.copy()
    zs_bad[mask == 0] = np.nan                                  # a masked measurement is never read
    mu, cov, _, _ = _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, Wm, Wc, lam + n, zs_bad, mask, x0, P0)
    assert rel(mu, mu_ref) < 1e-9 and rel(cov, cov_ref) < 1e-9


def test_quad_step_reports_a_covariance_that_is_not_positive_definite(quad_lib):
    n, m = 12, 2
    r, F, H, Q, R, x0, P0 = _model(n, m, 3)
    P0 = P0.copy()
    P0[5, 5] = -1.0
    Wm, Wc = ukf_oracle.merwe_weights(n, .5, 2., 0.)
    _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, Wm, Wc, .25 * n, r.standard_normal((2, m)), None, x0, P0, want_status=1)


def test_quad_step_julier_weights(quad_lib):
    """JulierSigmaPoints (sigma_points.py:358-372): equal weights within every pair, a different centre weight."""
    n, m, kappa = 11, 3, 1.5
    r, F, H, Q, R, x0, P0 = _model(n, m, 8)
    zs = r.standard_normal((15, m))
    W = np.full(2 * n + 1, .5 / (n + kappa))
    W[0] = kappa / (n + kappa)
    # the oracle's Merwe form with alpha = 1, beta = 0 is Julier's set: lambda = kappa, Wc0 = Wm0 + (1 - 1 + 0)
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, 1., 0., kappa)
    Wm, Wc = ukf_oracle.merwe_weights(n, 1., 0., kappa)
    assert np.allclose(Wm, W) and np.allclose(Wc, W)
    mu, cov, _, _ = _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, W, W, n + kappa, zs, None, x0, P0)
    assert rel(mu, mu_ref) < 1e-9 and rel(cov, cov_ref) < 1e-9


def _rts(lib, entry, n, F, Q, Wm, Wc, scale, Xs, Ps):
    T = Xs.shape[0]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    F, Q, Wm, Wc, Xs, Ps = map(c, (F, Q, Wm, Wc, Xs, Ps))
    xs, ps, Ks = np.full((T, n), np.nan), np.full((T, n, n), np.nan), np.full((T, n, n), np.nan)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    st = getattr(lib, entry)(ctypes.c_int(n), ctypes.c_long(T), p(F), p(Q), p(Wm), p(Wc), ctypes.c_double(scale),
                              p(Xs), p(Ps), p(xs), p(ps), p(Ks))
    assert st == 0, st
    return xs, ps, Ks


relrows = lambda a, b: float(np.max(np.max(np.abs(a - b).reshape(len(a), -1), axis=1) / np.max(np.abs(b).reshape(len(b), -1), axis=1)))  # noqa: E731


@pytest.mark.parametrize("entry", ["hc_ukf_quad_rts_v4", "hc_ukf_quad_rts_park_v4"])
@pytest.mark.parametrize("n", [4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("abk", [(.1, 2., None), (1., 2., .1)])
def test_quad_smoother_step_matches_the_oracle(quad_lib, n, abk, entry):
    """UKF.rts_smoother (UKF.py:714-739) through the oracle against the distributed backward step (ukf_quad_rts_step_v4: the
    gain's forward substitution fused into the factorisation of Pb, the rows of Pn - Pb and of K broadcast by their owners)."""
    alpha, beta, kappa = abk
    kappa = 3. - n if kappa is None else kappa
    m = 2
    r, F, H, Q, R, x0, P0 = _model(n, m, n * 10 + 3)
    T = 20
    zs = r.standard_normal((T, m))
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mu, cov = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, alpha, beta, kappa)
    xr, Pr, Kr = ukf_oracle.ukf_rts_smoother(mu, cov, lambda s, d: F @ s, 0.1, Q, alpha, beta, kappa)
    xs, ps, Ks = _rts(quad_lib, entry, n, F, Q, Wm, Wc, alpha ** 2 * (n + kappa), mu, cov)
    if entry.endswith("park_v4"):                              # parking moves values, not bits
        plain = _rts(quad_lib, "hc_ukf_quad_rts_v4", n, F, Q, Wm, Wc, alpha ** 2 * (n + kappa), mu, cov)
        assert all(np.array_equal(u, v) for u, v in zip((xs, ps, Ks), plain))
    assert relrows(xs, xr) < 1e-10 and relrows(ps, Pr) < 1e-10 and relrows(Ks[:-1], Kr[:-1]) < 1e-10
    assert np.array_equal(xs[-1], mu[-1]) and np.array_equal(ps[-1], cov[-1]) and not Ks[-1].any()
    assert relrows(ps, np.swapaxes(ps, 1, 2)) < 1e-13


@pytest.mark.parametrize("n", [4, 5, 7, 8, 9])
def test_quad_smoother_step_agrees_with_the_one_lane_step(quad_lib, n):
    one = ctypes.CDLL(os.path.join(HC, "libhostcheck.so"))
    r, F, H, Q, R, x0, P0 = _model(n, 2, 50 + n)
    zs = r.standard_normal((25, 2))
    alpha, beta, kappa = .3, 2., 3. - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mu, cov = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, alpha, beta, kappa)
    a = _rts(quad_lib, "hc_ukf_quad_rts_v4", n, F, Q, Wm, Wc, alpha ** 2 * (n + kappa), mu, cov)
    b = _rts(one, "hc_ukf_linear_rts_v4", n, F, Q, Wm, Wc, alpha ** 2 * (n + kappa), mu, cov)
    for u, v in zip(a, b):
        assert relrows(u[:-1], v[:-1]) < 1e-12


def test_quad_step_is_as_accurate_as_the_one_lane_step_on_ill_conditioned_covariances(quad_lib):
    """Covariances of condition 1e3 .. 1e7, states of size 1e-2 .. 1e3: the error against the oracle is the problem's own
    sensitivity (cond x eps), and the distributed step's must be the one-lane step's -- same sums, same order -- not worse."""
    one = ctypes.CDLL(os.path.join(HC, "libhostcheck.so"))
    worst = 0.0
    for seed in range(25):
        r = np.random.default_rng(1000 + seed)
        n, m = [(7, 3), (8, 4), (9, 3), (5, 2), (4, 2)][seed % 5]
        U, _ = np.linalg.qr(r.standard_normal((n, n)))
        P0 = (U * np.geomspace(1, 10.0 ** r.uniform(3, 7), n)) @ U.T
        P0 = (P0 + P0.T) / 2
        F = np.eye(n) + 0.2 * r.standard_normal((n, n))
        F /= max(1, 1.1 * np.max(np.abs(np.linalg.eigvals(F))))
        H, A, B = r.standard_normal((m, n)), r.standard_normal((n, n)), r.standard_normal((m, m))
        Q, R = 1e-3 * (A @ A.T / n + 0.1 * np.eye(n)), 0.3 * (B @ B.T / m + 0.2 * np.eye(m))
        x0, zs = r.standard_normal(n) * 10 ** r.uniform(-2, 3), r.standard_normal((15, m)) * 10
        alpha, kappa = float(r.choice([1.0, 0.5, 0.1, 0.3])), float(r.choice([0.0, 1.0]))
        Wm, Wc = ukf_oracle.merwe_weights(n, alpha, 2.0, kappa)
        mu, cov = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, alpha, 2.0, kappa)
        a = _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, Wm, Wc, alpha ** 2 * (n + kappa), zs, None, x0, P0)
        b = _run(one, "hc_ukf_linear_v4", n, m, F, H, Q, R, Wm, Wc, alpha ** 2 * (n + kappa), zs, None, x0, P0)
        eq, eo = max(rel(a[0], mu), rel(a[1], cov)), max(rel(b[0], mu), rel(b[1], cov))
        worst = max(worst, eq / max(eo, 1e-13))
    assert worst < 5.0, worst


@pytest.mark.parametrize("n", [8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_oct_smoother_step_matches_the_oracle_and_the_quad(quad_lib, n):
    """The same backward step on EIGHT lanes per track (two row slots per lane at dim_x 16; the exchange is still "the value lane
    o of my group holds"): against the oracle at 1e-10, and against the four-lane run -- same sums in the same order, only the
    ownership of the rows differs, so the two agree to the last bits of a few sums that rows computed by different lanes enter."""
    alpha, beta, kappa = .3, 2., 3. - n
    r, F, H, Q, R, x0, P0 = _model(n, 2, n * 7 + 1)
    zs = r.standard_normal((18, 2))
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mu, cov = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, alpha, beta, kappa)
    xr, Pr, Kr = ukf_oracle.ukf_rts_smoother(mu, cov, lambda s, d: F @ s, 0.1, Q, alpha, beta, kappa)
    a = _rts(quad_lib, "hc_ukf_oct_rts_v4", n, F, Q, Wm, Wc, alpha ** 2 * (n + kappa), mu, cov)
    b = _rts(quad_lib, "hc_ukf_quad_rts_v4", n, F, Q, Wm, Wc, alpha ** 2 * (n + kappa), mu, cov)
    assert relrows(a[0], xr) < 1e-10 and relrows(a[1], Pr) < 1e-10 and relrows(a[2][:-1], Kr[:-1]) < 1e-10
    for u, v in zip(a, b):
        assert relrows(u[:-1], v[:-1]) < 1e-12
    assert np.array_equal(a[0][-1], mu[-1]) and np.array_equal(a[1][-1], cov[-1]) and not a[2][-1].any()


@pytest.mark.parametrize("n,m", [(8, 4), (9, 3)] + [(n, m) for n in (10, 13, 14, 15, 16) for m in (1, 3, 4, 5, 6, 7, 8)])
def test_oct_filter_step_matches_the_oracle_and_the_quad(quad_lib, n, m):
    """The filter step on eight lanes per track against the oracle (1e-9, the package's UKF bar on this model) and against
    the four-lane run (same sums, same order; rows owned by other lanes)."""
    alpha, beta, kappa = .4, 2., 3. - n
    r, F, H, Q, R, x0, P0 = _model(n, m, n * 13 + m)
    T = 14
    zs = r.standard_normal((T, m))
    mask = np.ones(T, dtype=np.uint8)
    mask[5] = mask[6] = 0
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    zl = [z if k else None for z, k in zip(zs, mask)]
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0, P0, zl, lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, alpha, beta, kappa)
    a = _run(quad_lib, "hc_ukf_oct_v4", n, m, F, H, Q, R, Wm, Wc, alpha ** 2 * (n + kappa), zs, mask, x0, P0)
    b = _run(quad_lib, "hc_ukf_quad_v4", n, m, F, H, Q, R, Wm, Wc, alpha ** 2 * (n + kappa), zs, mask, x0, P0)
    assert rel(a[0], mu_ref) < 1e-9 and rel(a[1], cov_ref) < 1e-9
    assert rel(a[0], b[0]) < 1e-12 and rel(a[1], b[1]) < 1e-12
    assert np.array_equal(a[2], a[0][-1]) and np.array_equal(a[3], a[1][-1])
