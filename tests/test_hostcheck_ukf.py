"""The arithmetic of the fused linear UKF kernels (filterpy_amd/csrc/fk_ukf.hpp: ukf_linear_step_v3, ukf_linear_rts_gain_v3 /
_correct -- the very functions ukf_kernels.hip runs per lane) compiled for the host and held against the oracle's UKF
(oracle/ukf_oracle.py, pinned to the reference by tests/test_oracle_ukf.py) with fx = F x, hx = H x."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import ukf_oracle  # noqa: E402


def _v3(n, m, F, H, Q, R, Wm, Wc, scale, zs, mask, x0, P0, entry="hc_ukf_linear_v3"):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    T = zs.shape[0]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    F, H, Q, R, Wm, Wc, zs = map(c, (F, H, Q, R, Wm, Wc, zs))
    x, P = c(x0).copy(), c(P0).copy()
    means, covs = np.empty((T, n)), np.empty((T, n, n))
    mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    st = getattr(lib, entry)(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_long(T), p(F), p(H), p(Q), p(R), p(Wm), p(Wc),
                              ctypes.c_double(scale), p(zs), p(mk), p(x), p(P), p(means), p(covs))
    assert st == 0, st
    return means, covs, x, P


@pytest.mark.parametrize("n,m,entry", [(n, m, e) for e in ("hc_ukf_linear_v3", "hc_ukf_linear_v4") for n, m in
                                       [(2, 2), (3, 1), (4, 2), (5, 2), (6, 3), (7, 3), (8, 4), (9, 3), (9, 4)]])
@pytest.mark.parametrize("abk", [(.1, 2., None), (1e-3, 2., 0.), (1., 2., .1)])
def test_ukf_step_matches_the_oracle(n, m, entry, abk):
    """ukf_linear_step_v3 (images of the sigma points formed from the image of the factor) on the host against the oracle's
    UKF.batch_filter (UKF.py:364-491, 524-632)."""
    alpha, beta, kappa = abk
    kappa = 3. - n if kappa is None else kappa
    r = np.random.default_rng(n * 10 + m)
    T = 40
    F = np.eye(n) + 0.05 * np.triu(r.standard_normal((n, n)), 1)
    H = np.eye(m, n) + 0.1 * r.standard_normal((m, n))
    A = r.standard_normal((n, n))
    Q = 0.01 * np.eye(n) + 0.002 * A @ A.T
    B = r.standard_normal((m, m))
    R = 0.5 * np.eye(m) + 0.05 * B @ B.T
    x0 = r.standard_normal(n)
    P0 = 10.0 * np.eye(n)
    zs = r.standard_normal((T, m))
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R,
                                                  alpha, beta, kappa)
    mu, cov, xf, Pf = _v3(n, m, F, H, Q, R, Wm, Wc, lam + n, zs, None, x0, P0, entry)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))  # noqa: E731
    # Merwe's cancelling weights (Wm0 ~ -1e6 at alpha = 1e-3) amplify rounding: the package's UKF bar is 1e-9
    tol = 1e-9 if alpha > 1e-2 else 1e-6
    assert rel(mu, mu_ref) < tol and rel(cov, cov_ref) < tol
    assert np.allclose(xf, mu[-1]) and np.allclose(Pf, cov[-1])
    assert np.allclose(cov, np.swapaxes(cov, 1, 2))


@pytest.mark.parametrize("entry", ["hc_ukf_linear_v3", "hc_ukf_linear_v4"])
def test_ukf_missing_measurements_skip_the_update(entry):
    n, m, T = 4, 2, 12
    r = np.random.default_rng(5)
    F = np.eye(n) + 0.05 * np.triu(r.standard_normal((n, n)), 1)
    H = np.eye(m, n)
    Q, R = 0.01 * np.eye(n), 0.5 * np.eye(m)
    x0, P0 = r.standard_normal(n), 10.0 * np.eye(n)
    zs = r.standard_normal((T, m))
    mask = (np.arange(T) % 3 != 1).astype(np.uint8)
    alpha, beta, kappa = .1, 2., -1.
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    zl = [z if k else None for z, k in zip(zs, mask)]
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0, P0, zl, lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R,
                                                  alpha, beta, kappa)
    mu, cov, _, _ = _v3(n, m, F, H, Q, R, Wm, Wc, lam + n, zs, mask, x0, P0, entry)
    assert np.max(np.abs(mu - mu_ref)) / np.max(np.abs(mu_ref)) < 1e-9
    assert np.max(np.abs(cov - cov_ref)) / np.max(np.abs(cov_ref)) < 1e-9


def _rts(n, F, Q, Wm, Wc, scale, Xs, Ps, entry="hc_ukf_linear_rts_v3"):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    T = Xs.shape[0]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    F, Q, Wm, Wc, Xs, Ps = map(c, (F, Q, Wm, Wc, Xs, Ps))
    xs, ps, Ks = np.empty((T, n)), np.empty((T, n, n)), np.empty((T, n, n))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    st = getattr(lib, entry)(ctypes.c_int(n), ctypes.c_long(T), p(F), p(Q), p(Wm), p(Wc), ctypes.c_double(scale),
                               p(Xs), p(Ps), p(xs), p(ps), p(Ks))
    assert st == 0, st
    return xs, ps, Ks


@pytest.mark.parametrize("n,m,entry", [(n, 2, e) for e in ("hc_ukf_linear_rts_v3", "hc_ukf_linear_rts_v4") for n in range(2, 10)])
@pytest.mark.parametrize("abk", [(.1, 2., None), (1., 2., .1)])
def test_fused_ukf_smoother_step_matches_the_oracle(n, m, entry, abk):
    """fk_ukf.hpp ukf_linear_rts_gain_v3 / _correct (the arithmetic of fk_ukf_linear_rts_f64) on the host against the
    oracle's UKF.rts_smoother (UKF.py:714-739; oracle pinned to the live reference by tests/test_oracle_ukf.py)."""
    alpha, beta, kappa = abk
    kappa = 3. - n if kappa is None else kappa
    r = np.random.default_rng(n * 10 + m + 1)
    T = 30
    F = np.eye(n) + 0.05 * np.triu(r.standard_normal((n, n)), 1)
    H = np.eye(m, n) + 0.1 * r.standard_normal((m, n))
    A = r.standard_normal((n, n))
    Q = 0.01 * np.eye(n) + 0.002 * A @ A.T
    R = 0.5 * np.eye(m)
    x0, P0 = r.standard_normal(n), 10.0 * np.eye(n)
    zs = r.standard_normal((T, m))
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    mu, cov = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), lambda s, d: F @ s, lambda s: H @ s, 0.1, Q, R, alpha, beta, kappa)
    xr, Pr, Kr = ukf_oracle.ukf_rts_smoother(mu, cov, lambda s, d: F @ s, 0.1, Q, alpha, beta, kappa)
    xs, ps, Ks = _rts(n, F, Q, Wm, Wc, lam + n, mu, cov, entry)
    rel = lambda a, b: float(np.max(np.max(np.abs(a - b).reshape(len(a), -1), axis=1) / np.max(np.abs(b).reshape(len(b), -1), axis=1)))  # noqa: E731
    assert rel(xs, xr) < 1e-10 and rel(ps, Pr) < 1e-10 and rel(Ks[:-1], Kr[:-1]) < 1e-10
    assert np.array_equal(xs[-1], mu[-1]) and np.array_equal(ps[-1], cov[-1]) and not Ks[-1].any()
