"""fk_imm.hpp (the per-track IMM arithmetic of imm_kernels.hip) compiled for the host, against the
goldens frozen from the live filterpy.kalman.IMMEstimator.  No GPU needed."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, golden, rel_err_rows

TOL = 1e-10


def lib():
    L = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    L.hc_imm_batch.restype = ctypes.c_int
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hc_imm(g, p, n, m, nm, T=None, mmae=False):
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    zs = c(g[p + "zs"]) if T is None else c(g[p + "zs"][:T])
    T = zs.shape[0]
    xs, Ps = c(g[p + "xs0"]).copy(), c(g[p + "Ps0"]).copy()
    mu = c(g[p + "p0"]).copy() if mmae else c(g[p + "mu0"] / g[p + "mu0"].sum())
    M = c(np.eye(nm)) if mmae else c(g[p + "M"])
    x, P, MU = np.zeros((T, n)), np.zeros((T, n, n)), np.zeros((T, nm))
    xp, Pp, L = np.zeros((T, n)), np.zeros((T, n, n)), np.zeros((T, nm))
    st = lib().hc_imm_batch(n, m, nm, ctypes.c_long(T), _p(c(g[p + "Fs"])), _p(c(g[p + "Qs"])), _p(c(g[p + "Hs"])),
                            _p(c(g[p + "Rs"])), _p(M), _p(zs), _p(xs), _p(Ps), _p(mu), _p(x), _p(P),
                            _p(MU), _p(xp), _p(Pp), _p(L), int(mmae))
    return st, x, P, MU, xp, Pp, L, xs, Ps, mu


@pytest.mark.parametrize("n,m,nm", [(2, 1, 2), (4, 2, 2), (4, 2, 3), (6, 3, 2), (3, 2, 2), (5, 2, 3)])
def test_imm_math_vs_golden(n, m, nm):
    g = golden("imm")
    p = f"n{n}m{m}k{nm}_"
    st, x, P, MU, xp, Pp, L, xs, Ps, mu = hc_imm(g, p, n, m, nm)
    assert st == 0
    assert rel_err_rows(x, g[p + "x"]) < TOL and rel_err_rows(P, g[p + "P"]) < TOL
    assert rel_err_rows(xp, g[p + "xp"]) < TOL and rel_err_rows(Pp, g[p + "Pp"]) < TOL
    assert np.allclose(MU, g[p + "mu"], rtol=1e-9, atol=1e-14)
    assert np.allclose(L, g[p + "L"], rtol=1e-9, atol=1e-300)
    assert rel_err_rows(xs, g[p + "xs_final"]) < TOL and rel_err_rows(Ps, g[p + "Ps_final"]) < TOL
    assert np.allclose(mu, g[p + "mu"][-1], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("n,m,nm", [(2, 1, 2), (4, 2, 2), (4, 2, 3), (6, 3, 3), (3, 2, 2), (2, 1, 3)])
def test_mmae_math_vs_golden(n, m, nm):
    """the MMAE branch (no mixing, p *= likelihood, mmae.py's own covariance loop) against goldens
    frozen from the live filterpy.kalman.MMAEFilterBank."""
    g = golden("mmae")
    p = f"n{n}m{m}k{nm}_"
    st, x, P, PR, _, _, L, xs, Ps, pf = hc_imm(g, p, n, m, nm, mmae=True)
    assert st == 0
    assert rel_err_rows(x, g[p + "x"]) < TOL and rel_err_rows(P, g[p + "P"]) < TOL
    assert np.allclose(PR, g[p + "p"], rtol=1e-9, atol=1e-14)
    assert np.allclose(L, g[p + "L"], rtol=1e-9, atol=1e-300)
    assert rel_err_rows(xs, g[p + "xs_final"]) < TOL and rel_err_rows(Ps, g[p + "Ps_final"]) < TOL


@pytest.mark.parametrize("scale,sigmas", [(1e-300, 40.0), (1e-200, 41.0), (1e+250, 20.0), (1.0, 10.0)])
def test_likelihood_keeps_the_reference_range_at_the_ends_of_the_exponent_range(scale, sigmas):
    """the density is (2 pi)^(-m/2) |S|^(-1/2) exp(-q/2); formed as two separate factors, exp(-q/2) underflows to 0
    beyond q ~ 1490 even where a tiny |S| brings the density back into range (and inf * 0 would be NaN), while the reference's
    exp(-(m ln 2 pi + ln |S| + q) / 2) (kalman_filter.py:1213-1226) is finite.  The root's binary exponent now rides inside the
    one exponential: residuals of `sigmas` standard deviations under an S of size `scale` against the oracle."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import imm_oracle
    n, m, nm = 2, 1, 2
    F = np.array([[1.0, 1.0], [0.0, 1.0]])
    g = {"Fs": np.array([F, F]), "Qs": np.array([scale * 1e-3 * np.eye(n), scale * 2e-3 * np.eye(n)]),
         "Hs": np.array([[[1.0, 0.0]]] * nm), "Rs": np.array([[[scale]], [[2.0 * scale]]]),
         "xs0": np.zeros((nm, n)), "Ps0": np.array([scale * np.eye(n)] * nm), "mu0": np.array([0.5, 0.5]),
         "M": np.array([[0.9, 0.1], [0.2, 0.8]])}
    S0 = 2.0 * scale + scale * 1e-3 + scale                 # H (F P F' + Q) H' + R of filter 0 (mixing leaves equal states alone)
    g["zs"] = np.array([[sigmas * np.sqrt(S0)]])
    st, x, P, MU, xp, Pp, L, xs, Ps, mu = hc_imm(g, "", n, m, nm)
    ref = imm_oracle.imm_batch(g["xs0"], g["Ps0"], g["mu0"], g["M"], g["zs"], g["Fs"], g["Qs"], g["Hs"], g["Rs"])
    assert st == 0 and np.all(np.isfinite(L)) and np.all(np.isfinite(MU))
    assert np.all(ref[5] > 2.3e-308), ref[5]                # the case is in range for the reference
    assert np.allclose(L, ref[5], rtol=1e-9, atol=0.0), (L, ref[5])
    assert np.allclose(MU, ref[2], rtol=1e-9, atol=1e-14)
