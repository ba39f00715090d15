"""The per-track arithmetic of the HIP kernels (filterpy_amd/csrc/fk_math.hpp), compiled for the
host by the test-only harness tests/hostcheck, against the goldens generated from the live
reference.  Runs in the GPU-less build container; the same templates are instantiated by the
gfx950 kernels (tests/test_gpu_*.py check those on the GPU box)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, golden, rel_err_rows

TOL = 1e-10   # BASELINE.json: x/P within 1e-10 rel fp64

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def hc_batch(x0, P0, zs, F, Q, H, R, mask=None, alpha_sq=1.0, update_first=False):
    n, m = F.shape[0], H.shape[0]
    T = zs.shape[0]
    c = np.ascontiguousarray
    x, P = c(x0, dtype=float).copy(), c(P0, dtype=float).copy()
    mu, cov = np.zeros((T, n)), np.zeros((T, n, n))
    mup, covp = np.zeros((T, n)), np.zeros((T, n, n))
    mk = None if mask is None else c(mask, dtype=np.uint8)
    st = lib().hc_kf_batch(n, m, ctypes.c_long(T), _p(c(F)), _p(c(Q)), _p(c(H)), _p(c(R)), _p(c(zs)), _p(mk),
                           _p(x), _p(P), _p(mu), _p(cov), _p(mup), _p(covp),
                           ctypes.c_double(alpha_sq), int(update_first))
    return mu, cov, mup, covp, x, P, st


def hc_rts(Xs, Ps, F, Q):
    T, n = Xs.shape
    c = np.ascontiguousarray
    xs, Pso, K, Pp = np.zeros((T, n)), np.zeros((T, n, n)), np.zeros((T, n, n)), np.zeros((T, n, n))
    st = lib().hc_rts(n, ctypes.c_long(T), _p(c(F)), _p(c(Q)), _p(c(Xs)), _p(c(Ps)), _p(xs), _p(Pso), _p(K), _p(Pp))
    return xs, Pso, K, Pp, st


DIMS = [tuple(d) for d in golden("kf_dims")["dims"]]


@pytest.mark.parametrize("n,m", DIMS)
@pytest.mark.parametrize("variant", ["plain", "uf", "alpha", "miss"])
def test_kf_batch_vs_golden(n, m, variant):
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    kw = {}
    if variant == "uf":
        kw["update_first"] = True
    if variant == "alpha":
        kw["alpha_sq"] = 1.02 ** 2
    if variant == "miss":
        kw["mask"] = g[p + "mask"]
    mu, cov, mup, covp, xf, Pf, st = hc_batch(g[p + "x0"], g[p + "P0"], g[p + "zs"], g[p + "F"], g[p + "Q"],
                                              g[p + "H"], g[p + "R"], **kw)
    assert st == 0
    for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
        assert rel_err_rows(got, g[p + variant + "_" + key]) < TOL, key


@pytest.mark.parametrize("n,m", DIMS)
def test_rts_vs_golden(n, m):
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    xs, Ps, K, Pp, st = hc_rts(g[p + "plain_mu"], g[p + "plain_cov"], g[p + "F"], g[p + "Q"])
    assert st == 0
    for got, key in ((xs, "rts_x"), (Ps, "rts_P"), (K, "rts_K"), (Pp, "rts_Pp")):
        assert rel_err_rows(got, g[p + key]) < 1e-9, key   # n x n solve: cond(Pp) enters


def test_c1_vs_golden():
    g = golden("kf_c1")
    mu, cov, mup, covp, *_ = hc_batch(np.zeros(2), g["P0"], g["zs"].reshape(-1, 1), g["F"], g["Q"], g["H"], g["R"])
    for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
        assert rel_err_rows(got, g["1d_" + key]) < TOL


def test_sigma_points_vs_golden():
    g = golden("ukf_merwe")
    for ci, (n, m, alpha, beta, kappa) in enumerate(g["cases"]):
        n = int(n)
        lam = alpha ** 2 * (n + kappa) - n
        sig = np.zeros((2 * n + 1, n))
        x0, P0 = np.ascontiguousarray(g[f"c{ci}_x0"]), np.ascontiguousarray(g[f"c{ci}_P0"])
        st = lib().hc_sigma(n, ctypes.c_double(lam + n), _p(x0), _p(P0), _p(sig))
        assert st == 0
        assert rel_err_rows(sig, g[f"c{ci}_sigmas"]) < 1e-12


def hc_batch_sym(x0, P0, zs, F, Q, H, R, mask=None, alpha_sq=1.0):
    n, m = F.shape[0], H.shape[0]
    T = zs.shape[0]
    c = np.ascontiguousarray
    x, P = c(x0, dtype=float).copy(), c(P0, dtype=float).copy()
    mu, cov = np.zeros((T, n)), np.zeros((T, n, n))
    mup, covp = np.zeros((T, n)), np.zeros((T, n, n))
    mk = None if mask is None else c(mask, dtype=np.uint8)
    st = lib().hc_kf_batch_sym(n, m, ctypes.c_long(T), _p(c(F)), _p(c(Q)), _p(c(H)), _p(c(R)), _p(c(zs)), _p(mk),
                               _p(x), _p(P), _p(mu), _p(cov), _p(mup), _p(covp), ctypes.c_double(alpha_sq))
    return mu, cov, mup, covp, x, P, st


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (4, 2), (6, 3), (9, 3)])
@pytest.mark.parametrize("variant", ["plain", "alpha", "miss"])
def test_kf_packed_symmetric_vs_golden(n, m, variant):
    """fk_math_sym.hpp (packed symmetric P, row-streamed Joseph form) -- the arithmetic of the fast
    kernel -- against the live-reference goldens."""
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    kw = {}
    if variant == "alpha":
        kw["alpha_sq"] = 1.02 ** 2
    if variant == "miss":
        kw["mask"] = g[p + "mask"]
    mu, cov, mup, covp, xf, Pf, st = hc_batch_sym(g[p + "x0"], g[p + "P0"], g[p + "zs"], g[p + "F"], g[p + "Q"],
                                                  g[p + "H"], g[p + "R"], **kw)
    assert st == 0
    for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
        assert rel_err_rows(got, g[p + variant + "_" + key]) < TOL, key


def test_packed_symmetric_c1_1000_steps():
    g = golden("kf_c1")
    mu, cov, mup, covp, *_ = hc_batch_sym(np.zeros(2), g["P0"], g["zs"].reshape(-1, 1), g["F"], g["Q"], g["H"], g["R"])
    for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
        assert rel_err_rows(got, g["1d_" + key]) < TOL


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (4, 2), (6, 3), (9, 3)])
def test_rts_packed_symmetric_vs_golden(n, m):
    """rts_step_sym (packed symmetric, row-streamed) -- the arithmetic of the exact-dim RTS kernels."""
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    Xs, Ps = g[p + "plain_mu"], g[p + "plain_cov"]
    T = Xs.shape[0]
    c = np.ascontiguousarray
    xs, Pso, K, Pp = np.zeros((T, n)), np.zeros((T, n, n)), np.zeros((T, n, n)), np.zeros((T, n, n))
    st = lib().hc_rts_sym(n, ctypes.c_long(T), _p(c(g[p + "F"])), _p(c(g[p + "Q"])), _p(c(Xs)), _p(c(Ps)),
                          _p(xs), _p(Pso), _p(K), _p(Pp))
    assert st == 0
    for got, key in ((xs, "rts_x"), (Pso, "rts_P"), (K, "rts_K"), (Pp, "rts_Pp")):
        assert rel_err_rows(got, g[p + key]) < 1e-9, key


def test_det_from_reciprocal_pivots_over_the_whole_exponent_range():
    """logdet_from_dinv / rsqrt_det_from_dinv (fk_math.hpp; the IMM likelihood and the Saver's log-likelihood): one logarithm /
    no logarithm of the PRODUCT of the reciprocal pivots, carried as mantissa x 2^exponent -- against sums of logarithms in
    extended precision, for pivots from 1e-300 to 1e300 whose plain product would leave the range, odd and even exponent sums."""
    import ctypes
    from conftest import ROOT
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    rs = np.random.RandomState(11)
    worst_l = worst_r = 0.0
    for trial in range(4000):
        m = 1 + trial % 4
        lo, hi = [(-300, 300), (-5, 5), (-300, -200), (200, 300)][(trial // 4) % 4]
        dinv = 10.0 ** rs.uniform(lo, hi, size=m)
        ld, rd = ctypes.c_double(), ctypes.c_double()
        lib.hc_det_from_dinv(ctypes.c_int(m), dinv.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ld), ctypes.byref(rd))
        ref_ld = -float(np.sum(np.log(dinv.astype(np.longdouble))))                 # ln |S| = -sum ln (1 / d_i)
        assert abs(ld.value - ref_ld) <= 4e-16 * max(1.0, abs(ref_ld)) + 1e-15, (dinv, ld.value, ref_ld)
        ref_r = np.exp(np.longdouble(-0.5) * np.longdouble(ref_ld))
        if 1e-300 < ref_r < 1e300:
            worst_r = max(worst_r, abs(float(rd.value / ref_r) - 1.0))
        elif ref_r <= 1e-300:                                                        # down to the subnormals: to an ulp of those
            assert abs(rd.value - float(ref_r)) <= max(1e-13 * float(ref_r), 1e-323), (dinv, rd.value, ref_r)
        else:
            assert np.isinf(rd.value) or abs(float(rd.value / ref_r) - 1.0) < 1e-12, (dinv, rd.value, ref_r)
        worst_l = max(worst_l, abs(ld.value - ref_ld) / max(1.0, abs(ref_ld)))
    assert worst_r < 1e-13, worst_r


def _rts_bars(g, p):
    """max(1e-10, 2 * spread) per output; for the gain also 8 x the reference's own distance from the exactly rounded gain
    (tests/golden/make_rts_conditioning.py says why)"""
    sp = g[p + "spread"]
    bar = {k: max(TOL, 2.0 * float(v)) for k, v in zip(("xs", "Ps", "K", "Pp"), sp)}
    bar["K"] = max(bar["K"], 8.0 * float(g[p + "K_ref_err"]))
    return bar


@pytest.mark.parametrize("ci", range(13))
def test_rts_margin_on_badly_conditioned_models(ci):
    """the smoother's margin.  On models whose backward recursion is badly conditioned (cond(Pp)
    1e2 .. 1e9: tools/bench_configs.py's unstable random models, where the several-lane kernel measured 6.9e-11, and integrator
    chains) the kernels' arithmetic -- the LDL' solve of Pp where the reference calls numpy.linalg.inv -- stays inside
    max(1e-10, 2 x the reference's own spread under one-ulp input perturbations); the gain also inside 8 x the reference's own
    distance from the exactly rounded gain, against the reference and against that gain itself."""
    g = golden("rts_conditioning")
    p = f"c{ci}_"
    xs, Ps, K, Pp, st = hc_rts(g[p + "mu"], g[p + "cov"], g[p + "F"], g[p + "Q"])
    assert st == 0
    bar = _rts_bars(g, p)
    assert rel_err_rows(xs, g[p + "xs"]) < bar["xs"] and rel_err_rows(Ps, g[p + "Ps"]) < bar["Ps"]
    assert rel_err_rows(K[:-1], g[p + "K"][:-1]) < bar["K"] and rel_err_rows(K[:-1], g[p + "K_exact"][:-1]) < bar["K"]
    assert rel_err_rows(Pp[:-1], g[p + "Pp"][:-1]) < bar["Pp"]
