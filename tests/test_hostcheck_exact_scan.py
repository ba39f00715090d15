"""The exact parallel cumsum (filterpy_amd/csrc/fk_exact_scan.hpp) emulated on the host with the
GPU kernel's tile/thread/wave decomposition must equal numpy.cumsum BIT-FOR-BIT -- that is what
makes the resample indices bit-exact (BASELINE.json north_star)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT


def hc_cumsum(w):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    lib.hc_cumsum_exact.restype = ctypes.c_long
    w = np.ascontiguousarray(w, dtype=np.float64)
    cs = np.empty_like(w)
    segs = lib.hc_cumsum_exact(ctypes.c_long(w.size), w.ctypes.data_as(ctypes.c_void_p), cs.ctypes.data_as(ctypes.c_void_p))
    return cs, segs


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


CASES = {
    "uniform": lambda rs, N: rs.rand(N),
    "normalised": lambda rs, N: (lambda w: w / w.sum())(rs.rand(N)),
    "heavy_tail": lambda rs, N: np.exp(rs.randn(N) * 6.0),
    "sparse": lambda rs, N: rs.rand(N) * (rs.rand(N) < 0.03),
    "leading_zeros": lambda rs, N: np.concatenate([np.zeros(N // 2), rs.rand(N - N // 2)]),
    "tiny": lambda rs, N: rs.rand(N) * 1e-310,                # subnormal sums
    "ties": lambda rs, N: rs.randint(0, 3, N) * 2.0 ** -53 + (rs.rand(N) < 0.01),   # exact half-ulp ties
    "ties_everywhere": lambda rs, N: np.full(N, 2.0 ** -12 + 2.0 ** -54),           # every add in [1/2, 1) is a tie
    "rare_tie": lambda rs, N: np.where(np.arange(N) % 5000 == 4999, 2.0 ** -54, 0.0) + rs.randint(1, 1000, N) * 2.0 ** -40,
    "growing": lambda rs, N: 1.5 ** (np.arange(N) % 900),     # a binade crossing on most adds
    "huge_then_small": lambda rs, N: np.concatenate([[1e300], rs.rand(N - 1)]),
    "onehot": lambda rs, N: np.eye(1, N, N // 3)[0],
    "all_zero": lambda rs, N: np.zeros(N),
    "with_negative": lambda rs, N: rs.randn(N),               # not a valid weight vector, still exact
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("N", [1, 2, 7, 2047, 2048, 2049, 5000, 100003])
def test_exact_cumsum_bitwise(name, N):
    rs = np.random.RandomState(hash(name) % 1000 + N)
    w = np.asarray(CASES[name](rs, N), dtype=np.float64)[:N]
    cs, segs = hc_cumsum(w)
    ref = np.cumsum(w)
    assert np.array_equal(bits(cs), bits(ref)) or np.array_equal(cs, ref), (
        name, N, int(np.argmax(bits(cs) != bits(ref))))
    # zeros may differ in sign only where numpy gives -0.0 + 0.0; everything else bitwise
    nz = ref != 0
    assert np.array_equal(bits(cs)[nz], bits(ref)[nz])


def test_exact_cumsum_large_and_parallel_fraction():
    """2^20 normalised weights: bitwise equal, and almost all of it went through the parallel
    scan (a few dozen binade segments + per-tile restarts), not the serial fall-back."""
    rs = np.random.RandomState(7)
    N = 1 << 20
    w = rs.rand(N)
    w /= w.sum()
    cs, segs = hc_cumsum(w)
    assert np.array_equal(bits(cs), bits(np.cumsum(w)))
    assert segs < N // 2048 + 64


def _tile_search(cs, c_in, ps):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    cs = np.ascontiguousarray(cs, dtype=np.float64)
    ps = np.ascontiguousarray(ps, dtype=np.float64)
    out = np.empty(ps.size, dtype=np.int32)
    lib.hc_tile_search(ctypes.c_int(cs.size), cs.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(c_in),
                       ctypes.c_long(ps.size), ps.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


TILE_CASES = {
    "uniform": lambda r, n: r.random(n),
    "skewed": lambda r, n: r.random(n) ** 12,                       # the interpolated window misses often
    "one_heavy": lambda r, n: np.where(np.arange(n) == n // 3, 1e3, r.random(n)),
    "zeros": lambda r, n: np.where(r.random(n) < 0.7, 0.0, r.random(n)),   # runs of equal sums
    "all_zero": lambda r, n: np.zeros(n),                           # span 0: inv_span = inf / NaN guess
    "steps": lambda r, n: np.where(np.arange(n) % 17 == 0, 1.0, 0.0),
    "tiny": lambda r, n: r.random(n) * 1e-300,
}


@pytest.mark.parametrize("name", sorted(TILE_CASES))
@pytest.mark.parametrize("n", [1, 2, 7, 16, 17, 33, 1000, 2048])
def test_tile_upper_bound_equals_searchsorted(name, n):
    """fk::tile_upper_bound (interpolated window + branch-free steps, or the fallback search) must return
    #{j : cs[j] <= p} for every p -- inside the tile, on the sums themselves, below and above the tile."""
    r = np.random.default_rng(n * 131 + len(name))
    c_in = float(r.random() * 3)
    cs = c_in + np.cumsum(TILE_CASES[name](r, n))
    lo, hi = c_in, cs[-1]
    span = hi - lo if hi > lo else 1.0
    ps = np.concatenate([
        lo + span * r.random(4000),                       # inside
        cs, np.nextafter(cs, -np.inf), np.nextafter(cs, np.inf),       # on and next to every sum
        [lo, np.nextafter(lo, -np.inf), lo - span, hi + span, 0.0],
        np.linspace(lo, hi, 513),                         # evenly spaced, like systematic positions
    ])
    want = np.searchsorted(cs, ps, side="right").astype(np.int32)
    got = _tile_search(cs, c_in, ps)
    bad = np.nonzero(want != got)[0]
    assert bad.size == 0, (name, n, ps[bad[:5]], want[bad[:5]], got[bad[:5]])
