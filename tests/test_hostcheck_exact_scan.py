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


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("N", [2049, 40000, 300007])
def test_chunk_parallel_plan_bitwise(name, N):
    """The chunk-parallel plan/chain (approximate binade guess + verified O(1) chain steps) must give
    the same bits as the sequential sum for every kind of input, including the ones it cannot
    shortcut (negatives, zeros, subnormals)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    lib.hc_cumsum_chunked.restype = ctypes.c_long
    rs = np.random.RandomState(hash(name) % 1000 + N)
    w = np.ascontiguousarray(np.asarray(CASES[name](rs, N), dtype=np.float64)[:N])
    cs = np.empty_like(w)
    nshort = ctypes.c_long(0)
    nch = lib.hc_cumsum_chunked(ctypes.c_long(N), w.ctypes.data_as(ctypes.c_void_p), cs.ctypes.data_as(ctypes.c_void_p),
                                ctypes.byref(nshort))
    ref = np.cumsum(w)
    nz = ref != 0
    assert np.array_equal(cs, ref) and np.array_equal(bits(cs)[nz], bits(ref)[nz]), (name, N)
    if name in ("uniform", "normalised") and N >= 300000:
        assert nshort.value > 0.8 * nch        # almost every chunk took the O(1) step


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


def _lean_vs_general(w, cin, started=1, prelude=0, first_chunk=0):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    w = np.ascontiguousarray(w, dtype=np.float64)
    g, l = np.empty_like(w), np.full_like(w, np.nan)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    todo = lib.hc_tile_lean_vs_general(ctypes.c_int(w.size), p(w), ctypes.c_double(cin), ctypes.c_int(started),
                                       ctypes.c_int(prelude), ctypes.c_int(first_chunk), p(g), p(l))
    return todo, g, l


@pytest.mark.parametrize("n", [2048, 2047, 1000, 1])
def test_lean_output_route_is_the_general_one_or_declines(n):
    """resample_chunk_lean_kernel either reproduces tile_cumsum_exact bit for bit or flags the chunk for the
    general kernel -- and it must decline exactly the chunks it cannot prove: state not started, prelude
    pending, first chunk, a half-ulp tie, a sum leaving the binade, an invalid weight."""
    r = np.random.default_rng(n)
    accepted = 0
    for trial in range(60):
        w = r.random(n) / 8e6
        cin = float(r.random() * 0.9 + 0.05)                   # somewhere inside a binade, or crossing it
        todo, g, l = _lean_vs_general(w, cin)
        ref = cin + 0.0
        want = np.empty(n)
        for j in range(n):                                     # plain sequential fp64 adds = numpy.cumsum
            ref = ref + w[j]
            want[j] = ref
        assert np.array_equal(bits(g), bits(want))
        crosses = np.frexp(want[-1])[1] != np.frexp(cin)[1]
        if todo == 0:
            accepted += 1
            assert not crosses
            assert np.array_equal(bits(l), bits(want))
        else:
            assert crosses or _has_tie(w, cin)
    assert accepted > 30
    w = r.random(n) / 8e6
    assert _lean_vs_general(w, 0.3, started=0)[0] == 1
    assert _lean_vs_general(w, 0.3, prelude=5)[0] == 1
    assert _lean_vs_general(w, 0.3, first_chunk=1)[0] == 1
    assert _lean_vs_general(w, 0.0)[0] == 1
    assert _lean_vs_general(w, np.inf)[0] == 1
    for badv in (-1e-9, np.nan, np.inf):
        wb = w.copy()
        wb[n // 2] = badv
        assert _lean_vs_general(wb, 0.3)[0] == 1
    # an exact half-ulp tie: ulp(0.3) = 2^-54, so 1.5 ulp has a remainder of exactly half an ulp
    wt = w.copy()
    wt[n // 3] = 1.5 * 2.0 ** -54
    assert _lean_vs_general(wt, 0.3)[0] == 1


def _has_tie(w, cin):
    u = np.spacing(cin)
    t = w / u
    return bool(np.any(t - np.floor(t) == 0.5))


@pytest.mark.parametrize("name", sorted(TILE_CASES))
@pytest.mark.parametrize("n", [1, 9, 100, 2048])
def test_search_then_walk_equals_searchsorted(name, n):
    """The output scheme of the experimental lean kernel (one search per 8 consecutive slots, then walks over the
    guarded tile) returns #{j : cs[j] <= p} for every non-decreasing run of positions, also when many slots
    fall on one element or many elements lie between two slots."""
    r = np.random.default_rng(n * 7 + len(name))
    c_in = float(r.random() * 3)
    cs = c_in + np.cumsum(TILE_CASES[name](r, n))
    lo, hi = c_in, cs[-1]
    span = hi - lo if hi > lo else 1.0
    for ps in (np.sort(lo + span * r.random(4001)),                     # random, sorted
               np.linspace(lo, hi, 1237),                                # evenly spaced like systematic positions
               np.sort(np.concatenate([cs, cs, np.nextafter(cs, -np.inf)])),   # on the sums themselves, repeated
               np.linspace(lo - span, hi + span, 77)):                   # starting below, ending above the tile
        lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
        ps = np.ascontiguousarray(ps, dtype=np.float64)
        csc = np.ascontiguousarray(cs, dtype=np.float64)
        out = np.empty(ps.size, dtype=np.int32)
        lib.hc_tile_search_walk(ctypes.c_int(csc.size), csc.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(c_in),
                                ctypes.c_long(ps.size), ps.ctypes.data_as(ctypes.c_void_p),
                                out.ctypes.data_as(ctypes.c_void_p))
        want = np.searchsorted(cs, ps, side="right").astype(np.int32)
        assert np.array_equal(out, want), (name, n)
