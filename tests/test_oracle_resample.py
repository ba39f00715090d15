"""Pin the resampling oracle (NumPy forms + the C restatement) to the goldens frozen from the
live reference under np.random.seed: indices must be bit-identical."""
import hashlib
import sys
import os

import numpy as np
import pytest

from conftest import golden
from oracle import resample_oracle as ro

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from weights import weights_for  # noqa: E402


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _check(key, g, idx):
    assert str(idx.dtype) == str(g[key + "_dtype"]), key
    if key + "_idx" in g.files:
        assert np.array_equal(idx, g[key + "_idx"]), key
    else:
        assert np.array_equal(idx[:1024], g[key + "_head"]) and np.array_equal(idx[-1024:], g[key + "_tail"]), key
        assert np.array_equal(_sha(idx), g[key + "_sha"]), key


@pytest.mark.parametrize("name", ["sys", "strat", "multi", "resid"])
def test_seeded_forms_vs_goldens(name):
    g = golden("resample")
    fn = dict(sys=ro.systematic_seeded, strat=ro.stratified_seeded, multi=ro.multinomial_seeded,
              resid=ro.residual_seeded)[name]
    keys = sorted(k[:-len("_wseed")] for k in g.files if k.startswith(name + "_N") and k.endswith("_wseed"))
    assert keys
    for key in keys:
        N = int(key.split("_N")[1].split("_")[0])
        kind = key.split("_")[-1]
        if N > (1 << 20) or key + "_indexerror" in g.files:
            continue
        w = weights_for(N, int(g[key + "_wseed"]), kind)
        np.random.seed(int(g[key + "_useed"]))
        with np.errstate(all="ignore"):
            idx = fn(w)
        _check(key, g, idx)


@pytest.mark.parametrize("N", [1, 2, 3, 10, 64, 1000, 8000])
def test_literal_loop_equals_fast_forms_and_c(N):
    rs = np.random.RandomState(N)
    for kind in ("rand", "exp", "sparse"):
        w = weights_for(N, 100 + N, kind) if N >= 10 else weights_for(N, 100 + N, "rand")
        u = rs.rand()
        assert np.array_equal(ro.systematic_loop(w, u), ro.systematic_np(w, u))
        c, over = ro.systematic_c(w, u)
        assert over == 0 and np.array_equal(c, ro.systematic_np(w, u)) and c.dtype == np.int32
        uu = rs.rand(N)
        assert np.array_equal(ro.stratified_loop(w, uu), ro.stratified_np(w, uu))
        c, over = ro.stratified_c(w, uu)
        assert over == 0 and np.array_equal(c, ro.stratified_np(w, uu))


def test_c_cumsum_is_numpy_cumsum_bitwise():
    import ctypes
    lib = ro._lib()
    rs = np.random.RandomState(0)
    for N in (1, 7, 100003, 1 << 20):
        w = rs.rand(N)
        cs = np.empty(N)
        lib.oracle_cumsum(ctypes.c_int64(N), w.ctypes.data_as(ctypes.c_void_p), cs.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(cs.view(np.uint64), np.cumsum(w).view(np.uint64))


def test_overrun_reported_like_indexerror():
    w = np.full(1000, 0.5e-3)
    with pytest.raises(IndexError):
        ro.systematic_loop(w, 0.7)
    idx, over = ro.systematic_c(w, 0.7)
    assert over > 0 and (idx[-over:] == 1000).all()
