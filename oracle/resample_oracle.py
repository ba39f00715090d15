"""Restatement of filterpy.monte_carlo.resampling -- TEST INFRASTRUCTURE ONLY.

Follows rlabbe/filterpy v1.4.5 filterpy/monte_carlo/resampling.py:
  residual_resample :27-76, stratified_resample :80-114,
  systematic_resample :117-150, multinomial_resample :153-176.

The random draws are *inputs* here (``u``), so the same numbers can be handed
to the GPU kernel; ``*_seeded`` wrappers draw them from the process-global
NumPy MT19937 stream in exactly the order the reference does
(resampling.py:24 ``from numpy.random import random``).

Three forms, all required to agree bit-for-bit:
  *_loop : the literal pure-Python two-pointer loop (small N only);
  *_np   : np.cumsum + np.searchsorted(side='right') (equivalent, fast);
  C      : oracle/resample_oracle.c via ctypes (literal loops, at-scale).
Never imported by filterpy_amd/.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libresample_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C library not built: run `make -C oracle`")
        _LIB = ctypes.CDLL(path)
        for name in ("oracle_systematic", "oracle_stratified"):
            fn = getattr(_LIB, name)
            fn.restype = ctypes.c_int64
    return _LIB


def positions_systematic(N, u):
    """resampling.py:139  positions = (random() + np.arange(N)) / N"""
    return (u + np.arange(N)) / N


def positions_stratified(N, u):
    """resampling.py:103  positions = (random(N) + range(N)) / N"""
    return (np.asarray(u) + range(N)) / N


def _merge_loop(positions, cumulative_sum):
    """resampling.py:143-149 (same in :107-113): literal two-pointer merge.
    Raises IndexError like the reference when a position >= cumulative_sum[-1]."""
    N = len(positions)
    indexes = np.zeros(N, 'i')
    i, j = 0, 0
    while i < N:
        if positions[i] < cumulative_sum[j]:
            indexes[i] = j
            i += 1
        else:
            j += 1
    return indexes


def systematic_loop(weights, u):
    N = len(weights)
    return _merge_loop(positions_systematic(N, u), np.cumsum(weights))


def stratified_loop(weights, u):
    N = len(weights)
    return _merge_loop(positions_stratified(N, u), np.cumsum(weights))


def systematic_np(weights, u):
    """Vectorised equivalent of the merge: idx_i = #{j : cs_j <= pos_i}.
    (No IndexError: a position >= cs[-1] yields N.)"""
    N = len(weights)
    return np.searchsorted(np.cumsum(weights), positions_systematic(N, u),
                           side='right').astype(np.int32)


def stratified_np(weights, u):
    N = len(weights)
    return np.searchsorted(np.cumsum(weights), positions_stratified(N, u),
                           side='right').astype(np.int32)


def multinomial(weights, u):
    """resampling.py:174-176: cs[-1] = 1.; searchsorted(cs, u) (side left) -> intp."""
    cumulative_sum = np.cumsum(weights)
    cumulative_sum[-1] = 1.
    return np.searchsorted(cumulative_sum, u)


def residual(weights, u_fn):
    """resampling.py:56-76 restated literally, including
    ``residual = weights - num_copies`` (NOT N*weights - num_copies) and the
    builtin sequential ``sum``.  ``u_fn(k)`` returns the k uniforms drawn at :76."""
    weights = np.asarray(weights, dtype=float)
    N = len(weights)
    indexes = np.zeros(N, 'i')
    num_copies = (np.floor(N * weights)).astype(int)
    k = 0
    for i in range(N):
        for _ in range(num_copies[i]):
            indexes[k] = i
            k += 1
    residual_w = weights - num_copies
    residual_w /= sum(residual_w)
    cumulative_sum = np.cumsum(residual_w)
    cumulative_sum[-1] = 1.
    indexes[k:N] = np.searchsorted(cumulative_sum, u_fn(N - k))
    return indexes


def residual_parts(weights):
    """The deterministic part of residual_resample: (num_copies, k, normalised residual cumsum
    with cs[-1]=1).  Used to check the host-side split of the GPU path."""
    weights = np.asarray(weights, dtype=float)
    N = len(weights)
    num_copies = (np.floor(N * weights)).astype(int)
    residual_w = weights - num_copies
    residual_w /= sum(residual_w)
    cs = np.cumsum(residual_w)
    cs[-1] = 1.
    return num_copies, int(num_copies.sum()), cs


# ---- seeded wrappers: consume the global NumPy stream exactly like the reference ----

def systematic_seeded(weights):
    return systematic_np(weights, np.random.random())


def stratified_seeded(weights):
    return stratified_np(weights, np.random.random(len(weights)))


def multinomial_seeded(weights):
    return multinomial(weights, np.random.random(len(weights)))


def residual_seeded(weights):
    return residual(weights, np.random.random)


# ---- C forms (literal loops; status != 0 where the reference would IndexError) ----

def _c_call(name, weights, u):
    w = np.ascontiguousarray(weights, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    N = w.shape[0]
    idx = np.empty(N, dtype=np.int32)
    overrun = getattr(_lib(), name)(
        ctypes.c_int64(N), w.ctypes.data_as(ctypes.c_void_p),
        u.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p))
    return idx, int(overrun)


def systematic_c(weights, u):
    """(indexes int32, overrun) -- overrun = number of positions >= cs[-1]
    (the reference raises IndexError when overrun > 0; entries are then N)."""
    return _c_call("oracle_systematic", weights, [u])


def stratified_c(weights, u):
    return _c_call("oracle_stratified", weights, u)
