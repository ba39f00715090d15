"""Particle-filter resampling with filterpy's signatures (filterpy/monte_carlo/resampling.py:
residual_resample :27-76, stratified_resample :80-114, systematic_resample :117-150,
multinomial_resample :153-176), computed by the gfx950 kernels of resample_kernels.hip.

The uniforms are drawn on the host from the process-global NumPy MT19937 stream, in exactly
the order the reference draws them (resampling.py:24), and handed to the kernels; the
cumulative sum is reproduced bit-for-bit (fk_exact_scan.hpp), so the returned indices are
bit-identical to the reference's.

Every function also accepts a bank of filters, weights (F, N) -> indexes (F, N) (one
`random()` / `random(N)` per filter, in filter order).
"""
import numpy as np
from numpy.random import random

from .. import _engine as E

__all__ = ["residual_resample", "stratified_resample", "systematic_resample", "multinomial_resample"]


def _prep(weights):
    import torch
    E.require_gpu()
    if isinstance(weights, torch.Tensor):
        w = weights.to(dtype=torch.float64).contiguous()
        batched = w.dim() == 2
        w2 = w if batched else w.reshape(1, -1)
        return w2, batched
    w = np.asarray(weights, dtype=np.float64)
    batched = w.ndim == 2
    return E.dev(w.reshape((-1, w.shape[-1]) if batched else (1, -1))), batched


def _finish(idx, status, batched, name):
    from .. import _abi
    if bool((status & _abi.FK_STATUS_INTERNAL).any()):
        raise _abi.FilterHipError(f"{name}: an in-launch hand-off of the resampling kernel timed out")
    bad = status.nonzero()
    if bad.numel():
        # a position >= cumulative_sum[-1]: the reference's merge loop runs off the end
        # (resampling.py:109,145)
        raise IndexError(f"{name}: index {idx.shape[1]} is out of bounds for axis 0 with size {idx.shape[1]} "
                         f"(filter {int(bad[0])}: weights sum to less than the last position)")
    out = idx.cpu().numpy()
    return out if batched else out[0]


def systematic_resample(weights):
    """resampling.py:117-150.  One uniform per filter; returns int32 indexes."""
    import torch
    w, batched = _prep(weights)
    Fn, Np = w.shape
    u = E.dev(np.atleast_1d(random(Fn) if batched else random()))
    idx = torch.empty((Fn, Np), dtype=torch.int32, device=w.device)
    st = torch.zeros(Fn, dtype=torch.int32, device=w.device)
    E.resample_systematic(Fn, Np, w, u, idx, st)
    return _finish(idx, st, batched, "systematic_resample")


def stratified_resample(weights):
    """resampling.py:80-114.  N uniforms per filter; returns int32 indexes."""
    import torch
    w, batched = _prep(weights)
    Fn, Np = w.shape
    u = E.dev(np.stack([random(Np) for _ in range(Fn)]))
    idx = torch.empty((Fn, Np), dtype=torch.int32, device=w.device)
    st = torch.zeros(Fn, dtype=torch.int32, device=w.device)
    E.resample_stratified(Fn, Np, w, u, idx, st)
    return _finish(idx, st, batched, "stratified_resample")


def multinomial_resample(weights):
    """resampling.py:153-176: cumsum, cs[-1] = 1., searchsorted(cs, random(N)); returns intp."""
    import torch
    w, batched = _prep(weights)
    Fn, Np = w.shape
    u = E.dev(np.stack([random(Np) for _ in range(Fn)]))
    idx = torch.empty((Fn, Np), dtype=torch.int64, device=w.device)
    E.resample_multinomial(Fn, Np, Np, w, u, idx)
    out = idx.cpu().numpy().astype(np.intp, copy=False)
    return out if batched else out[0]


def residual_resample(weights):
    """resampling.py:27-76, restated literally on the device: floor(N w) deterministic copies,
    then a multinomial draw on `weights - num_copies` normalised by its sequential sum."""
    import torch
    w, batched = _prep(weights)
    if batched:
        return np.stack([residual_resample(wi) for wi in w])
    wd = w[0]
    N = wd.numel()
    num_copies = torch.floor(N * wd)                              # :61
    counts = num_copies.to(torch.int64)
    k = int(counts.sum())
    if k > N:        # the reference's fill loop writes indexes[k] past the end (resampling.py:63-66)
        raise IndexError(f"index {N} is out of bounds for axis 0 with size {N}")
    idx = torch.zeros(N, dtype=torch.int32, device=wd.device)
    if k:
        idx[:k] = torch.repeat_interleave(torch.arange(N, device=wd.device, dtype=torch.int32), counts)[:N]
    residual = (wd - num_copies).contiguous()                     # :70 (not N*w - copies)
    cs = torch.empty_like(residual)
    E.cumsum_exact(1, N, residual, cs)                            # builtin sum() == last sequential partial sum
    residual = (residual / cs[-1]).contiguous()                   # :71
    if N - k > 0:
        u = E.dev(random(N - k))
        tail = torch.empty(N - k, dtype=torch.int64, device=wd.device)
        E.resample_multinomial(1, N, N - k, residual, u, tail)    # cumsum, cs[-1]=1., searchsorted  (:72-76)
        idx[k:] = tail.to(torch.int32)
    return idx.cpu().numpy()
