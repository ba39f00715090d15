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
    """resampling.py:27-76 on the device, for one filter or a bank: floor(N w) deterministic copies
    (fk_resample_residual_fill_f64), then a multinomial draw on `weights - num_copies` normalised by its sequential sum
    (fk_resample_residual_draw_f64).  The host only draws the uniforms -- random(N - k) per filter, in filter order: how
    many the reference takes from numpy.random depends on the weights."""
    import torch
    w, batched = _prep(weights)
    Fn, Np = w.shape
    idx = torch.zeros((Fn, Np), dtype=torch.int32, device=w.device)
    k = torch.zeros(Fn, dtype=torch.int64, device=w.device)
    cs = torch.empty((Fn, Np), dtype=torch.float64, device=w.device)
    st = torch.zeros(Fn, dtype=torch.int32, device=w.device)
    E.resample_residual_fill(Fn, Np, w, idx, k, cs, st)
    kh = k.cpu().numpy()
    if (kh > Np).any():   # the reference's fill loop writes indexes[k] past the end (resampling.py:63-66)
        raise IndexError(f"index {Np} is out of bounds for axis 0 with size {Np}")
    draws = [random(int(Np - kf)) for kf in kh]
    if sum(len(d) for d in draws):
        uoff = np.concatenate([[0], np.cumsum([len(d) for d in draws])[:-1]]).astype(np.int64)
        E.resample_residual_draw(Fn, Np, cs, k, torch.as_tensor(uoff, device=w.device), E.dev(np.concatenate(draws)), idx)
    out = idx.cpu().numpy()
    return out if batched else out[0]
