"""unscented_transform with filterpy's signature (filterpy/kalman/unscented_transform.py:22-128),
computed by fk_ut_transform_f64."""
import numpy as np

from .. import _engine as E

__all__ = ["unscented_transform"]


def unscented_transform(sigmas, Wm, Wc, noise_cov=None, mean_fn=None, residual_fn=None, layout="soa"):
    """x = Wm . sigmas ; P = sum_k Wc[k] (sigmas[k]-x)(sigmas[k]-x)' (+ noise_cov).
    sigmas (k, n) -> (x (n,), P (n,n)); a bank (N, k, n) -> (x (N,n), P (N,n,n))."""
    if mean_fn is not None or not (residual_fn is None or residual_fn is np.subtract):
        raise NotImplementedError("custom mean_fn / residual_fn callables cannot run inside the HIP kernel")
    E.require_gpu()
    s = np.asarray(sigmas, dtype=np.float64)
    batched = s.ndim == 3
    sb = s.reshape((-1,) + s.shape[-2:])
    N, k, n = sb.shape
    if n > 16:
        raise NotImplementedError("unscented_transform: dim > 16")
    ds = E.to_records(sb, layout, 0)
    xo, Po = E.alloc_records((), N, n, layout), E.alloc_records((), N, n * n, layout)
    noise = None
    if noise_cov is not None:
        noise = E.dev(np.broadcast_to(np.asarray(noise_cov, dtype=np.float64), (n, n)).copy())
    E.ut_transform(n, k, N, layout, ds, E.dev(np.asarray(Wm, dtype=np.float64)),
                   E.dev(np.asarray(Wc, dtype=np.float64)), noise, xo, Po)
    x, P = E.from_records(xo, layout, 0, (n,)), E.from_records(Po, layout, 0, (n, n))
    return (x, P) if batched else (x[0], P[0])
