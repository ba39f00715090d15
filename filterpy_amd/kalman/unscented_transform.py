"""unscented_transform with filterpy's signature (filterpy/kalman/unscented_transform.py:22-128),
computed by fk_ut_transform_f64 (default mean / residual) or, with mean_fn / residual_fn callables, by
fk_ut_cross_variance_f64 over the residuals those callables return."""
import numpy as np

from .. import _engine as E

__all__ = ["unscented_transform"]


def unscented_transform(sigmas, Wm, Wc, noise_cov=None, mean_fn=None, residual_fn=None, layout="soa"):
    """x = Wm . sigmas ; P = sum_k Wc[k] (sigmas[k]-x)(sigmas[k]-x)' (+ noise_cov).
    sigmas (k, n) -> (x (n,), P (n,n)); a bank (N, k, n) -> (x (N,n), P (N,n,n))."""
    E.require_gpu()
    s = np.asarray(sigmas, dtype=np.float64)
    batched = s.ndim == 3
    sb = s.reshape((-1,) + s.shape[-2:])
    N, k, n = sb.shape
    if n > 16:
        raise NotImplementedError("unscented_transform: dim > 16")
    default_res = residual_fn is None or residual_fn is np.subtract
    if mean_fn is not None or not default_res:
        # unscented_transform.py:105-106, :120-123: the callables run per filter / per point like the reference; the
        # kernel sums Wc[k] outer(y_k, y_k) over the residuals they return (x = z = NULL mode, reference loop order)
        if mean_fn is None:
            xo, Po = E.alloc_records((), N, n, layout), E.alloc_records((), N, n * n, layout)
            E.ut_transform(n, k, N, layout, E.to_records(sb, layout, 0), E.dev(np.asarray(Wm, dtype=np.float64)),
                           E.dev(np.asarray(Wc, dtype=np.float64)), None, xo, Po)
            x = E.from_records(xo, layout, 0, (n,))
        else:
            x = np.array([mean_fn(sb[i].copy(), Wm) for i in range(N)], dtype=np.float64).reshape(N, n)
        if default_res:
            y = sb - x[:, None, :]
        else:
            y = np.array([[residual_fn(sb[i, j].copy(), x[i].copy()) for j in range(k)] for i in range(N)], dtype=np.float64)
        dy = E.to_records(y.reshape(N, k, n), layout, 0)
        Po = E.alloc_records((), N, n * n, layout)
        E.ut_cross_variance(n, n, k, N, layout, None, None, dy, dy, E.dev(np.asarray(Wc, dtype=np.float64)), Po)
        P = E.from_records(Po, layout, 0, (n, n))
        if noise_cov is not None:
            P = P + np.broadcast_to(np.asarray(noise_cov, dtype=np.float64), (n, n))
        return (x, P) if batched else (x[0], P[0])
    ds = E.to_records(sb, layout, 0)
    xo, Po = E.alloc_records((), N, n, layout), E.alloc_records((), N, n * n, layout)
    noise = None
    if noise_cov is not None:
        noise = E.dev(np.broadcast_to(np.asarray(noise_cov, dtype=np.float64), (n, n)).copy())
    E.ut_transform(n, k, N, layout, ds, E.dev(np.asarray(Wm, dtype=np.float64)),
                   E.dev(np.asarray(Wc, dtype=np.float64)), noise, xo, Po)
    x, P = E.from_records(xo, layout, 0, (n,)), E.from_records(Po, layout, 0, (n, n))
    return (x, P) if batched else (x[0], P[0])
