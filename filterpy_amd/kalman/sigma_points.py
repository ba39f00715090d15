"""Sigma-point generators with filterpy's interface (filterpy/kalman/sigma_points.py:
MerweScaledSigmaPoints :24-208, JulierSigmaPoints :211-383); the points are produced by the
gfx950 kernel fk_ut_sigma_points_f64 (Cholesky in-lane).  Weights are host-side scalars."""
import numpy as np

from .. import _engine as E

__all__ = ["MerweScaledSigmaPoints", "JulierSigmaPoints"]


def _sigma_points_gpu(n, scale, x, P, layout="soa"):
    """x (n,) or (N,n); P (n,n) or (N,n,n) or scalar -> (2n+1, n) or (N, 2n+1, n)."""
    import torch
    E.require_gpu()
    x = np.asarray(x, dtype=np.float64)
    batched = x.ndim == 2
    xb = x.reshape(-1, n)
    N = xb.shape[0]
    if np.isscalar(P) or np.ndim(P) == 0:
        Pb = np.broadcast_to(np.eye(n) * P, (N, n, n))
    else:
        Pb = np.broadcast_to(np.atleast_2d(np.asarray(P, dtype=np.float64)), (N, n, n))
    dx, dP = E.to_records(xb, layout, 0), E.to_records(np.ascontiguousarray(Pb), layout, 0)
    sig = E.alloc_records((), N, (2 * n + 1) * n, layout)
    st = torch.zeros(N, dtype=torch.int32, device=dx.device)
    E.ut_sigma_points(n, N, layout, scale, dx, dP, sig, st)
    E.raise_on_status(st, "sigma_points (scipy.linalg.cholesky would raise LinAlgError)")
    out = E.from_records(sig, layout, 0, (2 * n + 1, n))
    return out if batched else out[0]


def _sigma_points_hooked(n, scale, x, P, sqrt, subtract):
    """sigma_points.py:165-175 / :343-355 with the constructor's sqrt_method / subtract, per filter like the reference:
    U = sqrt(scale P) -- the caller's function, or (only `subtract` custom) the rows the kernel adds to a zero mean --
    then sigma_0 = x, sigma_{k+1} = subtract(x, -U[k]), sigma_{n+k+1} = subtract(x, U[k]).
    (Whole-bank conventions for these callables live with the filter: UnscentedKalmanFilter(vectorized= / device_callables=).)"""
    x = np.asarray(x, dtype=np.float64)
    batched = x.ndim == 2
    xb = x.reshape(-1, n)
    N = xb.shape[0]
    if np.isscalar(P) or np.ndim(P) == 0:
        Pb = np.broadcast_to(np.eye(n) * P, (N, n, n))
    else:
        Pb = np.broadcast_to(np.atleast_2d(np.asarray(P, dtype=np.float64)), (N, n, n))
    if sqrt is None:
        U = _sigma_points_gpu(n, scale, np.zeros((N, n)), Pb)[:, 1:n + 1, :]
    else:
        U = np.array([sqrt(scale * Pb[i]) for i in range(N)], dtype=np.float64)
    sub = np.subtract if subtract is None else subtract
    out = np.zeros((N, 2 * n + 1, n))
    for i in range(N):
        out[i, 0] = xb[i]
        for k in range(n):
            out[i, k + 1] = sub(xb[i].copy(), -U[i, k])
            out[i, n + k + 1] = sub(xb[i].copy(), U[i, k].copy())
    return out if batched else out[0]


def _default_sqrt(A, *args, **kw):
    """scipy.linalg.cholesky, imported on first use: what `points.sqrt` / `ukf.msqrt` are in the reference when no
    callable was passed (sigma_points.py:106-109, UKF.py:318-321) -- for user code that calls the attribute"""
    from scipy.linalg import cholesky
    return cholesky(A, *args, **kw)


class _Hooks(object):
    """sqrt_method / subtract of the reference constructors (sigma_points.py:106-116, :271-281)"""

    def _set_hooks(self, sqrt_method, subtract):
        self._sqrt = sqrt_method            # None = upper Cholesky factor inside the kernel (scipy.linalg.cholesky's convention)
        self._subtract = None if subtract is np.subtract else subtract
        # public attribute like the reference's (sigma_points.py:106-109: scipy.linalg.cholesky unless given); the
        # filter itself never calls it when it is the default -- _sqrt = None means "the factorisation inside the kernel"
        self.sqrt = _default_sqrt if sqrt_method is None else sqrt_method
        self.subtract = np.subtract if subtract is None else subtract

    def _points(self, scale, x_arr, P):
        if self._sqrt is None and self._subtract is None:
            return _sigma_points_gpu(self.n, scale, np.atleast_1d(x_arr), P)
        return _sigma_points_hooked(self.n, scale, np.atleast_1d(x_arr), P, self._sqrt, self._subtract)


class MerweScaledSigmaPoints(_Hooks):
    """filterpy/kalman/sigma_points.py:24-208."""

    def __init__(self, n, alpha, beta, kappa, sqrt_method=None, subtract=None):
        self._set_hooks(sqrt_method, subtract)
        self.n, self.alpha, self.beta, self.kappa = n, alpha, beta, kappa
        self._compute_weights()

    def num_sigmas(self):
        return 2 * self.n + 1

    def sigma_points(self, x, P):
        """sigma_points.py:124-177; also accepts a bank: x (N,n), P (N,n,n) -> (N, 2n+1, n)."""
        x_arr = np.asarray(x, dtype=np.float64)
        if (x_arr.ndim <= 1 and self.n != np.size(x)) or (x_arr.ndim == 2 and x_arr.shape[1] != self.n):
            raise ValueError("expected size(x) {}, but size is {}".format(self.n, np.size(x)))
        lambda_ = self.alpha ** 2 * (self.n + self.kappa) - self.n
        return self._points(lambda_ + self.n, x_arr, P)

    def _compute_weights(self):
        """sigma_points.py:180-192."""
        n = self.n
        lambda_ = self.alpha ** 2 * (n + self.kappa) - n
        c = .5 / (n + lambda_)
        self.Wc = np.full(2 * n + 1, c)
        self.Wm = np.full(2 * n + 1, c)
        self.Wc[0] = lambda_ / (n + lambda_) + (1 - self.alpha ** 2 + self.beta)
        self.Wm[0] = lambda_ / (n + lambda_)

    @property
    def scale(self):
        """lambda + n exactly as sigma_points() forms it (sigma_points.py:165-168): lambda_ is rounded first (it
        cancels against n: six digits at alpha = 1e-3), and the weights divide by the same rounded sum
        (:184-185) -- spread^2 * Wc stays exactly 1/2 only if this is the same floating-point number."""
        lambda_ = self.alpha ** 2 * (self.n + self.kappa) - self.n
        return lambda_ + self.n


class JulierSigmaPoints(_Hooks):
    """filterpy/kalman/sigma_points.py:211-383."""

    def __init__(self, n, kappa=0., sqrt_method=None, subtract=None):
        self._set_hooks(sqrt_method, subtract)
        self.n, self.kappa = n, kappa
        self._compute_weights()

    def num_sigmas(self):
        return 2 * self.n + 1

    def sigma_points(self, x, P):
        x_arr = np.asarray(x, dtype=np.float64)
        if (x_arr.ndim <= 1 and self.n != np.size(x)) or (x_arr.ndim == 2 and x_arr.shape[1] != self.n):
            raise ValueError("expected size(x) {}, but size is {}".format(self.n, np.size(x)))
        return self._points(self.n + self.kappa, x_arr, P)

    def _compute_weights(self):
        """sigma_points.py:360-372."""
        n, k = self.n, self.kappa
        self.Wm = np.full(2 * n + 1, .5 / (n + k))
        self.Wm[0] = k / (n + k)
        self.Wc = self.Wm

    @property
    def scale(self):
        return self.n + self.kappa
