"""NumPy restatement of the UKF sigma-point / unscented-transform path -- TEST INFRASTRUCTURE ONLY.

Follows rlabbe/filterpy v1.4.5: filterpy/kalman/sigma_points.py:99-192 (Merwe),
:211-383 (Julier), filterpy/kalman/unscented_transform.py:99-128,
filterpy/kalman/UKF.py:364-522 (predict/update/cross_variance), :524-632
(batch_filter), :634-739 (rts_smoother).  Never imported by filterpy_amd/.
"""
import numpy as np
from numpy import dot
from scipy.linalg import cholesky


def merwe_weights(n, alpha, beta, kappa):
    """MerweScaledSigmaPoints._compute_weights (sigma_points.py:180-192)."""
    lambda_ = alpha**2 * (n + kappa) - n
    c = .5 / (n + lambda_)
    Wc = np.full(2 * n + 1, c)
    Wm = np.full(2 * n + 1, c)
    Wc[0] = lambda_ / (n + lambda_) + (1 - alpha**2 + beta)
    Wm[0] = lambda_ / (n + lambda_)
    return Wm, Wc


def merwe_sigma_points(x, P, alpha, kappa, sqrt=cholesky, subtract=np.subtract):
    """MerweScaledSigmaPoints.sigma_points (sigma_points.py:153-177).

    U = sqrt((lambda+n) P) (default: upper Cholesky factor);  sigma_0 = x, sigma_{k+1} = subtract(x, -U[k]),
    sigma_{n+k+1} = subtract(x, U[k])  (sqrt_method / subtract: the constructor hooks, sigma_points.py:99-116).
    """
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = x.size
    P = np.eye(n) * P if np.isscalar(P) else np.atleast_2d(P)
    lambda_ = alpha**2 * (n + kappa) - n
    U = sqrt((lambda_ + n) * P)
    sigmas = np.zeros((2 * n + 1, n))
    sigmas[0] = x
    for k in range(n):
        sigmas[k + 1] = subtract(x, -U[k])
        sigmas[n + k + 1] = subtract(x, U[k])
    return sigmas


HOOK_DEFAULTS = dict(x_mean=None, z_mean=None, residual_x=np.subtract, residual_z=np.subtract, state_add=np.add,
                     sqrt=cholesky, subtract=np.subtract)


def hooks(**kw):
    """the constructor hooks of UnscentedKalmanFilter (UKF.py:284-340) and of the sigma-point class
    (sigma_points.py:99-116), defaults filled in"""
    unknown = set(kw) - set(HOOK_DEFAULTS)
    if unknown:
        raise TypeError(f"unknown hooks {sorted(unknown)}")
    return {**HOOK_DEFAULTS, **kw}


def julier_weights(n, kappa):
    """JulierSigmaPoints._compute_weights (sigma_points.py:360-372)."""
    W = np.full(2 * n + 1, .5 / (n + kappa))
    W[0] = kappa / (n + kappa)
    return W, W.copy()


def julier_sigma_points(x, P, kappa):
    """JulierSigmaPoints.sigma_points (sigma_points.py:328-357):
    U = cholesky((n+kappa) P); sigma_{k+1} = x - (-U[k]); sigma_{n+k+1} = x - U[k]."""
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = x.size
    P = np.eye(n) * P if np.isscalar(P) else np.atleast_2d(P)
    U = cholesky((n + kappa) * P)
    sigmas = np.zeros((2 * n + 1, n))
    sigmas[0] = x
    for k in range(n):
        sigmas[k + 1] = np.subtract(x, -U[k])
        sigmas[n + k + 1] = np.subtract(x, U[k])
    return sigmas


def unscented_transform(sigmas, Wm, Wc, noise_cov=None, mean_fn=None, residual_fn=None):
    """unscented_transform (unscented_transform.py:101-126): mean_fn replaces Wm . sigmas (:105-106); a residual_fn
    other than numpy.subtract takes the point-by-point loop (:120-123)."""
    x = np.dot(Wm, sigmas) if mean_fn is None else mean_fn(sigmas, Wm)
    if residual_fn is np.subtract or residual_fn is None:
        y = sigmas - x[np.newaxis, :]
        P = np.dot(y.T, np.dot(np.diag(Wc), y))
    else:
        kmax, n = sigmas.shape
        P = np.zeros((n, n))
        for k in range(kmax):
            y = residual_fn(sigmas[k], x)
            P += Wc[k] * np.outer(y, y)
    if noise_cov is not None:
        P += noise_cov
    return x, P


def cross_variance(x, z, sigmas_f, sigmas_h, Wc, residual_x=np.subtract, residual_z=np.subtract):
    """UnscentedKalmanFilter.cross_variance (UKF.py:493-504)."""
    Pxz = np.zeros((sigmas_f.shape[1], sigmas_h.shape[1]))
    for i in range(sigmas_f.shape[0]):
        dx = residual_x(sigmas_f[i], x)
        dz = residual_z(sigmas_h[i], z)
        Pxz += Wc[i] * np.outer(dx, dz)
    return Pxz


def ukf_predict(x, P, fx, dt, Q, Wm, Wc, alpha, kappa, hk=None):
    """UKF.predict (UKF.py:400-411): sigma points -> fx -> UT(+Q; x_mean, residual_x) -> regenerate sigmas."""
    hk = hooks() if hk is None else hk
    sigmas = merwe_sigma_points(x, P, alpha, kappa, hk["sqrt"], hk["subtract"])
    sigmas_f = np.array([fx(s, dt) for s in sigmas])
    x, P = unscented_transform(sigmas_f, Wm, Wc, Q, hk["x_mean"], hk["residual_x"])
    sigmas_f = merwe_sigma_points(x, P, alpha, kappa, hk["sqrt"], hk["subtract"])
    return x, P, sigmas_f


def ukf_update(x, P, sigmas_f, z, hx, R, Wm, Wc, inv=np.linalg.inv, hk=None):
    """UKF.update (UKF.py:462-481):  P = P - K (S K'); y = residual_z(z, zp); x = state_add(x, K y)."""
    hk = hooks() if hk is None else hk
    sigmas_h = np.atleast_2d([hx(s) for s in sigmas_f])
    zp, S = unscented_transform(sigmas_h, Wm, Wc, R, hk["z_mean"], hk["residual_z"])
    SI = inv(S)
    Pxz = cross_variance(x, zp, sigmas_f, sigmas_h, Wc, hk["residual_x"], hk["residual_z"])
    K = dot(Pxz, SI)
    y = hk["residual_z"](z, zp)
    x = hk["state_add"](x, dot(K, y))
    P = P - dot(K, dot(S, K.T))
    return x, P, K, y, S


def ukf_batch_filter(x0, P0, zs, fx, hx, dt, Q, R, alpha, beta, kappa, hk=None):
    """UKF.batch_filter (UKF.py:623-632): predict -> update per z; returns (means, covariances).
    A z that is None / all-NaN skips the update (UKF.py:440-444)."""
    n = len(x0)
    Wm, Wc = merwe_weights(n, alpha, beta, kappa)
    x, P = np.array(x0, dtype=float), np.array(P0, dtype=float)
    means = np.zeros((len(zs), n))
    covs = np.zeros((len(zs), n, n))
    for i, z in enumerate(zs):
        x, P, sigmas_f = ukf_predict(x, P, fx, dt, Q, Wm, Wc, alpha, kappa, hk)
        if z is not None and not np.all(np.isnan(np.asarray(z, dtype=float))):
            x, P, _, _, _ = ukf_update(x, P, sigmas_f, z, hx, R, Wm, Wc, hk=hk)
        means[i], covs[i] = x, P
    return means, covs


def ukf_rts_smoother(Xs, Ps, fx, dt, Q, alpha, beta, kappa, inv=np.linalg.inv, hk=None):
    """UKF.rts_smoother (UKF.py:714-739).  Quirk kept: always self.Q (UKF.py:720-722)."""
    hk = hooks() if hk is None else hk
    rx = hk["residual_x"]
    n, dim_x = Xs.shape
    Wm, Wc = merwe_weights(dim_x, alpha, beta, kappa)
    Ks = np.zeros((n, dim_x, dim_x))
    xs, ps = Xs.copy(), Ps.copy()
    for k in reversed(range(n - 1)):
        sigmas = merwe_sigma_points(xs[k], ps[k], alpha, kappa, hk["sqrt"], hk["subtract"])
        sigmas_f = np.array([fx(s, dt) for s in sigmas])
        xb, Pb = unscented_transform(sigmas_f, Wm, Wc, Q, hk["x_mean"], rx)
        Pxb = 0
        for i in range(2 * dim_x + 1):
            y = rx(sigmas_f[i], xb)
            z = rx(sigmas[i], Xs[k])
            Pxb = Pxb + Wc[i] * np.outer(z, y)
        K = dot(Pxb, inv(Pb))
        xs[k] += dot(K, rx(xs[k + 1], xb))
        ps[k] += dot(K, ps[k + 1] - Pb).dot(K.T)
        Ks[k] = K
    return xs, ps, Ks
