"""NumPy restatement of the linear Kalman filter hot path -- TEST INFRASTRUCTURE ONLY.

Follows rlabbe/filterpy v1.4.5, filterpy/kalman/kalman_filter.py, in the
reference's own association order.  One filter at a
time, exactly like the reference; ``*_tracks`` helpers loop over tracks so the
GPU result for a sample of tracks can be compared.

Never imported by filterpy_amd/ (see oracle/__init__.py).
"""
import numpy as np
from numpy import dot


def kf_predict(x, P, F, Q, B=None, u=None, alpha_sq=1.0):
    """KalmanFilter.predict  (kalman_filter.py:472-478).

    x = F x (+ B u iff B and u are both given);  P = alpha^2 * ((F P) F') + Q.
    """
    if B is not None and u is not None:
        x = dot(F, x) + dot(B, u)
    else:
        x = dot(F, x)
    P = alpha_sq * dot(dot(F, P), F.T) + Q
    return x, P


def kf_update(x, P, z, R, H, inv=np.linalg.inv):
    """KalmanFilter.update  (kalman_filter.py:533-556), Joseph form.

    Returns x, P, y, K, S, SI.  ``z is None`` (kalman_filter.py:515-520) is
    handled by the caller (state unchanged).
    """
    y = z - dot(H, x)
    PHT = dot(P, H.T)
    S = dot(H, PHT) + R
    SI = inv(S)
    K = dot(PHT, SI)
    x = x + dot(K, y)
    I_KH = np.eye(P.shape[0]) - dot(K, H)
    P = dot(dot(I_KH, P), I_KH.T) + dot(dot(K, R), K.T)
    return x, P, y, K, S, SI


def proc_update(x, P, z, R, H):
    """module-level update()  (kalman_filter.py:1473-1501): note the different
    association  S = (H P) H' + R ;  K = (P H') inv(S)."""
    y = z - dot(H, x)
    S = dot(dot(H, P), H.T) + R
    K = dot(dot(P, H.T), np.linalg.inv(S))
    x = x + dot(K, y)
    KH = dot(K, H)
    I_KH = np.eye(KH.shape[0]) - KH
    P = dot(dot(I_KH, P), I_KH.T) + dot(dot(K, R), K.T)
    return x, P, y, K, S


def _per_step(v, n):
    """None / single matrix / per-epoch list -> list of length n."""
    if isinstance(v, (list, tuple)):
        return list(v)
    v = None if v is None else np.asarray(v)
    if v is not None and v.ndim == 3:
        return list(v)
    return [v] * n


def kf_batch_filter(x0, P0, zs, F, Q, H, R, B=None, us=None, alpha_sq=1.0,
                    update_first=False, return_all=False):
    """KalmanFilter.batch_filter  (kalman_filter.py:940-993).

    F,Q,H,R,B may be one matrix (used every epoch) or a length-T sequence
    (the Fs/Qs/Hs/Rs/Bs lists).  ``zs[i] is None`` or a row of NaN marks a
    missing measurement (kalman_filter.py:515-520: update is skipped).
    x0 may be (n,) or (n,1); each z must then be (m,) or (m,1) respectively
    (batch_filter always passes H, so reshape_z is skipped: :527-529).
    Returns (means, covariances, means_p, covariances_p) like the reference;
    with return_all also the per-epoch histories (Ks, ys, Ss, SIs) as filterpy.common.Saver
    records them (helpers.py:121-152): after update(None) y = 0 and K, S, SI keep their previous
    values (zeros before the first update, kalman_filter.py:411-414, :515-520).
    """
    n_steps = len(zs)
    Fs, Qs, Hs, Rs, Bs = (_per_step(v, n_steps) for v in (F, Q, H, R, B))
    us = [None] * n_steps if us is None else list(us)
    x = np.array(x0, dtype=float)
    P = np.array(P0, dtype=float)
    dim_x = x.shape[0]
    means = np.zeros((n_steps,) + x.shape)
    means_p = np.zeros((n_steps,) + x.shape)
    covs = np.zeros((n_steps, dim_x, dim_x))
    covs_p = np.zeros((n_steps, dim_x, dim_x))
    dim_z = np.asarray(Hs[0]).shape[0]
    Ks = np.zeros((n_steps, dim_x, dim_z))
    ys = np.zeros((n_steps, dim_z))
    Ss = np.zeros((n_steps, dim_z, dim_z))
    SIs = np.zeros((n_steps, dim_z, dim_z))
    last = [np.zeros((dim_x, dim_z)), np.zeros((dim_z, dim_z)), np.zeros((dim_z, dim_z))]

    def missing(z):
        return z is None or (np.ndim(z) > 0 and np.all(np.isnan(np.asarray(z, dtype=float))))

    def do_update(i):
        nonlocal x, P
        z = zs[i]
        if missing(z):
            Ks[i], Ss[i], SIs[i] = last
            return
        z = np.asarray(z, dtype=float)
        if x.ndim == 2 and z.ndim == 1:
            z = z.reshape(-1, 1)
        x, P, y, K, S, SI = kf_update(x, P, z, Rs[i], Hs[i])
        Ks[i], ys[i], Ss[i], SIs[i] = K, np.ravel(y), S, SI
        last[:] = [K, S, SI]

    def do_predict(i):
        nonlocal x, P
        x, P = kf_predict(x, P, Fs[i], Qs[i], Bs[i], us[i], alpha_sq)

    for i in range(n_steps):
        if update_first:
            do_update(i)
            means[i], covs[i] = x, P
            do_predict(i)
            means_p[i], covs_p[i] = x, P
        else:
            do_predict(i)
            means_p[i], covs_p[i] = x, P
            do_update(i)
            means[i], covs[i] = x, P
    if return_all:
        return means, covs, means_p, covs_p, Ks, ys, Ss, SIs
    return means, covs, means_p, covs_p


def rts_smoother(Xs, Ps, F, Q, convention="class", inv=np.linalg.inv):
    """RTS smoother.

    convention="class":  KalmanFilter.rts_smoother (kalman_filter.py:1066-1072)
        uses Fs[k+1], Qs[k+1];
    convention="module": module rts_smoother (kalman_filter.py:1851-1856)
        uses Fs[k], Qs[k].
    Returns (x, P, K, Pp).
    """
    n = Xs.shape[0]
    dim_x = Xs.shape[1]
    Fs, Qs = _per_step(F, n), _per_step(Q, n)
    off = 1 if convention == "class" else 0
    K = np.zeros((n, dim_x, dim_x))
    x, P, Pp = Xs.copy(), Ps.copy(), Ps.copy()
    for k in range(n - 2, -1, -1):
        Fk, Qk = Fs[k + off], Qs[k + off]
        Pp[k] = dot(dot(Fk, P[k]), Fk.T) + Qk
        K[k] = dot(dot(P[k], Fk.T), inv(Pp[k]))
        x[k] += dot(K[k], x[k + 1] - dot(Fk, x[k]))
        P[k] += dot(dot(K[k], P[k + 1] - Pp[k]), K[k].T)
    return x, P, K, Pp


# --------------------------------------------------------------------------
# helpers looping the single-filter oracle over a batch of tracks
# --------------------------------------------------------------------------

def kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks, model_mode=0, B=None, us=None,
                           alpha_sq=1.0, update_first=False, mask=None):
    """Run kf_batch_filter for each track index in ``tracks``.

    x0 (N,n), P0 (N,n,n), zs (T,N,m), optional mask (T,N) (0 = missing).
    model_mode 0: F,Q,H,R shared (a,b); 1: per-track (N,a,b);
    2: per-track-per-step (T,N,a,b).  us: (T,N,nu) or None; B follows model_mode.
    Returns arrays shaped (T, len(tracks), ...).
    """
    outs = []
    T = zs.shape[0]
    for i in tracks:
        def pick(M):
            if M is None:
                return None
            M = np.asarray(M)
            if model_mode == 0:
                return M
            if model_mode == 1:
                return M[i]
            return M[:, i]
        z_i = [None if (mask is not None and not mask[t, i]) else zs[t, i] for t in range(T)]
        u_i = None if us is None else [us[t, i] for t in range(T)]
        outs.append(kf_batch_filter(x0[i], P0[i], z_i, pick(F), pick(Q), pick(H), pick(R),
                                    B=pick(B), us=u_i, alpha_sq=alpha_sq,
                                    update_first=update_first))
    return tuple(np.stack([o[k] for o in outs], axis=1) for k in range(4))


def rts_smoother_tracks(Xs, Ps, F, Q, tracks, convention="class", model_mode=0):
    """Xs (T,N,n), Ps (T,N,n,n) -> smoothed (T,len(tracks),...) x, P, K, Pp."""
    outs = []
    for i in tracks:
        def pick(M):
            M = np.asarray(M)
            if model_mode == 0:
                return M
            if model_mode == 1:
                return M[i]
            return M[:, i]
        outs.append(rts_smoother(Xs[:, i].copy(), Ps[:, i].copy(), pick(F), pick(Q), convention))
    return tuple(np.stack([o[k] for o in outs], axis=1) for k in range(4))


def log_likelihood(y, S):
    """KalmanFilter.log_likelihood (kalman_filter.py:1203-1211) = logpdf(x=y, cov=S)
    (filterpy/stats/stats.py:131-154 -> scipy.stats.multivariate_normal.logpdf, allow_singular)."""
    from scipy.stats import multivariate_normal
    return multivariate_normal.logpdf(np.asarray(y).flatten(), None, S, True)


def mahalanobis(y, SI):
    """KalmanFilter.mahalanobis (kalman_filter.py:1228-1240)."""
    y = np.asarray(y, dtype=float).reshape(-1, 1)
    return float(np.sqrt(float(dot(dot(y.T, SI), y).item())))


# --------------------------------------------------------------------------
# API variants
# --------------------------------------------------------------------------

def steadystate_filter(x0, zs, F, H, K, B=None, us=None):
    """T x { predict_steadystate (kalman_filter.py:582-589) ; update_steadystate (:654-660) }:
    x = F x (+ B u) ; y = z - H x ; x = x + K y.  z None: x unchanged, y = 0 (:647-652).
    Returns means (T,n), means_p (T,n), ys (T,m)."""
    x = np.array(x0, dtype=float)
    T = len(zs)
    m = np.asarray(H).shape[0]
    means, means_p, ys = np.zeros((T,) + x.shape), np.zeros((T,) + x.shape), np.zeros((T, m))
    for t in range(T):
        if B is not None:
            x = dot(F, x) + dot(B, us[t])
        else:
            x = dot(F, x)
        means_p[t] = x
        if zs[t] is not None:
            y = zs[t] - dot(H, x)
            x = x + dot(K, y)
            ys[t] = y
        means[t] = x
    return means, means_p, ys


def update_correlated(x, P, z, R, H, M, inv=np.linalg.inv):
    """KalmanFilter.update_correlated (kalman_filter.py:727-748)."""
    y = z - dot(H, x)
    PHT = dot(P, H.T)
    S = dot(H, PHT) + dot(H, M) + dot(M.T, H.T) + R
    SI = inv(S)
    K = dot(PHT + M, SI)
    x = x + dot(K, y)
    P = P - dot(K, dot(H, P) + M.T)
    return x, P, y, K, S, SI


def update_sequential(x, P, start, z_i, R, H):
    """KalmanFilter.update_sequential (kalman_filter.py:778-824) with R_i, H_i taken from R, H.
    x is a column vector (n,1) like the reference requires (z_i is reshaped to (length,1))."""
    length = 1 if np.isscalar(z_i) else len(z_i)
    z_i = np.reshape(z_i, [length, 1])
    stop = start + length
    R_i = R[start:stop, start:stop]
    H_i = np.reshape(H[start:stop], [length, x.shape[0]])
    y_i = z_i - dot(H_i, x)
    PHT = dot(P, H_i.T)
    S_i = dot(H_i, PHT) + R_i
    if length == 1:
        K_i = PHT * (1.0 / S_i)
    else:
        K_i = dot(PHT, np.linalg.inv(S_i))
    I_KH = np.eye(x.shape[0]) - dot(K_i, H_i)
    x = x + dot(K_i, y_i)
    P = dot(dot(I_KH, P), I_KH.T) + dot(dot(K_i, R_i), K_i.T)
    return x, P, y_i, K_i
