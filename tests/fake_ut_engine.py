"""CPU stand-ins for the unscented-transform entry points of libfilterhip.so, for HOST-LOGIC tests only: they let the
Python layer of filterpy_amd.kalman.UKF (record layouts, hook calling conventions, the order things are handed to the
kernels, the NULL-mean modes of the ABI) run on CPU tensors.  The arithmetic inside each stand-in is the oracle's
(tests may use it); the kernels themselves are tested under -m gpu."""
import numpy as np
import torch
from scipy.linalg import cholesky

from oracle import ukf_oracle as uo

CPU = torch.device("cpu")


def _get(rec, layout, shape):
    """records (one bank, no time axis) -> numpy (N, *shape)"""
    a = rec.detach().numpy()
    E = int(np.prod(shape))
    if layout == "soa":
        a = a.reshape(E, -1).T
    return np.array(a.reshape(-1, *shape))


def _put(rec, layout, arr):
    N = arr.shape[0]
    a = arr.reshape(N, -1)
    if layout == "soa":
        a = a.T
    rec.copy_(torch.as_tensor(np.ascontiguousarray(a)).reshape(rec.shape))


def install(monkeypatch):
    from filterpy_amd import _engine as E
    calls = []
    monkeypatch.setattr(E, "require_gpu", lambda: CPU)
    monkeypatch.setattr(E, "dev", lambda a, device=None: (a.to(dtype=torch.float64).contiguous() if isinstance(a, torch.Tensor)
                                                          else torch.as_tensor(np.array(a, dtype=np.float64))))
    monkeypatch.setattr(E, "alloc_records", lambda lead, N, Ee, layout, device=None:
                        torch.full((*lead, N, Ee) if layout == "aos" else (*lead, Ee, N), float("nan"), dtype=torch.float64))

    def ut_sigma_points(n, N, layout, scale, x, P, sig, status=None):
        calls.append("sigma")
        xs, Ps = _get(x, layout, (n,)), _get(P, layout, (n, n))
        out = np.zeros((N, 2 * n + 1, n))
        for i in range(N):
            U = cholesky(scale * Ps[i])
            out[i, 0] = xs[i]
            for k in range(n):
                out[i, k + 1] = xs[i] - (-U[k])
                out[i, n + k + 1] = xs[i] - U[k]
        _put(sig, layout, out)

    def ut_transform(n, k, N, layout, sig, Wm, Wc, noise, xo, Po):
        calls.append("ut")
        s = _get(sig, layout, (k, n))
        xs, Ps = np.zeros((N, n)), np.zeros((N, n, n))
        for i in range(N):
            xs[i], Ps[i] = uo.unscented_transform(s[i], Wm.numpy(), Wc.numpy(), None if noise is None else noise.numpy().reshape(n, n))
        _put(xo, layout, xs)
        _put(Po, layout, Ps)

    def ut_cross_variance(n, m, k, N, layout, x, z, sf, sh, Wc, Pxz):
        calls.append("cross" if x is not None else "cross_residuals")
        assert (x is None) == (z is None)
        f, h = _get(sf, layout, (k, n)), _get(sh, layout, (k, m))
        xs = np.zeros((N, n)) if x is None else _get(x, layout, (n,))
        zs = np.zeros((N, m)) if z is None else _get(z, layout, (m,))
        out = np.zeros((N, n, m))
        for i in range(N):
            out[i] = uo.cross_variance(xs[i], zs[i], f[i], h[i], Wc.numpy())
        _put(Pxz, layout, out)

    def ukf_correct(n, m, N, layout, Pxz, zp, S, z, x, P, K=None, status=None):
        calls.append("correct" if zp is not None else "correct_residual")
        pxz, Ss, zs = _get(Pxz, layout, (n, m)), _get(S, layout, (m, m)), _get(z, layout, (m,))
        zps = np.zeros((N, m)) if zp is None else _get(zp, layout, (m,))
        xs, Ps = _get(x, layout, (n,)), _get(P, layout, (n, n))
        Ks = np.zeros((N, n, m))
        for i in range(N):
            Ks[i] = pxz[i] @ np.linalg.inv(Ss[i])
            xs[i] = xs[i] + Ks[i] @ (zs[i] - zps[i])
            Ps[i] = Ps[i] - Ks[i] @ (Ss[i] @ Ks[i].T)
        _put(x, layout, xs)
        _put(P, layout, Ps)
        if K is not None:
            _put(K, layout, Ks)

    def ukf_rts_correct(n, N, layout, Pxb, xb, Pb, xn, Pn, x, P, K=None, status=None):
        calls.append("rts" if xb is not None else "rts_residual")
        pxb, pb, xns, pns = _get(Pxb, layout, (n, n)), _get(Pb, layout, (n, n)), _get(xn, layout, (n,)), _get(Pn, layout, (n, n))
        xbs = np.zeros((N, n)) if xb is None else _get(xb, layout, (n,))
        xs, Ps = _get(x, layout, (n,)), _get(P, layout, (n, n))
        Ks = np.zeros((N, n, n))
        for i in range(N):
            Ks[i] = pxb[i] @ np.linalg.inv(pb[i])
            xs[i] = xs[i] + Ks[i] @ (xns[i] - xbs[i])
            Ps[i] = Ps[i] + (Ks[i] @ (pns[i] - pb[i])) @ Ks[i].T
        _put(x, layout, xs)
        _put(P, layout, Ps)
        if K is not None:
            _put(K, layout, Ks)

    def ut_linear_map(n_in, n_out, k, N, layout, M, sig_in, sig_out):
        calls.append("linear_map")
        s = _get(sig_in, layout, (k, n_in))
        _put(sig_out, layout, s @ M.numpy().reshape(n_out, n_in).T)

    def ukf_linear_batch(n, m, N, T, layout, scale, F, H, Q, R, Wm, Wc, z, x, P, *, mask=None, means=None, covs=None, status=None,
                         paired=None):
        """fk_ukf_linear_batch_f64: T x { UKF.predict ; UKF.update } (UKF.py:364-491) with fx = F x, hx = H x, any weight set"""
        calls.append("fused_batch")
        Fh, Hh, Qh, Rh, wm, wc = F.numpy().reshape(n, n), H.numpy().reshape(m, n), Q.numpy().reshape(n, n), R.numpy().reshape(m, m), Wm.numpy(), Wc.numpy()
        zs = z.numpy().reshape(T, N, m) if layout == "aos" else np.swapaxes(z.numpy().reshape(T, m, N), 1, 2)
        mk = None if mask is None else mask.numpy().reshape(T, N)
        xs, Ps = _get(x, layout, (n,)), _get(P, layout, (n, n))
        mu, cov = np.zeros((T, N, n)), np.zeros((T, N, n, n))

        def sig(xx, PP):
            U = cholesky(scale * PP)
            return np.vstack([xx, xx + U, xx - U])
        for i in range(N):
            xi, Pi = xs[i].copy(), Ps[i].copy()
            for t in range(T):
                xi, Pi = uo.unscented_transform(sig(xi, Pi) @ Fh.T, wm, wc, Qh)
                sf = sig(xi, Pi)
                if mk is None or mk[t, i]:
                    sh = sf @ Hh.T
                    zp, S = uo.unscented_transform(sh, wm, wc, Rh)
                    Pxz = uo.cross_variance(xi, zp, sf, sh, wc)
                    K = Pxz @ np.linalg.inv(S)
                    xi = xi + K @ (zs[t, i] - zp)
                    Pi = Pi - K @ (S @ K.T)
                mu[t, i], cov[t, i] = xi, Pi
            xs[i], Ps[i] = xi, Pi
        _put(x, layout, xs)
        _put(P, layout, Ps)
        for rec, arr in ((means, mu), (covs, cov)):
            if rec is not None:
                a = arr.reshape(T, N, -1)
                rec.copy_(torch.as_tensor(np.ascontiguousarray(a if layout == "aos" else np.swapaxes(a, 1, 2))).reshape(rec.shape))

    def ukf_linear_rts(n, N, T, layout, scale, F, Q, Wm, Wc, Xs, Ps, xs, Ps_out, K=None, status=None, paired=None):
        """fk_ukf_linear_rts_f64: UKF.rts_smoother (UKF.py:714-739) with fx = F x"""
        calls.append("fused_rts")
        Fh, Qh, wm, wc = F.numpy().reshape(n, n), Q.numpy().reshape(n, n), Wm.numpy(), Wc.numpy()
        unrec = lambda t, d: (t.numpy().reshape(T, N, d) if layout == "aos" else np.swapaxes(t.numpy().reshape(T, d, N), 1, 2))  # noqa: E731
        X, Pm = unrec(Xs, n).copy(), unrec(Ps, n * n).reshape(T, N, n, n).copy()
        ox, oP, oK = X.copy(), Pm.copy(), np.zeros((T, N, n, n))
        for i in range(N):
            for k in reversed(range(T - 1)):
                U = cholesky(scale * oP[k, i])
                s = np.vstack([ox[k, i], ox[k, i] + U, ox[k, i] - U])
                sf = s @ Fh.T
                xb, Pb = uo.unscented_transform(sf, wm, wc, Qh)
                Pxb = 0
                for j in range(2 * n + 1):
                    Pxb = Pxb + wc[j] * np.outer(s[j] - X[k, i], sf[j] - xb)
                Kk = Pxb @ np.linalg.inv(Pb)
                ox[k, i] = ox[k, i] + Kk @ (ox[k + 1, i] - xb)
                oP[k, i] = oP[k, i] + (Kk @ (oP[k + 1, i] - Pb)) @ Kk.T
                oK[k, i] = Kk
        for rec, arr in ((xs, ox), (Ps_out, oP), (K, oK)):
            if rec is not None:
                a = arr.reshape(T, N, -1)
                rec.copy_(torch.as_tensor(np.ascontiguousarray(a if layout == "aos" else np.swapaxes(a, 1, 2))).reshape(rec.shape))

    monkeypatch.setattr(E, "ukf_linear_supported", lambda n, m, paired=False: True)
    monkeypatch.setattr(E, "ukf_linear_rts_supported", lambda n, paired=False: True)
    for name, fn in dict(ut_sigma_points=ut_sigma_points, ut_transform=ut_transform, ut_cross_variance=ut_cross_variance,
                         ukf_correct=ukf_correct, ukf_rts_correct=ukf_rts_correct, ut_linear_map=ut_linear_map,
                         ukf_linear_batch=ukf_linear_batch, ukf_linear_rts=ukf_linear_rts).items():
        monkeypatch.setattr(E, name, fn)
    return calls
