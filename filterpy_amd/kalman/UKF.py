"""Unscented Kalman filter with filterpy's call surface (filterpy/kalman/UKF.py:
__init__ :284-340, predict :364-411, update :413-491, cross_variance :493-504,
compute_process_sigmas :506-522, batch_filter :524-632), arithmetic on the GPU.

Two ways to run it:

* general fx / hx (Python callables, like the reference): sigma points, both unscented
  transforms, the cross variance and the K / x / P correction are gfx950 kernels
  (fk_ut_sigma_points_f64, fk_ut_transform_f64, fk_ut_cross_variance_f64, fk_ukf_correct_f64);
  the callables run on the host between them, once per sigma point like the reference
  (UKF.py:521-522, :462-466), or once per call on the whole (N, 2n+1, n) array when
  ``vectorized=True``;
* linear fx / hx given as matrices (``fx=F, hx=H`` NumPy arrays): batch_filter runs the
  fused kernel fk_ukf_linear_batch_f64, the whole predict/update loop on the GPU;
* ``device_callables=True`` (banks only): fx / hx are vectorised callables on GPU tensors --
  ``fx(sigmas (N, 2n+1, n), dt, **args) -> (N, 2n+1, n)``, ``hx(sigmas (N, 2n+1, n), **args) -> (N, 2n+1, m)``,
  torch tensors in, torch tensors out -- and the whole split path stays in HBM: sigma_kernel -> fx -> ut_kernel ->
  sigma_kernel -> hx -> ut_kernel -> cross_kernel -> ukf_correct_kernel, no host copy between them
  (UKF.py:506-522, :462-481); batch_filter / rts_smoother keep state, histories and sigma points as device
  records for all T steps and download (or hand over, ``device_outputs=True``) only the results.

``n_tracks=N`` turns the object into a bank of N independent filters (x (N,n), P (N,n,n),
z (N,m) / zs (T,N,m)).  ``rts_smoother`` (UKF.py:634-739) runs its backward loop on the host like
the reference, every step's arithmetic on the GPU.

Constructor hooks (UKF.py:284-340: x_mean_fn, z_mean_fn, residual_x, residual_z, state_add; sigma_points.py:99-116:
sqrt_method, subtract) are user code like fx / hx and run where fx / hx run, between the kernels: the filter forms the
means / residuals with them and hands the RESIDUALS to the kernels (fk_ut_cross_variance_f64 with x = z = NULL sums
Wc[i] outer(dx_i, dz_i) in the reference loop's order -- UKF.py:500-503, unscented_transform.py:120-123;
fk_ukf_correct_f64 with zp = NULL takes y = residual_z(z, zp); state_add receives K y).  Calling convention per mode:
  default            exactly the reference's: mean(sigmas (k, d), Wm) -> (d,); residual(a (d,), b (d,)) -> (d,);
                     state_add(x, dx); sqrt(A (n, n)) -> (n, n); subtract(x (n,), u (n,)) -- once per track / point;
  vectorized=True    once per call on NumPy arrays: mean(sigmas (N, k, d), Wm) -> (N, d); residual(a (N, k, d), b (N, 1, d))
                     and residual(a (N, d), b (N, d)); state_add(x (N, n), dx (N, n)); sqrt(A (N, n, n)); subtract(x (N, 1, n),
                     U (N, n, n));
  device_callables   the same shapes on float64 CUDA tensors; nothing leaves HBM.
sqrt_fn is stored as ``msqrt`` and, like in the reference (UKF.py:318-321), never used by the filter itself.
"""
import sys
from copy import deepcopy
from math import exp, log, sqrt

import numpy as np

from .. import _engine as E
from ..common.helpers import logpdf

__all__ = ["UnscentedKalmanFilter"]


class UnscentedKalmanFilter(object):
    def __init__(self, dim_x, dim_z, dt, hx, fx, points, sqrt_fn=None, x_mean_fn=None, z_mean_fn=None,
                 residual_x=None, residual_z=None, state_add=None, n_tracks=None, vectorized=False,
                 layout="soa", device_callables=False):
        if device_callables and n_tracks is None:
            raise ValueError("device_callables=True needs a bank: pass n_tracks=N (use N = 1 for one filter)")
        self._dim_x, self._dim_z = dim_x, dim_z
        self._N = n_tracks
        self._devcall = bool(device_callables)
        self._vec = vectorized or n_tracks is not None and not callable(fx)
        self._layout = layout
        shape = (dim_x,) if n_tracks is None else (n_tracks, dim_x)
        self.x = np.zeros(shape)
        self.P = np.eye(dim_x) if n_tracks is None else np.tile(np.eye(dim_x), (n_tracks, 1, 1))
        self.x_prior, self.P_prior = np.copy(self.x), np.copy(self.P)
        self.Q, self.R = np.eye(dim_x), np.eye(dim_z)
        self.points_fn = points
        self._dt = dt
        self._num_sigmas = points.num_sigmas()
        self.hx, self.fx = hx, fx
        self.x_mean, self.z_mean = x_mean_fn, z_mean_fn
        self._mode = "torch" if device_callables else ("vec" if vectorized else "loop")
        self._ut_fn = None
        self._log_likelihood = log(sys.float_info.min)
        self._likelihood = sys.float_info.min
        self._mahalanobis = None
        from .sigma_points import _default_sqrt
        self.msqrt = _default_sqrt if sqrt_fn is None else sqrt_fn        # kept, never used by the filter: UKF.py:318-321
        self.Wm, self.Wc = points.Wm, points.Wc
        self.residual_x = np.subtract if residual_x is None else residual_x
        self.residual_z = np.subtract if residual_z is None else residual_z
        self.state_add = np.add if state_add is None else state_add
        self.sigmas_f = np.zeros(((self._num_sigmas, dim_x) if n_tracks is None
                                 else (n_tracks, self._num_sigmas, dim_x)))
        self.sigmas_h = np.zeros(((self._num_sigmas, dim_z) if n_tracks is None
                                 else (n_tracks, self._num_sigmas, dim_z)))
        self.K = np.zeros((dim_x, dim_z))
        self.y = np.zeros(dim_z)
        self.z = np.array([[None] * dim_z]).T
        self.S = np.zeros((dim_z, dim_z))
        self.SI = np.zeros((dim_z, dim_z))
        self.inv = np.linalg.inv
        # copies of x, P "after update()" exist from construction on (UKF.py:360-362)
        self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)

    # ---------------------------------------------------------------- helpers --
    def _b(self, a, tail):
        """attribute -> bank-shaped (N, *tail) array"""
        N = self._N or 1
        return np.broadcast_to(np.asarray(a, dtype=np.float64), (N,) + tail).copy()

    def _apply(self, fn, sig, *args, **kw):
        """run a user callable over sigma points sig (N, k, d)"""
        if not callable(fn):                       # a matrix: linear model
            return sig @ np.asarray(fn, dtype=np.float64).T
        if self._vec:
            return np.asarray(fn(sig, *args, **kw), dtype=np.float64)
        out = [[fn(s, *args, **kw) for s in trk] for trk in sig]
        return np.asarray(out, dtype=np.float64).reshape(sig.shape[0], sig.shape[1], -1)

    def _unb(self, a):
        return a if self._N is not None else a[0]

    # ------------------------------------------------- device-resident split path --
    @property
    def _hooked(self):
        """any constructor hook that is not the reference's default (attributes may be reassigned after construction)"""
        pf = self.points_fn
        return (self.x_mean is not None or self.z_mean is not None or self.residual_x is not np.subtract
                or self.residual_z is not np.subtract or self.state_add is not np.add
                or getattr(pf, "_sqrt", None) is not None or getattr(pf, "_subtract", None) is not None)

    @property
    def _linear(self):
        """fx and hx were handed over as matrices"""
        return (not callable(self.fx)) and (not callable(self.hx))

    @property
    def _resident(self):
        # (a matrix model has no host callable to run between the kernels: its steps stay on the device too)
        return self._devcall or self._hooked or self._ut_fn is not None or not callable(self.fx) or not callable(self.hx)

    def _with_ut(self, UT):
        """context: a caller-supplied UT function (UKF.py:395-396, :447-448, :712-713) for the duration of one call"""
        from contextlib import contextmanager
        from .unscented_transform import unscented_transform

        @contextmanager
        def cm():
            old = self._ut_fn
            self._ut_fn = None if (UT is None or UT is unscented_transform) else UT
            try:
                yield
            finally:
                self._ut_fn = old
        return cm()

    def _rec_view(self, rec, k, d):
        """device records of N (k x d) blocks -> torch view (N, k, d), no copy"""
        N = self._N or 1
        if self._layout == "aos":
            return rec.view(N, k, d)
        return rec.view(k, d, N).permute(2, 0, 1)

    def _vec_view(self, rec, d):
        return self._rec_view(rec, 1, d)[:, 0, :]

    def _to_rec(self, t, k, d):
        """torch tensor (N, k, d) as the callable returned it -> device records in the bank's layout"""
        import torch
        N = self._N or 1
        if (not isinstance(t, torch.Tensor) or t.device.type != E.require_gpu().type or t.dtype != torch.float64
                or tuple(t.shape) != (N, k, d)):
            raise TypeError(f"device callables must return a float64 CUDA tensor shaped {(N, k, d)}")
        if self._layout == "aos":
            return t.contiguous().view(N, k * d)
        return t.permute(1, 2, 0).contiguous().view(k * d, N)

    @staticmethod
    def _tt(a, like):
        import torch
        return torch.as_tensor(np.array(a, dtype=np.float64, order="C"), device=like.device)

    def _user(self, fn, sig, *args, **kw):
        """fx / hx on a (N, k, d) device tensor, in the mode's calling convention"""
        if not callable(fn):                                   # a matrix: linear model, one kernel (fk_ut_linear_map_f64)
            M = np.asarray(fn, dtype=np.float64)
            N, k, d_in = sig.shape
            out = E.alloc_records((), N, k * M.shape[0], self._layout)
            E.ut_linear_map(d_in, M.shape[0], k, N, self._layout, E.dev(M), self._to_rec(sig, k, d_in), out)
            return self._rec_view(out, k, M.shape[0])
        if self._mode == "torch":
            return fn(sig, *args, **kw)
        return self._tt(self._apply(fn, sig.cpu().numpy(), *args, **kw), sig)

    def _pair(self, fn, default, a, b):
        """residual_x / residual_z / state_add / subtract: a (N, [k,] d), b (N, d) device tensors -> like a"""
        bb = b.unsqueeze(1) if a.dim() == 3 else b
        if fn is default:
            return a - bb if default is not np.add else a + bb
        if self._mode == "torch":
            return fn(a, bb)
        an, bn = a.cpu().numpy(), bb.cpu().numpy()
        if self._mode == "vec":
            return self._tt(np.broadcast_to(fn(an, bn), an.shape), a)
        out = np.empty(an.shape)
        if an.ndim == 3:
            for i in range(an.shape[0]):
                for j in range(an.shape[1]):
                    out[i, j] = fn(an[i, j].copy(), bn[i, 0].copy())
        else:
            for i in range(an.shape[0]):
                out[i] = fn(an[i].copy(), bn[i].copy())
        return self._tt(out, a)

    def _mean(self, fn, sig):
        """x_mean_fn / z_mean_fn on (N, k, d) -> (N, d)"""
        if self._mode == "torch":
            return fn(sig, E.dev(np.asarray(self.Wm, dtype=np.float64)))
        sn = sig.cpu().numpy()
        if self._mode == "vec":
            return self._tt(fn(sn, self.Wm), sig)
        return self._tt(np.array([fn(sn[i].copy(), self.Wm) for i in range(sn.shape[0])]), sig)

    def _sqrt(self, fn, A):
        if self._mode == "torch":
            return fn(A)
        An = A.cpu().numpy()
        if self._mode == "vec":
            return self._tt(fn(An), A)
        return self._tt(np.array([fn(An[i].copy()) for i in range(An.shape[0])]), A)

    def _dev_consts(self):
        n, m = self._dim_x, self._dim_z
        return dict(Wm=E.dev(np.asarray(self.Wm, dtype=np.float64)), Wc=E.dev(np.asarray(self.Wc, dtype=np.float64)),
                    Q=E.dev(np.broadcast_to(np.asarray(self.Q, dtype=np.float64), (n, n)).copy()),
                    R=E.dev(np.broadcast_to(np.asarray(self.R, dtype=np.float64), (m, m)).copy()))

    def _dev_sigmas(self, dx, dP, sig, st):
        """points_fn.sigma_points on device records.  Custom sqrt_method / subtract (sigma_points.py:106-116, :168-175):
        U = sqrt(scale P) by the caller's function -- or, with only `subtract` custom, read off the kernel's points of a
        zero mean (0 + u = u exactly) -- then sigma_{k+1} = subtract(x, -U[k]), sigma_{n+k+1} = subtract(x, U[k])."""
        import torch
        n, k, N, lay = self._dim_x, self._num_sigmas, self._N or 1, self._layout
        pf = self.points_fn
        sq, sb = getattr(pf, "_sqrt", None), getattr(pf, "_subtract", None)
        if sq is None and sb is None:
            E.ut_sigma_points(n, N, lay, pf.scale, dx, dP, sig, st)
            return
        x = self._vec_view(dx, n)
        if sq is None:
            E.ut_sigma_points(n, N, lay, pf.scale, torch.zeros_like(dx), dP, sig, st)
            U = self._rec_view(sig, k, n)[:, 1:n + 1, :].clone()
        else:
            U = self._sqrt(sq, pf.scale * self._rec_view(dP, n, n))
        sub = np.subtract if sb is None else sb
        pts = torch.cat([x.unsqueeze(1), self._sub_from(sub, x, -U), self._sub_from(sub, x, U)], 1)
        sig.copy_(self._to_rec(pts, k, n).reshape(sig.shape))

    def _sub_from(self, fn, x, U):
        """subtract(x, U[k]) for every row k: x (N, n), U (N, n, n) -> (N, n, n)"""
        xx = x.unsqueeze(1)
        if fn is np.subtract:
            return xx - U
        if self._mode == "torch":
            return fn(xx, U)
        xn, Un = xx.cpu().numpy(), U.cpu().numpy()
        if self._mode == "vec":
            return self._tt(np.broadcast_to(fn(xn, Un), Un.shape), U)
        out = np.empty(Un.shape)
        for i in range(Un.shape[0]):
            for j in range(Un.shape[1]):
                out[i, j] = fn(xn[i, 0].copy(), Un[i, j].copy())
        return self._tt(out, U)

    def _dev_ut(self, s_rec, k, d, Wm, Wc, noise, mean_fn, res_fn, x_out, P_out):
        """unscented_transform (unscented_transform.py:101-126) on device records; with a custom mean / residual the
        residuals are formed by the callables and summed by the cross kernel in the reference loop's order."""
        N, lay = self._N or 1, self._layout
        if self._ut_fn is not None:
            # the caller's own transform: UT(sigmas, Wm, Wc, noise_cov, mean_fn, residual_fn) -> (x, P), called like the
            # hooks are -- per filter on NumPy arrays, once on the NumPy bank (vectorized), or once on CUDA tensors
            s = self._rec_view(s_rec, k, d)
            mf = None if mean_fn is None else mean_fn
            rf = None if res_fn is np.subtract else res_fn
            if self._mode == "torch":
                xo, Po = self._ut_fn(s, Wm, Wc, None if noise is None else noise.view(d, d), mf, rf)
            else:
                sn = s.cpu().numpy()
                nz = None if noise is None else noise.cpu().numpy().reshape(d, d)
                if self._mode == "vec":
                    xo, Po = self._ut_fn(sn, self.Wm, self.Wc, nz, mf, rf)
                else:
                    outs = [self._ut_fn(sn[i].copy(), self.Wm, self.Wc, nz, mf, rf) for i in range(N)]
                    xo, Po = np.array([o[0] for o in outs]), np.array([o[1] for o in outs])
                xo, Po = self._tt(np.reshape(xo, (N, d)), s), self._tt(np.reshape(Po, (N, d, d)), s)
            x_out.copy_(self._to_rec(xo.reshape(N, 1, d), 1, d).reshape(x_out.shape))
            P_out.copy_(self._to_rec(Po.reshape(N, d, d), d, d).reshape(P_out.shape))
            return
        if mean_fn is None and res_fn is np.subtract:
            E.ut_transform(d, k, N, lay, s_rec, Wm, Wc, noise, x_out, P_out)
            return
        s = self._rec_view(s_rec, k, d)
        if mean_fn is None:
            E.ut_transform(d, k, N, lay, s_rec, Wm, Wc, None, x_out, P_out)      # x = Wm . sigmas; P rewritten below
        else:
            x_out.copy_(self._to_rec(self._mean(mean_fn, s).unsqueeze(1), 1, d).reshape(x_out.shape))
        y = self._to_rec(self._pair(res_fn, np.subtract, s, self._vec_view(x_out, d)), k, d)
        E.ut_cross_variance(d, d, k, N, lay, None, None, y, y, Wc, P_out)
        if noise is not None:
            self._rec_view(P_out, d, d).add_(noise.view(d, d))

    def _dev_predict(self, dx, dP, c, dt, st, fx=None, **fx_args):
        """UKF.py:400-411 on device records; returns the regenerated sigma points (records)."""
        n, k, N, lay = self._dim_x, self._num_sigmas, self._N or 1, self._layout
        fx = self.fx if fx is None else fx
        sig = E.alloc_records((), N, k * n, lay)
        self._dev_sigmas(dx, dP, sig, st)
        sv = self._rec_view(sig, k, n)
        sf = self._to_rec(self._user(fx, sv, dt, **fx_args) if callable(fx) else self._user(fx, sv), k, n)
        self._dev_ut(sf, k, n, c["Wm"], c["Wc"], c["Q"], self.x_mean, self.residual_x, dx, dP)
        self._dev_sigmas(dx, dP, sig, st)                                         # UKF.py:407
        return sig

    def _dev_update(self, dx, dP, sig, dz, c, st, dK=None, hx=None, R=None, **hx_args):
        """UKF.py:462-481 on device records (sig = the sigma points the predict left behind)."""
        import torch
        n, m, k, N, lay = self._dim_x, self._dim_z, self._num_sigmas, self._N or 1, self._layout
        hx = self.hx if hx is None else hx
        sh = self._to_rec(self._user(hx, self._rec_view(sig, k, n), **hx_args), k, m)
        zp, S = E.alloc_records((), N, m, lay), E.alloc_records((), N, m * m, lay)
        self._dev_ut(sh, k, m, c["Wm"], c["Wc"], c["R"] if R is None else R, self.z_mean, self.residual_z, zp, S)
        Pxz = E.alloc_records((), N, n * m, lay)
        rx, rz = self.residual_x, self.residual_z
        if rx is np.subtract and rz is np.subtract:
            E.ut_cross_variance(n, m, k, N, lay, dx, zp, sig, sh, c["Wc"], Pxz)
        else:
            ex = self._to_rec(self._pair(rx, np.subtract, self._rec_view(sig, k, n), self._vec_view(dx, n)), k, n)
            ez = self._to_rec(self._pair(rz, np.subtract, self._rec_view(sh, k, m), self._vec_view(zp, m)), k, m)
            E.ut_cross_variance(n, m, k, N, lay, None, None, ex, ez, c["Wc"], Pxz)
        zin, zpin = dz, zp
        if rz is not np.subtract:                                                 # y = residual_z(z, zp)  (UKF.py:474)
            y = self._pair(rz, np.subtract, self._vec_view(dz, m), self._vec_view(zp, m))
            zin, zpin = self._to_rec(y.unsqueeze(1), 1, m).reshape(dz.shape), None
        if self.state_add is np.add:
            E.ukf_correct(n, m, N, lay, Pxz, zpin, S, zin, dx, dP, dK, st)
        else:                                                                     # x = state_add(x, K y)  (UKF.py:477)
            x_old = self._vec_view(dx, n).clone()
            Ky = torch.zeros_like(dx)
            E.ukf_correct(n, m, N, lay, Pxz, zpin, S, zin, Ky, dP, dK, st)
            xn = self._pair(self.state_add, np.add, x_old, self._vec_view(Ky, n))
            dx.copy_(self._to_rec(xn.unsqueeze(1), 1, n).reshape(dx.shape))
        return sh, zp, S

    def _res_predict(self, dt, fx, **fx_args):
        """predict() with state resident for the step (device callables and / or hooks)"""
        import torch
        n, k, N, lay = self._dim_x, self._num_sigmas, self._N or 1, self._layout
        dx, dP = E.to_records(self._b(self.x, (n,)), lay, 0), E.to_records(self._b(self.P, (n, n)), lay, 0)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        self._sig_dev = self._dev_predict(dx, dP, self._dev_consts(), dt, st, fx, **fx_args)
        E.raise_on_status(st, "UnscentedKalmanFilter.predict")
        self.x, self.P = self._unb(E.from_records(dx, lay, 0, (n,))), self._unb(E.from_records(dP, lay, 0, (n, n)))
        self.sigmas_f = self._unb(E.from_records(self._sig_dev, lay, 0, (k, n)))
        self.x_prior, self.P_prior = np.copy(self.x), np.copy(self.P)

    def _res_update(self, z, R, hx, **hx_args):
        import torch
        n, m, k, N, lay = self._dim_x, self._dim_z, self._num_sigmas, self._N or 1, self._layout
        dx, dP = E.to_records(self._b(self.x, (n,)), lay, 0), E.to_records(self._b(self.P, (n, n)), lay, 0)
        sig = E.to_records(np.asarray(self.sigmas_f, dtype=np.float64).reshape(N, k, n), lay, 0)
        zb = np.asarray(z, dtype=np.float64).reshape(N, m)
        dz, dK = E.to_records(zb, lay, 0), E.alloc_records((), N, n * m, lay)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        c = self._dev_consts()
        Rd = None if R is None else E.dev(np.eye(m) * R if np.isscalar(R) else np.broadcast_to(np.asarray(R, dtype=np.float64), (m, m)).copy())
        sh, zp, S = self._dev_update(dx, dP, sig, dz, c, st, dK=dK, hx=hx, R=Rd, **hx_args)
        E.raise_on_status(st, "UnscentedKalmanFilter.update")
        zpn, Sn = E.from_records(zp, lay, 0, (m,)), E.from_records(S, lay, 0, (m, m))
        self.sigmas_h = self._unb(E.from_records(sh, lay, 0, (k, m)))
        self.S, self.SI = self._unb(Sn), self._unb(np.linalg.inv(Sn))
        self.K = self._unb(E.from_records(dK, lay, 0, (n, m)))
        if self.residual_z is np.subtract:
            self.y = self._unb(zb - zpn)
        else:
            self.y = self._unb(self._pair(self.residual_z, np.subtract, self._vec_view(dz, m), self._vec_view(zp, m)).cpu().numpy())
        self.x, self.P = self._unb(E.from_records(dx, lay, 0, (n,))), self._unb(E.from_records(dP, lay, 0, (n, n)))

    def _dev_batch_filter(self, zs, Rs, dts, device_outputs):
        import torch
        n, m, N, lay = self._dim_x, self._dim_z, self._N or 1, self._layout
        T = len(zs)
        c = self._dev_consts()
        dx, dP = E.to_records(self._b(self.x, (n,)), lay, 0), E.to_records(self._b(self.P, (n, n)), lay, 0)
        if isinstance(zs, torch.Tensor):           # already device records [T][N][m] (aos) / [T][m][N] (soa)
            dzs, present = zs, [True] * T
        else:
            present = [z is not None for z in zs]
            zarr = np.zeros((T, N, m))
            for i, z in enumerate(zs):
                if z is not None:
                    zarr[i] = np.asarray(z, dtype=np.float64).reshape(N, m)
            dzs = E.to_records(zarr, lay, 1)
        means, covs = E.alloc_records((T,), N, n, lay), E.alloc_records((T,), N, n * n, lay)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        k = self._num_sigmas
        last_upd = max([t for t in range(T) if present[t]], default=-1)
        last = prior = None
        for t in range(T):
            dt = self._dt if dts is None else dts[t]
            sig = self._dev_predict(dx, dP, c, dt, st)
            if t == T - 1:
                prior = (dx.clone(), dP.clone())
            if present[t]:
                Rt = None
                if Rs is not None:
                    r = Rs[t]
                    Rt = E.dev(np.eye(m) * r if np.isscalar(r) else np.broadcast_to(np.asarray(r, dtype=np.float64), (m, m)).copy())
                dK = E.alloc_records((), N, n * m, lay) if t == last_upd else None
                sh, zp, S = self._dev_update(dx, dP, sig, dzs[t], c, st, dK=dK, R=Rt)
                if t == last_upd:
                    last = (sh, zp, S, dK, dzs[t])
            means[t].copy_(dx.reshape(means[t].shape))      # (aos records keep the host array's trailing shape)
            covs[t].copy_(dP.reshape(covs[t].shape))
        E.raise_on_status(st, "UnscentedKalmanFilter.batch_filter")
        self.x, self.P = self._unb(E.from_records(dx, lay, 0, (n,))), self._unb(E.from_records(dP, lay, 0, (n, n)))
        # leave the object where the reference's per-epoch loop leaves it (UKF.py:623-632 -> :400-411, :462-491): the
        # attributes of the last predict and of the last update that ran
        if T:
            self.sigmas_f = self._unb(E.from_records(sig, lay, 0, (k, n)))
            self.x_prior = self._unb(E.from_records(prior[0], lay, 0, (n,)))
            self.P_prior = self._unb(E.from_records(prior[1], lay, 0, (n, n)))
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
        if last is not None:
            sh, zp, S, dK, dz = last
            zpn, Sn = E.from_records(zp, lay, 0, (m,)), E.from_records(S, lay, 0, (m, m))
            zn = E.from_records(dz.reshape(zp.shape), lay, 0, (m,))
            self.sigmas_h = self._unb(E.from_records(sh, lay, 0, (k, m)))
            self.S, self.SI = self._unb(Sn), self._unb(np.linalg.inv(Sn))
            self.K = self._unb(E.from_records(dK, lay, 0, (n, m)))
            if self.residual_z is np.subtract:
                self.y = self._unb(zn - zpn)
            else:
                self.y = self._unb(self._pair(self.residual_z, np.subtract, self._vec_view(dz.reshape(zp.shape), m),
                                              self._vec_view(zp, m)).cpu().numpy())
            self._log_likelihood = self._likelihood = self._mahalanobis = None
        if T:
            self.z = np.array([[None] * m]).T if last_upd != T - 1 else (zn if self._N is not None else zn[0])
        if device_outputs:
            return means, covs
        mu, cov = E.from_records(means, lay, 1, (n,)), E.from_records(covs, lay, 1, (n, n))
        return (mu, cov) if self._N is not None else (mu[:, 0], cov[:, 0])

    # ---------------------------------------------------------------- predict --
    def compute_process_sigmas(self, dt, fx=None, **fx_args):
        """UKF.py:506-522."""
        fx = self.fx if fx is None else fx
        n = self._dim_x
        if self._resident:
            import torch
            k, N, lay = self._num_sigmas, self._N or 1, self._layout
            dx, dP = E.to_records(self._b(self.x, (n,)), lay, 0), E.to_records(self._b(self.P, (n, n)), lay, 0)
            st = torch.zeros(N, dtype=torch.int32, device=dx.device)
            sig = E.alloc_records((), N, k * n, lay)
            self._dev_sigmas(dx, dP, sig, st)
            E.raise_on_status(st, "UnscentedKalmanFilter.compute_process_sigmas")
            sv = self._rec_view(sig, k, n)
            sf = self._to_rec(self._user(fx, sv, dt, **fx_args) if callable(fx) else self._user(fx, sv), k, n)
            self.sigmas_f = self._unb(E.from_records(sf, lay, 0, (k, n)))
            return
        sig = self.points_fn.sigma_points(self._b(self.x, (n,)), self._b(self.P, (n, n)))
        sf = self._apply(fx, sig, dt, **fx_args) if callable(fx) else self._apply(fx, sig)
        self.sigmas_f = self._unb(sf)

    def predict(self, dt=None, UT=None, fx=None, **fx_args):
        """UKF.py:364-411: sigma points -> fx -> UT(+Q) -> regenerate sigma points."""
        from .unscented_transform import unscented_transform
        dt = self._dt if dt is None else dt
        n = self._dim_x
        with self._with_ut(UT):
            if self._resident:
                return self._res_predict(dt, fx, **fx_args)
        self.compute_process_sigmas(dt, fx, **fx_args)
        sf = np.asarray(self.sigmas_f).reshape(-1, self._num_sigmas, n)
        x, P = unscented_transform(sf, self.Wm, self.Wc, self.Q, layout=self._layout)
        self.x, self.P = self._unb(x), self._unb(P)
        self.sigmas_f = self._unb(self.points_fn.sigma_points(x, P))
        self.x_prior, self.P_prior = np.copy(self.x), np.copy(self.P)

    # ----------------------------------------------------------------- update --
    def update(self, z, R=None, UT=None, hx=None, **hx_args):
        """UKF.py:413-491."""
        import torch
        from .unscented_transform import unscented_transform
        if z is None:
            self.z = np.array([[None] * self._dim_z]).T
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
            return
        hx = self.hx if hx is None else hx
        n, m, k = self._dim_x, self._dim_z, self._num_sigmas
        N = self._N or 1
        with self._with_ut(UT):
            resident = self._resident
            if resident:
                self._res_update(z, R, hx, **hx_args)
        if resident:
            self.z = deepcopy(z)
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
            self._log_likelihood = self._likelihood = self._mahalanobis = None
            return
        if R is None:
            R = self.R
        elif np.isscalar(R):
            R = np.eye(m) * R
        sf = np.asarray(self.sigmas_f, dtype=np.float64).reshape(N, k, n)
        sh = self._apply(hx, sf, **hx_args).reshape(N, k, m)
        self.sigmas_h = self._unb(sh)
        zp, S = unscented_transform(sh, self.Wm, self.Wc, R, layout=self._layout)
        lay = self._layout
        xb, Pb = self._b(self.x, (n,)), self._b(self.P, (n, n))
        zb = np.asarray(z, dtype=np.float64).reshape(N, m)
        dx, dP = E.to_records(xb, lay, 0), E.to_records(Pb, lay, 0)
        dzp, dS, dz = E.to_records(zp, lay, 0), E.to_records(S, lay, 0), E.to_records(zb, lay, 0)
        dPxz = E.alloc_records((), N, n * m, lay)
        E.ut_cross_variance(n, m, k, N, lay, dx, dzp, E.to_records(sf, lay, 0), E.to_records(sh, lay, 0),
                            E.dev(np.asarray(self.Wc, dtype=np.float64)), dPxz)
        dK = E.alloc_records((), N, n * m, lay)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.ukf_correct(n, m, N, lay, dPxz, dzp, dS, dz, dx, dP, dK, st)
        E.raise_on_status(st, "UnscentedKalmanFilter.update")
        self.S = self._unb(S)
        self.SI = self._unb(np.linalg.inv(S))         # attribute only; the kernel used a Cholesky solve
        self.K = self._unb(E.from_records(dK, lay, 0, (n, m)))
        self.y = self._unb(zb - zp)
        self.x = self._unb(E.from_records(dx, lay, 0, (n,)))
        self.P = self._unb(E.from_records(dP, lay, 0, (n, n)))
        self.z = deepcopy(z)
        self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
        self._log_likelihood = None
        self._likelihood = None
        self._mahalanobis = None

    def cross_variance(self, x, z, sigmas_f, sigmas_h):
        """UKF.py:493-504 for one filter."""
        n, m, k = sigmas_f.shape[1], sigmas_h.shape[1], sigmas_f.shape[0]
        lay = self._layout
        out = E.alloc_records((), 1, n * m, lay)
        if self.residual_x is not np.subtract or self.residual_z is not np.subtract:
            if self._mode != "loop":
                raise ValueError("cross_variance() is the reference's one-filter call: hooks in the per-vector convention")
            ex = np.array([self.residual_x(np.array(s, dtype=np.float64), np.array(x, dtype=np.float64)) for s in sigmas_f])
            ez = np.array([self.residual_z(np.array(s, dtype=np.float64), np.array(z, dtype=np.float64)) for s in sigmas_h])
            E.ut_cross_variance(n, m, k, 1, lay, None, None, E.to_records(ex[None], lay, 0), E.to_records(ez[None], lay, 0),
                                E.dev(np.asarray(self.Wc, dtype=np.float64)), out)
            return E.from_records(out, lay, 0, (n, m))[0]
        E.ut_cross_variance(n, m, k, 1, lay, E.to_records(np.reshape(x, (1, n)), lay, 0),
                            E.to_records(np.reshape(z, (1, m)), lay, 0),
                            E.to_records(np.asarray(sigmas_f)[None], lay, 0),
                            E.to_records(np.asarray(sigmas_h)[None], lay, 0),
                            E.dev(np.asarray(self.Wc, dtype=np.float64)), out)
        return E.from_records(out, lay, 0, (n, m))[0]

    # ----------------------------------------------------------- batch_filter --
    def batch_filter(self, zs, Rs=None, dts=None, UT=None, saver=None, device_outputs=False):
        """UKF.py:524-632: predict -> update per measurement; returns (means, covariances).
        With linear fx/hx matrices and no per-epoch Rs/dts/saver the whole loop is ONE fused
        kernel launch."""
        import torch
        from .unscented_transform import unscented_transform
        n, m = self._dim_x, self._dim_z
        fused = (self._linear and Rs is None and dts is None and saver is None and UT is None and not self._devcall
                 and not self._hooked and not isinstance(zs, torch.Tensor)
                 and E.ukf_linear_supported(n, m, n >= 10 and E.pair_weights(self.Wm, self.Wc, n)))
        with self._with_ut(UT):
            if self._resident and saver is None and not fused:
                return self._dev_batch_filter(zs, Rs, dts, device_outputs)
        try:
            z0 = zs[0]
        except TypeError:
            raise TypeError('zs must be list-like') from None
        if self._N is None:
            if m == 1:
                if not (np.isscalar(z0) or (np.ndim(z0) == 1 and len(z0) == 1)):
                    raise TypeError('zs must be a list of scalars or 1D, 1 element arrays')
            elif len(z0) != m:
                raise TypeError('each element in zs must be a 1D array of length {}'.format(m))
        T = len(zs)
        N = self._N or 1
        if fused:
            lay = self._layout
            zarr = np.zeros((T, N, m))
            mask = np.ones((T, N), dtype=np.uint8)
            for i, zi in enumerate(zs):
                if zi is None:
                    mask[i] = 0
                else:
                    zarr[i] = np.asarray(zi, dtype=np.float64).reshape(N, m)
            x_init, P_init = np.array(self.x, dtype=np.float64), np.array(self.P, dtype=np.float64)
            dx, dP = E.to_records(self._b(self.x, (n,)), lay, 0), E.to_records(self._b(self.P, (n, n)), lay, 0)
            means, covs = E.alloc_records((T,), N, n, lay), E.alloc_records((T,), N, n * n, lay)
            st = torch.zeros(N, dtype=torch.int32, device=dx.device)
            dm = None if mask.all() else torch.as_tensor(mask, device=dx.device)
            E.ukf_linear_batch(n, m, N, T, lay, self.points_fn.scale, E.dev(self.fx), E.dev(self.hx),
                               E.dev(np.broadcast_to(self.Q, (n, n)).copy()), E.dev(np.broadcast_to(self.R, (m, m)).copy()),
                               E.dev(np.asarray(self.Wm, dtype=np.float64)), E.dev(np.asarray(self.Wc, dtype=np.float64)),
                               E.to_records(zarr, lay, 1), dx, dP, mask=dm, means=means, covs=covs, status=st,
                               paired=E.pair_weights(self.Wm, self.Wc, n))
            E.raise_on_status(st, "UnscentedKalmanFilter.batch_filter")
            x_end = self._unb(E.from_records(dx, lay, 0, (n,)))
            P_end = self._unb(E.from_records(dP, lay, 0, (n, n)))
            mu, cov = E.from_records(means, lay, 1, (n,)), E.from_records(covs, lay, 1, (n, n))
            # The reference's loop leaves the LAST epoch's by-products on the filter (UKF.py:623-632: x_prior / P_prior, sigmas_f,
            # z, x_post / P_post, the lazy likelihoods reset) and -- update(None) returns early, UKF.py:443-447 -- the K, S, SI,
            # y, sigmas_h of the last epoch that HAD a measurement; the fused launch keeps all of them in registers.  Those
            # epochs (one, or two when the call ends on missing measurements: ADVICE r4) are replayed through predict() /
            # update() from the state before them; x / P themselves stay the fused launch's, so that means[-1] is self.x bit
            # for bit like in the reference.
            if T >= 1:
                def before(t):
                    return (x_init, P_init) if t == 0 else (self._unb(np.array(mu[t - 1])), self._unb(np.array(cov[t - 1])))
                seen = [t for t in range(T) if zs[t] is not None]
                for t in ([seen[-1]] if seen and seen[-1] != T - 1 else []) + [T - 1]:
                    self.x, self.P = before(t)
                    self.predict()
                    self.update(zs[t])
            self.x, self.P = x_end, P_end
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
            return (mu, cov) if self._N is not None else (mu[:, 0], cov[:, 0])
        Rs = [self.R] * T if Rs is None else Rs
        dts = [self._dt] * T if dts is None else dts
        means = np.zeros((T,) + np.shape(self.x))
        covs = np.zeros((T,) + np.shape(self.P))
        for i, (z, r, dt) in enumerate(zip(zs, Rs, dts)):
            self.predict(dt=dt, UT=UT)
            self.update(z, r, UT=UT)
            means[i], covs[i] = self.x, self.P
            if saver is not None:
                saver.save()
        return (means, covs)

    def _dev_rts_smoother(self, Xs, Ps, dts, device_outputs):
        """The backward pass with everything resident: Xs / Ps are uploaded once (or arrive as device records
        [T][N][..]), every step is sigma_kernel -> fx -> ut_kernel -> cross_kernel -> ukf_rts_kernel on device
        records, and only the results come back."""
        import torch
        n, k, N, lay = self._dim_x, self._num_sigmas, self._N or 1, self._layout
        c = self._dev_consts()
        single_shape = None
        if isinstance(Xs, torch.Tensor):
            dXs, dPs = Xs, Ps
            T = int(Xs.shape[0])
        else:
            Xs = np.asarray(Xs, dtype=np.float64)
            T = Xs.shape[0]
            single_shape = Xs.shape if self._N is None else None
            dXs = E.to_records(Xs.reshape(T, N, n), lay, 1)
            dPs = E.to_records(np.asarray(Ps, dtype=np.float64).reshape(T, N, n, n), lay, 1)
        if dts is None:
            dts = [self._dt] * T
        elif np.isscalar(dts):
            dts = [dts] * T
        xs, ps = dXs.clone(), dPs.clone()
        Ks = torch.zeros_like(ps)
        sig = E.alloc_records((), N, k * n, lay)
        xb, Pb, Pxb = E.alloc_records((), N, n, lay), E.alloc_records((), N, n * n, lay), E.alloc_records((), N, n * n, lay)
        st = torch.zeros(N, dtype=torch.int32, device=xs.device)
        rx = self.residual_x
        for j in reversed(range(T - 1)):
            self._dev_sigmas(xs[j], ps[j], sig, st)
            sv = self._rec_view(sig, k, n)
            sf = self._to_rec(self._user(self.fx, sv, dts[j]) if callable(self.fx) else self._user(self.fx, sv), k, n)
            self._dev_ut(sf, k, n, c["Wm"], c["Wc"], c["Q"], self.x_mean, rx, xb, Pb)
            if rx is np.subtract:
                E.ut_cross_variance(n, n, k, N, lay, dXs[j], xb, sig, sf, c["Wc"], Pxb)
                E.ukf_rts_correct(n, N, lay, Pxb, xb, Pb, xs[j + 1], ps[j + 1], xs[j], ps[j], Ks[j], st)
            else:                        # UKF.py:722-731 with a custom residual_x: z around Xs[k], y around xb, x[k+1] - xb
                ez = self._to_rec(self._pair(rx, np.subtract, sv, self._vec_view(dXs[j], n)), k, n)
                ey = self._to_rec(self._pair(rx, np.subtract, self._rec_view(sf, k, n), self._vec_view(xb, n)), k, n)
                E.ut_cross_variance(n, n, k, N, lay, None, None, ez, ey, c["Wc"], Pxb)
                dn = self._pair(rx, np.subtract, self._vec_view(xs[j + 1], n), self._vec_view(xb, n))
                E.ukf_rts_correct(n, N, lay, Pxb, None, Pb, self._to_rec(dn.unsqueeze(1), 1, n).reshape(xb.shape), ps[j + 1],
                                  xs[j], ps[j], Ks[j], st)
        E.raise_on_status(st, "UnscentedKalmanFilter.rts_smoother")
        if device_outputs:
            return xs, ps, Ks
        out = E.from_records(xs, lay, 1, (n,)), E.from_records(ps, lay, 1, (n, n)), E.from_records(Ks, lay, 1, (n, n))
        if single_shape is not None:
            return out[0][:, 0].reshape(single_shape), out[1][:, 0], out[2][:, 0]
        return out

    def rts_smoother(self, Xs, Ps, Qs=None, dts=None, UT=None, device_outputs=False):
        """UKF.py:634-739: backward pass over the filter output.  Per step k (from the end): sigma
        points of (xs[k], ps[k]) -> fx -> UT (+ self.Q: the reference passes self.Q, its Qs argument
        is never read, UKF.py:717-719) -> cross variance of the sigma points around Xs[k] and their
        images around xb -> fk_ukf_rts_correct_f64 (K = Pxb inv(Pb); x += K (x[k+1] - xb);
        P += K (P[k+1] - Pb) K').  Xs (T, n) / Ps (T, n, n), or (T, N, n) / (T, N, n, n) for a bank.
        Returns (xs, Ps, Ks)."""
        import torch
        from .unscented_transform import unscented_transform
        if len(Xs) != len(Ps):
            raise ValueError('Xs and Ps must have the same length')
        fused = (not callable(self.fx) and dts is None and UT is None and not self._devcall and not self._hooked
                 and not isinstance(Xs, torch.Tensor)
                 and E.ukf_linear_rts_supported(self._dim_x, self._dim_x >= 10 and E.pair_weights(self.Wm, self.Wc, self._dim_x)))
        with self._with_ut(UT):
            if (self._resident or isinstance(Xs, torch.Tensor)) and not fused:
                return self._dev_rts_smoother(Xs, Ps, dts, device_outputs)
        if fused:
            # linear fx given as a matrix: the whole backward loop is ONE fused launch (fk_ukf_linear_rts_f64)
            n, N, lay = self._dim_x, self._N or 1, self._layout
            Xa = np.asarray(Xs, dtype=np.float64)
            T = Xa.shape[0]
            dX = E.to_records(Xa.reshape(T, N, n), lay, 1)
            dP = E.to_records(np.asarray(Ps, dtype=np.float64).reshape(T, N, n, n), lay, 1)
            oxs, ops, oK = E.alloc_records((T,), N, n, lay), E.alloc_records((T,), N, n * n, lay), E.alloc_records((T,), N, n * n, lay)
            st = torch.zeros(N, dtype=torch.int32, device=dX.device)
            E.ukf_linear_rts(n, N, T, lay, self.points_fn.scale, E.dev(np.asarray(self.fx, dtype=np.float64)),
                             E.dev(np.broadcast_to(np.asarray(self.Q, dtype=np.float64), (n, n)).copy()),
                             E.dev(np.asarray(self.Wm, dtype=np.float64)), E.dev(np.asarray(self.Wc, dtype=np.float64)),
                             dX, dP, oxs, ops, oK, st, paired=E.pair_weights(self.Wm, self.Wc, n))
            E.raise_on_status(st, "UnscentedKalmanFilter.rts_smoother")
            xs, ps, Ks = E.from_records(oxs, lay, 1, (n,)), E.from_records(ops, lay, 1, (n, n)), E.from_records(oK, lay, 1, (n, n))
            if self._N is None:
                return xs[:, 0].reshape(Xa.shape), ps[:, 0], Ks[:, 0]
            return xs, ps, Ks
        n, k = self._dim_x, self._num_sigmas
        N = self._N or 1
        Xs = np.asarray(Xs, dtype=np.float64)
        T = Xs.shape[0]
        if dts is None:
            dts = [self._dt] * T
        elif np.isscalar(dts):
            dts = [dts] * T
        lay = self._layout
        X = Xs.reshape(T, N, n)
        xs, ps = X.copy(), np.asarray(Ps, dtype=np.float64).reshape(T, N, n, n).copy()
        Ks = np.zeros((T, N, n, n))
        dWc = E.dev(np.asarray(self.Wc, dtype=np.float64))
        for j in reversed(range(T - 1)):
            sig = self.points_fn.sigma_points(xs[j], ps[j]).reshape(N, k, n)
            sf = (self._apply(self.fx, sig, dts[j]) if callable(self.fx) else self._apply(self.fx, sig)).reshape(N, k, n)
            xb, Pb = unscented_transform(sf, self.Wm, self.Wc, self.Q, layout=lay)
            dxb, dPb = E.to_records(xb.reshape(N, n), lay, 0), E.to_records(Pb.reshape(N, n, n), lay, 0)
            dPxb = E.alloc_records((), N, n * n, lay)
            E.ut_cross_variance(n, n, k, N, lay, E.to_records(X[j], lay, 0), dxb, E.to_records(sig, lay, 0),
                                E.to_records(sf, lay, 0), dWc, dPxb)
            dx, dP = E.to_records(xs[j], lay, 0), E.to_records(ps[j], lay, 0)
            dK = E.alloc_records((), N, n * n, lay)
            st = torch.zeros(N, dtype=torch.int32, device=dx.device)
            E.ukf_rts_correct(n, N, lay, dPxb, dxb, dPb, E.to_records(xs[j + 1], lay, 0),
                              E.to_records(ps[j + 1], lay, 0), dx, dP, dK, st)
            E.raise_on_status(st, "UnscentedKalmanFilter.rts_smoother")
            xs[j] = E.from_records(dx, lay, 0, (n,))
            ps[j] = E.from_records(dP, lay, 0, (n, n))
            Ks[j] = E.from_records(dK, lay, 0, (n, n))
        if self._N is None:
            return xs[:, 0].reshape(Xs.shape), ps[:, 0], Ks[:, 0]
        return xs, ps, Ks

    # ------------------------------------------------------------- properties --
    @property
    def log_likelihood(self):
        if self._log_likelihood is None:
            self._log_likelihood = logpdf(x=self.y, cov=self.S)
        return self._log_likelihood

    @property
    def likelihood(self):
        if self._likelihood is None:
            self._likelihood = exp(self.log_likelihood)
            if self._likelihood == 0:
                self._likelihood = sys.float_info.min
        return self._likelihood

    @property
    def mahalanobis(self):
        if self._mahalanobis is None:
            y = np.asarray(self.y, dtype=float).reshape(-1)
            self._mahalanobis = sqrt(float(y @ self.SI @ y))
        return self._mahalanobis
