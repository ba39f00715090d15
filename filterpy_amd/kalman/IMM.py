"""Interacting multiple model estimator with filterpy's call surface (filterpy/kalman/IMM.py:
__init__ :124-158, update :160-186, predict :188-222, _compute_state_estimate :224-237,
_compute_mixing_probabilities :239-249), the whole bank arithmetic in one gfx950 kernel
(fk_imm_batch_f64, filterpy_amd/csrc/imm_kernels.hip).

The reference loops over its filters in Python and lets each ``KalmanFilter`` do its own
predict/update; here the filters only supply their model (F, Q, H, R) and their state (x, P),
one lane runs the mixing, every filter's predict/update, the likelihoods and the mode
probabilities of one track, and the filters' ``x``/``P`` are written back after every call so code
that inspects ``imm.filters[j].x`` keeps working.

``n_tracks=N`` turns the object into N independent IMMs that share the models: x (N, n),
P (N, n, n), mu (N, n_models), z (N, m); ``batch_filter(zs)`` (no reference counterpart: the
reference has no batch method on IMMEstimator) runs T x {predict; update} in ONE launch and returns
the per-step estimates.

Reach (one lane owns a track's whole bank): linear ``KalmanFilter``-like filters of the same dim_x <= 16 and
dim_z <= 8, 2 to 16 filters (9 to 16: the rolled general kernel, bank in scratch memory), ``predict(u)`` with every filter's own B (dim_u <= 4), ``update(None)`` (IMM.py:171-186 as the
reference runs it on top of kalman_filter.py:511-520: the filters keep x and P, each likelihood is the density of a zero
residual under that filter's last S).  Register-resident classes: (2,1), (4,2), (6,3) x {2, 3}; (9,4) x 2..8 and (16,8) x 2
unrolled with spills; (16,8) x 3..8 rolled.
"""
import numpy as np
import torch

from .. import _engine as E

__all__ = ["IMMEstimator"]

_PHASE_STEP, _PHASE_PREDICT, _PHASE_UPDATE = 0, 1, 2


def _split_missing(zs, nt, m):
    """zs: array (T, [nt,] m) or a list whose entries may be None (a missing measurement for the whole bank, like
    the reference's update(None)) -> (array (T, nt, m) with zeros at the gaps, present (T, nt) or None)."""
    if isinstance(zs, np.ndarray) and zs.dtype != object:
        return np.asarray(zs, dtype=np.float64), None
    T = len(zs)
    out = np.zeros((T, nt, m))
    present = np.ones((T, nt), dtype=bool)
    for t, z in enumerate(zs):
        if z is None:
            present[t] = False
        else:
            out[t] = np.asarray(z, dtype=np.float64).reshape(nt, m)
    return out, (None if present.all() else present)


class IMMEstimator(object):
    def __init__(self, filters, mu, M, n_tracks=None, layout="soa"):
        if len(filters) < 2:
            raise ValueError('filters must contain at least two filters')
        x_shape = np.shape(filters[0].x)
        for f in filters:
            if x_shape != np.shape(f.x):
                raise ValueError('All filters must have the same state dimension')
        self._init_bank(filters, n_tracks, layout)
        mu = np.asarray(mu, dtype=np.float64)
        self.mu = mu / np.sum(mu, axis=-1, keepdims=True)          # IMM.py:129
        if n_tracks is not None:
            self.mu = np.broadcast_to(self.mu, (n_tracks, self.N)).copy()
        self.M = np.asarray(M, dtype=np.float64)
        self.likelihood = np.zeros(self.N if n_tracks is None else (n_tracks, self.N))
        self._compute_mixing_probabilities()
        x, P = self._estimate_host()
        self._set_estimate(x, P)
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()
        self.x_post, self.P_post = self.x.copy(), self.P.copy()

    def _init_bank(self, filters, n_tracks, layout):
        """dimensions and the host copy of the bank state (shared with MMAEFilterBank)"""
        self.filters = filters
        self.N = len(filters)
        self.M = None
        if self.N > 16:
            raise NotImplementedError("the IMM kernel is built for banks of 2 to 16 filters (filterpy_amd/csrc/fk_dims_imm.def)")
        if any(getattr(f, "inv", np.linalg.inv) is not np.linalg.inv for f in filters):
            # (KalmanFilter.update honours a non-default `inv`; the bank kernels apply S^-1 themselves -- loud, not ignored)
            raise NotImplementedError("a filter of the bank has a non-default `inv`: the IMM / MMAE kernels apply S^-1 in-lane")
        self._nt = n_tracks
        self._layout = layout
        nt = n_tracks or 1
        x0 = np.asarray(filters[0].x, dtype=np.float64)
        self._column = n_tracks is None and x0.ndim == 2          # (n, 1) states like the reference
        self._n = x0.shape[-1] if n_tracks is not None and x0.ndim == 2 else x0.reshape(-1).shape[0]
        if n_tracks is not None and x0.ndim == 2 and x0.shape[0] != n_tracks:
            raise ValueError("per-track filter states must be shaped (n_tracks, dim_x)")
        self._m = np.atleast_2d(np.asarray(filters[0].H, dtype=np.float64)).shape[0]
        n, m = self._n, self._m
        if n > 16 or m > 8:
            raise NotImplementedError("the IMM kernel is built for dim_x <= 16 and dim_z <= 8")

        # log-density of a zero residual under each filter's last real update's S (-inf before any: with S = 0 the
        # reference's density is 0, floored at float_info.min) -- what update(None) turns into the filter's likelihood;
        # kept across launches
        self._ll0 = np.full((nt, self.N), -np.inf)
        # bank state on the host, (nt, n_models, ...)
        self._xs = np.stack([np.broadcast_to(np.asarray(f.x, dtype=np.float64).reshape(-1, n), (nt, n))
                             for f in filters], axis=1).copy()
        self._Ps = np.stack([np.broadcast_to(np.asarray(f.P, dtype=np.float64), (nt, n, n))
                             for f in filters], axis=1).copy()

    # ----------------------------------------------------------------- host bookkeeping --
    def _compute_mixing_probabilities(self):
        """IMM.py:239-249 -- kept on the host only for the ``cbar`` / ``omega`` attributes; the
        kernel recomputes both from mu."""
        mu = self.mu.reshape(-1, self.N)
        cbar = mu @ self.M
        omega = (self.M[None, :, :] * mu[:, :, None]) / cbar[:, None, :]
        if self._nt is None:
            self.cbar, self.omega = cbar[0], omega[0]
        else:
            self.cbar, self.omega = cbar, omega

    def _estimate_host(self):
        """IMM.py:224-237 on the host (constructor only)."""
        mu = self.mu.reshape(-1, self.N)
        x = np.einsum("tj,tjn->tn", mu, self._xs)
        y = self._xs - x[:, None, :]
        P = np.einsum("tj,tjab->tab", mu, y[..., :, None] * y[..., None, :] + self._Ps)
        return x, P

    def _shape_x(self, x):
        if self._nt is not None:
            return x
        return x[0].reshape(-1, 1) if self._column else x[0]

    def _set_estimate(self, x, P):
        self.x = self._shape_x(x)
        self.P = P if self._nt is not None else P[0]

    def _models(self, R=None, H=None):
        """stacked device models; R / H override every filter's own for this launch
        (MMAEFilterBank.update(z, R, H), mmae.py:160-186; a scalar R means R * I, kalman_filter.py:525)"""
        n, m = self._n, self._m
        F = np.stack([np.asarray(f.F, dtype=np.float64).reshape(n, n) for f in self.filters])
        Q = np.stack([np.asarray(f.Q, dtype=np.float64).reshape(n, n) for f in self.filters])
        Hs = np.stack([np.asarray(f.H if H is None else H, dtype=np.float64).reshape(m, n) for f in self.filters])
        if R is not None and np.isscalar(R):
            R = np.eye(m) * R
        Rs = np.stack([np.asarray(f.R if R is None else R, dtype=np.float64).reshape(m, m) for f in self.filters])
        out = [E.dev(np.ascontiguousarray(a)) for a in (F, Q, Hs, Rs)]
        return out + [None if self.M is None else E.dev(np.ascontiguousarray(self.M))]

    def _pull_from_filters(self):
        """The reference reads f.x / f.P at every call, so user edits between calls must count."""
        nt, n = self._nt or 1, self._n
        for j, f in enumerate(self.filters):
            self._xs[:, j] = np.broadcast_to(np.asarray(f.x, dtype=np.float64).reshape(-1, n), (nt, n))
            self._Ps[:, j] = np.broadcast_to(np.asarray(f.P, dtype=np.float64), (nt, n, n))

    def _push_to_filters(self):
        for j, f in enumerate(self.filters):
            if self._nt is None:
                f.x = self._xs[0, j].reshape(-1, 1).copy() if self._column else self._xs[0, j].copy()
                f.P = self._Ps[0, j].copy()
            else:
                f.x, f.P = self._xs[:, j].copy(), self._Ps[:, j].copy()

    def _launch(self, phase, zs, T, want_prior, want_post, mmae=False, R=None, H=None, present=None, us=None):
        """One launch (`present`: (T, nt) booleans, False = that measurement is None; None = all there).  A long run
        without missing measurements goes through the fast instantiations except for its LAST step, which the
        general kernel runs so that the zero-residual log-densities (self._ll0) stay current for a later update(None)."""
        nt, nm = self._nt or 1, self.N
        if phase == _PHASE_STEP and present is None and not mmae and us is None and T >= 2:
            zs = np.ascontiguousarray(zs, dtype=np.float64).reshape(T, nt, self._m)
            a = self._launch1(phase, zs[:-1], T - 1, want_prior, want_post, mmae, R, H, None, track_ll0=False)
            b = self._launch1(phase, zs[-1:], 1, want_prior, want_post, mmae, R, H, None, track_ll0=True)
            return {k: np.concatenate([a[k], b[k]], axis=0) for k in a}
        return self._launch1(phase, zs, T, want_prior, want_post, mmae, R, H, present, track_ll0=phase != _PHASE_PREDICT, us=us)

    def _launch1(self, phase, zs, T, want_prior, want_post, mmae, R, H, present, track_ll0, us=None):
        nt, n, m, nm, lay = self._nt or 1, self._n, self._m, self.N, self._layout
        self._pull_from_filters()
        F, Q, H, R, M = self._models(R, H)
        xs = E.to_records(self._xs.reshape(nt, nm * n), lay, 0)
        Ps = E.to_records(self._Ps.reshape(nt, nm * n * n), lay, 0)
        mu = E.to_records(self.mu.reshape(nt, nm), lay, 0)
        z = None if zs is None else E.to_records(np.ascontiguousarray(zs, dtype=np.float64).reshape(T, nt, m), lay, 1)
        out = {}
        if want_post:
            out.update(x_out=E.alloc_records((T,), nt, n, lay), P_out=E.alloc_records((T,), nt, n * n, lay),
                       mu_out=E.alloc_records((T,), nt, nm, lay), likelihood_out=E.alloc_records((T,), nt, nm, lay))
        if want_prior:
            out.update(x_prior_out=E.alloc_records((T,), nt, n, lay), P_prior_out=E.alloc_records((T,), nt, n * n, lay))
        status = torch.zeros(nt, dtype=torch.int32, device=xs.device)
        zmask = None if present is None else torch.as_tensor(np.ascontiguousarray(present, dtype=np.uint8).reshape(T, nt),
                                                              device=xs.device)
        ll0 = E.to_records(self._ll0.reshape(nt, nm), lay, 0) if track_ll0 else None
        ctrl = {}
        if us is not None:
            # predict(u): every filter's own B (kalman_filter.py:472-475; a filter without B ignores u like the reference)
            U = np.ascontiguousarray(us, dtype=np.float64).reshape(T, nt, -1)
            nu = U.shape[2]
            if nu > 4:
                raise NotImplementedError("the IMM kernel takes dim_u <= 4")
            Bs = np.stack([np.zeros((n, nu)) if getattr(f, "B", None) is None else
                           np.asarray(f.B, dtype=np.float64).reshape(n, nu) for f in self.filters])
            ctrl = dict(nu=nu, B=E.dev(Bs), u=E.to_records(U, lay, 1))
        E.imm_batch(n, m, nm, nt, T, lay, F, Q, H, R, M, z, xs, Ps, mu, status=status, phase=phase, mmae=mmae,
                    zmask=zmask, ll0=ll0, **ctrl, **out)
        E.raise_on_status(status, "IMMEstimator")
        if ll0 is not None:
            self._ll0 = E.from_records(ll0, lay, 0, (nm,)).copy()
        self._xs = E.from_records(xs, lay, 0, (nm, n)).copy()
        self._Ps = E.from_records(Ps, lay, 0, (nm, n, n)).copy()
        mu_h = E.from_records(mu, lay, 0, (nm,)).copy()
        self.mu = mu_h if self._nt is not None else mu_h[0]
        self._push_to_filters()
        shapes = dict(x_out=(n,), P_out=(n, n), mu_out=(nm,), likelihood_out=(nm,), x_prior_out=(n,), P_prior_out=(n, n))
        return {k: E.from_records(v, lay, 1, shapes[k]) for k, v in out.items()}

    # ---------------------------------------------------------------------- reference API --
    def predict(self, u=None):
        """IMM.py:188-222: mixed initial conditions, every filter's predict, prior estimate."""
        us = None if u is None else np.asarray(u, dtype=np.float64).reshape(1, self._nt or 1, -1)
        o = self._launch(_PHASE_PREDICT, None, 1, True, False, us=us)
        self._set_estimate(o["x_prior_out"][0], o["P_prior_out"][0])
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()

    def update(self, z):
        """IMM.py:160-186: every filter's update, likelihoods, mode probabilities, posterior estimate."""
        nt = self._nt or 1
        if z is None:
            # every filter's update(None) keeps x, P; the likelihoods are those of a zero residual under each filter's
            # last S, and the mode probabilities are re-weighted with them (IMM.py:171-186 as the reference runs it)
            o = self._launch(_PHASE_UPDATE, np.zeros((1, nt, self._m)), 1, False, True, present=np.zeros((1, nt), dtype=bool))
        else:
            o = self._launch(_PHASE_UPDATE, np.asarray(z, dtype=np.float64), 1, False, True)
        L = o["likelihood_out"][0]
        self.likelihood = L if self._nt is not None else L[0]
        self._compute_mixing_probabilities()
        self._set_estimate(o["x_out"][0], o["P_out"][0])
        self.x_post, self.P_post = self.x.copy(), self.P.copy()

    def batch_filter(self, zs, return_priors=False, us=None):
        """T x { predict(); update(zs[t]) } in one launch.

        zs: (T, m) for a single IMM, (T, N, m) for a bank.  Returns (xs, Ps, mus): the estimate
        and the mode probabilities after every update, shaped (T, ...) like ``x``, ``P``, ``mu``;
        with ``return_priors`` also the estimates after every predict.  The object ends in the
        state the reference reaches after the same sequence of calls."""
        zs, present = _split_missing(zs, self._nt or 1, self._m)
        T = zs.shape[0]
        if T == 0:
            raise ValueError("zs is empty")
        o = self._launch(_PHASE_STEP, zs, T, True, True, present=present, us=us)
        L = o["likelihood_out"][-1]
        self.likelihood = L if self._nt is not None else L[0]
        self._compute_mixing_probabilities()
        self._set_estimate(o["x_prior_out"][-1], o["P_prior_out"][-1])
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()
        self._set_estimate(o["x_out"][-1], o["P_out"][-1])
        self.x_post, self.P_post = self.x.copy(), self.P.copy()

        def shp(a, is_x=False):
            if self._nt is not None:
                return a
            a = a[:, 0]
            return a.reshape(T, -1, 1) if (is_x and self._column) else a
        res = (shp(o["x_out"], True), shp(o["P_out"]), shp(o["mu_out"]))
        if return_priors:
            res += (shp(o["x_prior_out"], True), shp(o["P_prior_out"]))
        return res

    def __repr__(self):
        names = ("x", "P", "x_prior", "P_prior", "x_post", "P_post", "N", "mu", "M", "cbar", "likelihood", "omega")
        return "\n".join(["IMMEstimator object"] + [f"{k} = {getattr(self, k)!r}" for k in names])
