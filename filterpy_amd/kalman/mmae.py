"""Multiple-model adaptive estimator bank with filterpy's call surface (filterpy/kalman/mmae.py:
__init__ :104-137, predict :140-158, update :160-212) on the batched IMM kernel
(fk_imm_batch_f64 with FK_IMM_FLAG_MMAE): every filter predicts and updates from its own state,
p_i *= likelihood_i (normalised), x = sum p_i x_i, and P exactly as the reference's loop computes it
(mmae.py:205-207 zips the components of x with the filters; reproduced, not corrected).

``n_tracks=N`` runs N independent banks that share the filters' models; ``batch_filter(zs)`` runs
T x {predict; update} in one launch.  Restrictions as for ``IMMEstimator``: 2 or 3 linear filters,
dim_x <= 6, dim_z <= 3, no control input, every measurement present.
"""
from copy import deepcopy

import numpy as np

from .IMM import IMMEstimator

__all__ = ["MMAEFilterBank"]


class MMAEFilterBank(object):
    def __init__(self, filters, p, dim_x, H=None, n_tracks=None, layout="soa"):
        if len(filters) != len(np.atleast_1d(p)) and n_tracks is None:
            raise ValueError('length of filters and p must be the same')
        if dim_x < 1:
            raise ValueError('dim_x must be >= 1')
        self.filters = filters
        self.dim_x = dim_x
        self.H = None if H is None else np.copy(H)
        # the launch plumbing (record packing, models, write-back into the filters) is IMMEstimator's
        self._eng = IMMEstimator.__new__(IMMEstimator)
        self._eng._init_bank(filters, n_tracks, layout)
        nt = n_tracks or 1
        self.p = np.asarray(p, dtype=np.float64)                   # NOT normalised here (mmae.py:111)
        if n_tracks is not None:
            self.p = np.broadcast_to(self.p, (nt, len(filters))).copy()
        self.z = np.copy(getattr(filters[0], "z", 0))
        self.x = np.copy(filters[0].x)
        self.P = np.copy(filters[0].P)
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()
        self.x_post, self.P_post = self.x.copy(), self.P.copy()

    def _run(self, phase, zs, T, want_post, R=None, H=None, present=None, us=None):
        e = self._eng
        e.mu = self.p
        o = e._launch(phase, zs, T, False, want_post, mmae=True, R=R, H=H, present=present, us=us)
        self.p = e.mu
        return o

    def predict(self, u=0):
        """mmae.py:140-158: every filter predicts; the prior is a copy of the last estimate."""
        us = None
        if np.any(np.asarray(u) != 0):
            us = np.asarray(u, dtype=np.float64).reshape(1, self._eng._nt or 1, -1)
        self._run(1, None, 1, False, us=us)
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()

    def update(self, z, R=None, H=None):
        """mmae.py:160-212."""
        if H is None:
            H = self.H
        if z is None:
            # mmae.py:184-187 with z = None: the filters keep x, P, p_i *= the density of a zero residual under
            # filter i's last S (kalman_filter.py:511-520, :1203-1226)
            nt = self._eng._nt or 1
            o = self._run(2, np.zeros((1, nt, self._eng._m)), 1, True, R=R, H=H, present=np.zeros((1, nt), dtype=bool))
        else:
            o = self._run(2, np.asarray(z, dtype=np.float64), 1, True, R=R, H=H)
        e = self._eng
        e._set_estimate(o["x_out"][0], o["P_out"][0])
        self.x, self.P = e.x, e.P
        self.z = deepcopy(z)
        self.x_post, self.P_post = self.x.copy(), self.P.copy()

    def batch_filter(self, zs):
        """T x { predict(); update(zs[t]) } in one launch (no reference counterpart).  Returns
        (xs, Ps, ps): estimate and model probabilities after every update."""
        from .IMM import _split_missing
        last_z = zs[-1] if len(zs) else None
        zs, present = _split_missing(zs, self._eng._nt or 1, self._eng._m)
        T = zs.shape[0]
        if T == 0:
            raise ValueError("zs is empty")
        o = self._run(0, zs, T, True, H=self.H, present=present)
        e = self._eng
        if T > 1:
            e._set_estimate(o["x_out"][-2], o["P_out"][-2])
            self.x_prior, self.P_prior = e.x.copy(), e.P.copy()
        else:
            self.x_prior, self.P_prior = self.x.copy(), self.P.copy()
        e._set_estimate(o["x_out"][-1], o["P_out"][-1])
        self.x, self.P = e.x, e.P
        self.z = deepcopy(last_z)
        self.x_post, self.P_post = self.x.copy(), self.P.copy()
        if e._nt is not None:
            return o["x_out"], o["P_out"], o["mu_out"]
        xs = o["x_out"][:, 0]
        return (xs.reshape(T, -1, 1) if e._column else xs), o["P_out"][:, 0], o["mu_out"][:, 0]

    def __repr__(self):
        return "\n".join(["MMAEFilterBank object", f"dim_x = {self.dim_x!r}", f"x = {self.x!r}",
                          f"P = {self.P!r}", f"log-p = {self.p!r}"])
