#!/usr/bin/env python3
"""Generate golden vectors from the LIVE reference (rlabbe/filterpy v1.4.5).

Run in the build container only (the reference is mounted read-only at
/root/reference and does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_goldens.py

Writes tests/golden/*.npz (inputs + the reference's outputs, seeds recorded).
NumPy 2.2.6 / SciPy 1.15.3 at generation time.  The goldens pin the oracle
(oracle/*.py, oracle/resample_oracle.c) and are the fixed point the HIP
kernels are compared against on the GPU box.
"""
import hashlib
import os
import sys

import numpy as np

REF = os.environ.get("FILTERPY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from filterpy.kalman import KalmanFilter, UnscentedKalmanFilter, MerweScaledSigmaPoints  # noqa: E402
from filterpy.kalman import JulierSigmaPoints, unscented_transform  # noqa: E402
import filterpy.kalman.kalman_filter as kfmod  # noqa: E402
from filterpy.monte_carlo import (systematic_resample, stratified_resample,  # noqa: E402
                                  multinomial_resample, residual_resample)

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)


def spd(rs, n, scale=1.0):
    A = rs.randn(n, n)
    return scale * (A @ A.T / n + 0.5 * np.eye(n))


def stable_F(rs, n):
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    return F / max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))


def make_kf(n, m, x0, P0, F, Q, H, R, dim_u=0, B=None, alpha=1.0):
    kf = KalmanFilter(dim_x=n, dim_z=m, dim_u=dim_u)
    kf.x = x0.copy()
    kf.P = P0.copy()
    kf.F, kf.Q, kf.H, kf.R = F.copy(), Q.copy(), H.copy(), R.copy()
    if B is not None:
        kf.B = B.copy()
    if alpha != 1.0:
        kf.alpha = alpha
    return kf


# ---------------------------------------------------------------- C1 ------
def gen_c1():
    """BASELINE config 1: KalmanFilter(2,1) constant velocity, 1000 z."""
    d = {}
    F = np.array([[1., 1.], [0., 1.]])
    H = np.array([[1., 0.]])
    P0 = np.eye(2) * 100.
    R = np.array([[4.]])
    Q = np.array([[.25, .5], [.5, 1.]]) * 0.01   # Q_discrete_white_noise(2, 1., 0.01)
    zs = np.arange(1000.) + np.random.RandomState(0).randn(1000) * 2
    d.update(F=F, H=H, P0=P0, R=R, Q=Q, zs=zs)
    for tag, x0, zlist in (("1d", np.zeros(2), list(zs)),
                           ("col", np.zeros((2, 1)), list(zs))):
        kf = make_kf(2, 1, x0, P0, F, Q, H, R)
        mu, cov, mup, covp = kf.batch_filter(zlist)
        xs, Ps, Ks, Pps = kf.rts_smoother(mu, cov)
        for k, v in dict(mu=mu, cov=cov, mup=mup, covp=covp, xs=xs, Ps=Ps, Ks=Ks, Pps=Pps).items():
            d[f"{tag}_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "kf_c1.npz"), **d)


# ---------------------------------------------------------------- dims ----
DIMS = [(1, 1), (2, 1), (2, 2), (3, 1), (4, 2), (4, 4), (5, 2), (6, 3), (8, 4), (9, 3), (11, 1), (16, 5)]


def gen_dims():
    """Random models for many (dim_x, dim_z): plain / update_first / alpha / control /
    missing measurements, class batch_filter + class rts_smoother (k+1 convention)."""
    d = {"dims": np.array(DIMS)}
    T = 40
    for (n, m) in DIMS:
        rs = np.random.RandomState(1000 + 17 * n + m)
        F, Q, H, R = stable_F(rs, n), spd(rs, n, 0.1), rs.randn(m, n), spd(rs, m, 0.5)
        P0, x0 = spd(rs, n, 5.0), rs.randn(n)
        zs = rs.randn(T, m) * 3
        B, us = rs.randn(n, 2), rs.randn(T, 2)
        p = f"n{n}m{m}_"
        d.update({p + "F": F, p + "Q": Q, p + "H": H, p + "R": R, p + "P0": P0, p + "x0": x0,
                  p + "zs": zs, p + "B": B, p + "us": us})
        variants = {
            "plain": dict(),
            "uf": dict(update_first=True),
            "alpha": dict(alpha=1.02),
            "ctrl": dict(control=True),
        }
        for vname, opt in variants.items():
            kf = make_kf(n, m, x0, P0, F, Q, H, R, dim_u=2,
                         B=B if opt.get("control") else None, alpha=opt.get("alpha", 1.0))
            kw = {}
            if opt.get("control"):
                kw["us"] = list(us)
            mu, cov, mup, covp = kf.batch_filter(list(zs), update_first=opt.get("update_first", False), **kw)
            d.update({p + vname + "_mu": mu, p + vname + "_cov": cov,
                      p + vname + "_mup": mup, p + vname + "_covp": covp})
            d[p + vname + "_xfinal"] = kf.x.copy()
            d[p + vname + "_Pfinal"] = kf.P.copy()
            if vname == "plain":
                xs, Ps, Ks, Pps = kf.rts_smoother(mu, cov)
                d.update({p + "rts_x": xs, p + "rts_P": Ps, p + "rts_K": Ks, p + "rts_Pp": Pps})
                # module-level smoother uses Fs[k], Qs[k]  (kalman_filter.py:1851-1856)
                xs2, Ps2, Ks2, Pps2 = kfmod.rts_smoother(mu, cov, [F] * T, [Q] * T)
                d.update({p + "rtsm_x": xs2, p + "rtsm_P": Ps2, p + "rtsm_K": Ks2, p + "rtsm_Pp": Pps2})
        # missing measurements: column-vector state so the zs list may hold None for any m
        mask = rs.rand(T) > 0.3
        kf = make_kf(n, m, x0.reshape(n, 1), P0, F, Q, H, R)
        zl = [zs[t].reshape(m, 1) if mask[t] else None for t in range(T)]
        # np.size(zs,0) on a ragged list fails under NumPy>=1.24;
        # feed an object array so the reference's own loop runs.
        arr = np.empty(T, dtype=object)
        for t in range(T):
            arr[t] = zl[t]
        zl = arr
        mu, cov, mup, covp = kf.batch_filter(zl)
        d.update({p + "mask": mask, p + "miss_mu": mu[..., 0], p + "miss_cov": cov,
                  p + "miss_mup": mup[..., 0], p + "miss_covp": covp})
    np.savez_compressed(os.path.join(OUT, "kf_dims.npz"), **d)


# ------------------------------------------------------ per-step models ---
def gen_models():
    """Per-epoch Fs/Qs/Hs/Rs lists (model_mode 2) through class and module batch_filter,
    and class (k+1) / module (k) rts_smoother with per-epoch Fs,Qs."""
    d = {}
    T = 30
    for (n, m) in [(2, 1), (4, 2), (6, 3), (9, 3)]:
        rs = np.random.RandomState(2000 + 13 * n + m)
        Fs = np.array([stable_F(rs, n) for _ in range(T)])
        Qs = np.array([spd(rs, n, 0.1) for _ in range(T)])
        Hs = rs.randn(T, m, n)
        Rs = np.array([spd(rs, m, 0.5) for _ in range(T)])
        P0, x0 = spd(rs, n, 5.0), rs.randn(n)
        zs = rs.randn(T, m) * 3
        p = f"n{n}m{m}_"
        kf = make_kf(n, m, x0, P0, Fs[0], Qs[0], Hs[0], Rs[0])
        mu, cov, mup, covp = kf.batch_filter(list(zs), Fs=list(Fs), Qs=list(Qs), Hs=list(Hs), Rs=list(Rs))
        xs, Ps, Ks, Pps = kf.rts_smoother(mu, cov, Fs=list(Fs), Qs=list(Qs))
        mu2, cov2, mup2, covp2 = kfmod.batch_filter(x0, P0, list(zs), list(Fs), list(Qs), list(Hs), list(Rs))
        xm, Pm, Km, Ppm = kfmod.rts_smoother(mu2, cov2, list(Fs), list(Qs))
        d.update({p + "Fs": Fs, p + "Qs": Qs, p + "Hs": Hs, p + "Rs": Rs, p + "P0": P0, p + "x0": x0,
                  p + "zs": zs, p + "mu": mu, p + "cov": cov, p + "mup": mup, p + "covp": covp,
                  p + "rts_x": xs, p + "rts_P": Ps, p + "rts_K": Ks, p + "rts_Pp": Pps,
                  p + "mod_mu": mu2, p + "mod_cov": cov2, p + "mod_mup": mup2, p + "mod_covp": covp2,
                  p + "rtsm_x": xm, p + "rtsm_P": Pm, p + "rtsm_K": Km, p + "rtsm_Pp": Ppm})
    np.savez_compressed(os.path.join(OUT, "kf_models.npz"), **d)


# ------------------------------------------------ single-step + module fns
def gen_steps():
    """Single predict()/update() calls (class and module), incl. kwarg overrides,
    scalar-attribute Q/R quirk, y/K/S/SI and the lazy
    log_likelihood / likelihood / mahalanobis properties."""
    d = {}
    for (n, m) in [(1, 1), (2, 1), (4, 2), (6, 3), (9, 3)]:
        rs = np.random.RandomState(3000 + 11 * n + m)
        F, Q, H, R = stable_F(rs, n), spd(rs, n, 0.1), rs.randn(m, n), spd(rs, m, 0.5)
        P0, x0, z = spd(rs, n, 5.0), rs.randn(n), rs.randn(m)
        p = f"n{n}m{m}_"
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        kf.predict()
        d.update({p + "F": F, p + "Q": Q, p + "H": H, p + "R": R, p + "P0": P0, p + "x0": x0, p + "z": z,
                  p + "xp": kf.x.copy(), p + "Pp": kf.P.copy()})
        kf.update(z)
        d.update({p + "x": kf.x.copy(), p + "P": kf.P.copy(), p + "y": kf.y.copy(), p + "K": kf.K.copy(),
                  p + "S": kf.S.copy(), p + "SI": kf.SI.copy(),
                  p + "loglik": np.float64(kf.log_likelihood), p + "lik": np.float64(kf.likelihood),
                  p + "maha": np.float64(kf.mahalanobis)})
        # module-level twins
        xp, Pp = kfmod.predict(x0, P0, F, Q)
        x2, P2, y2, K2, S2, ll2 = kfmod.update(xp, Pp, z, R, H, return_all=True)
        d.update({p + "m_xp": xp, p + "m_Pp": Pp, p + "m_x": x2, p + "m_P": P2, p + "m_y": y2,
                  p + "m_K": K2, p + "m_S": S2, p + "m_ll": np.float64(ll2)})
        # scalar attribute quirk: Q scalar attribute is added to every element
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        kf.Q = 0.37
        kf.predict()
        d[p + "qscal_attr_P"] = kf.P.copy()
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        kf.predict(Q=0.37)           # kwarg scalar -> eye*q
        d[p + "qscal_kw_P"] = kf.P.copy()
        kf.update(z, R=0.81)          # kwarg scalar -> eye*r
        d[p + "rscal_kw_x"] = kf.x.copy()
        d[p + "rscal_kw_P"] = kf.P.copy()
    np.savez_compressed(os.path.join(OUT, "kf_steps.npz"), **d)



# ------------------------------------------------------- Saver histories ---
def gen_saver():
    """per-epoch attribute histories as filterpy.common.Saver records them during
    batch_filter(saver=...) -- K, y, S, SI, priors/posts and the lazy log_likelihood / likelihood /
    mahalanobis properties -- including epochs with a missing measurement."""
    from filterpy.common import Saver
    d = {}
    T = 25
    for (n, m) in [(2, 1), (4, 2), (6, 3), (9, 3)]:
        rs = np.random.RandomState(6000 + 7 * n + m)
        F, Q, H, R = stable_F(rs, n), spd(rs, n, 0.1), rs.randn(m, n), spd(rs, m, 0.5)
        P0, x0 = spd(rs, n, 5.0), rs.randn(n, 1)
        zs = rs.randn(T, m, 1) * 3
        mask = rs.rand(T) > 0.25
        zl = np.empty(T, dtype=object)
        for t in range(T):
            zl[t] = zs[t] if mask[t] else None
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        s = Saver(kf)
        mu, cov, mup, covp = kf.batch_filter(zl, saver=s)
        p = f"n{n}m{m}_"
        d.update({p + "F": F, p + "Q": Q, p + "H": H, p + "R": R, p + "P0": P0, p + "x0": x0, p + "zs": zs,
                  p + "mask": mask, p + "mu": mu, p + "cov": cov})
        for key in ("x", "P", "x_prior", "P_prior", "x_post", "P_post", "K", "y", "S", "SI"):
            d[p + key] = np.array(s[key], dtype=float)
        for key in ("log_likelihood", "likelihood", "mahalanobis"):
            d[p + key] = np.array(s[key], dtype=float)
    np.savez_compressed(os.path.join(OUT, "kf_saver.npz"), **d)



# ------------------------------------------------------- KF API variants ----
def gen_variants():
    """predict_steadystate/update_steadystate, update_correlated, update_sequential
    of the live filterpy.kalman.KalmanFilter."""
    d = {}
    T = 25
    cases = [(2, 1), (4, 2), (6, 3), (9, 3), (3, 2)]
    for (n, m) in cases:
        rs = np.random.RandomState(9500 + 7 * n + m)
        F, H = stable_F(rs, n), rs.randn(m, n)
        Q, R, P0, x0 = spd(rs, n, 0.05), spd(rs, m, 0.5), spd(rs, n, 3.0), rs.randn(n)
        zs = rs.randn(T, m) * 2
        q = f"n{n}m{m}_"
        d.update({q + "F": F, q + "H": H, q + "Q": Q, q + "R": R, q + "P0": P0, q + "x0": x0, q + "zs": zs})
        # --- steady state: converge a gain with the ordinary filter, then run with it fixed
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        for t in range(60):
            kf.predict()
            kf.update(zs[t % T])
        K, Pss = kf.K.copy(), kf.P.copy()
        B = rs.randn(n, 2)
        us = rs.randn(T, 2)
        for tag, ctrl in (("ss_", False), ("ssu_", True)):
            kf = make_kf(n, m, x0, P0, F, Q, H, R)
            kf.K, kf.P = K.copy(), Pss.copy()
            if ctrl:
                kf.B = B.copy()
            X, XP, Y = [], [], []
            for t in range(T):
                if ctrl:
                    kf.predict_steadystate(u=us[t])
                else:
                    kf.predict_steadystate()
                XP.append(kf.x.copy())
                if t == 7:
                    kf.update_steadystate(None)
                else:
                    kf.update_steadystate(zs[t])
                X.append(kf.x.copy()); Y.append(np.ravel(kf.y).copy())
            d.update({q + tag + "x": np.array(X), q + tag + "xp": np.array(XP), q + tag + "y": np.array(Y)})
            assert np.array_equal(kf.P, Pss)
        d.update({q + "K": K, q + "Pss": Pss, q + "B": B, q + "us": us})
        # --- correlated noise: three consecutive updates with a non-zero M
        M = 0.1 * rs.randn(n, m)
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        kf.M = M.copy()
        cx, cP, cK, cS, cy = [], [], [], [], []
        for t in range(3):
            kf.predict()
            if t == 2:
                kf.update_correlated(zs[t], R=2.0 * R, H=0.5 * H)
            else:
                kf.update_correlated(zs[t])
            cx.append(kf.x.copy()); cP.append(kf.P.copy()); cK.append(kf.K.copy()); cS.append(kf.S.copy())
            cy.append(np.ravel(kf.y).copy())
        d.update({q + "M": M, q + "corr_x": np.array(cx), q + "corr_P": np.array(cP), q + "corr_K": np.array(cK),
                  q + "corr_S": np.array(cS), q + "corr_y": np.array(cy)})
        # --- sequential: the components of one z one at a time (column-vector state, as the reference needs)
        kf = make_kf(n, m, x0.reshape(-1, 1), P0, F, Q, H, R)
        kf.predict()
        sx, sP = [], []
        for i in range(m):
            kf.update_sequential(i, zs[0][i])
            sx.append(kf.x.copy().ravel()); sP.append(kf.P.copy())
        d.update({q + "seq_x": np.array(sx), q + "seq_P": np.array(sP), q + "seq_K": kf.K.copy(),
                  q + "seq_y": np.ravel(kf.y).copy()})
        if m >= 2:    # a 2-component block
            kf = make_kf(n, m, x0.reshape(-1, 1), P0, F, Q, H, R)
            kf.predict()
            kf.update_sequential(m - 2, zs[0][m - 2:])
            d.update({q + "seqb_x": kf.x.copy().ravel(), q + "seqb_P": kf.P.copy()})
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "kf_variants.npz"), **d)

# ------------------------------------------------------------------ IMM ----
IMM_CASES = [(2, 1, 2), (4, 2, 2), (4, 2, 3), (6, 3, 2), (3, 2, 2), (5, 2, 3)]


def gen_imm():
    """filterpy.kalman.IMMEstimator (IMM.py) over T steps of predict(); update(z)."""
    from filterpy.kalman import IMMEstimator
    d = {}
    T = 30
    for (n, m, nm) in IMM_CASES:
        rs = np.random.RandomState(8000 + 11 * n + 3 * m + nm)
        Fs = [stable_F(rs, n) for _ in range(nm)]
        Qs = [spd(rs, n, 0.05 * (j + 1)) for j in range(nm)]
        H = rs.randn(m, n)
        Hs = [H.copy() for _ in range(nm)]
        Rs = [spd(rs, m, 0.5) for _ in range(nm)]
        xs0 = [rs.randn(n) for _ in range(nm)]
        Ps0 = [spd(rs, n, 3.0) for _ in range(nm)]
        mu0 = rs.rand(nm) + 0.2
        Mt = rs.rand(nm, nm) + np.eye(nm) * 3
        Mt /= Mt.sum(axis=1, keepdims=True)
        zs = rs.randn(T, m) * 2
        filters = []
        for j in range(nm):
            filters.append(make_kf(n, m, xs0[j], Ps0[j], Fs[j], Qs[j], Hs[j], Rs[j]))
        imm = IMMEstimator(filters, mu0, Mt)
        X, P, MU, XP, PP, L = [], [], [], [], [], []
        for t in range(T):
            imm.predict()
            XP.append(imm.x.copy()); PP.append(imm.P.copy())
            imm.update(zs[t])
            X.append(imm.x.copy()); P.append(imm.P.copy()); MU.append(imm.mu.copy()); L.append(imm.likelihood.copy())
        p = f"n{n}m{m}k{nm}_"
        d.update({p + "Fs": np.array(Fs), p + "Qs": np.array(Qs), p + "Hs": np.array(Hs), p + "Rs": np.array(Rs),
                  p + "xs0": np.array(xs0), p + "Ps0": np.array(Ps0), p + "mu0": mu0, p + "M": Mt, p + "zs": zs,
                  p + "x": np.array(X), p + "P": np.array(P), p + "mu": np.array(MU), p + "xp": np.array(XP),
                  p + "Pp": np.array(PP), p + "L": np.array(L),
                  p + "xs_final": np.array([f.x for f in filters]), p + "Ps_final": np.array([f.P for f in filters])})
    d["cases"] = np.array(IMM_CASES)
    np.savez_compressed(os.path.join(OUT, "imm.npz"), **d)


def gen_mmae():
    """filterpy.kalman.MMAEFilterBank (mmae.py) over T steps of predict(); update(z)."""
    from filterpy.kalman import MMAEFilterBank
    d = {}
    T = 30
    cases = [(2, 1, 2), (4, 2, 2), (4, 2, 3), (6, 3, 3), (3, 2, 2), (2, 1, 3)]
    for (n, m, nm) in cases:
        rs = np.random.RandomState(9000 + 11 * n + 3 * m + nm)
        Fs = [stable_F(rs, n) for _ in range(nm)]
        Qs = [spd(rs, n, 0.05 * (j + 1)) for j in range(nm)]
        H = rs.randn(m, n)
        Hs = [H.copy() for _ in range(nm)]
        Rs = [spd(rs, m, 0.5) for _ in range(nm)]
        xs0 = [rs.randn(n) for _ in range(nm)]
        Ps0 = [spd(rs, n, 3.0) for _ in range(nm)]
        p0 = rs.rand(nm) + 0.2
        p0 /= p0.sum()
        zs = rs.randn(T, m) * 2
        filters = [make_kf(n, m, xs0[j], Ps0[j], Fs[j], Qs[j], Hs[j], Rs[j]) for j in range(nm)]
        bank = MMAEFilterBank(filters, p0.copy(), dim_x=n)
        X, P, PR, L = [], [], [], []
        for t in range(T):
            bank.predict()
            bank.update(zs[t])
            X.append(bank.x.copy()); P.append(bank.P.copy()); PR.append(bank.p.copy())
            L.append(np.array([f.likelihood for f in filters]))
        q = f"n{n}m{m}k{nm}_"
        d.update({q + "Fs": np.array(Fs), q + "Qs": np.array(Qs), q + "Hs": np.array(Hs), q + "Rs": np.array(Rs),
                  q + "xs0": np.array(xs0), q + "Ps0": np.array(Ps0), q + "p0": p0, q + "zs": zs,
                  q + "x": np.array(X), q + "P": np.array(P), q + "p": np.array(PR), q + "L": np.array(L),
                  q + "xs_final": np.array([f.x for f in filters]), q + "Ps_final": np.array([f.P for f in filters])})
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "mmae.npz"), **d)

# ---------------------------------------------------------------- UKF -----
UKF_CASES = [(1, 1, .5, 2., 1.), (2, 1, .1, 2., -1.), (4, 2, 1e-3, 2., 0.), (6, 3, .1, 2., -3.), (4, 2, 1., 2., .1)]


def gen_ukf():
    d = {"cases": np.array(UKF_CASES)}
    T = 30
    for ci, (n, m, alpha, beta, kappa) in enumerate(UKF_CASES):
        n, m = int(n), int(m)
        rs = np.random.RandomState(4000 + ci)
        pts = MerweScaledSigmaPoints(n, alpha, beta, kappa)
        F, H = stable_F(rs, n), rs.randn(m, n)
        Q, R, P0, x0 = spd(rs, n, 0.01), spd(rs, m, 0.5), spd(rs, n, 5.0), rs.randn(n)
        zs = rs.randn(T, m) * 3
        p = f"c{ci}_"
        sig = pts.sigma_points(x0, P0)
        ux, uP = unscented_transform(sig, pts.Wm, pts.Wc, Q)
        d.update({p + "F": F, p + "H": H, p + "Q": Q, p + "R": R, p + "P0": P0, p + "x0": x0, p + "zs": zs,
                  p + "Wm": pts.Wm, p + "Wc": pts.Wc, p + "sigmas": sig, p + "ut_x": ux, p + "ut_P": uP})
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=lambda x: H @ x, fx=lambda x, dt: F @ x, points=pts)
        ukf.x, ukf.P, ukf.Q, ukf.R = x0.copy(), P0.copy(), Q.copy(), R.copy()
        # one explicit step with the intermediates
        ukf.predict()
        d.update({p + "s1_xp": ukf.x.copy(), p + "s1_Pp": ukf.P.copy(), p + "s1_sigmas_f": ukf.sigmas_f.copy()})
        ukf.update(zs[0])
        d.update({p + "s1_x": ukf.x.copy(), p + "s1_P": ukf.P.copy(), p + "s1_K": ukf.K.copy(),
                  p + "s1_S": ukf.S.copy(), p + "s1_y": ukf.y.copy()})
        ukf.x, ukf.P = x0.copy(), P0.copy()
        mu, cov = ukf.batch_filter(list(zs) if m > 1 else [np.array([z[0]]) for z in zs])
        xs, Ps, Ks = ukf.rts_smoother(mu, cov)
        d.update({p + "mu": mu, p + "cov": cov, p + "rts_x": xs, p + "rts_P": Ps, p + "rts_K": Ks})
        # linear KF on the same model (the reference's own relational pin, test_ukf.py:948-978)
        kf = make_kf(n, m, x0, P0, F, Q, H, R)
        kmu, kcov, _, _ = kf.batch_filter(list(zs))
        d.update({p + "kf_mu": kmu, p + "kf_cov": kcov})
    # Julier
    rs = np.random.RandomState(4100)
    jp = JulierSigmaPoints(4, kappa=0.5)
    P0, x0 = spd(rs, 4, 2.0), rs.randn(4)
    d.update(jul_P0=P0, jul_x0=x0, jul_Wm=jp.Wm, jul_Wc=jp.Wc, jul_sigmas=jp.sigma_points(x0, P0))
    np.savez_compressed(os.path.join(OUT, "ukf_merwe.npz"), **d)


# ------------------------------------------------------------ resample ----
def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


from weights import weights_for  # noqa: E402  (tests/golden/weights.py)


RS_SMALL = [1, 2, 3, 10, 64, 1000, 8000]
RS_BIG = [100003, 1 << 20]
RS_HUGE = [8000000]


def gen_resample():
    d = {}
    fns = dict(sys=systematic_resample, strat=stratified_resample,
               multi=multinomial_resample, resid=residual_resample)
    cases = []
    for N in RS_SMALL + RS_BIG:
        for kind in (("rand", "onehot", "sparse", "exp") if N >= 10 else ("rand",)):
            cases.append((N, kind))
    for ci, (N, kind) in enumerate(cases):
        wseed = 5000 + ci
        w = weights_for(N, wseed, kind)
        for name, fn in fns.items():
            if name in ("multi", "resid") and N > 100003:
                continue
            if name == "resid" and N > 8000:
                continue
            useed = 7000 + ci
            np.random.seed(useed)
            try:
                idx = fn(w)
            except IndexError:
                idx = None
            key = f"{name}_N{N}_{kind}"
            d[key + "_wseed"] = np.int64(wseed)
            d[key + "_useed"] = np.int64(useed)
            if idx is None:
                d[key + "_indexerror"] = np.int64(1)
                continue
            d[key + "_dtype"] = np.array(str(idx.dtype))
            if N <= 8000:
                d[key + "_idx"] = idx
                if N <= 1000:
                    d[key + "_w"] = w
            else:
                d[key + "_sha"] = sha(idx)
                d[key + "_head"] = idx[:1024].copy()
                d[key + "_tail"] = idx[-1024:].copy()
    # the N=8e6 vectors: flips of a blocked scan only show up at this size (SURVEY §7 hard part 1)
    for ci, N in enumerate(RS_HUGE):
        for si in range(2):
            wseed, useed = 5900 + si, 7900 + si
            w = weights_for(N, wseed, "rand")
            for name in ("sys", "strat"):
                np.random.seed(useed)
                u = np.random.random() if name == "sys" else np.random.random(N)
                pos = (u + np.arange(N)) / N
                # searchsorted(side='right') == the reference's two-pointer loop (pinned at smaller N
                # against the live loop in tests/test_oracle_resample.py; the loop itself needs ~5 s here)
                np.random.seed(useed)
                idx = fns[name](w)
                assert np.array_equal(idx, np.searchsorted(np.cumsum(w), pos, side='right'))
                key = f"{name}_N{N}_rand{si}"
                # where a pairwise (blocked) summation would flip the index
                cs_blk = np.add.accumulate(w.reshape(-1, 1000), axis=1)
                off = np.concatenate([[0.0], np.cumsum(cs_blk[:, -1])[:-1]])
                idx_blk = np.searchsorted((cs_blk + off[:, None]).ravel(), pos, side='right')
                flips = np.nonzero(idx_blk != idx)[0]
                d[key + "_wseed"], d[key + "_useed"] = np.int64(wseed), np.int64(useed)
                d[key + "_sha"] = sha(idx.astype(np.int32))
                d[key + "_head"], d[key + "_tail"] = idx[:1024].copy(), idx[-1024:].copy()
                d[key + "_flip_pos"] = flips[:4096]
                d[key + "_flip_idx"] = idx[flips[:4096]]
                d[key + "_dtype"] = np.array(str(idx.dtype))
                print(key, "blocked-scan flips:", len(flips))
    # cs[-1] < 1 edge: position >= cumsum[-1] -> the reference raises IndexError
    N = 100003
    w = weights_for(N, 5000 + cases.index((100003, "rand")), "rand")
    d["edge_cs_last"] = np.float64(np.cumsum(w)[-1])
    np.savez_compressed(os.path.join(OUT, "resample.npz"), **d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "dims", "models", "steps", "saver", "variants", "imm", "mmae", "ukf", "resample"]
    for w in which:
        print("generating", w)
        globals()["gen_" + w]()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
