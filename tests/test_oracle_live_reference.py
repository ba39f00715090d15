"""The oracle against the LIVE reference on seeded random cases -- beyond the frozen goldens.

The GPU parity tests on random banks compare the kernels with the ORACLE (the reference cannot travel to the GPU box), so the
oracle has to be the reference everywhere in the space those tests draw from, not only at the golden cases.  Where the reference
checkout is present (the build container; `FILTERPY_REFERENCE` or /root/reference) this module imports filterpy itself and runs
both on the same inputs: random (dim_x, dim_z) up to (16, 8), shared and per-epoch models, control input, fading memory,
update_first, missing measurements, both smoother index conventions, Merwe / Julier unscented filters with their smoother,
IMM / MMAE banks with missing measurements, the four resamplers.  Bar: 1e-13 where the oracle calls the same NumPy routines in
the same order (bit-equal indices for the resamplers), 1e-11 behind a matrix inverse chain.  Elsewhere: skipped (the goldens pin).
Nothing here touches the GPU or the product package."""
import os
import sys

import numpy as np
import pytest

from conftest import rel_err_rows
from oracle import imm_oracle, kf_oracle, resample_oracle, ukf_oracle

REF = os.environ.get("FILTERPY_REFERENCE", "/root/reference")
TIGHT = 1e-13
# (the reference imports deprecated scipy namespaces; N = 1, 2 make residual_resample divide 0 by 0 in both implementations)
pytestmark = [pytest.mark.filterwarnings("ignore::DeprecationWarning"), pytest.mark.filterwarnings("ignore::RuntimeWarning")]


@pytest.fixture(scope="module")
def ref():
    """the reference package, imported for this module only (path and modules removed again afterwards)"""
    if not os.path.isdir(os.path.join(REF, "filterpy")):
        pytest.skip("no reference checkout here: the oracle's pin on this box is tests/golden/")
    os.environ.setdefault("MPLBACKEND", "Agg")
    before = set(sys.modules)
    sys.path.insert(0, REF)
    old_flag, sys.dont_write_bytecode = sys.dont_write_bytecode, True          # /root/reference is read-only by contract
    try:
        import filterpy.kalman as K
        import filterpy.monte_carlo as M
        import filterpy.kalman.kalman_filter as KM
        yield type("Ref", (), dict(K=K, M=M, KM=KM))
    finally:
        sys.dont_write_bytecode = old_flag
        sys.path.remove(REF)
        for name in set(sys.modules) - before:
            if name == "filterpy" or name.startswith("filterpy."):
                del sys.modules[name]


def spd(rs, k, scale=1.0):
    a = rs.randn(k, k)
    return scale * (a @ a.T / k + 0.5 * np.eye(k))


def stable_F(rs, n):
    a = rs.randn(n, n)
    return 0.95 * a / max(1.0, np.max(np.abs(np.linalg.eigvals(a))))


CASES = [(1, 1), (2, 1), (3, 2), (4, 2), (5, 4), (6, 3), (7, 2), (9, 3), (9, 4), (10, 2), (12, 5), (14, 6), (16, 8)]


@pytest.mark.parametrize("n,m", CASES)
@pytest.mark.parametrize("variant", ["plain", "per_step", "ctrl_alpha", "update_first_missing"])
def test_kalman_batch_filter_and_smoothers(ref, n, m, variant):
    """KalmanFilter.batch_filter (kalman_filter.py:826-993) and both rts_smoother conventions (:995-1074, :1792-1858)"""
    rs = np.random.RandomState(1000 * n + 10 * m + len(variant))
    T, nu = 12, 2
    x0, P0 = rs.randn(n), spd(rs, n, 4.0)
    F, Q, H, R = stable_F(rs, n), spd(rs, n, 0.05), rs.randn(m, n), spd(rs, m, 0.5)
    zs = [rs.randn(m) for _ in range(T)]
    kw_ref, kw_or = {}, {}
    Fa, Qa, Ha, Ra = F, Q, H, R
    if variant == "per_step":
        Fa, Qa = [stable_F(rs, n) for _ in range(T)], [spd(rs, n, 0.05) for _ in range(T)]
        Ha, Ra = [rs.randn(m, n) for _ in range(T)], [spd(rs, m, 0.5) for _ in range(T)]
        kw_ref = dict(Fs=Fa, Qs=Qa, Hs=Ha, Rs=Ra)
    kf = ref.K.KalmanFilter(dim_x=n, dim_z=m, dim_u=nu if variant == "ctrl_alpha" else 0)
    kf.x, kf.P, kf.F, kf.Q, kf.H, kf.R = x0.copy(), P0.copy(), F, Q, H, R
    if variant == "ctrl_alpha":
        B, us = rs.randn(n, nu), [rs.randn(nu) for _ in range(T)]
        kf.B, kf.alpha = B, 1.03
        kw_ref = dict(Bs=[B] * T, us=us)
        kw_or = dict(B=B, us=us, alpha_sq=1.03 ** 2)
    zs_ref = list(zs)
    if variant == "update_first_missing":
        # column-vector state so that the list may hold None at any dim_z, handed over as an object array: np.size(zs, 0) on a
        # ragged LIST fails under NumPy >= 1.24 (tests/golden/make_goldens.py does the same)
        x0 = x0.reshape(n, 1)
        kf.x = x0.copy()
        zs = [None if t in (0, 4, 5, T - 1) else z.reshape(m, 1) for t, z in enumerate(zs)]
        zs_ref = np.empty(T, dtype=object)
        for t in range(T):
            zs_ref[t] = zs[t]
        kw_ref, kw_or = dict(update_first=True), dict(update_first=True)
    want = kf.batch_filter(zs_ref, **kw_ref)
    got = kf_oracle.kf_batch_filter(x0, P0, list(zs), Fa, Qa, Ha, Ra, **kw_or)
    for g, w, key in zip(got, want, ("means", "covs", "means_p", "covs_p")):
        assert g.shape == np.asarray(w).shape, key
        assert rel_err_rows(g, np.asarray(w)) < TIGHT, key
    # smoothers on the reference's own histories: the class convention uses Fs[k+1], the module function Fs[k]
    mu, cov = np.asarray(want[0]), np.asarray(want[1])
    Fl = Fa if isinstance(Fa, list) else [F] * T
    Ql = Qa if isinstance(Qa, list) else [Q] * T
    for conv, out in (("class", kf.rts_smoother(mu, cov, Fs=Fl, Qs=Ql)), ("module", ref.KM.rts_smoother(mu, cov, Fl, Ql))):
        mine = kf_oracle.rts_smoother(mu, cov, Fl, Ql, conv)
        for g, w, key in zip(mine, out, ("x", "P", "K", "Pp")):
            assert rel_err_rows(g, np.asarray(w)) < 1e-11, (conv, key)


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3), (9, 4), (16, 8)])
def test_single_steps_and_byproducts(ref, n, m):
    """predict / update with every by-product (kalman_filter.py:437-561), log-likelihood and mahalanobis (:1203-1239),
    update_correlated (:670-752), the steady-state pair (:563-668)"""
    rs = np.random.RandomState(77 * n + m)
    kf = ref.K.KalmanFilter(dim_x=n, dim_z=m)
    x, P = rs.randn(n), spd(rs, n, 3.0)
    F, Q, H, R = stable_F(rs, n), spd(rs, n, 0.1), rs.randn(m, n), spd(rs, m)
    kf.x, kf.P, kf.F, kf.Q, kf.H, kf.R = x.copy(), P.copy(), F, Q, H, R
    kf.predict()
    xp, Pp = kf_oracle.kf_predict(x, P, F, Q)
    assert rel_err_rows(xp[None], kf.x[None]) < TIGHT and rel_err_rows(Pp[None], kf.P[None]) < TIGHT
    z = rs.randn(m)
    kf.update(z)
    xu, Pu, y, K, S, SI = kf_oracle.kf_update(xp, Pp, z, R, H)
    for g, w, key in ((xu, kf.x, "x"), (Pu, kf.P, "P"), (y, kf.y, "y"), (K, kf.K, "K"), (S, kf.S, "S"), (SI, kf.SI, "SI")):
        assert rel_err_rows(np.ravel(g)[None], np.ravel(w)[None]) < TIGHT, key
    assert abs(kf_oracle.log_likelihood(y, S) - kf.log_likelihood) <= 1e-12 * max(1.0, abs(kf.log_likelihood))
    assert abs(kf_oracle.mahalanobis(y, SI) - kf.mahalanobis) <= 1e-12 * max(1.0, abs(kf.mahalanobis))
    # update_correlated on a fresh prior
    kf2 = ref.K.KalmanFilter(dim_x=n, dim_z=m)
    Mx = 0.1 * rs.randn(n, m)
    kf2.x, kf2.P, kf2.H, kf2.R, kf2.M = xp.copy(), Pp.copy(), H, R, Mx
    kf2.update_correlated(z)
    xc, Pc = kf_oracle.update_correlated(xp, Pp, z, R, H, Mx)[:2]
    assert rel_err_rows(xc[None], kf2.x[None]) < TIGHT and rel_err_rows(Pc[None], kf2.P[None]) < TIGHT
    # steady-state pair with the gain the update just produced
    kf3 = ref.K.KalmanFilter(dim_x=n, dim_z=m)
    kf3.x, kf3.F, kf3.H, kf3.K = x.copy(), F, H, np.asarray(K).copy()
    zs = [rs.randn(m) for _ in range(6)]
    xs = []
    for zz in zs:
        kf3.predict_steadystate()
        kf3.update_steadystate(zz)
        xs.append(kf3.x.copy())
    mine = kf_oracle.steadystate_filter(x, np.array(zs), F, H, np.asarray(K))
    assert rel_err_rows(np.asarray(mine[0] if isinstance(mine, tuple) else mine), np.array(xs)) < TIGHT


@pytest.mark.parametrize("n,m,points", [(2, 1, "merwe"), (3, 2, "merwe"), (6, 3, "merwe"), (9, 3, "merwe"), (12, 5, "merwe"),
                                        (16, 8, "merwe"), (4, 2, "julier"), (10, 4, "julier")])
def test_unscented_filter_and_smoother(ref, n, m, points):
    """UKF.batch_filter / rts_smoother (UKF.py:524-739) with linear fx / hx, Merwe and Julier points, a missing measurement"""
    rs = np.random.RandomState(31 * n + m)
    T, dt = 8, 0.1
    F, H = np.eye(n) + 0.1 * stable_F(rs, n), rs.randn(m, n)
    Q, R = spd(rs, n, 0.02), spd(rs, m, 0.3)
    x0, P0 = rs.randn(n), spd(rs, n, 2.0)
    zs = [rs.randn(m) for _ in range(T)]
    zs[3] = None
    zs_ref = np.empty(T, dtype=object)                                     # (an object array: np.size(zs, 0) on a ragged list fails)
    for t in range(T):
        zs_ref[t] = zs[t]
    fx, hx = (lambda x, dt_: F @ x), (lambda x: H @ x)
    if points == "merwe":
        alpha, beta, kappa = 0.3, 2.0, 3.0 - n
        pts = ref.K.MerweScaledSigmaPoints(n, alpha, beta, kappa)
        Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    else:
        kappa = 1.5
        pts = ref.K.JulierSigmaPoints(n, kappa)
        Wm, Wc = ukf_oracle.julier_weights(n, kappa)
    assert np.array_equal(Wm, pts.Wm) and np.array_equal(Wc, pts.Wc)
    sig = pts.sigma_points(x0, P0)
    mine = (ukf_oracle.merwe_sigma_points(x0, P0, alpha, kappa) if points == "merwe" else ukf_oracle.julier_sigma_points(x0, P0, kappa))
    assert rel_err_rows(mine, sig) < TIGHT
    xt, Pt = ref.K.unscented_transform(sig, pts.Wm, pts.Wc, Q)
    xo, Po = ukf_oracle.unscented_transform(sig, Wm, Wc, Q)
    assert rel_err_rows(xo[None], xt[None]) < TIGHT and rel_err_rows(Po[None], Pt[None]) < TIGHT
    if points != "merwe":
        return                                                             # (the oracle's filter loop is written for Merwe points)
    ukf = ref.K.UnscentedKalmanFilter(dim_x=n, dim_z=m, dt=dt, hx=hx, fx=fx, points=pts)
    ukf.x, ukf.P, ukf.Q, ukf.R = x0.copy(), P0.copy(), Q, R
    mu, cov = ukf.batch_filter(zs_ref)
    mo, co = ukf_oracle.ukf_batch_filter(x0, P0, list(zs), fx, hx, dt, Q, R, alpha, beta, kappa)
    assert rel_err_rows(mo, mu) < 1e-12 and rel_err_rows(co, cov) < 1e-12
    xs, Ps, Ks = ukf.rts_smoother(mu, cov)
    xo, Po, Ko = ukf_oracle.ukf_rts_smoother(mu, cov, fx, dt, Q, alpha, beta, kappa)
    for g, w, key in ((xo, xs, "x"), (Po, Ps, "P"), (Ko, Ks, "K")):
        assert rel_err_rows(g, w) < 1e-10, key


@pytest.mark.parametrize("n,m,nm", [(2, 1, 2), (4, 2, 3), (6, 3, 2), (9, 4, 5), (9, 3, 8), (16, 8, 2), (12, 6, 4)])
def test_imm_and_mmae_banks(ref, n, m, nm):
    """IMMEstimator (IMM.py:124-249) and MMAEFilterBank (mmae.py:140-212) with a missing measurement"""
    rs = np.random.RandomState(500 + 17 * n + 3 * m + nm)
    T = 7
    Fs, Qs = [stable_F(rs, n) for _ in range(nm)], [spd(rs, n, 0.05) for _ in range(nm)]
    Hs, Rs = [rs.randn(m, n) for _ in range(nm)], [spd(rs, m, 0.5) for _ in range(nm)]
    xs0, Ps0 = [rs.randn(n) for _ in range(nm)], [spd(rs, n, 2.0) for _ in range(nm)]
    mu0 = rs.rand(nm) + 0.1
    mu0 /= mu0.sum()
    Mt = rs.rand(nm, nm) + 0.2
    Mt /= Mt.sum(axis=1, keepdims=True)
    zs = [rs.randn(m) for _ in range(T)]
    zs[2] = None

    def bank():
        out = []
        for j in range(nm):
            f = ref.K.KalmanFilter(dim_x=n, dim_z=m)
            f.x, f.P, f.F, f.Q, f.H, f.R = xs0[j].copy(), Ps0[j].copy(), Fs[j], Qs[j], Hs[j], Rs[j]
            out.append(f)
        return out
    imm = ref.K.IMMEstimator(bank(), mu0.copy(), Mt.copy())
    want = dict(x=[], P=[], mu=[], xp=[], Pp=[])
    for z in zs:
        imm.predict()
        want["xp"].append(imm.x_prior.copy())
        want["Pp"].append(imm.P_prior.copy())
        imm.update(z)
        want["x"].append(imm.x.copy())
        want["P"].append(imm.P.copy())
        want["mu"].append(imm.mu.copy())
    x, P, mu, xp, Pp, _ = imm_oracle.imm_batch(xs0, Ps0, mu0, Mt, zs, Fs, Qs, Hs, Rs)
    for g, key in ((x, "x"), (P, "P"), (mu, "mu"), (xp, "xp"), (Pp, "Pp")):
        assert rel_err_rows(g, np.array(want[key])) < 1e-11, key
    mm = ref.K.MMAEFilterBank(bank(), mu0.copy(), dim_x=n)
    wx, wP, wp = [], [], []
    for z in zs:
        mm.predict()
        mm.update(z)
        wx.append(np.array(mm.x).copy())
        wP.append(np.array(mm.P).copy())
        wp.append(np.array(mm.p).copy())
    x, P, p, _ = imm_oracle.mmae_batch(xs0, Ps0, mu0, zs, Fs, Qs, Hs, Rs)
    assert rel_err_rows(x, np.array(wx)) < 1e-11 and rel_err_rows(P, np.array(wP)) < 1e-11 and rel_err_rows(p, np.array(wp)) < 1e-11


@pytest.mark.parametrize("N", [1, 2, 7, 64, 1000, 8000, 40000])
@pytest.mark.parametrize("family", ["uniform", "peaked", "zeros_and_ties"])
def test_resamplers_bit_equal(ref, N, family):
    """systematic / stratified / multinomial / residual_resample (resampling.py:27-176): identical indices under the same seed"""
    rs = np.random.RandomState(9 * N + len(family))
    if family == "uniform":
        w = rs.rand(N)
    elif family == "peaked":
        w = rs.rand(N) ** 12 + 1e-300
    else:
        w = np.floor(rs.rand(N) * 4.0) / 4.0
        w[rs.randint(N)] += 1.0
    w = w / w.sum()
    for name, mine in (("systematic_resample", resample_oracle.systematic_seeded), ("stratified_resample", resample_oracle.stratified_seeded),
                       ("multinomial_resample", resample_oracle.multinomial_seeded), ("residual_resample", resample_oracle.residual_seeded)):
        np.random.seed(1234 + N)
        want = getattr(ref.M, name)(w.copy())
        np.random.seed(1234 + N)
        got = mine(w.copy())
        assert np.asarray(got).dtype == np.asarray(want).dtype and np.array_equal(got, want), (name, family, N)
