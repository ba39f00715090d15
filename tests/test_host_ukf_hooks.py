"""Host logic of the UKF constructor hooks (UKF.py:284-340, sigma_points.py:99-116, unscented_transform.py:105-123) in
the three calling conventions, with CPU stand-ins for the kernels (tests/fake_ut_engine.py): what is checked here is the
Python layer -- layouts, which residuals reach which entry point, NULL-mean modes, state_add / residual_z plumbing --
against the numbers frozen from the live reference (tests/golden/ukf_hooks.npz).  The kernels' own arithmetic is
checked on the GPU (tests/test_gpu_ukf_hooks.py)."""
import numpy as np
import pytest
import torch

import fake_ut_engine
import ukf_hook_model as hm
from conftest import golden, rel_err_rows


_make = hm.make_filter


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("mode", ["loop", "vec", "torch"])
def test_hooked_bank_matches_the_reference(monkeypatch, mode, layout):
    calls = fake_ut_engine.install(monkeypatch)
    g = golden("ukf_hooks")
    T, N = g["zs"].shape[:2]
    kf = _make(g, mode, layout, N)
    kf.x, kf.P = g["x0"].copy(), g["P0"].copy()
    zs = [g["zs"][t] for t in range(T)]
    mu, cov = kf.batch_filter(zs)
    keep = [i for i in range(N) if i != 1]                   # track 1 of the golden skips one measurement (below)
    assert rel_err_rows(mu[:, keep].reshape(-1, 3), g["mu"][:, keep].reshape(-1, 3)) < 1e-10
    assert rel_err_rows(cov[:, keep].reshape(-1, 9), g["cov"][:, keep].reshape(-1, 9)) < 1e-10
    assert np.array_equal(kf.x, mu[-1]) and np.array_equal(kf.P, cov[-1])
    # every hook went through the residual entry points, none through the default-subtraction ones
    assert {"cross_residuals", "correct_residual"} <= set(calls) and "cross" not in calls and "correct" not in calls
    assert "sigma" not in calls                               # sqrt_method given: the Cholesky kernel is not used
    calls.clear()
    xs, Ps, Ks = kf.rts_smoother(g["mu"], g["cov"])
    assert rel_err_rows(xs.reshape(-1, 3), g["rts_x"].reshape(-1, 3)) < 1e-10
    assert rel_err_rows(Ps.reshape(-1, 9), g["rts_P"].reshape(-1, 9)) < 1e-10
    assert rel_err_rows(Ks[:-1].reshape(-1, 9), g["rts_K"][:-1].reshape(-1, 9)) < 1e-10
    assert "rts_residual" in calls and "rts" not in calls


def test_single_filter_step_api_with_hooks_and_missing_measurement(monkeypatch):
    """one filter (no n_tracks), predict() / update() / update(None) one call at a time, reference convention"""
    from filterpy_amd.kalman import MerweScaledSigmaPoints, UnscentedKalmanFilter
    fake_ut_engine.install(monkeypatch)
    g = golden("ukf_hooks")
    T = g["zs"].shape[0]
    i = 1
    pts = MerweScaledSigmaPoints(3, float(g["alpha"]), float(g["beta"]), float(g["kappa"]), sqrt_method=hm.sqrt_lower_t,
                                 subtract=hm.sigma_subtract)
    kf = UnscentedKalmanFilter(3, 2, float(g["dt"]), hm.hx, hm.fx, pts, x_mean_fn=hm.x_mean, z_mean_fn=hm.z_mean,
                               residual_x=hm.residual_x, residual_z=hm.residual_z, state_add=hm.state_add)
    kf.x, kf.P, kf.Q, kf.R = g["x0"][i].copy(), g["P0"][i].copy(), g["Q"], g["R"]
    for t in range(T):
        kf.predict()
        assert kf.x.shape == (3,) and kf.sigmas_f.shape == (7, 3) and np.array_equal(kf.x_prior, kf.x)
        kf.update(None if t == 4 else g["zs"][t, i])
        assert rel_err_rows(kf.x[None], g["mu"][t, i][None]) < 1e-10
        assert rel_err_rows(kf.P.reshape(1, -1), g["cov"][t, i].reshape(1, -1)) < 1e-10
        if t != 4:
            assert kf.K.shape == (3, 2) and kf.S.shape == (2, 2) and kf.y.shape == (2,) and abs(kf.y[1]) <= np.pi
    xs, Ps, Ks = kf.rts_smoother(g["mu"][:, i], g["cov"][:, i])
    assert xs.shape == (T, 3) and rel_err_rows(xs, g["rts_x"][:, i]) < 1e-10


def test_partial_hooks(monkeypatch):
    """only some hooks set: a mean hook alone keeps plain subtraction for the residuals; default sqrt with a custom
    `subtract` reads U off the Cholesky kernel; the standalone unscented_transform / sigma_points calls"""
    from filterpy_amd.kalman import MerweScaledSigmaPoints, unscented_transform
    calls = fake_ut_engine.install(monkeypatch)
    g = golden("ukf_hooks")
    alpha, beta, kappa, dt = float(g["alpha"]), float(g["beta"]), float(g["kappa"]), float(g["dt"])
    pts = MerweScaledSigmaPoints(3, alpha, beta, kappa, subtract=hm.sigma_subtract)
    sig = pts.sigma_points(g["x0"][0], g["P0"][0])
    assert "sigma" in calls
    assert rel_err_rows(sig, g["sigmas0"]) < 1e-14            # (scipy's upper factor vs numpy's lower one: same to rounding)
    sf = np.array([hm.fx(s, dt) for s in g["sigmas0"]])
    x, P = unscented_transform(sf, g["Wm"], g["Wc"], g["Q"], hm.x_mean, hm.residual_x)
    assert np.array_equal(x, g["ut_x"]) and rel_err_rows(P.reshape(1, -1), g["ut_P"].reshape(1, -1)) < 1e-14
    x, P = unscented_transform(sf, g["Wm"], g["Wc"], g["Q"], hm.x_mean, None)
    assert np.array_equal(x, g["ut_meanonly_x"]) and rel_err_rows(P.reshape(1, -1), g["ut_meanonly_P"].reshape(1, -1)) < 1e-13
    # a bank through the same public call
    xb, Pb = unscented_transform(np.stack([sf, sf]), g["Wm"], g["Wc"], g["Q"], hm.x_mean, hm.residual_x)
    assert xb.shape == (2, 3) and np.array_equal(xb[1], g["ut_x"]) and np.array_equal(Pb[0], Pb[1])


def test_caller_supplied_ut_function(monkeypatch):
    """UT= (UKF.py:395-396, :447-448, :712-713): the caller's transform replaces unscented_transform in predict, update,
    batch_filter and rts_smoother.  Here it is the ORACLE's restatement of the reference transform wrapped in a counter,
    so the results must equal the hooked run frozen from the live reference."""
    from filterpy_amd.kalman import MerweScaledSigmaPoints, UnscentedKalmanFilter
    fake_ut_engine.install(monkeypatch)
    from oracle import ukf_oracle as uo
    g = golden("ukf_hooks")
    T, i = g["zs"].shape[0], 2
    calls = []

    def my_ut(sigmas, Wm, Wc, noise_cov, mean_fn, residual_fn):
        calls.append(sigmas.shape)
        return uo.unscented_transform(np.asarray(sigmas), np.asarray(Wm), np.asarray(Wc), noise_cov, mean_fn, residual_fn)
    pts = MerweScaledSigmaPoints(3, float(g["alpha"]), float(g["beta"]), float(g["kappa"]), sqrt_method=hm.sqrt_lower_t,
                                 subtract=hm.sigma_subtract)
    kf = UnscentedKalmanFilter(3, 2, float(g["dt"]), hm.hx, hm.fx, pts, x_mean_fn=hm.x_mean, z_mean_fn=hm.z_mean,
                               residual_x=hm.residual_x, residual_z=hm.residual_z, state_add=hm.state_add)
    kf.x, kf.P, kf.Q, kf.R = g["x0"][i].copy(), g["P0"][i].copy(), g["Q"], g["R"]
    mu, cov = kf.batch_filter([g["zs"][t, i] for t in range(T)], UT=my_ut)
    assert len(calls) == 2 * T and calls[0] == (7, 3) and calls[1] == (7, 2)
    assert rel_err_rows(mu, g["mu"][:, i]) < 1e-10 and rel_err_rows(cov.reshape(T, -1), g["cov"][:, i].reshape(T, -1)) < 1e-10
    xs, Ps, Ks = kf.rts_smoother(g["mu"][:, i], g["cov"][:, i], UT=my_ut)
    assert len(calls) == 2 * T + (T - 1)
    assert rel_err_rows(xs, g["rts_x"][:, i]) < 1e-10 and rel_err_rows(Ps.reshape(T, -1), g["rts_P"][:, i].reshape(T, -1)) < 1e-10
    # a filter WITHOUT hooks takes the caller's UT as well (and forgets it after the call)
    pts2 = MerweScaledSigmaPoints(3, .5, 2., 0.)
    kf2 = UnscentedKalmanFilter(3, 2, float(g["dt"]), hm.hx, hm.fx, pts2)
    kf2.x, kf2.P, kf2.Q, kf2.R = g["x0"][i].copy(), g["P0"][i].copy(), g["Q"], g["R"]
    n0 = len(calls)
    kf2.predict(UT=my_ut)
    assert len(calls) == n0 + 1 and kf2._ut_fn is None
    kf2.predict()
    assert len(calls) == n0 + 1


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_step_api_in_device_mode_equals_batch(monkeypatch, layout):
    """device_callables=True: predict() / update() one call at a time (state up and down per call) give what batch_filter
    gives with everything resident; update(None) skips; a per-call R and hx override reach the kernels"""
    fake_ut_engine.install(monkeypatch)
    g = golden("ukf_hooks")
    T, N = g["zs"].shape[:2]
    kf = hm.make_filter(g, "torch", layout, N)
    kf.x, kf.P = g["x0"].copy(), g["P0"].copy()
    for t in range(T):
        kf.predict()
        assert kf.x.shape == (N, 3) and kf.sigmas_f.shape == (N, 7, 3)
        kf.update(g["zs"][t])
        keep = [i for i in range(N) if i != 1]
        assert rel_err_rows(kf.x[keep], g["mu"][t, keep]) < 1e-10
        assert rel_err_rows(kf.P[keep].reshape(len(keep), -1), g["cov"][t, keep].reshape(len(keep), -1)) < 1e-10
        assert kf.K.shape == (N, 3, 2) and kf.S.shape == (N, 2, 2) and kf.y.shape == (N, 2)
    x_before = kf.x.copy()
    kf.update(None)
    assert np.array_equal(kf.x, x_before) and kf.z.shape == (2, 1)
    # per-call overrides: a different R changes the gain, a different hx the innovation
    kf.predict()
    k0 = None
    for R in (g["R"], 10.0 * g["R"]):
        kf2 = hm.make_filter(g, "torch", layout, N)
        kf2.x, kf2.P = kf.x.copy(), kf.P.copy()
        kf2.sigmas_f = kf.sigmas_f.copy()
        kf2.update(g["zs"][0], R=R)
        k0 = kf2.K if k0 is None else k0
    assert not np.allclose(k0, kf2.K)


@pytest.mark.parametrize("mode", ["loop", "vec", "torch"])
def test_resident_batch_filter_leaves_the_last_epochs_attributes(monkeypatch, mode):
    """after batch_filter on the resident path the object holds what the reference's per-epoch loop leaves
    behind (UKF.py:623-632): sigmas_f / x_prior / P_prior of the last predict, sigmas_h / K / S / SI / y / z of the
    last update, posts = the final state -- i.e. exactly what the same calls made one at a time leave."""
    fake_ut_engine.install(monkeypatch)
    g = golden("ukf_hooks")
    T, N = g["zs"].shape[:2]
    zs = [g["zs"][t] for t in range(T)]
    for miss_last in (False, True):
        if miss_last:
            zs[-1] = None
        a, b = _make(g, mode, "soa", N), _make(g, mode, "soa", N)
        for kf in (a, b):
            kf.x, kf.P = g["x0"].copy(), g["P0"].copy()
        a.batch_filter(zs)
        for z in zs:
            b.predict()
            b.update(z)
        for name in ("x", "P", "sigmas_f", "sigmas_h", "K", "S", "SI", "y", "x_prior", "P_prior", "x_post", "P_post"):
            va, vb = np.asarray(getattr(a, name), dtype=float), np.asarray(getattr(b, name), dtype=float)
            assert va.shape == vb.shape, name
            assert np.allclose(va, vb, rtol=1e-12, atol=1e-14), (name, miss_last)
        assert np.shape(a.z) == np.shape(b.z) and (miss_last or np.allclose(np.asarray(a.z, float), np.asarray(b.z, float)))
        assert abs(a.log_likelihood - b.log_likelihood) < 1e-9 if N == 1 else True
