"""Pin oracle/ukf_oracle.py to the goldens frozen from the live reference (UKF.py,
sigma_points.py, unscented_transform.py)."""
import numpy as np

from conftest import golden, rel_err_rows
from oracle import ukf_oracle as uo


def test_merwe_everything():
    g = golden("ukf_merwe")
    for ci, (n, m, alpha, beta, kappa) in enumerate(g["cases"]):
        n, m = int(n), int(m)
        p = f"c{ci}_"
        Wm, Wc = uo.merwe_weights(n, alpha, beta, kappa)
        assert np.array_equal(Wm, g[p + "Wm"]) and np.array_equal(Wc, g[p + "Wc"])
        sig = uo.merwe_sigma_points(g[p + "x0"], g[p + "P0"], alpha, kappa)
        assert np.array_equal(sig, g[p + "sigmas"])
        ux, uP = uo.unscented_transform(sig, Wm, Wc, g[p + "Q"])
        assert np.array_equal(ux, g[p + "ut_x"]) and np.array_equal(uP, g[p + "ut_P"])
        F, H = g[p + "F"], g[p + "H"]
        fx, hx = (lambda x, dt: F @ x), (lambda x: H @ x)
        xp, Pp, sf = uo.ukf_predict(g[p + "x0"], g[p + "P0"], fx, 1.0, g[p + "Q"], Wm, Wc, alpha, kappa)
        assert np.array_equal(xp, g[p + "s1_xp"]) and np.array_equal(Pp, g[p + "s1_Pp"])
        assert np.array_equal(sf, g[p + "s1_sigmas_f"])
        x, P, K, y, S = uo.ukf_update(xp, Pp, sf, g[p + "zs"][0], hx, g[p + "R"], Wm, Wc)
        for got, key in ((x, "s1_x"), (P, "s1_P"), (K, "s1_K"), (S, "s1_S"), (y, "s1_y")):
            assert np.allclose(got, g[p + key], rtol=1e-12, atol=1e-14), key
        mu, cov = uo.ukf_batch_filter(g[p + "x0"], g[p + "P0"], list(g[p + "zs"]), fx, hx, 1.0, g[p + "Q"], g[p + "R"],
                                      alpha, beta, kappa)
        tol = 1e-11 if alpha >= 0.1 else 1e-7
        assert rel_err_rows(mu, g[p + "mu"]) < tol and rel_err_rows(cov, g[p + "cov"]) < tol
        xs, Ps, Ks = uo.ukf_rts_smoother(g[p + "mu"], g[p + "cov"], fx, 1.0, g[p + "Q"], alpha, beta, kappa)
        assert rel_err_rows(xs, g[p + "rts_x"]) < max(tol, 1e-9) and rel_err_rows(Ps, g[p + "rts_P"]) < max(tol, 1e-9)
        # the reference's own relational pin (test_ukf.py:948-978): UKF == KF on a linear model
        if alpha >= 0.1:
            assert np.allclose(g[p + "mu"], g[p + "kf_mu"], atol=1e-7)


def test_julier():
    g = golden("ukf_merwe")
    Wm, Wc = uo.julier_weights(4, 0.5)
    assert np.array_equal(Wm, g["jul_Wm"]) and np.array_equal(Wc, g["jul_Wc"])
    assert np.array_equal(uo.julier_sigma_points(g["jul_x0"], g["jul_P0"], 0.5), g["jul_sigmas"])


def test_constructor_hooks_against_the_live_reference():
    """every hook set (tests/ukf_hook_model.py; frozen by tests/golden/make_ukf_hooks_golden.py): the oracle's hooked
    sigma points / UT are bit-identical, its filter and smoother agree to rounding over 25 steps across the +-pi wrap"""
    import ukf_hook_model as hm
    g = golden("ukf_hooks")
    hk = uo.hooks(**hm.HOOKS)
    alpha, beta, kappa, dt = float(g["alpha"]), float(g["beta"]), float(g["kappa"]), float(g["dt"])
    Wm, Wc = uo.merwe_weights(3, alpha, beta, kappa)
    assert np.array_equal(Wm, g["Wm"]) and np.array_equal(Wc, g["Wc"])
    sig = uo.merwe_sigma_points(g["x0"][0], g["P0"][0], alpha, kappa, hk["sqrt"], hk["subtract"])
    assert np.array_equal(sig, g["sigmas0"])
    sf = np.array([hm.fx(s, dt) for s in sig])
    ux, uP = uo.unscented_transform(sf, Wm, Wc, g["Q"], hm.x_mean, hm.residual_x)
    assert np.array_equal(ux, g["ut_x"]) and np.array_equal(uP, g["ut_P"])
    ux, uP = uo.unscented_transform(sf, Wm, Wc, g["Q"], hm.x_mean, None)
    assert np.array_equal(ux, g["ut_meanonly_x"]) and np.array_equal(uP, g["ut_meanonly_P"])
    T, N = g["zs"].shape[:2]
    for i in range(N):
        zs = [g["zs"][t, i] for t in range(T)]
        if i == 1:
            zs[4] = None
        mu, cov = uo.ukf_batch_filter(g["x0"][i], g["P0"][i], zs, hm.fx, hm.hx, dt, g["Q"], g["R"], alpha, beta, kappa, hk)
        assert rel_err_rows(mu, g["mu"][:, i]) < 1e-12 and rel_err_rows(cov.reshape(T, -1), g["cov"][:, i].reshape(T, -1)) < 1e-12
        xs, Ps, Ks = uo.ukf_rts_smoother(g["mu"][:, i], g["cov"][:, i], hm.fx, dt, g["Q"], alpha, beta, kappa, hk=hk)
        assert rel_err_rows(xs, g["rts_x"][:, i]) < 1e-12
        assert rel_err_rows(Ps.reshape(T, -1), g["rts_P"][:, i].reshape(T, -1)) < 1e-12
        assert rel_err_rows(Ks.reshape(T, -1)[:-1], g["rts_K"][:, i].reshape(T, -1)[:-1]) < 1e-12
    # the headings really cross the wrap (the hooks are exercised, not idle)
    assert (np.abs(np.diff(g["mu"][:, :, 2], axis=0)) > 3.0).any()


def test_merwe_large_dims_against_the_live_reference():
    """dim_x 7..16, dim_z 1..8 (tests/golden/make_ukf_dims_golden.py): the padded kernel classes 8 / 12 / 16"""
    g = golden("ukf_dims")
    for ci, (n, m, alpha, beta, kappa) in enumerate(g["cases"]):
        n, m = int(n), int(m)
        p = f"c{ci}_"
        Wm, Wc = uo.merwe_weights(n, alpha, beta, kappa)
        assert np.array_equal(Wm, g[p + "Wm"]) and np.array_equal(Wc, g[p + "Wc"])
        sig = uo.merwe_sigma_points(g[p + "x0"], g[p + "P0"], alpha, kappa)
        assert np.array_equal(sig, g[p + "sigmas"])
        ux, uP = uo.unscented_transform(sig, Wm, Wc, g[p + "Q"])
        assert np.array_equal(ux, g[p + "ut_x"]) and np.array_equal(uP, g[p + "ut_P"])
        F, H = g[p + "F"], g[p + "H"]
        fx, hx = (lambda x, dt: F @ x), (lambda x: H @ x)
        xp, Pp, sf = uo.ukf_predict(g[p + "x0"], g[p + "P0"], fx, 1.0, g[p + "Q"], Wm, Wc, alpha, kappa)
        assert np.array_equal(xp, g[p + "s1_xp"]) and np.array_equal(Pp, g[p + "s1_Pp"])
        sh = np.array([hx(s) for s in sf])
        assert np.array_equal(sh, g[p + "s1_sigmas_h"])
        Pxz = uo.cross_variance(xp, np.dot(Wm, sh), sf, sh, Wc)
        assert np.array_equal(Pxz, g[p + "s1_Pxz"])
        x, P, K, y, S = uo.ukf_update(xp, Pp, sf, g[p + "zs"][0], hx, g[p + "R"], Wm, Wc)
        for got, key in ((x, "s1_x"), (P, "s1_P"), (K, "s1_K"), (S, "s1_S"), (y, "s1_y")):
            assert np.allclose(got, g[p + key], rtol=1e-12, atol=1e-14), key
        zs = list(g[p + "zs"]) if m > 1 else [np.array([z[0]]) for z in g[p + "zs"]]
        mu, cov = uo.ukf_batch_filter(g[p + "x0"], g[p + "P0"], zs, fx, hx, 1.0, g[p + "Q"], g[p + "R"], alpha, beta, kappa)
        assert rel_err_rows(mu, g[p + "mu"]) < 1e-12 and rel_err_rows(cov, g[p + "cov"]) < 1e-12
        xs, Ps, Ks = uo.ukf_rts_smoother(g[p + "mu"], g[p + "cov"], fx, 1.0, g[p + "Q"], alpha, beta, kappa)
        assert rel_err_rows(xs, g[p + "rts_x"]) < 1e-12 and rel_err_rows(Ps, g[p + "rts_P"]) < 1e-12
        assert rel_err_rows(Ks[:-1], g[p + "rts_K"][:-1]) < 1e-12
