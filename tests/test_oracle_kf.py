"""Pin the NumPy oracle (oracle/kf_oracle.py) to the goldens frozen from the live reference.
Tight bar (1e-13): the oracle restates the reference's operation order with the same NumPy
routines, so it should agree to rounding."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows
from oracle import kf_oracle

TIGHT = 1e-13
DIMS = [tuple(d) for d in golden("kf_dims")["dims"]]


def test_c1_config():
    """BASELINE configs[0]: KalmanFilter dim_x=2 dim_z=1, batch_filter over 1000 z + RTS."""
    g = golden("kf_c1")
    for tag, x0 in (("1d", np.zeros(2)), ("col", np.zeros((2, 1)))):
        zs = list(g["zs"])
        out = kf_oracle.kf_batch_filter(x0, g["P0"], zs, g["F"], g["Q"], g["H"], g["R"])
        for got, key in zip(out, ("mu", "cov", "mup", "covp")):
            assert got.shape == g[f"{tag}_{key}"].shape
            assert rel_err_rows(got, g[f"{tag}_{key}"]) < TIGHT
        sm = kf_oracle.rts_smoother(out[0], out[1], g["F"], g["Q"], "class")
        for got, key in zip(sm, ("xs", "Ps", "Ks", "Pps")):
            assert rel_err_rows(got, g[f"{tag}_{key}"]) < 1e-12


@pytest.mark.parametrize("n,m", DIMS)
def test_batch_variants(n, m):
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    base = (g[p + "x0"], g[p + "P0"], list(g[p + "zs"]), g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    runs = {
        "plain": kf_oracle.kf_batch_filter(*base),
        "uf": kf_oracle.kf_batch_filter(*base, update_first=True),
        "alpha": kf_oracle.kf_batch_filter(*base, alpha_sq=1.02 ** 2),
        "ctrl": kf_oracle.kf_batch_filter(*base, B=g[p + "B"], us=list(g[p + "us"])),
        "miss": kf_oracle.kf_batch_filter(base[0], base[1],
                                          [z if k else None for z, k in zip(g[p + "zs"], g[p + "mask"])], *base[3:]),
    }
    for variant, out in runs.items():
        for got, key in zip(out, ("mu", "cov", "mup", "covp")):
            assert rel_err_rows(got, g[p + variant + "_" + key]) < TIGHT, (variant, key)
    sm = kf_oracle.rts_smoother(runs["plain"][0], runs["plain"][1], g[p + "F"], g[p + "Q"], "class")
    sm2 = kf_oracle.rts_smoother(runs["plain"][0], runs["plain"][1], g[p + "F"], g[p + "Q"], "module")
    for got, got2, key in zip(sm, sm2, ("x", "P", "K", "Pp")):
        assert rel_err_rows(got, g[p + "rts_" + key]) < 1e-11, key
        assert rel_err_rows(got2, g[p + "rtsm_" + key]) < 1e-11, key


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3), (9, 3)])
def test_per_step_models_and_conventions(n, m):
    g = golden("kf_models")
    p = f"n{n}m{m}_"
    out = kf_oracle.kf_batch_filter(g[p + "x0"], g[p + "P0"], list(g[p + "zs"]), g[p + "Fs"], g[p + "Qs"],
                                    g[p + "Hs"], g[p + "Rs"])
    for got, key in zip(out, ("mu", "cov", "mup", "covp")):
        assert rel_err_rows(got, g[p + key]) < TIGHT
        assert rel_err_rows(got, g[p + "mod_" + key]) < 1e-11      # module twin: other association order
    cls = kf_oracle.rts_smoother(g[p + "mu"], g[p + "cov"], g[p + "Fs"], g[p + "Qs"], "class")
    mod = kf_oracle.rts_smoother(g[p + "mod_mu"], g[p + "mod_cov"], g[p + "Fs"], g[p + "Qs"], "module")
    for a, b, key in zip(cls, mod, ("x", "P", "K", "Pp")):
        assert rel_err_rows(a, g[p + "rts_" + key]) < 1e-11
        assert rel_err_rows(b, g[p + "rtsm_" + key]) < 1e-11


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (4, 2), (6, 3), (9, 3)])
def test_single_steps(n, m):
    g = golden("kf_steps")
    p = f"n{n}m{m}_"
    xp, Pp = kf_oracle.kf_predict(g[p + "x0"], g[p + "P0"], g[p + "F"], g[p + "Q"])
    assert np.allclose(xp, g[p + "xp"], rtol=TIGHT, atol=0) and rel_err_rows(Pp[None], g[p + "Pp"][None]) < TIGHT
    x, P, y, K, S, SI = kf_oracle.kf_update(xp, Pp, g[p + "z"], g[p + "R"], g[p + "H"])
    for got, key in ((x, "x"), (P, "P"), (y, "y"), (K, "K"), (S, "S"), (SI, "SI")):
        assert rel_err_rows(np.atleast_2d(got)[None], np.atleast_2d(g[p + key])[None]) < TIGHT, key
    x2, P2, y2, K2, S2 = kf_oracle.proc_update(g[p + "m_xp"], g[p + "m_Pp"], g[p + "z"], g[p + "R"], g[p + "H"])
    for got, key in ((x2, "m_x"), (P2, "m_P"), (K2, "m_K"), (S2, "m_S")):
        assert rel_err_rows(np.atleast_2d(got)[None], np.atleast_2d(g[p + key])[None]) < TIGHT, key


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3), (9, 3)])
def test_saver_histories(n, m):
    """the histories filterpy.common.Saver records during batch_filter(saver=...)."""
    g = golden("kf_saver")
    p = f"n{n}m{m}_"
    zl = [z if k else None for z, k in zip(g[p + "zs"], g[p + "mask"])]
    mu, cov, mup, covp, Ks, ys, Ss, SIs = kf_oracle.kf_batch_filter(
        g[p + "x0"], g[p + "P0"], zl, g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"], return_all=True)
    assert rel_err_rows(mu, g[p + "x"]) < TIGHT and rel_err_rows(cov, g[p + "P"]) < TIGHT
    assert rel_err_rows(mup, g[p + "x_prior"]) < TIGHT and rel_err_rows(covp, g[p + "P_prior"]) < TIGHT
    for got, key in ((Ks, "K"), (Ss, "S"), (SIs, "SI")):
        assert np.allclose(got, g[p + key], rtol=1e-12, atol=1e-14), key
    assert np.allclose(ys, g[p + "y"][..., 0], rtol=1e-12, atol=1e-13)
    ll = np.array([kf_oracle.log_likelihood(y, S) if S.any() else g[p + "log_likelihood"][i]
                   for i, (y, S) in enumerate(zip(ys, Ss))])
    assert np.allclose(ll, g[p + "log_likelihood"], rtol=1e-10, atol=1e-10)
    mh = np.array([kf_oracle.mahalanobis(y, SI) for y, SI in zip(ys, SIs)])
    assert np.allclose(mh, g[p + "mahalanobis"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("n,m", [(2, 2), (4, 2), (6, 3), (9, 3), (3, 1)])
def test_scalar_attributes_vs_live_reference(n, m):
    """Scalar Q / R: raw in direct predict() / update() calls (q on every element of P, r on every element of S but
    r K K' in the Joseph term), eye * value inside batch_filter (tests/golden/make_scalar_attr_golden.py)."""
    g = golden("kf_scalar_attr")
    p = f"n{n}m{m}_"
    F, H, P0, x0, zs = (g[p + k] for k in ("F", "H", "P0", "x0", "zs"))
    q, r = float(g[p + "q"]), float(g[p + "r"])
    xp, Pp = kf_oracle.kf_predict(x0, P0, F, q)
    assert np.allclose(Pp, g[p + "step_Pp"], rtol=1e-13, atol=1e-14)
    x, P, y, K, S, SI = kf_oracle.kf_update(xp, Pp, zs[0], r, H)
    for got, key in ((x, "step_x"), (P, "step_P"), (y, "step_y"), (K, "step_K"), (S, "step_S"), (SI, "step_SI")):
        assert np.allclose(got, g[p + key], rtol=1e-12, atol=1e-13), key
    x, P = x0, P0
    for t, z in enumerate(zs):
        x, P = kf_oracle.kf_predict(x, P, F, q)
        x, P = kf_oracle.kf_update(x, P, z, r, H)[:2]
        assert np.allclose(x, g[p + "loop_x"][t], rtol=1e-11, atol=1e-12) and np.allclose(P, g[p + "loop_P"][t], rtol=1e-11, atol=1e-12)
    mu, cov, mup, covp = kf_oracle.kf_batch_filter(x0, P0, list(zs), F, q * np.eye(n), H, r * np.eye(m))
    for got, key in ((mu, "bf_mu"), (cov, "bf_cov"), (mup, "bf_mup"), (covp, "bf_covp")):
        assert np.allclose(got, g[p + key], rtol=1e-11, atol=1e-12), key
    x2, P2, y2, K2, S2 = kf_oracle.proc_update(xp, Pp, zs[0], r, H)
    for got, key in ((x2, "mod_x"), (P2, "mod_P"), (K2, "mod_K"), (S2, "mod_S")):
        assert np.allclose(got, g[p + key], rtol=1e-12, atol=1e-13), key


@pytest.mark.parametrize("n,m,nu", [(9, 3, 2), (12, 2, 3), (16, 4, 1)])
def test_every_batch_filter_argument_at_once_vs_live_reference(n, m, nu):
    """Fs / Qs / Hs / Rs / Bs lists + us + update_first + missing measurements in ONE call, and rts_smoother with
    Fs / Qs lists, at the sizes of the several-lanes-per-track kernels (tests/golden/make_kf_combo_golden.py): the oracle
    the GPU tests of the VAR instantiations compare against is the reference here too."""
    g = golden("kf_combo")
    p = f"n{n}m{m}_"
    zs = [None if np.isnan(z).all() else z for z in g[p + "zs"]]
    for uf, q in ((False, p + "pu_"), (True, p + "uf_")):
        out = kf_oracle.kf_batch_filter(g[p + "x0"], g[p + "P0"], zs, g[p + "Fs"], g[p + "Qs"], g[p + "Hs"], g[p + "Rs"],
                                        B=g[p + "Bs"], us=list(g[p + "us"]), update_first=uf)
        for got, key in zip(out, ("mu", "cov", "mup", "covp")):
            assert rel_err_rows(got.reshape(len(zs), -1), g[q + key].reshape(len(zs), -1)) < 1e-12, (uf, key)
    sm = kf_oracle.rts_smoother(g[p + "pu_mu"], g[p + "pu_cov"], g[p + "Fs"], g[p + "Qs"], "class")
    for got, key in zip(sm, ("x", "P", "K", "Pp")):
        assert rel_err_rows(got.reshape(len(zs), -1), g[p + "rts_" + key].reshape(len(zs), -1)) < 1e-11, key
