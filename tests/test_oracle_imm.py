"""Pin oracle/imm_oracle.py to the goldens frozen from the live filterpy.kalman.IMMEstimator."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows
from oracle import imm_oracle


@pytest.mark.parametrize("n,m,nm", [(2, 1, 2), (4, 2, 2), (4, 2, 3), (6, 3, 2), (3, 2, 2), (5, 2, 3)])
def test_imm_vs_golden(n, m, nm):
    g = golden("imm")
    p = f"n{n}m{m}k{nm}_"
    x, P, mu, xp, Pp, L = imm_oracle.imm_batch(g[p + "xs0"], g[p + "Ps0"], g[p + "mu0"], g[p + "M"], g[p + "zs"],
                                               g[p + "Fs"], g[p + "Qs"], g[p + "Hs"], g[p + "Rs"])
    assert rel_err_rows(x, g[p + "x"]) < 1e-12 and rel_err_rows(P, g[p + "P"]) < 1e-12
    assert rel_err_rows(xp, g[p + "xp"]) < 1e-12 and rel_err_rows(Pp, g[p + "Pp"]) < 1e-12
    assert np.allclose(mu, g[p + "mu"], rtol=1e-11, atol=1e-15)
    assert np.allclose(L, g[p + "L"], rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("n,m,nm", [(2, 1, 2), (4, 2, 2), (4, 2, 3), (6, 3, 3), (3, 2, 2), (2, 1, 3)])
def test_mmae_vs_golden(n, m, nm):
    g = golden("mmae")
    p = f"n{n}m{m}k{nm}_"
    x, P, pr, L = imm_oracle.mmae_batch(g[p + "xs0"], g[p + "Ps0"], g[p + "p0"], g[p + "zs"],
                                        g[p + "Fs"], g[p + "Qs"], g[p + "Hs"], g[p + "Rs"])
    assert rel_err_rows(x, g[p + "x"]) < 1e-12 and rel_err_rows(P, g[p + "P"]) < 1e-12
    assert np.allclose(pr, g[p + "p"], rtol=1e-11, atol=1e-15)
    assert np.allclose(L, g[p + "L"], rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("kind", ["imm", "mmae"])
def test_missing_measurements_vs_live_reference(kind):
    """update(None): the filters keep x, P; their likelihood is the density of a zero residual under the S of their last
    real update (1 before any) and the mode probabilities are re-weighted with it (tests/golden/make_imm_missing_golden.py)."""
    g = golden("imm_missing")
    miss = set(int(t) for t in g["missing"])
    for n, m, nm in g["cases"]:
        p = f"n{n}m{m}k{nm}_"
        zs = [None if t in miss else z for t, z in enumerate(g[p + "zs"])]
        Hs = [g[p + "H"]] * int(nm)
        q = p + kind + "_"
        if kind == "imm":
            x, P, mu, xp, Pp, L = imm_oracle.imm_batch(g[p + "xs0"], g[p + "Ps0"], g[p + "mu0"], g[p + "M"], zs,
                                                      g[p + "Fs"], g[p + "Qs"], Hs, g[p + "Rs"])
            assert np.allclose(L, g[q + "L"], rtol=1e-10, atol=1e-300)
        else:
            mu0 = g[p + "mu0"] / g[p + "mu0"].sum()
            x, P, mu, L = imm_oracle.mmae_batch(g[p + "xs0"], g[p + "Ps0"], mu0, zs, g[p + "Fs"], g[p + "Qs"], Hs, g[p + "Rs"])
        assert np.allclose(x, g[q + "x"], rtol=1e-11, atol=1e-12) and np.allclose(P, g[q + "P"], rtol=1e-11, atol=1e-12)
        assert np.allclose(mu, g[q + "mu"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("kind", ["imm", "mmae"])
def test_control_input_vs_live_reference(kind):
    """predict(u): x = F x + B u in every filter with its own B (tests/golden/make_imm_control_golden.py)."""
    g = golden("imm_control")
    for n, m, nm, nu in g["cases"]:
        p = f"n{n}m{m}k{nm}_"
        Hs = [g[p + "H"]] * int(nm)
        q = p + kind + "_"
        if kind == "imm":
            x, P, mu = imm_oracle.imm_batch(g[p + "xs0"], g[p + "Ps0"], g[p + "mu0"], g[p + "M"], list(g[p + "zs"]),
                                            g[p + "Fs"], g[p + "Qs"], Hs, g[p + "Rs"], Bs=g[p + "Bs"], us=g[p + "us"])[:3]
        else:
            mu0 = g[p + "mu0"] / g[p + "mu0"].sum()
            x, P, mu = imm_oracle.mmae_batch(g[p + "xs0"], g[p + "Ps0"], mu0, list(g[p + "zs"]), g[p + "Fs"], g[p + "Qs"], Hs,
                                             g[p + "Rs"], Bs=g[p + "Bs"], us=g[p + "us"])[:3]
        assert np.allclose(x, g[q + "x"], rtol=1e-11, atol=1e-12) and np.allclose(P, g[q + "P"], rtol=1e-11, atol=1e-12)
        assert np.allclose(mu, g[q + "mu"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("name", ["imm_big", "imm_banks16"])
def test_big_banks_vs_live_reference(name):
    """banks of up to eight filters, dim_x up to 9, dim_z up to 4 (tests/golden/make_imm_big_golden.py); nine to sixteen filters,
    dim_x up to 16, dim_z up to 8 (make_imm_banks16_golden.py)"""
    g = golden(name)
    for n, m, nm in g["imm_cases"]:
        p = f"imm_n{n}m{m}k{nm}_"
        x, P, mu, xp, Pp, L = imm_oracle.imm_batch(g[p + "xs0"], g[p + "Ps0"], g[p + "mu0"], g[p + "M"], g[p + "zs"],
                                                   g[p + "Fs"], g[p + "Qs"], g[p + "Hs"], g[p + "Rs"])
        assert rel_err_rows(x, g[p + "x"]) < 1e-12 and rel_err_rows(P, g[p + "P"]) < 1e-12
        assert rel_err_rows(xp, g[p + "xp"]) < 1e-12 and rel_err_rows(Pp, g[p + "Pp"]) < 1e-12
        assert np.allclose(mu, g[p + "mu"], rtol=1e-11, atol=1e-15) and np.allclose(L, g[p + "L"], rtol=1e-11, atol=1e-300)
    for n, m, nm in g["mmae_cases"]:
        p = f"mmae_n{n}m{m}k{nm}_"
        x, P, pr, L = imm_oracle.mmae_batch(g[p + "xs0"], g[p + "Ps0"], g[p + "p0"], g[p + "zs"],
                                            g[p + "Fs"], g[p + "Qs"], g[p + "Hs"], g[p + "Rs"])
        assert rel_err_rows(x, g[p + "x"]) < 1e-12 and rel_err_rows(P, g[p + "P"]) < 1e-12
        assert np.allclose(pr, g[p + "p"], rtol=1e-11, atol=1e-15) and np.allclose(L, g[p + "L"], rtol=1e-11, atol=1e-300)
