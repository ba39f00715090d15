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
