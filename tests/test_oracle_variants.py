"""Pin the N4 restatements in oracle/kf_oracle.py (steady state, correlated noise, sequential
update) to goldens frozen from the live filterpy.kalman.KalmanFilter."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows
from oracle import kf_oracle

CASES = [(2, 1), (4, 2), (6, 3), (9, 3), (3, 2)]


@pytest.mark.parametrize("n,m", CASES)
def test_steadystate_vs_golden(n, m):
    g = golden("kf_variants")
    q = f"n{n}m{m}_"
    zs = [None if t == 7 else g[q + "zs"][t] for t in range(len(g[q + "zs"]))]
    x, xp, y = kf_oracle.steadystate_filter(g[q + "x0"], zs, g[q + "F"], g[q + "H"], g[q + "K"])
    assert rel_err_rows(x, g[q + "ss_x"]) < 1e-13 and rel_err_rows(xp, g[q + "ss_xp"]) < 1e-13
    assert np.allclose(y, g[q + "ss_y"], rtol=1e-12, atol=1e-13)
    x, xp, y = kf_oracle.steadystate_filter(g[q + "x0"], zs, g[q + "F"], g[q + "H"], g[q + "K"], B=g[q + "B"], us=g[q + "us"])
    assert rel_err_rows(x, g[q + "ssu_x"]) < 1e-13 and rel_err_rows(xp, g[q + "ssu_xp"]) < 1e-13


@pytest.mark.parametrize("n,m", CASES)
def test_update_correlated_vs_golden(n, m):
    g = golden("kf_variants")
    q = f"n{n}m{m}_"
    x, P = g[q + "x0"], g[q + "P0"]
    for t in range(3):
        x, P = kf_oracle.kf_predict(x, P, g[q + "F"], g[q + "Q"])
        R, H = (2.0 * g[q + "R"], 0.5 * g[q + "H"]) if t == 2 else (g[q + "R"], g[q + "H"])
        x, P, y, K, S, SI = kf_oracle.update_correlated(x, P, g[q + "zs"][t], R, H, g[q + "M"])
        assert rel_err_rows(x, g[q + "corr_x"][t]) < 1e-12 and rel_err_rows(P, g[q + "corr_P"][t]) < 1e-12
        assert rel_err_rows(K, g[q + "corr_K"][t]) < 1e-12 and rel_err_rows(S, g[q + "corr_S"][t]) < 1e-12


@pytest.mark.parametrize("n,m", CASES)
def test_update_sequential_vs_golden(n, m):
    g = golden("kf_variants")
    q = f"n{n}m{m}_"
    x, P = kf_oracle.kf_predict(g[q + "x0"].reshape(-1, 1), g[q + "P0"], g[q + "F"], g[q + "Q"])
    xb, Pb = x, P
    for i in range(m):
        x, P, y, K = kf_oracle.update_sequential(x, P, i, g[q + "zs"][0][i], g[q + "R"], g[q + "H"])
        assert rel_err_rows(x.ravel(), g[q + "seq_x"][i]) < 1e-12 and rel_err_rows(P, g[q + "seq_P"][i]) < 1e-12
    if m >= 2:
        x, P, _, _ = kf_oracle.update_sequential(xb, Pb, m - 2, g[q + "zs"][0][m - 2:], g[q + "R"], g[q + "H"])
        assert rel_err_rows(x.ravel(), g[q + "seqb_x"]) < 1e-12 and rel_err_rows(P, g[q + "seqb_P"]) < 1e-12
