"""Module-level predict_steadystate / update_steadystate (kalman_filter.py:1511-1568, 1624-1660) through
fk_kf_steadystate_f64 on the GPU, against the formulas."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_module_steadystate_functions():
    import filterpy_amd.kalman as K
    r = np.random.default_rng(11)
    for n, m in ((2, 1), (4, 2), (6, 3)):
        x = r.standard_normal(n)
        F = np.eye(n) + 0.1 * r.standard_normal((n, n))
        H = r.standard_normal((m, n))
        G = 0.3 * r.standard_normal((n, m))
        z = r.standard_normal(m)
        B = r.standard_normal((n, 2))
        u = r.standard_normal(2)
        np.testing.assert_allclose(K.predict_steadystate(x, F), F @ x, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(K.predict_steadystate(x, F, u, B), F @ x + B @ u, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(K.update_steadystate(x, z, G, H), x + G @ (z - H @ x), rtol=1e-13, atol=1e-15)
        col = K.update_steadystate(x.reshape(n, 1), z.reshape(m, 1), G, H)
        assert col.shape == (n, 1)
    assert K.update_steadystate(1., 2., 0.5) == 1.5
    assert K.predict_steadystate(2., 3.) == 6.
