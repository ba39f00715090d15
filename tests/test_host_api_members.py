"""Host-side members of the KalmanFilter mirror that need no kernel of their own: test_matrix_dimensions,
log_likelihood_of, and the module-level predict_steadystate / update_steadystate.  The last two launch
fk_kf_steadystate_f64; here (no GPU) their argument plumbing is checked with `_Core.steadystate` replaced by
a NumPy stand-in with the same contract, against the reference's formulas (kalman_filter.py:1511-1568,
1624-1660).  tests/test_gpu_zz_module_steadystate.py repeats the comparison through the real kernel."""
import numpy as np
import pytest

import filterpy_amd.kalman.kalman_filter as kfm
from filterpy_amd.kalman import KalmanFilter


def _fake_steadystate(n, m, N, T, x, F, H, K, z, mask=None, B=None, us=None, nu=0, k_per_track=False, layout="soa"):
    """NumPy stand-in for _Core.steadystate (same arguments, same returns) for one step."""
    assert N == 1 and T == 1 and x.shape == (1, n)
    xs = x.copy()
    y = None
    if F is not None:
        assert F.shape == (n, n)
        xs = xs @ F.T
        if B is not None:
            assert B.shape == (n, nu) and us.shape == (1, 1, nu)
            xs = xs + us[0] @ B.T
    if z is not None:
        assert H.shape == (m, n) and K.shape == (n, m) and z.shape == (1, 1, m)
        y = z[0] - xs @ H.T
        xs = xs + y @ K.T
        y = y[None]
    return xs, None, None, y


@pytest.fixture
def fake_engine(monkeypatch):
    monkeypatch.setattr(kfm._Core, "steadystate", staticmethod(_fake_steadystate))


def test_update_steadystate_function(fake_engine):
    r = np.random.default_rng(3)
    n, m = 4, 2
    x = r.standard_normal(n)
    H = r.standard_normal((m, n))
    K = r.standard_normal((n, m))
    z = r.standard_normal(m)
    want = x + K @ (z - H @ x)
    np.testing.assert_allclose(kfm.update_steadystate(x, z, K, H), want, rtol=1e-14)
    # column-vector state keeps its shape
    got = kfm.update_steadystate(x.reshape(n, 1), z.reshape(m, 1), K, H)
    assert got.shape == (n, 1)
    np.testing.assert_allclose(got[:, 0], want, rtol=1e-14)
    # scalars in, scalar out (kalman_filter.py:1547)
    out = kfm.update_steadystate(1., 2., 0.5)
    assert isinstance(out, float) and out == 1.5
    # a missing measurement changes nothing
    assert kfm.update_steadystate(x, None, K, H) is x


def test_predict_steadystate_function(fake_engine):
    r = np.random.default_rng(4)
    n = 3
    x = r.standard_normal(n)
    F = r.standard_normal((n, n))
    np.testing.assert_allclose(kfm.predict_steadystate(x, F), F @ x, rtol=1e-14)
    B = r.standard_normal((n, 2))
    u = r.standard_normal(2)
    np.testing.assert_allclose(kfm.predict_steadystate(x, F, u, B), F @ x + B @ u, rtol=1e-14)
    assert kfm.predict_steadystate(2., 3.) == 6.
    assert kfm.predict_steadystate(2., 3., u=1., B=0.5) == 6.5
    col = kfm.predict_steadystate(x.reshape(n, 1), F)
    assert col.shape == (n, 1)


def test_test_matrix_dimensions():
    kf = KalmanFilter(dim_x=4, dim_z=2)
    kf.test_matrix_dimensions()
    kf.test_matrix_dimensions(z=np.zeros((2, 1)))
    with pytest.raises(AssertionError):
        kf.test_matrix_dimensions(z=np.zeros(3))
    with pytest.raises(AssertionError):
        kf.test_matrix_dimensions(H=np.zeros((2, 3)))
    with pytest.raises(AssertionError):
        kf.test_matrix_dimensions(R=np.eye(3))
    with pytest.raises(AssertionError):
        kf.test_matrix_dimensions(F=np.eye(3))
    with pytest.raises(AssertionError):
        kf.test_matrix_dimensions(Q=np.eye(5))
    kf.P = np.eye(3)
    with pytest.raises(AssertionError):
        kf.test_matrix_dimensions()
    # 1-D state, scalar measurement of a 1 x n H
    k1 = KalmanFilter(dim_x=2, dim_z=1)
    k1.x = np.zeros(2)
    k1.test_matrix_dimensions(z=3.)
    k1.test_matrix_dimensions(z=np.array([3.]))
    k1.test_matrix_dimensions(R=5.)
    with pytest.raises(AssertionError):
        k1.test_matrix_dimensions(z=np.zeros(2))


def test_log_likelihood_of():
    import math
    import sys
    from scipy.stats import multivariate_normal
    kf = KalmanFilter(dim_x=2, dim_z=2)
    kf.x = np.array([[1.], [2.]])
    kf.H = np.eye(2)
    kf.S = np.array([[2., .3], [.3, 1.]])
    z = np.array([[1.5], [1.]])
    want = multivariate_normal.logpdf(z.ravel(), mean=np.array([1., 2.]), cov=kf.S)
    assert abs(kf.log_likelihood_of(z) - want) < 1e-13
    assert kf.log_likelihood_of(None) == math.log(sys.float_info.min)
