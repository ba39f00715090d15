"""batch_filter(saver=filterpy_amd.common.Saver(kf)) through the real kernel: the histories the package's Saver
collects equal the ones filterpy.common.Saver recorded from the reference (tests/golden/kf_saver.npz).
(tests/test_host_saver_batch.py is the same comparison with the launch replaced by the oracle, on the CPU.)"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2)])
def test_package_saver_histories(n, m):
    from filterpy_amd.common import Saver
    from filterpy_amd.kalman import KalmanFilter
    g = golden("kf_saver")
    p = f"n{n}m{m}_"
    kf = KalmanFilter(n, m)
    kf.x, kf.P = g[p + "x0"].copy(), g[p + "P0"].copy()
    kf.F, kf.Q, kf.H, kf.R = g[p + "F"].copy(), g[p + "Q"].copy(), g[p + "H"].copy(), g[p + "R"].copy()
    s = Saver(kf, skip_private=True)
    zl = [z if k else None for z, k in zip(g[p + "zs"], g[p + "mask"])]
    kf.batch_filter(zl, saver=s)
    assert len(s) == len(zl)
    for k in ("x", "P", "x_prior", "P_prior", "x_post", "P_post", "K", "y", "S", "SI"):
        got, ref = np.array(s[k], dtype=float), g[p + k]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-11), k
    assert np.allclose(np.array(s["log_likelihood"], dtype=float), g[p + "log_likelihood"], rtol=1e-10, atol=1e-10)
    assert np.allclose(np.array(s["mahalanobis"], dtype=float), g[p + "mahalanobis"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3)])
@pytest.mark.parametrize("update_first", [False, True])
def test_attributes_after_batch_filter_equal_the_epoch_loop(n, m, update_first):
    """Without a saver batch_filter still leaves K, y, S, SI, z, the priors / posteriors and fresh lazy likelihoods
    at the last epoch, like the reference's loop of predict()/update() (kalman_filter.py:940-993)."""
    from filterpy_amd.kalman import KalmanFilter
    g = golden("kf_saver")
    p = f"n{n}m{m}_"

    def make():
        kf = KalmanFilter(n, m)
        kf.x, kf.P = g[p + "x0"].copy(), g[p + "P0"].copy()
        kf.F, kf.Q, kf.H, kf.R = g[p + "F"].copy(), g[p + "Q"].copy(), g[p + "H"].copy(), g[p + "R"].copy()
        return kf

    zl_all = [z if k else None for z, k in zip(g[p + "zs"], g[p + "mask"])]
    cuts = {len(zl_all)} | {i + 1 for i, z in enumerate(zl_all) if z is None and i > 0}
    for T in sorted(cuts)[-3:]:
        zl = zl_all[:T]
        a, b = make(), make()
        a.batch_filter(zl, update_first=update_first)
        for z in zl:
            if update_first:
                b.update(z)
                b.predict()
            else:
                b.predict()
                b.update(z)
        for k in ("x", "P", "K", "y", "S", "SI", "x_prior", "P_prior", "x_post", "P_post"):
            ga, gb = np.asarray(getattr(a, k), dtype=float), np.asarray(getattr(b, k), dtype=float)
            assert ga.shape == gb.shape, (k, T, ga.shape, gb.shape)
            assert np.allclose(ga, gb, rtol=1e-10, atol=1e-12), (k, T)
        assert np.shape(a.z) == np.shape(b.z) and np.array_equal(a.z == None, b.z == None)  # noqa: E711
        if zl[-1] is not None:
            assert np.array_equal(np.asarray(a.z, dtype=float), np.asarray(b.z, dtype=float))
        assert np.allclose(a.log_likelihood, b.log_likelihood, rtol=1e-10, atol=1e-10)
        assert np.allclose(a.mahalanobis, b.mahalanobis, rtol=1e-10, atol=1e-10)
