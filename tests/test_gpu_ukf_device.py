"""UnscentedKalmanFilter(device_callables=True): general fx / hx as vectorised callables on GPU tensors, the whole split
path (sigma_kernel -> fx -> ut_kernel -> sigma_kernel -> hx -> ut_kernel -> cross_kernel -> ukf_correct_kernel, and the
smoother's backward pass) resident in HBM (VERDICT r1 missing #2; filterpy/kalman/UKF.py:506-522, :462-481, :714-737).
Held against the goldens of the live reference run with the same models as Python lambdas."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows, ukf_tol

pytestmark = pytest.mark.gpu


def _cases():
    g = golden("ukf_merwe")
    return [(ci, int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4])) for ci, c in enumerate(g["cases"])]


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_device_callables_batch_filter_and_smoother(layout):
    import torch
    from filterpy_amd import _engine as E
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    g = golden("ukf_merwe")
    N = 70
    calls = {"fx": 0, "hx": 0}
    for ci, n, m, alpha, beta, kappa in _cases():
        if n > 9:
            continue
        p = f"c{ci}_"
        Fd, Hd = E.dev(g[p + "F"]), E.dev(g[p + "H"])

        def fx(sig, dt):
            assert isinstance(sig, torch.Tensor) and sig.is_cuda and tuple(sig.shape) == (N, 2 * n + 1, n)
            calls["fx"] += 1
            return torch.matmul(sig, Fd.T)

        def hx(sig):
            assert sig.is_cuda and tuple(sig.shape) == (N, 2 * n + 1, n)
            calls["hx"] += 1
            return torch.matmul(sig, Hd.T)
        ukf = UnscentedKalmanFilter(n, m, dt=1.0, hx=hx, fx=fx, points=MerweScaledSigmaPoints(n, alpha, beta, kappa),
                                    n_tracks=N, layout=layout, device_callables=True)
        ukf.x, ukf.P = np.tile(g[p + "x0"], (N, 1)), np.tile(g[p + "P0"], (N, 1, 1))
        ukf.Q, ukf.R = g[p + "Q"].copy(), g[p + "R"].copy()
        zs = np.tile(g[p + "zs"][:, None, :], (1, N, 1))
        T = zs.shape[0]
        mu, cov = ukf.batch_filter(zs)
        assert mu.shape == (T, N, n) and cov.shape == (T, N, n, n)
        for trk in (0, 63, 64, N - 1):
            assert rel_err_rows(mu[:, trk], g[p + "mu"]) < ukf_tol(ci, "mu"), (ci, trk)
            assert rel_err_rows(cov[:, trk], g[p + "cov"]) < ukf_tol(ci, "cov"), (ci, trk)
        assert rel_err_rows(ukf.x[[0, N - 1]], np.tile(g[p + "mu"][-1], (2, 1))) < ukf_tol(ci, "mu")
        # device in, device out: records stay in HBM
        ukf.x, ukf.P = np.tile(g[p + "x0"], (N, 1)), np.tile(g[p + "P0"], (N, 1, 1))
        dmu, dcov = ukf.batch_filter(E.to_records(zs, layout, 1), device_outputs=True)
        assert dmu.is_cuda and dcov.is_cuda
        assert np.array_equal(E.from_records(dmu, layout, 1, (n,)), mu) and np.array_equal(E.from_records(dcov, layout, 1, (n, n)), cov)
        # the smoother on the reference's own filter output
        Xs, Ps = np.tile(g[p + "mu"][:, None], (1, N, 1)), np.tile(g[p + "cov"][:, None], (1, N, 1, 1))
        xs, ps, Ks = ukf.rts_smoother(Xs, Ps)
        for trk in (0, N - 1):
            assert rel_err_rows(xs[:, trk], g[p + "rts_x"]) < ukf_tol(ci, "rts_x"), ci
            assert rel_err_rows(ps[:, trk], g[p + "rts_P"]) < ukf_tol(ci, "rts_P"), ci
            assert rel_err_rows(Ks[:-1, trk], g[p + "rts_K"][:-1]) < ukf_tol(ci, "rts_K"), ci
        dxs, dps, dKs = ukf.rts_smoother(E.to_records(Xs, layout, 1), E.to_records(Ps, layout, 1), device_outputs=True)
        assert dxs.is_cuda and np.array_equal(E.from_records(dxs, layout, 1, (n,)), xs)
    assert calls["fx"] > 0 and calls["hx"] > 0


def test_device_callables_nonlinear_model_vs_host_callables():
    """A genuinely non-linear pair (range / bearing-free polar-ish measurement, quadratic drift): the device-resident
    path and the host-callable path run the same kernels around the same arithmetic and agree to rounding."""
    import torch
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    n, m, N, T = 4, 2, 33, 12
    rs = np.random.RandomState(5)

    def fx_np(s, dt):          # (N, k, n) -> (N, k, n)
        o = s.copy()
        o[..., 0] += dt * s[..., 1] + 0.01 * s[..., 1] ** 2
        o[..., 2] += dt * s[..., 3]
        return o

    def hx_np(s):
        return np.stack([np.sqrt(1.0 + s[..., 0] ** 2 + s[..., 2] ** 2), s[..., 0] - 0.5 * s[..., 2]], axis=-1)

    def fx_t(s, dt):
        o = s.clone()
        o[..., 0] += dt * s[..., 1] + 0.01 * s[..., 1] ** 2
        o[..., 2] += dt * s[..., 3]
        return o

    def hx_t(s):
        return torch.stack([torch.sqrt(1.0 + s[..., 0] ** 2 + s[..., 2] ** 2), s[..., 0] - 0.5 * s[..., 2]], dim=-1)
    x0, P0 = rs.randn(N, n), np.tile(np.eye(n) * 2.0, (N, 1, 1))
    zs = rs.randn(T, N, m) + np.array([3.0, 0.0])
    out = []
    for dev in (False, True):
        ukf = UnscentedKalmanFilter(n, m, dt=0.5, hx=hx_t if dev else hx_np, fx=fx_t if dev else fx_np,
                                    points=MerweScaledSigmaPoints(n, .5, 2., 0.), n_tracks=N, vectorized=not dev,
                                    device_callables=dev)
        ukf.x, ukf.P = x0.copy(), P0.copy()
        ukf.Q, ukf.R = 0.05 * np.eye(n), 0.3 * np.eye(m)
        out.append(ukf.batch_filter(zs if dev else list(zs)))
    assert rel_err_rows(out[1][0].reshape(T * N, n), out[0][0].reshape(T * N, n)) < 1e-10
    assert rel_err_rows(out[1][1].reshape(T * N, -1), out[0][1].reshape(T * N, -1)) < 1e-10


def test_device_callables_need_a_bank_and_cuda_tensors():
    import torch
    from filterpy_amd.kalman import UnscentedKalmanFilter, MerweScaledSigmaPoints
    pts = MerweScaledSigmaPoints(2, .5, 2., 0.)
    with pytest.raises(ValueError):
        UnscentedKalmanFilter(2, 1, 1.0, hx=lambda s: s, fx=lambda s, dt: s, points=pts, device_callables=True)
    ukf = UnscentedKalmanFilter(2, 1, 1.0, hx=lambda s: s[..., :1].cpu(), fx=lambda s, dt: s, points=pts, n_tracks=3,
                                device_callables=True)
    with pytest.raises(TypeError):
        ukf.batch_filter(np.zeros((2, 3, 1)))
