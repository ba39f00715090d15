"""KalmanFilter.batch_filter(zs, saver=Saver(kf)) with the package's own Saver, on the CPU: the kernel launch
(`_Core.batch`) is replaced by the NumPy oracle behind the same contract, so what is tested is the host flow
-- argument marshalling, the per-epoch replay of x / P / priors / posteriors / K / y / S / SI into the filter
object and the Saver reading them (incl. the lazily evaluated likelihoods) -- against the histories
filterpy.common.Saver recorded from the reference's own epoch loop (tests/golden/kf_saver.npz).  The same
comparison through the real kernel is tests/test_gpu_api.py::test_saver_histories_drop_in."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden

sys.path.insert(0, ROOT)
from oracle import kf_oracle  # noqa: E402  (tests may use the oracle)

import filterpy_amd.kalman.kalman_filter as kfm  # noqa: E402
from filterpy_amd.common import Saver  # noqa: E402


def _fake_batch(n, m, N, T, x0, P0, z, mask, F, Q, H, R, mode, B=None, us=None, nu=0, alpha_sq=1.0,
                update_first=False, layout="soa", want_outputs=True, device_outputs=False, extras=()):
    assert N == 1 and mode == kfm.FK_MODEL_SHARED and not device_outputs
    zs = [None if (mask is not None and not mask[t, 0]) else z[t, 0] for t in range(T)]
    mu, cov, mup, covp, Ks, ys, Ss, SIs = kf_oracle.kf_batch_filter(
        x0[0], P0[0], zs, F, Q, H, R, alpha_sq=alpha_sq, update_first=update_first, return_all=True)
    res = [mu[:, None], cov[:, None], mup[:, None], covp[:, None], mu[-1][None], cov[-1][None]]
    if extras:
        hist = dict(y=ys[:, None], K=Ks[:, None], S=Ss[:, None], SI=SIs[:, None])
        res.append({k: hist[k] for k in extras})
    return res


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3)])
def test_batch_filter_fills_the_package_saver(monkeypatch, n, m):
    monkeypatch.setattr(kfm._Core, "batch", staticmethod(_fake_batch))
    g = golden("kf_saver")
    p = f"n{n}m{m}_"
    kf = kfm.KalmanFilter(dim_x=n, dim_z=m)
    kf.x, kf.P = g[p + "x0"].copy(), g[p + "P0"].copy()
    kf.F, kf.Q, kf.H, kf.R = g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"]
    s = Saver(kf)
    zl = [zz if k else None for zz, k in zip(g[p + "zs"], g[p + "mask"])]
    mu, cov, mup, covp = kf.batch_filter(zl, saver=s)
    assert len(s) == len(zl)
    assert np.allclose(mu, g[p + "mu"], rtol=1e-12, atol=1e-13)
    for k in ("x", "P", "x_prior", "P_prior", "x_post", "P_post", "K", "y", "S", "SI"):
        got, ref = np.array(s[k], dtype=float), g[p + k]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-12), k
    assert np.allclose(np.array(s["log_likelihood"], dtype=float), g[p + "log_likelihood"], rtol=1e-9, atol=1e-9)
    assert np.allclose(np.array(s["mahalanobis"], dtype=float), g[p + "mahalanobis"], rtol=1e-9, atol=1e-11)
    assert np.allclose(np.array(s["likelihood"], dtype=float), g[p + "likelihood"], rtol=1e-8, atol=1e-300)
    s.to_array()
    assert s.x.shape == g[p + "x"].shape and s.P.shape == g[p + "P"].shape


def _fake_update(n, m, N, x, P, z, H, R, mode, mask=None, layout="soa", flags=0, inv=None):
    assert N == 1 and mode == kfm.FK_MODEL_SHARED and mask is None
    xn, Pn, y, K, S, SI = kf_oracle.kf_update(x[0], P[0], np.asarray(z).reshape(m), R, H)
    return xn[None], Pn[None], y[None], K[None], S[None], SI[None]


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3)])
@pytest.mark.parametrize("drop_tail", [0, 1, 3])
def test_attributes_after_batch_filter_without_saver(monkeypatch, n, m, drop_tail):
    """without a saver the reference's per-epoch loop still leaves K, y, S, SI, z at the last
    epoch and clears the cached likelihoods (kalman_filter.py:511-561, :940-993); the histories recorded by
    filterpy.common.Saver from the live reference (kf_saver.npz) give the expected last values."""
    monkeypatch.setattr(kfm._Core, "batch", staticmethod(_fake_batch))
    monkeypatch.setattr(kfm._Core, "update", staticmethod(_fake_update))
    g = golden("kf_saver")
    p = f"n{n}m{m}_"
    kf = kfm.KalmanFilter(dim_x=n, dim_z=m)
    kf.x, kf.P = g[p + "x0"].copy(), g[p + "P0"].copy()
    kf.F, kf.Q, kf.H, kf.R = g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"]
    zl = [zz if k else None for zz, k in zip(g[p + "zs"], g[p + "mask"])]
    T = len(zl) - drop_tail
    # cut the run so that it ends on a measured epoch (drop_tail = 0 may or may not) or on a missing one
    while drop_tail == 3 and T > 1 and zl[T - 1] is not None:
        T -= 1
    zl = zl[:T]
    kf._log_likelihood = kf._likelihood = kf._mahalanobis = 123.0   # stale cache from "before the call"
    kf.batch_filter(zl)
    for k in ("K", "S", "SI", "y"):
        got, ref = np.asarray(getattr(kf, k), dtype=float), g[p + k][T - 1]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-12), (k, T)
    if zl[-1] is None:
        assert kf.z.shape == (m, 1) and all(v is None for v in kf.z.ravel())
    else:
        assert np.array_equal(np.asarray(kf.z, dtype=float), np.asarray(zl[-1], dtype=float))
    assert kf._log_likelihood is None and kf._likelihood is None and kf._mahalanobis is None
    assert np.allclose(kf.log_likelihood, g[p + "log_likelihood"][T - 1], rtol=1e-9, atol=1e-9)
    assert np.allclose(kf.mahalanobis, g[p + "mahalanobis"][T - 1], rtol=1e-9, atol=1e-11)
