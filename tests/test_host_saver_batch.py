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
