"""CPU dry run of tests/test_gpu_ukf_mlg.py: the same test functions, with the two C-ABI calls they make replaced by the host model
of the kernels (tests/hostcheck/hostcheck_quad.cpp: the kernels' own step functions, lanes as fibers) and the device buffers by
CPU tensors.  What this checks is the TEST file -- shapes, layouts, reference runs, masks, tolerances, status expectations --
before it costs GPU minutes (a probe with a column-major mask did: profiles/r04/lease_q); the kernels' shells it cannot see."""
import ctypes
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, _build

HC = os.path.join(ROOT, "tests", "hostcheck")


def _host_lib():
    so, src = os.path.join(HC, "libhostcheck_quad.so"), os.path.join(HC, "hostcheck_quad.cpp")
    deps = [src] + [os.path.join(ROOT, "filterpy_amd", "csrc", h) for h in ("fk_ukf_quad.hpp", "fk_ukf.hpp", "fk_math.hpp", "fk_math_sym.hpp")]
    _build(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-w", "-o", so, src], so, deps)
    return ctypes.CDLL(so)


def _install_fused(mp, lib):
    """the two fused C-ABI calls of the several-lane UKF, served by the host model of the kernels"""
    from filterpy_amd import _engine as E
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731

    def put(t, arr, layout, lead):
        """arr: lead + (N, rec...) -> into the records tensor t"""
        a = arr.reshape(*arr.shape[:lead + 1], -1)
        t.copy_(torch.as_tensor(a if layout == "aos" else np.swapaxes(a, -1, -2).copy()).reshape(t.shape))

    def fake_batch(n, m, N, T, layout, scale, F, H, Q, R, Wm, Wc, z, x, P, *, mask=None, means=None, covs=None, status=None, paired=None):
        assert paired and 10 <= n <= 16 or 7 <= n <= 9
        zs, x0, P0 = E.from_records(z, layout, 1, (m,)), E.from_records(x, layout, 0, (n,)), E.from_records(P, layout, 0, (n, n))
        mk = None if mask is None else mask.contiguous().numpy()
        assert mk is None or (mk.shape == (T, N) and mk.dtype == np.uint8)
        Fh, Hh, Qh, Rh, wm, wc = (c(v.numpy()) for v in (F, H, Q, R, Wm, Wc))
        mu, cov, xe, Pe, st = np.empty((T, N, n)), np.empty((T, N, n, n)), np.empty((N, n)), np.empty((N, n, n)), np.zeros(N, np.int32)
        entry = "hc_ukf_oct_v4" if (n >= 13 and m >= 5) else "hc_ukf_quad_v4"        # the library's default routing
        for i in range(N):
            xi, Pi, a, b = c(x0[i]).copy(), c(P0[i]).copy(), np.empty((T, n)), np.empty((T, n, n))
            zi, mi = c(zs[:, i]), (None if mk is None else np.ascontiguousarray(mk[:, i]))
            rc = getattr(lib, entry)(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_long(T), p(Fh), p(Hh), p(Qh), p(Rh), p(wm), p(wc),
                                     ctypes.c_double(scale), p(zi), p(mi), p(xi), p(Pi), p(a), p(b))
            st[i] = 16 if rc == -2 else (rc & 0xff)              # -2: the pair table check (FK_STATUS_BAD_WEIGHTS on every track)
            if not (np.all(np.isfinite(xi)) and np.all(np.isfinite(Pi))):
                st[i] |= 2
            mu[:, i], cov[:, i], xe[i], Pe[i] = a, b, xi, Pi
        put(x, xe, layout, 0)
        put(P, Pe, layout, 0)
        if means is not None:
            put(means, mu, layout, 1)
        if covs is not None:
            put(covs, cov, layout, 1)
        if status is not None:
            status.copy_(torch.as_tensor(st))

    def fake_rts(n, N, T, layout, scale, F, Q, Wm, Wc, Xs, Ps, xs, Ps_out, K=None, status=None, paired=None):
        assert paired
        X, Pm = E.from_records(Xs, layout, 1, (n,)), E.from_records(Ps, layout, 1, (n, n))
        Fh, Qh, wm, wc = (c(v.numpy()) for v in (F, Q, Wm, Wc))
        ox, oP, oK, st = np.empty((T, N, n)), np.empty((T, N, n, n)), np.empty((T, N, n, n)), np.zeros(N, np.int32)
        entry = "hc_ukf_oct_rts_v4" if n >= 13 else "hc_ukf_quad_rts_v4"
        for i in range(N):
            a, b, k = np.empty((T, n)), np.empty((T, n, n)), np.empty((T, n, n))
            st[i] = getattr(lib, entry)(ctypes.c_int(n), ctypes.c_long(T), p(Fh), p(Qh), p(wm), p(wc), ctypes.c_double(scale),
                                        p(c(X[:, i])), p(c(Pm[:, i])), p(a), p(b), p(k)) & 0xff
            ox[:, i], oP[:, i], oK[:, i] = a, b, k
        put(xs, ox, layout, 1)
        put(Ps_out, oP, layout, 1)
        if K is not None:
            put(K, oK, layout, 1)
        if status is not None:
            status.copy_(torch.as_tensor(st))

    mp.setattr(E, "ukf_linear_batch", fake_batch)
    mp.setattr(E, "ukf_linear_rts", fake_rts)


@pytest.fixture(scope="module")
def gpu_tests(monkeypatch_module):
    lib = _host_lib()
    from filterpy_amd import _engine as E
    mp = monkeypatch_module
    mp.setattr(E, "require_gpu", lambda: torch.device("cpu"))
    real_dev = E.dev
    mp.setattr(E, "dev", lambda a, device=None: real_dev(a, device).clone())      # (an upload copies; torch.from_numpy on the CPU aliases)
    _install_fused(mp, lib)
    import test_gpu_ukf_mlg
    return importlib.reload(test_gpu_ukf_mlg)


@pytest.fixture(scope="module")
def monkeypatch_module():
    mp = pytest.MonkeyPatch()
    yield mp
    mp.undo()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(10, 1), (12, 3), (13, 5), (16, 4), (16, 8)])
def test_dry_every_instantiation(gpu_tests, n, m, layout):
    gpu_tests.test_every_instantiation_vs_oracle(n, m, layout)


@pytest.mark.parametrize("N", [1, 2, 17, 65])
def test_dry_bank_sizes(gpu_tests, N):
    gpu_tests.test_bank_sizes(N, "soa")
    gpu_tests.test_bank_sizes(N, "aos")


def test_dry_status_bits_and_small_alpha(gpu_tests):
    gpu_tests.test_status_bits()
    gpu_tests.test_small_alpha()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n", [10, 13, 16])
def test_dry_smoother(gpu_tests, n, layout):
    gpu_tests.test_smoother_every_dim_vs_oracle(n, layout)


def test_dry_smoother_bank_sizes(gpu_tests):
    for N in (1, 17):
        gpu_tests.test_smoother_bank_sizes_and_without_gain(N, "soa")
        gpu_tests.test_smoother_bank_sizes_and_without_gain(N, "aos")


@pytest.mark.parametrize("n,m", [(7, 1), (8, 4), (9, 3), (9, 4)])
def test_dry_small_dims(gpu_tests, n, m):
    """(the FK_UKF_MLG_MIN_NX=7 A/B: dim_x 7..9 on the four-lane kernels)"""
    gpu_tests._small_dims_filter_vs_oracle(n, m, "soa")
    gpu_tests._small_dims_filter_vs_oracle(n, m, "aos")
    if m == 4 or n == 7:
        gpu_tests.test_small_dims_smoother_vs_oracle(n, "aos")


@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_dry_python_api(monkeypatch, layout):
    """UnscentedKalmanFilter.batch_filter / rts_smoother with matrix models at dim_x >= 10 and the switch on: the class's routing,
    marshalling and last-epoch replay around the fused calls (the building blocks served by tests/fake_ut_engine.py's stand-ins),
    against the live-reference goldens."""
    import fake_ut_engine
    fake_ut_engine.install(monkeypatch)
    _install_fused(monkeypatch, _host_lib())
    from filterpy_amd import _engine as E

    def fake_linear_map(n_in, n_out, k, N, lay, M, sig_in, sig_out):      # out[i] = M in[i] for every sigma point of every track
        a = fake_ut_engine._get(sig_in, lay, (k, n_in))
        fake_ut_engine._put(sig_out, lay, a @ M.numpy().T)
    monkeypatch.setattr(E, "ut_linear_map", fake_linear_map)
    import test_gpu_ukf_mlg
    mod = importlib.reload(test_gpu_ukf_mlg)
    mod.test_python_api_routes_matrix_models_here(layout)
    mod.test_python_api_smoother_routes_here(layout)
