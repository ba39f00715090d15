"""The resampler's division-free slot boundaries (filterpy_amd/csrc/fk_resample_math.hpp, compiled for the host by
tests/hostcheck/hostcheck_rs.cpp with -ffp-contract=off) against the brute-force count with real IEEE divisions:
n(c) = #{ i : fl(fl(u_i + i) / N) < c }  for the positions of resampling.py:103 / :139."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HC = os.path.join(ROOT, "tests", "hostcheck")


@pytest.fixture(scope="module")
def lib():
    so, src = os.path.join(HC, "libhostcheck_rs.so"), os.path.join(HC, "hostcheck_rs.cpp")
    deps = [src] + [os.path.join(ROOT, "filterpy_amd", "csrc", h) for h in ("fk_resample_math.hpp", "fk_exact_scan.hpp", "fk_math.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-o", so, src])
    return ctypes.CDLL(so)


def _call(lib, name, Np, u, c):
    c = np.ascontiguousarray(c, dtype=np.float64)
    out = np.empty(len(c), dtype=np.int32)
    vp = ctypes.c_void_p
    if np.ndim(u) == 0:
        getattr(lib, name)(ctypes.c_int(Np), ctypes.c_double(float(u)), ctypes.c_long(len(c)), c.ctypes.data_as(vp), out.ctypes.data_as(vp))
    else:
        u = np.ascontiguousarray(u, dtype=np.float64)
        getattr(lib, name)(ctypes.c_int(Np), u.ctypes.data_as(vp), ctypes.c_long(len(c)), c.ctypes.data_as(vp), out.ctypes.data_as(vp))
    return out


def _probe_values(rs, pos, Np, K):
    sel = rs.randint(0, Np, size=min(Np, K))
    x = pos[sel]
    parts = [x, np.nextafter(x, np.inf), np.nextafter(x, -np.inf), (x + pos[np.minimum(sel + 1, Np - 1)]) / 2,
             rs.rand(K), rs.rand(K // 8) * 1e-9, 1 + rs.rand(64), np.array([1.0, 2.0, 1e300, 2.0 ** -500]),
             np.arange(0, Np + 1, max(1, Np // 3000)) / Np, np.nextafter(np.arange(1, Np + 1, max(1, Np // 3000)) / Np, 0)]
    y = x
    for _ in range(4):
        y = np.nextafter(y, np.inf)
        parts.append(y)
    c = np.concatenate(parts)
    return c[c > 0]


@pytest.mark.parametrize("Np", [1, 2, 3, 7, 64, 1000, 8000, 100003, 1 << 20, 8000000, 15625 * 512 + 1])
def test_systematic_boundaries_equal_the_division(lib, Np):
    rs = np.random.RandomState(Np % 9973)
    for u in list(rs.rand(2)) + [0.0, np.nextafter(1.0, 0), 0.5, 2.0 ** -60]:
        pos = (u + np.arange(Np)) / Np                       # resampling.py:139
        c = _probe_values(rs, pos, Np, 8000)
        ref = np.searchsorted(pos, c, side="left")           # number of positions < c
        assert np.array_equal(_call(lib, "hc_n_boundary_sys", Np, u, c), ref)
        assert np.array_equal(_call(lib, "hc_n_boundary_fast_sys", Np, u, c), ref)


@pytest.mark.parametrize("Np", [1, 2, 5, 64, 1000, 8000, 100003, 1 << 20, 8000000])
def test_stratified_boundaries_equal_the_division(lib, Np):
    rs = np.random.RandomState(Np % 9967)
    for _ in range(3):
        us = rs.rand(Np)
        us[rs.randint(0, Np, size=max(1, Np // 50))] = np.nextafter(1.0, 0)
        us[rs.randint(0, Np, size=max(1, Np // 50))] = 0.0
        pos = (us + np.arange(Np)) / Np                      # resampling.py:103
        assert np.all(np.diff(pos) >= 0)
        c = _probe_values(rs, pos, Np, 8000)
        ref = np.searchsorted(pos, c, side="left")
        assert np.array_equal(_call(lib, "hc_n_boundary_strat", Np, us, c), ref)
        assert np.array_equal(_call(lib, "hc_n_boundary_fast_strat", Np, us, c), ref)
