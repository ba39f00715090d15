"""pytest configuration: registers the `gpu` marker, puts the repo root on sys.path and
builds the test-only helper libraries (oracle C restatement, host math harness)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _build(cmd, target, sources):
    if os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(s) for s in sources):
        return
    subprocess.check_call(cmd)


@pytest.fixture(scope="session", autouse=True)
def _helpers_built():
    _build(["make", "-s", "-C", os.path.join(ROOT, "oracle")],
           os.path.join(ROOT, "oracle", "_build", "libresample_oracle.so"),
           [os.path.join(ROOT, "oracle", "resample_oracle.c")])
    hc = os.path.join(ROOT, "tests", "hostcheck")
    _build(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-w",
            "-o", os.path.join(hc, "libhostcheck.so"), os.path.join(hc, "hostcheck.cpp")],
           os.path.join(hc, "libhostcheck.so"),
           [os.path.join(hc, "hostcheck.cpp")] + [os.path.join(ROOT, "filterpy_amd", "csrc", h) for h in
                                                 ("fk_math.hpp", "fk_math_sym.hpp", "fk_imm.hpp", "fk_exact_scan.hpp", "fk_ukf.hpp")])


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    """normwise relative error per array: max|a-b| / max(|b|)  (SURVEY §7 hard part 4)."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = np.max(np.abs(b)) if b.size else 1.0
    return float(np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0)) if b.size else 0.0


def rel_err_rows(a, b):
    """worst normwise relative error over the leading axis (one matrix/vector per step)."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    scale = np.max(np.abs(b2), axis=1)
    scale[scale == 0] = 1.0
    return float(np.max(np.max(np.abs(a2 - b2), axis=1) / scale))
