"""Which measurement / R / H shapes KalmanFilter.update accepts, and what shapes x and y have afterwards, must be
the reference's (SURVEY Appendix C `kf_shapes`; filterpy/kalman/tests/test_kf.py:529-575, 699-718).  The table
tests/golden/kf_shapes.json was frozen from the live reference by tests/golden/make_shapes_golden.py; all of this
is host logic in front of the kernel launch, so the launch itself (`_Core.update`) is replaced by a NumPy
stand-in with the same contract and the test runs without a GPU."""
import json
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_shapes_golden import run  # noqa: E402

import filterpy_amd.kalman.kalman_filter as kfm  # noqa: E402


def _fake_update(n, m, N, x, P, z, H, R, mode, mask=None, layout="soa", flags=0, inv=None):
    assert N == 1 and x.shape == (1, n) and P.shape == (1, n, n) and z.shape == (1, m)
    assert H.shape == (m, n) and R.shape == (m, m)
    y = z[0] - H @ x[0]
    S = H @ P[0] @ H.T + R
    SI = np.linalg.inv(S)
    K = P[0] @ H.T @ SI
    A = np.eye(n) - K @ H
    return ((x[0] + K @ y)[None], (A @ P[0] @ A.T + K @ R @ K.T)[None], y[None], K[None], S[None], SI[None])


with open(os.path.join(ROOT, "tests", "golden", "kf_shapes.json")) as _f:
    TABLE = json.load(_f)


def test_table_is_the_case_list():
    from shape_cases import CASES
    assert [tuple(map(lambda v: tuple(v) if isinstance(v, list) else v, t["case"])) for t in TABLE] == \
           [tuple(c) for c in CASES]


def test_update_shape_table(monkeypatch):
    """Without an H override the mirror must do exactly what the reference does (same exception class, same
    shapes of x and y afterwards).  With an explicit ``H=`` the reference skips reshape_z
    (kalman_filter.py:530-533) and lets numpy broadcasting decide, so a z of the wrong orientation turns the
    STATE into a matrix (x (4,) -> (4, 2), (3, 1) -> (3, 3, 2), ...).  The mirror reproduces every case in
    which the reference keeps x's shape; where the reference's x stops being a state vector it either treats
    z as the dim_z values it holds or raises ValueError -- never a broadcast state."""
    monkeypatch.setattr(kfm._Core, "update", staticmethod(_fake_update))
    bad = []
    n_exact = n_deviating = 0
    for t in TABLE:
        case = t["case"]
        case = (case[0], case[1], case[2], tuple(case[3]), case[4], case[5])
        n, xnd, h_override = case[0], case[2], case[5] is not None
        x_shape = [n] if xnd == 1 else [n, 1]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = run(kfm.KalmanFilter, case)
        want = {k: v for k, v in t.items() if k != "case"}
        regular = want["outcome"] != "ok" or want["x_shape"] == x_shape
        if not h_override or regular:
            n_exact += 1
            if got != want:
                bad.append((case, want, got))
        else:
            n_deviating += 1
            if not (got["outcome"] == "ValueError" or (got["outcome"] == "ok" and got["x_shape"] == x_shape)):
                bad.append((case, want, got))
    assert not bad, "%d of %d cases differ, first: %r" % (len(bad), len(TABLE), bad[:5])
    assert n_exact >= 600 and n_deviating <= 180, (n_exact, n_deviating)
