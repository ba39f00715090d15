"""filterpy_amd.common.Saver records what filterpy.common.Saver records (filterpy/common/helpers.py:27-219): key
order, history lengths, what to_array() / flatten() produce and when to_array() refuses -- against the table
tests/golden/saver_toy.json frozen from the live reference by tests/golden/make_saver_golden.py -- and, watching
the KalmanFilter mirror, the reference's set of public attributes."""
import json
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_saver_golden import OPTIONS, record  # noqa: E402

from filterpy_amd.common import Saver  # noqa: E402


def test_saver_matches_the_reference_table():
    with open(os.path.join(ROOT, "tests", "golden", "saver_toy.json")) as f:
        table = json.load(f)
    assert len(table) == len(OPTIONS)
    for want, opt in zip(table, OPTIONS):
        got = json.loads(json.dumps(record(Saver, opt)))        # tuples -> lists like the stored table
        want = {k: v for k, v in want.items() if k != "options"}
        assert got == want, (opt, got, want)


def test_saver_on_the_kalman_filter_mirror():
    from filterpy_amd.kalman import KalmanFilter
    kf = KalmanFilter(dim_x=4, dim_z=2)
    s = Saver(kf, skip_private=True)
    for k in range(3):
        kf.x = kf.x + 1.0          # attribute edits only: no kernel launch in a CPU test
        s.save()
    assert len(s) == 3
    # the reference's public attributes and properties (kalman_filter.py:399-435, 1203-1257)
    for name in ("x", "P", "Q", "B", "F", "H", "R", "K", "y", "S", "SI", "z", "x_prior", "P_prior", "x_post", "P_post",
                 "dim_x", "dim_z", "dim_u", "inv", "alpha", "likelihood", "log_likelihood", "mahalanobis"):
        assert name in s.keys, name
        assert len(s[name]) == 3
    assert not [k for k in s.keys if k.startswith("_")]
    s.to_array(flatten=True)
    assert s.x.shape == (3, 4) and s.P.shape == (3, 4, 4)
    assert np.array_equal(s.x[:, 0], [1.0, 2.0, 3.0])
