"""Freeze the reference's accept / reject behaviour for measurement and model shapes (SURVEY Appendix C `kf_shapes`;
filterpy/kalman/tests/test_kf.py:529-575, 699-718) into tests/golden/kf_shapes.json.

    PYTHONPATH=/root/reference python tests/golden/make_shapes_golden.py

For every case: construct KalmanFilter(dim_x, dim_z) with x as a vector or a column, call update(z) (optionally
with R / H overrides) and record "ok" + the shapes of x and y afterwards, or the exception class."""
import json
import os
import warnings

import numpy as np

from shape_cases import CASES, build_z  # noqa: E402  (same directory)


def run(KalmanFilter, case):
    n, m, xnd, zspec, rspec, hspec = case
    kf = KalmanFilter(dim_x=n, dim_z=m)
    if xnd == 1:
        kf.x = np.zeros(n)
    kw = {}
    if rspec == "scalar":
        kw["R"] = 2.5
    elif rspec == "matrix":
        kw["R"] = np.eye(m) * 2.5
    if hspec == "matrix":
        kw["H"] = np.ones((m, n))
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kf.update(build_z(zspec), **kw)
        return {"outcome": "ok", "x_shape": list(np.shape(kf.x)), "y_shape": list(np.shape(kf.y))}
    except Exception as e:  # noqa: BLE001  (the table records which exception the reference raises)
        return {"outcome": type(e).__name__}


if __name__ == "__main__":
    from filterpy.kalman import KalmanFilter
    out = [{"case": list(c), **run(KalmanFilter, c)} for c in CASES]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kf_shapes.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(len(out), "cases ->", path, {k: sum(1 for o in out if o["outcome"] == k) for k in {o["outcome"] for o in out}})
