"""The case list shared by make_shapes_golden.py (live reference) and tests/test_host_shapes.py (the mirror):
(dim_x, dim_z, ndim of x, (form of z, dim_z), R override, H override)."""
import numpy as np

Z_FORMS = ["scalar", "py_list", "tuple", "list_of_list_row", "list_of_list_col", "arr0d", "arr1d", "arr_row",
           "arr_col", "arr3d", "arr1d_short", "arr1d_long", "arr_row_long"]


def build_z(zspec):
    form, m = zspec
    v = [3.0 + i for i in range(m)]
    return {
        "scalar": 3.0,
        "py_list": list(v),
        "tuple": tuple(v),
        "list_of_list_row": [list(v)],
        "list_of_list_col": [[a] for a in v],
        "arr0d": np.array(3.0),
        "arr1d": np.array(v),
        "arr_row": np.array([v]),
        "arr_col": np.array([v]).T,
        "arr3d": np.array([[v]]),
        "arr1d_short": np.array(v[:-1]),
        "arr1d_long": np.array(v + [9.0]),
        "arr_row_long": np.array([v + [9.0]]),
    }[form]


CASES = [(n, m, xnd, (form, m), r, h)
         for (n, m) in [(1, 1), (2, 1), (3, 1), (3, 2), (4, 2), (3, 3)]
         for xnd in (1, 2)
         for form in Z_FORMS
         for (r, h) in [(None, None), ("scalar", None), ("matrix", None), (None, "matrix"), ("matrix", "matrix")]]
