"""Freeze what filterpy.common.Saver (filterpy/common/helpers.py:27-219) records for a small scripted object into
tests/golden/saver_toy.json: key order, lengths, and the type / shape of every attribute after to_array() and
after to_array(flatten=True), for each combination of constructor options.

    PYTHONPATH=/root/reference python tests/golden/make_saver_golden.py"""
import json
import os

import numpy as np

OPTIONS = [dict(), dict(save_current=True), dict(skip_private=True), dict(skip_callable=True),
           dict(ignore=("P", "twice")), dict(skip_private=True, skip_callable=True, ignore=("x",)),
           dict(ragged=True)]


class Toy:
    def __init__(self):
        self.x = np.zeros((3, 1))
        self.P = np.eye(3)
        self._priv = 1.0
        self.fn = np.sin
        self.s = 0.0
        self.v = np.zeros((1, 1))
        self.w = np.zeros(2)
        self.r = np.zeros(1)

    @property
    def energy(self):
        return float((self.x.T @ self.x).item())

    @property
    def twice(self):
        return 2 * self.s


def describe(v):
    return ["ndarray", list(v.shape)] if isinstance(v, np.ndarray) else [type(v).__name__, len(v)]


def record(Saver, opt):
    opt = dict(opt)
    ragged = opt.pop("ragged", False)
    t = Toy()
    s = Saver(t, **opt)
    for k in range(5):
        t.x = t.x + k
        t.P = t.P * 1.1
        t.s += 1
        t.v = t.v + 1
        t.w = t.w + k
        if ragged:
            t.r = np.zeros(k + 1)          # changes shape every epoch: to_array must refuse
        s.save()
    out = {"keys": list(s.keys), "len": len(s), "x_last": np.asarray(s["x"][-1]).ravel().tolist() if "x" in s.keys else None}
    for flat in (False, True):
        try:
            s.to_array(flatten=flat)
            err = None
        except ValueError as e:
            err = str(e)
        out["flat" if flat else "array"] = {"error": err, "attrs": {k: describe(getattr(s, k)) for k in s.keys}}
    out["repr_keys"] = repr(s).split("\n")[1]
    return out


if __name__ == "__main__":
    from filterpy.common import Saver
    res = [{"options": {k: (list(v) if isinstance(v, tuple) else v) for k, v in o.items()}, **record(Saver, o)} for o in OPTIONS]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "saver_toy.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(len(res), "option sets ->", path)
