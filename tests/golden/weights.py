"""Seeded synthetic weight vectors shared by the golden generator and the resampling tests."""
import numpy as np


def weights_for(N, seed, kind="rand"):
    rs = np.random.RandomState(seed)
    if kind == "rand":
        w = rs.rand(N)
    elif kind == "onehot":
        w = np.zeros(N)
        w[rs.randint(N)] = 1.0
    elif kind == "sparse":
        w = rs.rand(N) * (rs.rand(N) < 0.05)
        w[0] = 0.0
        if w.sum() == 0:
            w[N // 2] = 1.0
    elif kind == "exp":
        w = np.exp(rs.randn(N) * 4.0)           # heavy-tailed: a few particles dominate
    elif kind == "tiny":
        w = rs.rand(N) * 1e-300                 # unnormalised, near-denormal
        return w
    w /= w.sum()
    return w
