"""Which of the three is off at dim_x >= 7, NumPy order: kf_fast with extras, the generic kernel with extras, or neither
(compared with kf_fast without extras and with the oracle on a few tracks)."""
import os
import sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_kf import _run_ex, _per_track
from gpu_util import run_kf_batch
from oracle import kf_oracle


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


for (n, m) in ((6, 1), (7, 1), (7, 2), (8, 4), (9, 1), (9, 3)):
    for layout in ("soa", "aos"):
        rs = np.random.RandomState(100 * n + m)
        N, T = 333, 12
        A = rs.randn(N, n, n)
        x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
        zs = rs.randn(T, N, m) * 2
        F = np.eye(n) + 0.05 * rs.randn(n, n)
        B = rs.randn(n, n)
        Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
        H = rs.randn(m, n)
        R = 0.5 * np.eye(m)
        mask = rs.rand(T, N) > 0.25
        mask[0] = True
        kw = {"mask": mask} if os.environ.get("WITH_MASK") else {}
        fast, hf = _run_ex(x0, P0, zs, F, Q, H, R, layout, **kw)
        os.environ["FK_NO_FAST_EX"] = "1"
        gen, hg = _run_ex(x0, P0, zs, F, Q, H, R, layout, **kw)
        del os.environ["FK_NO_FAST_EX"]
        plain = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, **kw)
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=[0, 64, 300], **kw)
        print((n, m), layout, "means: fast_ex vs plain", rel(fast[0], plain[0]), "gen_ex vs plain", rel(gen[0], plain[0]),
              "plain vs oracle", rel(plain[0][:, [0, 64, 300]], ref[0]),
              "| K: fast vs gen", rel(hf["K"], hg["K"]), "| covs fast vs plain", rel(fast[1], plain[1]), "gen vs plain", rel(gen[1], plain[1]), flush=True)
        bad = np.argwhere(np.abs(gen[0] - plain[0]) > 1e-8 * np.max(np.abs(plain[0])))
        if len(bad):
            print("   gen_ex differs at (t, track, comp):", bad[:6].tolist(), "count", len(bad))
        bad = np.argwhere(np.abs(fast[0] - plain[0]) > 1e-8 * np.max(np.abs(plain[0])))
        if len(bad):
            print("   fast_ex differs at (t, track, comp):", bad[:6].tolist(), "count", len(bad))
