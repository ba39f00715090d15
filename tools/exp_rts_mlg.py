#!/usr/bin/env python3
"""Time (and give rocprofv3 something to count on) the four-lane smoother at a few dims: RTS_DIMS=14,16 RTS_N=60000."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    from filterpy_amd import _engine as E
    from tools.bench_configs import timeit
    N, T = int(os.environ.get("RTS_N", 60000)), int(os.environ.get("RTS_T", 50))
    layout = os.environ.get("RTS_LAYOUT", "soa")
    dev = torch.device("cuda")
    for n in [int(v) for v in os.environ.get("RTS_DIMS", "14,16").split(",")]:
        rs = np.random.RandomState(n)
        F = np.eye(n) + 0.05 * rs.randn(n, n)
        Q = 0.05 * np.eye(n)
        g = torch.Generator(device=dev)
        g.manual_seed(4)
        Xs = torch.randn((T, n, N) if layout == "soa" else (T, N, n), generator=g, device=dev, dtype=torch.float64)
        A = rs.randn(n, n)
        P1 = A @ A.T / n + np.eye(n)
        Ps = E.to_records(np.tile(P1, (T, 8, 1, 1)), layout, 1)
        Ps = (Ps.repeat(1, 1, N // 8) if layout == "soa" else Ps.reshape(T, 8, n * n).repeat(1, N // 8, 1)).contiguous()
        Nn = Ps.shape[-1] if layout == "soa" else Ps.shape[1]
        Xs = Xs[..., :Nn].contiguous() if layout == "soa" else Xs[:, :Nn].contiguous()
        o = [E.alloc_records((T,), Nn, n, layout)] + [E.alloc_records((T,), Nn, n * n, layout) for _ in range(3)]
        st = torch.zeros(Nn, dtype=torch.int32, device=dev)
        desc = dict(n=n, m=1, nu=0, model_mode=0, N=Nn, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
        dF, dQ = E.dev(F), E.dev(Q)
        ms = timeit(lambda: E.kf_rts(desc, dF, dQ, Xs, Ps, o[0], o[1], o[2], o[3], convention=0, status=st), warm=1, reps=3)
        assert not st.any()
        print(json.dumps(dict(kernel=f"rts n={n} {layout}", N=Nn, T=T, ms=ms, us_per_step_per_round=ms * 1e3 / T / max(1, -(-Nn // 16 // 1024)),
                              frac=Nn * T * 8 * (2 * n + 4 * n * n) / (ms * 1e-3) / 8e12)), flush=True)


if __name__ == "__main__":
    main()
