#!/usr/bin/env python3
"""One IMM bank shape, the per-step outputs switched on and off (the pointers are run-time choices of imm_lanes / imm_quad): how much of
a launch is the arithmetic and how much the stores and what they make the step wait for.
    python tools/bench_imm_outputs.py --dims 16x8x2 --N 50000 --T 20 [--layout soa]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="16x8x2")
    ap.add_argument("--N", type=int, default=50000)
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--layout", default="soa")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import torch
    from filterpy_amd import _engine as E
    n, m, nm = (int(v) for v in a.dims.split("x"))
    N, T, layout = a.N, a.T, a.layout
    rs = np.random.RandomState(100 * n + 10 * m + nm)
    Fs = np.array([np.eye(n) + 0.03 * (j + 1) * rs.randn(n, n) for j in range(nm)])
    Qs = np.array([0.05 * (j + 1) * np.eye(n) for j in range(nm)])
    Hs = np.array([np.eye(m, n)] * nm)
    Rs = np.array([0.5 * np.eye(m)] * nm)
    M = np.full((nm, nm), 0.05 / (nm - 1)) + (0.95 - 0.05 / (nm - 1)) * np.eye(nm)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    xs0 = torch.zeros((N, nm * n) if layout == "aos" else (nm * n, N), dtype=torch.float64, device=dev)
    Ps0 = (4.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, nm)
    Ps0 = Ps0.contiguous() if layout == "aos" else Ps0.T.contiguous()
    mu0 = torch.full((N, nm) if layout == "aos" else (nm, N), 1.0 / nm, dtype=torch.float64, device=dev)
    xs, Ps, mu = xs0.clone(), Ps0.clone(), mu0.clone()
    sizes = dict(x_out=n, P_out=n * n, mu_out=nm, likelihood_out=nm, x_prior_out=n, P_prior_out=n * n)
    allout = {k: E.alloc_records((T,), N, v, layout) for k, v in sizes.items()}
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    d = [E.dev(x) for x in (Fs, Qs, Hs, Rs, M)]
    for name, keys in (("none", ()), ("mu", ("mu_out",)), ("x,mu", ("x_out", "mu_out")), ("x,P,mu", ("x_out", "P_out", "mu_out")),
                       ("all six", tuple(sizes))):
        out = {k: allout[k] for k in keys}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = []
        for it in range(a.iters + 2):
            xs.copy_(xs0); Ps.copy_(Ps0); mu.copy_(mu0)
            ev[0].record()
            E.imm_batch(n, m, nm, N, T, layout, *d, z, xs, Ps, mu, status=st, **out)
            ev[1].record()
            torch.cuda.synchronize()
            if it >= 2:
                best.append(ev[0].elapsed_time(ev[1]))
        print(json.dumps({"bank": a.dims, "N": N, "T": T, "layout": layout, "outputs": name, "ms_median": sorted(best)[len(best) // 2],
                          "ms_min": min(best)}), flush=True)


if __name__ == "__main__":
    main()
