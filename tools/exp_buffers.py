#!/usr/bin/env python3
"""Which array's backing decides the headline kernel's time, and does a simple probe of a buffer predict it?
(tools/exp_alloc.py: the same kernel on re-allocated arrays at the SAME virtual addresses took 5.40 .. 6.95 ms.)
Candidates: 8 buffers of the covariance size, 8 of the mean size.  One array at a time is moved over the candidates;
each candidate is probed alone: streaming write (fill_), streaming read (sum), random 8-byte gather."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    import torch
    from filterpy_amd import _engine as E
    from bench import c2_model, gpu_clocks
    N, T, n, m = 1_000_000, 100, 4, 2
    K = int(os.environ.get("CANDIDATES", "8"))
    dev = torch.device("cuda")
    F, Q, H, R = c2_model()
    dF, dQ, dH, dR = (E.dev(M, dev) for M in (F, Q, H, R))
    x0 = torch.zeros((N, n), dtype=torch.float64, device=dev)
    P0 = (100.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1).contiguous()
    x, P = x0.clone(), P0.clone()
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS["aos"], update_first=0, alpha_sq=1.0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    z = torch.randn((T, N, m), generator=g, device=dev, dtype=torch.float64)
    C = [torch.empty((T, N, n * n), dtype=torch.float64, device=dev) for _ in range(K)]
    M = [torch.empty((T, N, n), dtype=torch.float64, device=dev) for _ in range(K)]
    c = gpu_clocks()
    print(json.dumps({"gpu": {k: c.get(k) for k in ("oam_id", "asic_serial")}, "C": [hex(t.data_ptr()) for t in C],
                      "M": [hex(t.data_ptr()) for t in M]}), flush=True)

    def timed(fn, reps=5):
        ts = []
        for r in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        return round(float(np.median(ts)), 4)

    def kernel(me, co, mp, cp):
        def run():
            x.copy_(x0)
            P.copy_(P0)
            E.kf_batch_filter(desc, dF, dQ, dH, dR, z, x, P, means=M[me], covs=C[co], means_p=M[mp], covs_p=C[cp], status=st)
        return timed(run)

    idxC = torch.randint(0, C[0].numel(), (1 << 24,), generator=g, device=dev)
    idxM = torch.randint(0, M[0].numel(), (1 << 24,), generator=g, device=dev)
    for name, bufs, idx in (("C", C, idxC), ("M", M, idxM)):
        for i, b in enumerate(bufs):
            flat = b.view(-1)
            print(json.dumps({"probe": f"{name}{i}", "fill_ms": timed(lambda: flat.fill_(1.0), 3), "sum_ms": timed(lambda: flat.sum(), 3),
                              "gather_ms": timed(lambda: flat[idx].sum(), 3)}), flush=True)
    print(json.dumps({"kernel": "base M0 C0 M1 C1", "ms": kernel(0, 0, 1, 1)}), flush=True)
    for i in range(K):
        print(json.dumps({"kernel": f"covs = C{i} (covs_p = C{(i + 1) % K})", "ms": kernel(0, i, 1, (i + 1) % K)}), flush=True)
    for i in range(2, K):
        print(json.dumps({"kernel": f"covs = C{i}, covs_p = C1", "ms": kernel(0, i, 1, 1)}), flush=True)
    for i in range(2, K):
        print(json.dumps({"kernel": f"covs_p = C{i}, covs = C0", "ms": kernel(0, 0, 1, i)}), flush=True)
    for i in range(2, K):
        print(json.dumps({"kernel": f"means = M{i}", "ms": kernel(i, 0, 1, 1)}), flush=True)
    for i in range(2, K):
        print(json.dumps({"kernel": f"means_p = M{i}", "ms": kernel(0, 0, i, 1)}), flush=True)
    # only ONE output array live: the others NULL is not a call the specialised kernel takes; instead all four outputs on the
    # same candidate pair, to see a pair's own speed
    for i in range(K):
        print(json.dumps({"kernel": f"all on pair {i}: M{i} C{i} M{(i + 1) % K} C{(i + 1) % K}", "ms": kernel(i, i, (i + 1) % K, (i + 1) % K)}), flush=True)
    print(json.dumps({"kernel": "base again", "ms": kernel(0, 0, 1, 1)}), flush=True)


if __name__ == "__main__":
    main()
