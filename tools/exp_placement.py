#!/usr/bin/env python3
"""Does the headline kernel's time depend on WHERE its five arrays sit relative to each other?  (the same
command on the same GPU measured 5.35 and 6.10 ms in two processes of one lease.)

One slab is allocated once; z / means / covs / means_p / covs_p are views into it at chosen byte offsets, so that inside this
process the physical backing does not change and only the relative placement does.  Prints one JSON line per placement."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    import torch
    from filterpy_amd import _engine as E
    from bench import c2_model
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=1_000_000)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--random", type=int, default=24)
    a = ap.parse_args()
    N, T, n, m = a.tracks, a.T, 4, 2
    dev = torch.device("cuda")
    F, Q, H, R = c2_model()
    dF, dQ, dH, dR = (E.dev(M, dev) for M in (F, Q, H, R))
    sizes = dict(z=T * N * m * 8, means=T * N * n * 8, covs=T * N * n * n * 8, means_p=T * N * n * 8, covs_p=T * N * n * n * 8)
    MiB = 1 << 20
    slack = 1024 * MiB
    order = ["z", "means", "covs", "means_p", "covs_p"]
    home, off = {}, 0
    for k in order:
        home[k] = off
        off += (sizes[k] + slack + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
    slab = torch.empty(off, dtype=torch.uint8, device=dev)
    pad = (-slab.data_ptr()) % (2 * MiB)
    print(json.dumps({"slab_bytes": off, "slab_ptr_mod_2MiB": slab.data_ptr() % (2 * MiB), "homes": home}), flush=True)
    shapes = dict(z=(T, N, m), means=(T, N, n), covs=(T, N, n * n), means_p=(T, N, n), covs_p=(T, N, n * n))
    x0 = torch.zeros((N, n), dtype=torch.float64, device=dev)
    P0 = (100.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1).contiguous()
    x, P = x0.clone(), P0.clone()
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS["aos"], update_first=0, alpha_sq=1.0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    zsrc = torch.randn((T, N, m), generator=g, device=dev, dtype=torch.float64)

    def view(k, delta):
        o = pad + home[k] + delta
        return slab[o:o + sizes[k]].view(torch.float64).view(shapes[k])

    def measure(deltas, label):
        v = {k: view(k, deltas.get(k, 0)) for k in order}
        v["z"].copy_(zsrc)
        ts = []
        for r in range(a.reps + 1):
            x.copy_(x0)
            P.copy_(P0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            E.kf_batch_filter(desc, dF, dQ, dH, dR, v["z"], x, P, means=v["means"], covs=v["covs"], means_p=v["means_p"],
                              covs_p=v["covs_p"], status=st)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        print(json.dumps({"label": label, "deltas": deltas, "ms": round(float(np.median(ts)), 4), "min": round(min(ts), 4),
                          "max": round(max(ts), 4)}), flush=True)

    for i in range(3):
        measure({}, "home")
    pows = [256 << k for k in range(0, 23)]                     # 256 B .. 1 GiB
    for k in ("covs_p", "means_p", "covs", "z"):
        for d in pows:
            if d < slack:
                measure({k: d}, f"{k} + 2^k")
    for d in (3 * 256, 5 * 4096, 7 * 65536, 3 * MiB, 5 * MiB, 37 * MiB, 101 * MiB, 333 * MiB, 777 * MiB):
        measure({"covs_p": d}, "covs_p + odd")
    rs = np.random.RandomState(0)
    for i in range(a.random):
        measure({k: int(rs.randint(0, slack // 256)) * 256 for k in order}, "random")
    measure({}, "home")
    # control: separately allocated arrays, as bench.py does
    del slab
    torch.cuda.empty_cache()
    sep = {k: torch.empty(shapes[k], dtype=torch.float64, device=dev) for k in order}
    sep["z"].copy_(zsrc)
    ts = []
    for r in range(a.reps + 1):
        x.copy_(x0)
        P.copy_(P0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        E.kf_batch_filter(desc, dF, dQ, dH, dR, sep["z"], x, P, means=sep["means"], covs=sep["covs"], means_p=sep["means_p"],
                          covs_p=sep["covs_p"], status=st)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    print(json.dumps({"label": "separate allocations", "ptrs_mod_1GiB": {k: sep[k].data_ptr() % (1 << 30) for k in order},
                      "ptrs": {k: hex(sep[k].data_ptr()) for k in order}, "ms": round(float(np.median(ts)), 4)}), flush=True)


if __name__ == "__main__":
    main()
