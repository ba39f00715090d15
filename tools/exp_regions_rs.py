#!/usr/bin/env python3
"""The one-pass resampling kernel (125 x 8e6: 8 GB of weights read, 4 GB of indices written) over where the two arrays sit in
one arena -- the same question tools/exp_regions.py asks of batch_filter's two covariance histories."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    import torch
    from filterpy_amd import _engine as E
    Fn, Np = 125, 8_000_000
    dev = torch.device("cuda")
    SLAB, STEP = 176 << 30, 16 << 30
    slab = torch.empty(SLAB, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    w0 = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
    w0 /= w0.sum(dim=1, keepdim=True)
    u = torch.rand(Fn, generator=g, device=dev, dtype=torch.float64)
    st = torch.zeros(Fn, dtype=torch.int32, device=dev)
    wbytes, ibytes = Fn * Np * 8, Fn * Np * 4
    base = (-slab.data_ptr()) % (2 << 20)

    def timed(ow, oi, reps=3):
        w = slab[base + ow:base + ow + wbytes].view(torch.float64).view(Fn, Np)
        idx = slab[base + oi:base + oi + ibytes].view(torch.int32).view(Fn, Np)
        w.copy_(w0)
        ts = []
        for r in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            E.resample_systematic(Fn, Np, w, u, idx, st)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        return round(float(np.median(ts)), 3)

    offs = list(range(0, SLAB - (16 << 30), STEP))
    for ow in offs:
        row = {}
        for oi in offs:
            if oi != ow:
                row[oi >> 30] = timed(ow, oi)
        print(json.dumps({"w_at_GiB": ow >> 30, "ms_by_idx_at_GiB": row}), flush=True)
    # control: separate allocations
    del slab
    torch.cuda.empty_cache()
    idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        E.resample_systematic(Fn, Np, w0, u, idx, st)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    print(json.dumps({"separate allocations ms": round(float(np.median(ts)), 3)}), flush=True)


if __name__ == "__main__":
    main()
