#!/usr/bin/env python3
"""Map of the headline kernel's time over WHERE in one large slab its two covariance histories sit (tools/exp_buffers.py: the
time is decided by the PAIR of buffers behind covs and covs_p; buffers fall into classes, a pair from one class is slow).
One slab of SLAB_GB is allocated once (its physical backing stays put); covs at offset a, covs_p at offset b, both on a grid
of STEP_GB; z / means / means_p in separate small allocations.  Prints one JSON line per (a, b)."""
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
    SLAB = int(os.environ.get("SLAB_GB", "192")) << 30
    STEP = int(os.environ.get("STEP_GB", "8")) << 30
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
    means = torch.empty((T, N, n), dtype=torch.float64, device=dev)
    means_p = torch.empty((T, N, n), dtype=torch.float64, device=dev)
    slab = torch.empty(SLAB, dtype=torch.uint8, device=dev)
    csize = T * N * n * n * 8
    c = gpu_clocks()
    print(json.dumps({"gpu": {k: c.get(k) for k in ("oam_id", "asic_serial")}, "slab": hex(slab.data_ptr()), "slab_GiB": SLAB >> 30}), flush=True)

    def cview(off):
        return slab[off:off + csize].view(torch.float64).view(T, N, n * n)

    def timed(a, b, reps=3):
        ca, cb = cview(a), cview(b)
        ts = []
        for r in range(reps + 1):
            x.copy_(x0)
            P.copy_(P0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            E.kf_batch_filter(desc, dF, dQ, dH, dR, z, x, P, means=means, covs=ca, means_p=means_p, covs_p=cb, status=st)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        return round(float(np.median(ts)), 3)

    offs = list(range(0, SLAB - csize, STEP))
    need = -(-csize // STEP)                         # grid steps a history covers
    for ia, a in enumerate(offs):
        row = {}
        for ib, b in enumerate(offs):
            if abs(ia - ib) >= need:
                row[b >> 30] = timed(a, b)
        print(json.dumps({"covs_at_GiB": a >> 30, "ms_by_covs_p_at_GiB": row}), flush=True)
    # finer: covs at 0, covs_p swept in 1 GiB steps
    row = {}
    for b in range(need * STEP, SLAB - csize, 1 << 30):
        row[b >> 30] = timed(0, b)
    print(json.dumps({"covs_at_GiB": 0, "fine_ms_by_covs_p_at_GiB": row}), flush=True)


if __name__ == "__main__":
    main()
