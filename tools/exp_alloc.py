#!/usr/bin/env python3
"""The headline kernel against HOW its arrays were allocated, inside one process (the same command on one
GPU measured 5.35 and 6.10 ms in two processes; relative placement inside one slab moves it by 1 %, tools/exp_placement.py).
Each trial allocates the five arrays in one of several ways, times the kernel, and runs a translation probe on the largest
output array: a random 8-byte gather over the whole array (one TLB entry per access unless the entries are large), next to a
streaming read of the same array.  Prints one JSON line per trial."""
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
    dev = torch.device("cuda")
    F, Q, H, R = c2_model()
    dF, dQ, dH, dR = (E.dev(M, dev) for M in (F, Q, H, R))
    order = ["z", "means", "covs", "means_p", "covs_p"]
    shapes = dict(z=(T, N, m), means=(T, N, n), covs=(T, N, n * n), means_p=(T, N, n), covs_p=(T, N, n * n))
    nbytes = {k: int(np.prod(s)) * 8 for k, s in shapes.items()}
    x0 = torch.zeros((N, n), dtype=torch.float64, device=dev)
    P0 = (100.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1).contiguous()
    x, P = x0.clone(), P0.clone()
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS["aos"], update_first=0, alpha_sq=1.0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    idx = torch.randint(0, nbytes["covs"] // 8, (1 << 24,), generator=g, device=dev)
    c = gpu_clocks()
    print(json.dumps({"gpu": {k: c.get(k) for k in ("oam_id", "asic_serial", "vbios")}}), flush=True)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=5):
        ts = []
        for r in range(reps + 1):
            a, b = ev(), ev()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    def trial(label, arrays, keep):
        arrays["z"].normal_(generator=g)

        def kernel():
            x.copy_(x0)
            P.copy_(P0)
            E.kf_batch_filter(desc, dF, dQ, dH, dR, arrays["z"], x, P, means=arrays["means"], covs=arrays["covs"],
                              means_p=arrays["means_p"], covs_p=arrays["covs_p"], status=st)
        ms = timed(kernel)
        flat = arrays["covs"].view(-1)
        gather_ms = timed(lambda: flat[idx].sum(), reps=3)
        stream_ms = timed(lambda: flat.sum(), reps=3)
        free, total = torch.cuda.mem_get_info()
        print(json.dumps({"trial": label, "kernel_ms": round(ms, 4), "gather_16M_ms": round(gather_ms, 4),
                          "stream_read_ms": round(stream_ms, 4), "ptrs": {k: hex(arrays[k].data_ptr()) for k in order},
                          "free_GB": round(free / 1e9, 1)}), flush=True)

    def separate(rev=False):
        ks = order[::-1] if rev else order
        return {k: torch.empty(shapes[k], dtype=torch.float64, device=dev) for k in ks}

    def slab(chunks=1):
        tot = sum(nbytes.values()) + 5 * (2 << 20)
        s = torch.empty(tot, dtype=torch.uint8, device=dev)
        out, off = {"_slab": s}, (-s.data_ptr()) % (2 << 20)
        for k in order:
            out[k] = s[off:off + nbytes[k]].view(torch.float64).view(shapes[k])
            off += (nbytes[k] + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        return out

    def release(a):
        a.clear()
        torch.cuda.empty_cache()

    for rep in range(2):
        a = separate()
        trial("separate", a, None)
        release(a)
        a = slab()
        trial("one slab", a, None)
        release(a)
        a = separate(rev=True)
        trial("separate, reversed order", a, None)
        release(a)
    # fragment the free VRAM: many 3 MiB blocks, every other one freed, then allocate
    blocks = [torch.empty(3 << 20, dtype=torch.uint8, device=dev) for _ in range(4000)]
    del blocks[::2]
    torch.cuda.empty_cache()
    a = separate()
    trial("separate, after freeing every other of 4000 x 3 MiB", a, None)
    release(a)
    del blocks
    torch.cuda.empty_cache()
    # hold a large allocation first (what an earlier tenant of the process would do), then allocate
    hold = torch.empty(100 << 30, dtype=torch.uint8, device=dev)
    a = separate()
    trial("separate, behind a 100 GiB allocation", a, None)
    release(a)
    del hold
    torch.cuda.empty_cache()
    a = separate()
    trial("separate, again", a, None)
    release(a)
    # the caching allocator keeps the blocks: same arrays re-used without returning them to the driver
    a = separate()
    trial("separate (kept)", a, None)
    b = separate()
    trial("second set while the first is alive", b, None)


if __name__ == "__main__":
    main()
