#!/usr/bin/env python3
"""The one-pass resampling path (filterpy_amd/csrc/resample_onepass.hip) against the C oracle (the reference's merge
loop, literally) and against the one-workgroup-per-filter path (resample_kernel), bit for bit, on weight families that exercise every
route of the kernel; then both paths are timed.  GPU box only.

    python tools/exp_onepass.py [--shapes 125x8000000,1000x8000,...] [--iters 10] [--quick]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

FAMILIES = ("uniform", "heavy_tail", "zeros", "leading_zeros", "one_heavy", "ties", "sum_half", "unnormalised",
            "all_zero_filter", "negative", "nan")


def weights(kind, Fn, Np, dev, gen):
    import torch
    w = torch.rand((Fn, Np), generator=gen, device=dev, dtype=torch.float64)
    if kind == "heavy_tail":
        w = w ** 12
    elif kind == "zeros":
        w = torch.where(torch.rand((Fn, Np), generator=gen, device=dev) < 0.7, torch.zeros_like(w), w)
    elif kind == "leading_zeros":
        w[:, : (Np * 3) // 10] = 0.0
    elif kind == "one_heavy":
        w[:, Np // 3] = 1e4
    elif kind == "ties":            # multiples of 2^-40: many exact half-ulp remainders
        w = torch.floor(w * 2 ** 20) * 2.0 ** -40
    w = w / w.sum(dim=1, keepdim=True)
    if kind == "sum_half":          # positions run past cumsum[-1]: IndexError in the reference
        w = w * 0.5
    elif kind == "unnormalised":
        w = w * 1e6
    elif kind == "all_zero_filter":
        w[0] = 0.0
    elif kind == "negative":        # garbage in: the reference's loop still defines the answer
        w[-1, Np // 2] = -0.25 / Np
        if Fn > 1:
            w[0, 0] = -1e-9
    elif kind == "nan":
        w[-1, Np // 5] = float("nan")
    return w.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="3x65536,7x2049,5x1,3x2,4x100,64x8000,8x1000003,1x8000000,16x8000000")
    ap.add_argument("--time-shapes", default="125x8000000,1000x8000,125x8000,8x8000000,1x8000000")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--check-filters", type=int, default=3, help="filters per case held against the C oracle")
    a = ap.parse_args()

    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    dev = torch.device("cuda")
    ok_all = True

    def run(path, strat, Fn, Np, w, u, idx, st):
        # "onepass": every length through resample_onepass.hip; "serial": one workgroup per filter (resample_kernel)
        os.environ.pop("FK_RESAMPLE_PATH", None)
        os.environ.pop("FK_RESAMPLE_SERIAL", None)
        if path == "onepass":
            os.environ["FK_RESAMPLE_PATH"] = "onepass"
        else:
            os.environ["FK_RESAMPLE_SERIAL"] = "1"
        (E.resample_stratified if strat else E.resample_systematic)(Fn, Np, w, u, idx, st)

    for shape in a.shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        for strat in (0, 1):
            for kind in FAMILIES:
                if kind in ("negative", "nan") and Np < 8:
                    continue
                g = torch.Generator(device=dev)
                g.manual_seed(11 + strat)
                w = weights(kind, Fn, Np, dev, g)
                u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
                new = torch.full((Fn, Np), -7, dtype=torch.int32, device=dev)
                st = torch.zeros(Fn, dtype=torch.int32, device=dev)
                run("onepass", strat, Fn, Np, w, u, new, st)
                torch.cuda.synchronize()
                rec = {"shape": shape, "stratified": strat, "weights": kind, "status": sorted(set(st.cpu().tolist()))}
                # the C oracle on a few filters (first, last, middle): indices and the IndexError condition
                good = True
                for f in sorted({0, Fn - 1, Fn // 2})[: a.check_filters]:
                    wf, uf = w[f].cpu().numpy(), (u[f].cpu().numpy() if strat else np.array([float(u[f])]))
                    ref, overrun = (ro.stratified_c if strat else ro.systematic_c)(wf, uf)
                    got = new[f].cpu().numpy()
                    valid = ref < Np                      # slots the reference fills before it raises
                    same = bool(np.array_equal(got[valid], ref[valid])) and (bool(st[f] & 4) == (overrun > 0))
                    if not same:
                        good = False
                        rec.setdefault("mismatch", []).append(
                            {"filter": f, "first_diff": int(np.flatnonzero(got[valid] != ref[valid])[0]) if (got[valid] != ref[valid]).any() else -1,
                             "n_diff": int((got[valid] != ref[valid]).sum()), "status": int(st[f]), "overrun": int(overrun)})
                # the per-filter path on everything (valid inputs only)
                if kind not in ("negative", "nan"):
                    old = torch.full((Fn, Np), -9, dtype=torch.int32, device=dev)
                    st2 = torch.zeros(Fn, dtype=torch.int32, device=dev)
                    run("serial", strat, Fn, Np, w, u, old, st2)
                    torch.cuda.synchronize()
                    okf = (st == 0) & (st2 == 0)
                    same_all = bool(torch.equal(new[okf], old[okf])) and bool(torch.equal(st != 0, st2 != 0))
                    rec["equals_serial_path"] = same_all
                    good &= same_all
                rec["ok"] = good
                ok_all &= good
                print(json.dumps(rec), flush=True)

    for shape in a.time_shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        for strat in (0, 1):
            g = torch.Generator(device=dev)
            g.manual_seed(5)
            w = weights("uniform", Fn, Np, dev, g)
            u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
            idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
            st = torch.zeros(Fn, dtype=torch.int32, device=dev)
            rec = {"time_shape": shape, "stratified": strat}
            for path in ("onepass", "serial"):
                run(path, strat, Fn, Np, w, u, idx, st)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(a.iters):
                    run(path, strat, Fn, Np, w, u, idx, st)
                t1.record()
                torch.cuda.synchronize()
                ms = t0.elapsed_time(t1) / a.iters
                rec[path + "_ms"] = round(ms, 4)
                rec[path + "_frac_of_8TBs"] = round((20.0 if strat else 12.0) * Fn * Np / (ms * 1e-3) / 8e12, 4)
            print(json.dumps(rec), flush=True)
    print("ALL OK" if ok_all else "MISMATCHES")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
