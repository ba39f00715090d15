#!/usr/bin/env python3
"""A/B of build-time variants of the one-pass resampling kernel (filterpy_amd/csrc/resample_onepass.hip).

    python tools/exp_rs_variants.py --build  name=SRC:-DMACRO=V,-DMACRO2=V ...     # here (hipcc cross-compiles)
    python tools/exp_rs_variants.py --run    [--shapes ...] [--time-shapes ...]     # on the GPU box

Every variant becomes filterpy_amd/csrc/exp_build/librsv_<name>.so (resampling units only; the shipped libfilterhip.so is
not touched).  SRC is `cur` (the tree's resample_onepass.hip) or a path to another copy of that file.  --run checks every
variant bit for bit against the FIRST one on several weight families (every length forced through the one-pass kernel),
then times them all, interleaved, on the timing shapes."""
import argparse
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "filterpy_amd", "csrc")
OUT = os.path.join(CSRC, "exp_build")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    for old in glob.glob(os.path.join(OUT, "librsv_*.so")):
        os.remove(old)
    procs = []
    for spec in specs:
        name, rest = spec.split("=", 1)
        src, _, defs = rest.partition(":")
        src = os.path.join(CSRC, "resample_onepass.hip") if src == "cur" else os.path.abspath(src)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-gpu-rdc", "-I" + CSRC] + [d for d in defs.split(",") if d] + \
              ["-o", os.path.join(OUT, f"librsv_{name}.so"), src, os.path.join(CSRC, "resample_kernels.hip"),
               os.path.join(CSRC, "resample_whole.hip"), "-x", "hip", os.path.join(CSRC, "fk_host.cpp")]
        procs.append((name, subprocess.Popen(cmd, cwd=CSRC)))
    for name, p in procs:
        assert p.wait() == 0, name
        print("built", name)


def family(kind, Fn, Np, dev, gen):
    import torch
    w = torch.rand((Fn, Np), generator=gen, device=dev, dtype=torch.float64)
    if kind == "heavy_tail":
        w = w ** 12
    elif kind == "zeros":
        w = torch.where(torch.rand((Fn, Np), generator=gen, device=dev) < 0.7, torch.zeros_like(w), w)
    elif kind == "ties":
        w = torch.floor(w * 2 ** 20) * 2.0 ** -40
    w = w / w.sum(dim=1, keepdim=True)
    if kind == "sum_half":
        w = w * 0.5
    return w.contiguous()


class Lib:
    def __init__(self, path):
        self.name = os.path.basename(path)[len("librsv_"):-3]
        self.h = ctypes.CDLL(path)
        self.h.fk_resample_workspace_bytes.restype = ctypes.c_size_t

    def go(self, strat, Fn, Np, w, u, idx, st, ws):
        p = ctypes.c_void_p
        fn = self.h.fk_resample_stratified_f64 if strat else self.h.fk_resample_systematic_f64
        rc = fn(ctypes.c_int64(Fn), ctypes.c_int64(Np), p(w.data_ptr()), p(u.data_ptr()), p(idx.data_ptr()),
                p(st.data_ptr()), p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), p(0))
        assert rc == 0, (self.name, rc)


def run(shapes, time_shapes, iters, path):
    import torch
    if path:
        os.environ["FK_RESAMPLE_PATH"] = path
    libs = [Lib(p) for p in sorted(glob.glob(os.path.join(OUT, "librsv_*.so")))]
    base = [l for l in libs if l.name.startswith("base")] or libs[:1]
    libs = base + [l for l in libs if l not in base]
    dev = torch.device("cuda")
    ok_all = True
    for shape in shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        nb = max(l.h.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np)) for l in libs)
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
        for strat in (0, 1):
            for kind in ("uniform", "heavy_tail", "zeros", "ties", "sum_half"):
                g = torch.Generator(device=dev)
                g.manual_seed(17 + strat)
                w = family(kind, Fn, Np, dev, g)
                u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
                ref, ref_st = None, None
                for l in libs:
                    idx = torch.full((Fn, Np), -7, dtype=torch.int32, device=dev)
                    st = torch.zeros(Fn, dtype=torch.int32, device=dev)
                    l.go(strat, Fn, Np, w, u, idx, st, ws)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref, ref_st = idx, st
                        continue
                    same = bool(torch.equal(idx, ref)) and bool(torch.equal(st, ref_st))
                    ok_all &= same
                    if not same:
                        print(json.dumps({"MISMATCH": l.name, "shape": shape, "stratified": strat, "weights": kind,
                                          "n_diff": int((idx != ref).sum()), "status": sorted(set(st.cpu().tolist()))}), flush=True)
        print(json.dumps({"checked": shape, "ok_so_far": ok_all}), flush=True)
    for shape in time_shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        nb = max(l.h.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np)) for l in libs)
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
        for strat in (0, 1):
            g = torch.Generator(device=dev)
            g.manual_seed(5)
            w = family("uniform", Fn, Np, dev, g)
            u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
            idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
            st = torch.zeros(Fn, dtype=torch.int32, device=dev)
            rec = {"time_shape": shape, "stratified": strat}
            for rep in range(2):                  # two interleaved rounds: box drift shows as a difference between them
                for l in libs:
                    l.go(strat, Fn, Np, w, u, idx, st, ws)
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for _ in range(iters):
                        l.go(strat, Fn, Np, w, u, idx, st, ws)
                    t1.record()
                    torch.cuda.synchronize()
                    rec.setdefault(l.name, []).append(round(t0.elapsed_time(t1) / iters, 4))
            best = {k: min(v) for k, v in rec.items() if isinstance(v, list)}
            rec["frac_of_8TBs"] = {k: round((20.0 if strat else 12.0) * Fn * Np / (v * 1e-3) / 8e12, 4) for k, v in best.items()}
            print(json.dumps(rec), flush=True)
    print("ALL EQUAL" if ok_all else "MISMATCHES")
    return 0 if ok_all else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="*", default=None)
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--shapes", default="3x65536,64x8000,5x2049,16x8000000")
    ap.add_argument("--time-shapes", default="125x8000000,1000x8000")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--path", default="onepass", help="FK_RESAMPLE_PATH for the run ('' = default dispatch)")
    a = ap.parse_args()
    if a.build is not None:
        build(a.build)
    if a.run:
        sys.exit(run(a.shapes, a.time_shapes, a.iters, a.path))
