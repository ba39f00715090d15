"""EXPERIMENTAL (written at the end of round 1 without GPU time left; never run on a GPU yet).
Builds filterpy_amd/csrc/experimental/resample_lean2.hip into build/libfk_exp.so and compares it with the shipped
resampler: indices must be bit-identical, then both are timed.

    python tools/exp_lean2.py --build                 # here (hipcc cross-compiles)
    python tools/exp_lean2.py --run                   # on the GPU box
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "filterpy_amd", "csrc")
LIB = os.path.join(CSRC, "exp_build", "libfk_exp.so")


def build():
    os.makedirs(os.path.join(CSRC, "exp_build"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-o", LIB, os.path.join(CSRC, "experimental", "resample_lean2.hip"),
                           "-x", "hip", os.path.join(CSRC, "fk_host.cpp")], cwd=CSRC)
    print("built", LIB)


def weights(kind, Fn, Np, dev, gen):
    import torch
    w = torch.rand((Fn, Np), generator=gen, device=dev, dtype=torch.float64)
    if kind == "heavy_tail":
        w = w ** 12
    elif kind == "zeros":
        w = torch.where(torch.rand((Fn, Np), generator=gen, device=dev) < 0.7, torch.zeros_like(w), w)
    elif kind == "one_heavy":
        w[:, Np // 3] = 1e4
    elif kind == "ties":            # multiples of 2^-40: many exact half-ulp remainders
        w = torch.floor(w * 2 ** 20) * 2.0 ** -40
    return w / w.sum(dim=1, keepdim=True)


def run(shapes, iters):
    import torch
    from filterpy_amd import _engine as E
    lib = ctypes.CDLL(LIB)
    lib.fk_resample_workspace_bytes.restype = ctypes.c_size_t
    p = ctypes.c_void_p
    dev = torch.device("cuda")
    ok_all = True
    for shape in shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        for strat in (0, 1):
            for kind in ("uniform", "heavy_tail", "zeros", "one_heavy", "ties"):
                g = torch.Generator(device=dev)
                g.manual_seed(7)
                w = weights(kind, Fn, Np, dev, g)
                u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
                ref = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
                new = torch.full((Fn, Np), -1, dtype=torch.int32, device=dev)
                st = torch.zeros(Fn, dtype=torch.int32, device=dev)
                shipped = E.resample_stratified if strat else E.resample_systematic
                shipped(Fn, Np, w, u, ref, st)
                nb = lib.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np))
                ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)

                def exp():
                    rc = lib.fk_exp_resample_lean2_f64(ctypes.c_int32(strat), ctypes.c_int64(Fn), ctypes.c_int64(Np),
                                                       p(w.data_ptr()), p(u.data_ptr()), p(new.data_ptr()), p(0),
                                                       p(ws.data_ptr()), ctypes.c_size_t(nb), p(0))
                    assert rc == 0, rc
                exp()
                torch.cuda.synchronize()
                same = bool(torch.equal(ref, new))
                ok_all &= same
                rec = {"shape": shape, "stratified": strat, "weights": kind, "bit_identical": same}
                if not same:
                    rec["first_diff"] = int((ref != new).flatten().nonzero()[0])
                if kind == "uniform":
                    for name, fn in (("shipped_ms", lambda: shipped(Fn, Np, w, u, ref, st)), ("lean2_ms", exp)):
                        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        fn()
                        t0.record()
                        for _ in range(iters):
                            fn()
                        t1.record()
                        torch.cuda.synchronize()
                        rec[name] = round(t0.elapsed_time(t1) / iters, 4)
                print(json.dumps(rec))
    print("ALL BIT-IDENTICAL" if ok_all else "MISMATCHES")
    return 0 if ok_all else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--shapes", default="125x8000000,8x1000003,3x65536")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    if a.build:
        build()
    rc = run(a.shapes, a.iters) if a.run else 0
    if not (a.build or a.run):
        ap.print_help()
        rc = 2
    sys.exit(rc)
