"""Where does resample_chunk_lean_kernel spend its time?  Builds an INSTRUMENTED copy of the resampling unit
(-DFK_RS_PHASE_CLOCKS -> filterpy_amd/csrc/build/librs_phase.so; the shipped libfilterhip.so carries none
of it), runs systematic resampling and prints the share of wave-0 clock ticks per phase:

    python tools/rs_phase.py --build            # here (hipcc cross-compiles)
    python tools/rs_phase.py --run [--shape 125x8000000]      # on the GPU box

phases (slot): issue_loads (5) kernel entry -> tile loads issued, wait_loads (6) -> loads landed and written
to LDS, barrier (0) -> the workgroup's other waves arrived, count_lo (1) count_below(carry-in), cumsum (2)
exact tile cumsum, count_hi (3) count_below(carry-out), outputs (4) division + tile search + store."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "filterpy_amd", "csrc")
LIB = os.path.join(CSRC, "build", "librs_phase.so")
PHASES = ["barrier", "count_lo", "cumsum", "count_hi", "outputs", "issue_loads", "wait_loads"]


def build():
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-DFK_RS_PHASE_CLOCKS", "-o", LIB, os.path.join(CSRC, "resample_kernels.hip"), "-x", "hip",
           os.path.join(CSRC, "fk_host.cpp")]
    subprocess.check_call(cmd, cwd=CSRC)
    print("built", LIB)


def run(shape, iters):
    import torch
    lib = ctypes.CDLL(LIB)
    Fn, Np = (int(v) for v in shape.split("x"))
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
    w /= w.sum(dim=1, keepdim=True)
    u = torch.rand(Fn, generator=g, device=dev, dtype=torch.float64)
    idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
    lib.fk_resample_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np))
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
    p = ctypes.c_void_p

    def go():
        rc = lib.fk_resample_systematic_f64(ctypes.c_int64(Fn), ctypes.c_int64(Np), p(w.data_ptr()), p(u.data_ptr()),
                                            p(idx.data_ptr()), p(0), p(ws.data_ptr()), ctypes.c_size_t(nb), p(0))
        assert rc == 0, rc
    out = (ctypes.c_ulonglong * 8)()
    go()
    torch.cuda.synchronize()
    lib.fk_debug_rs_phases(out)          # clear
    for _ in range(iters):
        go()
    torch.cuda.synchronize()
    assert lib.fk_debug_rs_phases(out) == 0
    ticks = [int(v) for v in out[:len(PHASES)]]
    tot = float(sum(ticks)) or 1.0
    nblocks = iters * Fn * ((Np + 2047) // 2048)
    print(json.dumps({"shape": shape, "share": {k: round(t / tot, 3) for k, t in zip(PHASES, ticks)},
                      "ticks_per_workgroup": {k: round(t / nblocks, 1) for k, t in zip(PHASES, ticks)}}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--shape", default="125x8000000")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    if a.build:
        build()
    if a.run:
        run(a.shape, a.iters)
    if not (a.build or a.run):
        ap.print_help()
        sys.exit(2)
