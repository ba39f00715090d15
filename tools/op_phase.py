"""Where does resample_onepass_kernel spend its time?  Builds an INSTRUMENTED copy of the resampling units
(-DFK_OP_CLOCKS -> filterpy_amd/csrc/exp_build/libop_phase.so; the shipped libfilterhip.so carries none of it), runs
systematic resampling and prints thread 0's shader-clock ticks per phase and workgroup:

    python tools/op_phase.py --build            # here (hipcc cross-compiles)
    python tools/op_phase.py --run [--shapes 125x8000000,1000x8000]      # on the GPU box
"""
import argparse
import ctypes
import json
import os
import subprocess

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "filterpy_amd", "csrc")
LIB = os.path.join(CSRC, "exp_build", "libop_phase%s.so")
PHASES = ["ticket", "load+stage", "stage1 | tile boundaries", "increments", "stage2 | exact scan", "general_scan | quick boundaries", "boundaries", "emission"]
COUNTS = {8: "general", 9: "quick", 10: "zeros", 11: "workgroups"}


def build(extra, tag):
    os.makedirs(os.path.join(CSRC, "exp_build"), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-gpu-rdc", "-DFK_OP_CLOCKS"] + extra + ["-o", LIB % tag, os.path.join(CSRC, "resample_onepass.hip"),
           os.path.join(CSRC, "resample_kernels.hip"), os.path.join(CSRC, "resample_whole.hip"), "-x", "hip", os.path.join(CSRC, "fk_host.cpp")]
    subprocess.check_call(cmd, cwd=CSRC)
    print("built", LIB % tag)


def run(shapes, iters, strat, tag):
    import torch
    lib = ctypes.CDLL(LIB % tag)
    dev = torch.device("cuda")
    for shape in shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
        w /= w.sum(dim=1, keepdim=True)
        u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
        lib.fk_resample_workspace_bytes.restype = ctypes.c_size_t
        nb = lib.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np))
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
        p = ctypes.c_void_p
        fn = lib.fk_resample_stratified_f64 if strat else lib.fk_resample_systematic_f64

        def go():
            rc = fn(ctypes.c_int64(Fn), ctypes.c_int64(Np), p(w.data_ptr()), p(u.data_ptr()),
                    p(idx.data_ptr()), p(0), p(ws.data_ptr()), ctypes.c_size_t(nb), p(0))
            assert rc == 0, rc
        out = (ctypes.c_ulonglong * 16)()
        go()
        torch.cuda.synchronize()
        lib.fk_debug_op_phases(out)          # clear
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            go()
        t1.record()
        torch.cuda.synchronize()
        assert lib.fk_debug_op_phases(out) == 0
        v = [int(x) for x in out]
        nwg = float(v[11]) or 1.0
        tot = float(sum(v[1:8])) or 1.0
        print(json.dumps({"tag": tag, "shape": shape, "stratified": strat, "ms_per_call": round(t0.elapsed_time(t1) / iters, 4),
                          "ticks_per_workgroup": {k: round(t / nwg, 1) for k, t in zip(PHASES, v[:8])},
                          "share": {k: round(t / tot, 3) for k, t in zip(PHASES, v[:8])},
                          "total_ticks_per_workgroup": round(tot / nwg, 1),
                          "counts_per_call": {n: v[s] / iters for s, n in COUNTS.items()},
                          "segmented_scan_ticks_per_general_chunk": {n: round(v[s] / max(v[8], 1), 1) for s, n in
                                                                     ((12, "classify+scans"), (13, "lists"), (14, "check+chain"), (15, "sums"))},
                          "dirty_per_general_chunk": round(v[0] / max(v[8], 1), 2),
                          "v2": os.environ.get("FK_OP_V2", "1") != "0", "predicted_chunks_per_call": v[12] / iters,
                          "v2_slow_chunks_per_call": {"no valid guess": v[13] / iters, "look-back missed": v[14] / iters, "end of positions": v[15] / iters},
                          "env": {k: os.environ[k] for k in ("FK_OP_V2", "FK_OP_PRED_BACK", "FK_OP_POLLS", "FK_OP_WAVES", "FK_OP_LB") if k in os.environ}}), flush=True)


WH_PHASES = ["weights+sums", "classify+scans", "lists", "check+chain", "boundaries", "heads", "window scan", "stores", "plain-prefix boundaries"]


def run_whole(shapes, iters, strat, tag):
    """resample_whole_kernel (Np <= 8192): ticks between its seven barriers, thread 0 of every workgroup"""
    import torch
    lib = ctypes.CDLL(LIB % tag)
    dev = torch.device("cuda")
    for shape in shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
        w /= w.sum(dim=1, keepdim=True)
        u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
        p = ctypes.c_void_p
        fn = lib.fk_resample_stratified_f64 if strat else lib.fk_resample_systematic_f64

        def go():
            rc = fn(ctypes.c_int64(Fn), ctypes.c_int64(Np), p(w.data_ptr()), p(u.data_ptr()), p(idx.data_ptr()), p(0), p(0),
                    ctypes.c_size_t(0), p(0))
            assert rc == 0, rc
        out = (ctypes.c_ulonglong * 16)()
        go()
        torch.cuda.synchronize()
        lib.fk_debug_wh_phases(out)          # clear
        for _ in range(iters):
            go()
        torch.cuda.synchronize()
        assert lib.fk_debug_wh_phases(out) == 0
        v = [int(x) for x in out]
        nwg = float(v[11]) or 1.0
        print(json.dumps({"tag": tag, "kernel": "resample_whole_kernel", "shape": shape, "stratified": strat,
                          "ticks_per_workgroup": {k: round(t / nwg, 1) for k, t in zip(WH_PHASES, v[:9])},
                          "total_ticks_per_workgroup": round(sum(v[:9]) / nwg, 1), "exact_rounds": int(v[12]),
                          "dirty_per_exact_round": round(v[10] / max(v[12], 1), 2),
                          "workgroups": int(nwg)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--whole", action="store_true", help="with --run: the phases of resample_whole_kernel (shapes with Np <= 8192)")
    ap.add_argument("--shapes", default="125x8000000,1000x8000,8x8000000")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--stratified", type=int, default=0)
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    if a.build:
        build(["-D" + d for d in a.define], a.tag)
    if a.run and a.whole:
        run_whole(a.shapes, a.iters, a.stratified, a.tag)
    elif a.run:
        run(a.shapes, a.iters, a.stratified, a.tag)
    if not (a.build or a.run):
        ap.print_help()
