"""Time fk_resample_systematic_f64 alone (no gather, no collective) on a few (filters, particles) shapes.
    python tools/bench_resample.py [--shapes 125x8000000,1000x8000] [--iters 10]
Prints one JSON line per shape; `tools/bench_c5.py` times the whole BASELINE configs[4] step instead."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from filterpy_amd import _engine as E  # noqa: E402

if os.environ.get("FK_LIB"):      # an experimental build of the library (A/B of compile-time choices in one lease); tools only
    from filterpy_amd import _abi
    _abi.LIB_PATH = os.path.abspath(os.environ["FK_LIB"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="125x8000000,8x8000000,1000x8000,125x8000")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda")
    for shape in a.shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
        w /= w.sum(dim=1, keepdim=True)
        u = torch.rand(Fn, generator=g, device=dev, dtype=torch.float64)
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
        st = torch.zeros(Fn, dtype=torch.int32, device=dev)
        for _ in range(2):
            E.resample_systematic(Fn, Np, w, u, idx, st)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.iters):
            E.resample_systematic(Fn, Np, w, u, idx, st)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.iters
        # algorithmic bytes: w read once (8 B) + int32 index written (4 B) per particle
        print(json.dumps({"filters": Fn, "particles": Np, "ms": round(ms, 4),
                          "particles_per_s": Fn * Np / ms * 1e3,
                          "frac_hbm": 12.0 * Fn * Np / (ms * 1e-3) / 8e12}))


if __name__ == "__main__":
    main()
