"""Per-chunk timeline of the one-pass resampling kernel (instrumented build of tools/op_phase.py --build): every chunk's thread 0
stamps the 100 MHz wall clock on its way through the kernel; the stamps of one call go to an .npy (chunks x 16) for
tools/op_timeline.py --show.  What a call with FEW LONG vectors waits for is a chain through the chunks that cross a binade;
the phase totals of op_phase.py cannot show a chain, this does.

    python tools/op_timeline.py --run --shapes 1x2000000,1x8000000 --out gpurun_out/tl      # on the GPU box
    python tools/op_timeline.py --show gpurun_out/tl/tl_1x2000000.npy
slots: 0 start, 1 sums done, 3 look-back done (fast), 4 fast path stored; op_chunk_slow: 5 entry, 6 approximate carry-in, 7 increments,
8 exact carry-in, 9 carry-out published (general scan), 10 boundaries done; 11 = 100 + quick; 12 flags | g << 8; 13 verdict
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LIB = os.path.join(ROOT, "filterpy_amd", "csrc", "exp_build", "libop_phase.so")


def run(shapes, out):
    import torch
    lib = ctypes.CDLL(LIB)
    dev = torch.device("cuda")
    os.makedirs(out, exist_ok=True)
    for shape in shapes.split(","):
        Fn, Np = (int(v) for v in shape.split("x"))
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
        w /= w.sum(dim=1, keepdim=True)
        u = torch.rand((Fn,), generator=g, device=dev, dtype=torch.float64)
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
        lib.fk_resample_workspace_bytes.restype = ctypes.c_size_t
        nb = lib.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np))
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
        nch = (Np + 2047) // 2048
        tl = torch.zeros((Fn * nch, 16), dtype=torch.int64, device=dev)
        p = ctypes.c_void_p

        def go():
            rc = lib.fk_resample_systematic_f64(ctypes.c_int64(Fn), ctypes.c_int64(Np), p(w.data_ptr()), p(u.data_ptr()),
                                                p(idx.data_ptr()), p(0), p(ws.data_ptr()), ctypes.c_size_t(nb), p(0))
            assert rc == 0, rc
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        assert lib.fk_debug_set_timeline(p(tl.data_ptr())) == 0
        go()
        torch.cuda.synchronize()
        assert lib.fk_debug_set_timeline(p(0)) == 0
        if Fn * nch <= 100000:
            np.save(os.path.join(out, "tl_%s.npy" % shape), tl.cpu().numpy())
        show(tl.cpu().numpy(), Fn, brief=True)


def show(tl, Fn=1, brief=False):
    n = tl.shape[0]
    t0 = tl[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    end = np.maximum(tl[:, 4], tl[:, 10])
    slow = tl[:, 5] > 0
    print("chunks %d, slow %d, whole call %.1f us (first start to last end)" % (n, int(slow.sum()), us(end.max())))
    print("starts: median %.1f us, last %.1f us" % (np.median(us(tl[:, 0])), us(tl[:, 0].max())))
    ks = np.nonzero(slow)[0]
    print("slow chunks (us since the first start): k  start sums | entry approx incr exact carry done | quick flags verdict")
    rows = ks if not brief else ks[:: max(1, len(ks) // 40)]
    for k in rows:
        r = tl[k]
        f = lambda s: ("%7.1f" % us(r[s])) if r[s] else "      -"
        print("%6d %s %s | %s %s %s %s %s %s | %d %x %d | prep %d fail %d D %d early %s" % (k, f(0), f(1), f(5), f(6), f(7), f(8), f(9), f(10), int(r[11]) - 100, int(r[12]) & 255, int(r[13]),
              int(r[15]) & 1, (int(r[15]) >> 1) & 1, int(r[15]) >> 8, f(14)))
    gen = slow & (tl[:, 11] == 100)
    if gen.any():
        d = tl[gen]
        print("general chunks %d: entry->approx %.1f, ->prepared %.1f, ->exact %.1f, ->scan done %.1f, ->done %.1f us (means); declined %d, not prepared %d" % (
            int(gen.sum()), np.mean(d[:, 6] - d[:, 5]) / 100, np.mean(d[:, 7] - d[:, 6]) / 100, np.mean(d[:, 8] - d[:, 7]) / 100,
            np.mean(d[:, 9] - d[:, 8]) / 100, np.mean(d[:, 10] - d[:, 9]) / 100, int(((d[:, 15] >> 1) & 1).sum()), int(((d[:, 15] & 1) == 0).sum())))
        pk = d[:, 2].astype(np.uint64)
        if pk.any():
            parts = [((pk >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(np.float64) * 16 for i in range(4)]
            print("  seg_prepare, core clocks (thread 0, means): to barrier 1 %.0f, to 2 %.0f, to 3 %.0f, to the end %.0f" % tuple(np.mean(x[pk > 0]) for x in parts))
    q1 = slow & (tl[:, 11] == 101)
    if q1.any():
        d = tl[q1]
        print("quick slow chunks %d: entry->approx %.1f, ->incr %.1f, ->exact %.1f, ->done %.1f us (means)" % (
            int(q1.sum()), np.mean(d[:, 6] - d[:, 5]) / 100, np.mean(d[:, 7] - d[:, 6]) / 100, np.mean(d[:, 8] - d[:, 7]) / 100, np.mean(d[:, 10] - d[:, 8]) / 100))
    fast = ~slow
    if fast.any():
        d = tl[fast]
        print("fast chunks: sums->look-back done median %.1f us, max %.1f; look-back done->stored median %.1f us" % (
            np.median((d[:, 3] - d[:, 1]) / 100.0), ((d[:, 3] - d[:, 1]) / 100.0).max(), np.median((d[:, 4] - d[:, 3]) / 100.0)))
        # when did each fast chunk get its carry-in, by position
        step = max(1, n // 24)
        print("look-back done at (us), every %d-th chunk:" % step, " ".join("%d:%.0f" % (k, us(max(tl[k, 3], tl[k, 8]))) for k in range(0, n, step)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--shapes", default="1x2000000,1x8000000")
    ap.add_argument("--out", default="gpurun_out/tl")
    ap.add_argument("--show")
    a = ap.parse_args()
    if a.run:
        run(a.shapes, a.out)
    elif a.show:
        show(np.load(a.show))
    else:
        sys.exit(__doc__)
