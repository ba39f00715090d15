#!/usr/bin/env python3
"""BASELINE.json configs[4]: monte_carlo systematic_resample, 1e3 filters sharded across the GPUs
of one node, RCCL all-gather of the posterior means over xGMI.

    python tools/bench_c5.py [--filters 1000] [--particles 8000] [--dim 4] [--steps 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29512 tools/bench_c5.py --particles 8000000 --filters 1000

Filters are independent: rank r resamples filters [lo_r, hi_r) with no data-path collective; the
only exchange is one all-gather of the (filters, dim) posterior means per step (the caller-side
"resample from index" of docs/monte_carlo/resampling.rst is a gather + mean: fk_resample_gather_mean_f64,
reported apart from the resampling kernel).  Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filters", type=int, default=1000)
    ap.add_argument("--particles", type=int, default=8000)
    ap.add_argument("--dim", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gpus", type=int, default=0, help="ranks (one per GPU); without a launcher the script re-executes "
                    "itself under torch.distributed.run like bench.py does")
    ap.add_argument("--force-dist", action="store_true", help="one rank: still create the 1-rank RCCL group and run the "
                    "all-gather / barrier / all-reduce on it (executes the N-GPU code path on a one-GPU box)")
    a = ap.parse_args()

    from filterpy_amd import parallel
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        parallel.relaunch_under_torchrun(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:])
    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    if a.gpus and a.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch one rank per GPU")

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world = parallel.init_from_env(backend="nccl", device=dev, force=a.force_dist)
    lo, hi = parallel.shard_bounds(a.filters, rank, world)
    per = a.filters // world                       # equal shards for the all-gather
    hi = lo + per
    Fn, Np, d = per, a.particles, a.dim

    g = torch.Generator(device=dev)
    g.manual_seed(100 + rank)
    w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
    w /= w.sum(dim=1, keepdim=True)
    rs = np.random.RandomState(7 + rank)
    u = E.dev(rs.rand(Fn), dev)                    # the host draws the uniforms (numpy.random in the reference)
    particles = torch.randn((Fn, Np, d), generator=g, device=dev, dtype=torch.float64)
    idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
    st = torch.zeros(Fn, dtype=torch.int32, device=dev)
    # the posterior means are all-gathered beside the next step (parallel.SummaryExchange: side stream, two mean buffers and
    # two gathered buffers); the resampler of step k + 1 never waits for the collective of step k
    mbuf = [torch.empty((Fn, d), dtype=torch.float64, device=dev) for _ in range(2)]
    ex = parallel.SummaryExchange(like=mbuf[0], depth=2)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]

    def step(k, e=None):
        slot = k % 2
        ex.acquire(slot)
        if e:
            e[0].record()
        E.resample_systematic(Fn, Np, w, u, idx, st)
        if e:
            e[1].record()
        # posterior mean of the resampled set: gather + mean fused (fk_resample_gather_mean_f64)
        E.resample_gather_mean(Fn, Np, d, particles, idx, mbuf[slot])
        if e:
            e[2].record()
        ex.post(mbuf[slot], slot, timed=e is not None)

    for k in range(a.warmup):
        step(k)
    ex.drain()
    parallel.barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k, ev[k])
    ex.drain()
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    assert not st.any()
    last = (a.steps - 1) % 2
    means, gathered = mbuf[last], ex.gathered[last]
    # parity (not timed), bit-exact: the first, a middle and the last filter of this rank against the reference's merge loop
    # (C restatement, oracle/resample_oracle.c), and EVERY filter of the rank against the independent tile-by-tile kernel of
    # round 1 (FK_RESAMPLE_SERIAL=1: no inter-workgroup protocol at all), every index compared on the GPU -- at 125 x 8e6 the
    # one-pass kernel has 488 k chunks in flight, which is where its hand-off protocol matters
    import hashlib
    checked = sorted({0, Fn // 2, Fn - 1})
    for f in checked:
        ref, over = ro.systematic_c(w[f].cpu().numpy(), float(u[f]))
        assert over == 0 and np.array_equal(idx[f].cpu().numpy(), ref), f"resample indices of filter {f} differ from the oracle"
    exact = True
    os.environ["FK_RESAMPLE_SERIAL"] = "1"
    idx2 = torch.full_like(idx, -1)
    E.resample_systematic(Fn, Np, w, u, idx2, st)
    torch.cuda.synchronize()
    del os.environ["FK_RESAMPLE_SERIAL"]
    assert not st.any()
    differ = (idx != idx2).any(dim=1).nonzero().flatten().tolist()          # every index of every filter, compared on the GPU
    assert not differ, f"filters {differ[:8]} differ between the dispatched kernel and the serial kernel"
    sha = {f: hashlib.sha256(idx[f].cpu().numpy().tobytes()).hexdigest()[:16] for f in checked}
    del idx2
    for f in checked:
        m_ref = particles[f].index_select(0, idx[f].long()).mean(dim=0)
        assert torch.allclose(means[f], m_ref, rtol=1e-11, atol=1e-13), "posterior mean differs from the gathered mean"
    if rank == 0:
        rs_ms = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
        gm_ms = float(np.median([e[1].elapsed_time(e[2]) for e in ev]))
        total = float(a.steps) * per * world * Np
        print(json.dumps({
            "metric": "particles resampled / s (systematic_resample)", "value": total / elapsed, "unit": "particles/s",
            "n_gpus": world, "steps": a.steps, "ms_per_step": 1e3 * elapsed / a.steps,
            "config": {"workload": f"BASELINE configs[4]: {per * world} filters x {Np} particles, {per} filters per GPU, "
                                   f"all-gather of ({per * world}, {d}) posterior means per step"},
            "resample_kernel_ms": rs_ms, "gather_mean_kernel_ms": gm_ms,
            "resample_GBs_algorithmic": 12.0 * per * Np / (rs_ms * 1e-3) / 1e9, "bit_exact_vs_oracle": exact,
            "parity": f"filters {checked} vs the merge loop (oracle, C); all {Fn} filters, every index, vs the serial kernel",
            "index_sha256_16": sha,
            "allgather_ms": ex.gather_ms(), "exchange": "overlapped with the next step (side stream, double-buffered)",
            "collectives": (f"{world}-rank {torch.distributed.get_backend()} group" if parallel.collectives_active() else "none (single process)"),
            "gather_ok": bool(torch.equal(gathered[rank], means))}), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
