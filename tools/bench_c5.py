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
    means = torch.empty((Fn, d), dtype=torch.float64, device=dev)
    gathered = torch.empty((world, Fn, d), dtype=torch.float64, device=dev)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]

    def step(e=None):
        if e:
            e[0].record()
        E.resample_systematic(Fn, Np, w, u, idx, st)
        if e:
            e[1].record()
        # posterior mean of the resampled set: gather + mean fused (fk_resample_gather_mean_f64)
        E.resample_gather_mean(Fn, Np, d, particles, idx, means)
        if e:
            e[2].record()
        parallel.allgather_summary(means, gathered)

    for _ in range(a.warmup):
        step()
    parallel.barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(ev[k])
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    assert not st.any()
    # parity: first filter of this rank against the oracle, bit-exact
    ref = ro.systematic_np(w[0].cpu().numpy(), float(u[0]))
    exact = bool(np.array_equal(idx[0].cpu().numpy(), ref))
    assert exact, "resample indices differ from the oracle"
    m_ref = particles[0].index_select(0, idx[0].long()).mean(dim=0)
    assert torch.allclose(means[0], m_ref, rtol=1e-11, atol=1e-13), "posterior mean differs from the gathered mean"
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
            "collectives": (f"{world}-rank {torch.distributed.get_backend()} group" if parallel.collectives_active() else "none (single process)"),
            "gather_ok": bool(torch.equal(gathered[rank], means))}), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
