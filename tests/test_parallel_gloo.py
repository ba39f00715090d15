"""world_size-2 gloo test (CPU) of the multi-GPU path: contiguous track sharding with no
data-path collective + all-gather of the summary state + max-over-ranks timing."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from filterpy_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_tracks, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_bounds(n_tracks, rank, world)
    # every rank "filters" its own shard: the summary state of track i is a known function of i
    full = torch.arange(n_tracks * 4, dtype=torch.float64).reshape(n_tracks, 4) * 0.5
    per = n_tracks // world
    local = full[rank * per:(rank + 1) * per].clone()          # equal shards for the all-gather
    gathered = parallel.allgather_summary(local)
    ok = gathered.shape == (world, per, 4) and torch.equal(gathered.reshape(-1, 4), full[:world * per])
    tmax = parallel.max_over_ranks(1.0 + rank)
    parallel.barrier()
    q.put((rank, lo, hi, bool(ok), tmax))
    torch.distributed.destroy_process_group()


def test_shard_bounds_cover_and_are_contiguous():
    for n, world in ((10, 3), (1_000_000, 8), (7, 8), (125, 1)):
        b = [parallel.shard_bounds(n, r, world) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allgather():
    world, n_tracks = 2, 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_tracks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 500), (500, 1000)]
    assert all(r[3] for r in res) and all(abs(r[4] - 2.0) < 1e-12 for r in res)


def test_single_process_needs_no_group():
    t = torch.ones(3, 2)
    assert parallel.allgather_summary(t).shape == (1, 3, 2)
    assert parallel.max_over_ranks(3.5) == 3.5
