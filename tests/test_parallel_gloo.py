"""world_size-2 gloo test (CPU) of the multi-GPU path: contiguous track sharding with no
data-path collective + all-gather of the summary state + max-over-ranks timing."""
import os

import pytest
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


def _overlap_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_from_env(backend="gloo")
    N, n, steps = 4096, 4, 7
    local = [torch.empty(N, n, dtype=torch.float64) for _ in range(2)]
    ex = parallel.SummaryExchange(like=local[0], depth=2)
    want = lambda k, r: (torch.arange(N * n, dtype=torch.float64).reshape(N, n) + r * N * n) * 0.25 + 100.0 * k  # noqa: E731
    ok, in_flight = True, 0
    for k in range(steps):
        slot = k % 2
        ex.acquire(slot)                       # the collective of step k - 2 is done with local[slot]
        local[slot].copy_(want(k, rank))       # "kernel" of step k
        ex.post(local[slot], slot)             # returns at once; step k + 1 overwrites the OTHER buffer meanwhile
        if k >= 1:
            prev = (k - 1) % 2
            w = ex._work[prev]
            in_flight += int(w is not None and not w.is_completed())
            g = ex.wait(prev)                  # step k - 1: complete and in rank order although step k has been written
            ok = ok and g.shape == (world, N, n) and all(torch.equal(g[r], want(k - 1, r)) for r in range(world))
    g = ex.wait((steps - 1) % 2)
    ok = ok and all(torch.equal(g[r], want(steps - 1, r)) for r in range(world))
    ex.drain()
    parallel.barrier()
    q.put((rank, bool(ok), in_flight))
    torch.distributed.destroy_process_group()


def test_two_rank_overlapped_exchange_is_rank_ordered_and_complete():
    """the per-step all-gather runs beside the next step (parallel.SummaryExchange: double-buffered local
    state and gathered state; on the GPU a side stream + events, here gloo's async work handles).  While step k's buffer is
    being written the gathered state of step k - 1 must come out complete and in rank order, for every step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_single_process_exchange_is_a_copy():
    local = [torch.full((5, 3), float(k)) for k in range(2)]
    ex = parallel.SummaryExchange(like=local[0], depth=2)
    for k in range(4):
        ex.acquire(k % 2)
        local[k % 2].fill_(10.0 + k)
        ex.post(local[k % 2], k % 2)
    assert ex.world == 1 and torch.equal(ex.wait(1)[0], torch.full((5, 3), 13.0)) and torch.equal(ex.wait(0)[0], torch.full((5, 3), 12.0))
    assert ex.gather_ms() is None


def test_single_process_needs_no_group():
    t = torch.ones(3, 2)
    assert parallel.allgather_summary(t).shape == (1, 3, 2)
    assert parallel.max_over_ranks(3.5) == 3.5


# ---- the command the driver runs: `python bench.py --gpus N` must use N ranks -----------------------------
def _run_bench(*argv, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, *argv], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    import json
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_self_spawns_one_rank_per_gpu():
    """bare `python bench.py --gpus 2` re-executes itself under torch.distributed.run with 2 ranks (gloo + a stub
    kernel here: --selftest-cpu), shards, all-gathers the summary state and prints ONE line with n_gpus = 2."""
    r = _run_bench("bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["gather_ok"] is True and line["steps"] == 3 and line["data"] == "selftest-stub"


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_under_the_drivers_torchrun_command(world):
    """the driver's own launch line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --
    at every N of its scaling run (1 is the plain call)"""
    r = _run_bench("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1",
                   "--selftest-cpu", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == world and line["gather_ok"] is True and line["collectives"] is True


@pytest.mark.parametrize("world", [2, 3])
def test_bench_strong_scaling_shards_one_total(world):
    """--scaling strong: --tracks is the TOTAL, rank r owns parallel.shard_bounds(total, r, world) -- 1001
    tracks here, so the shards are ragged and the summary leaves through the zero-padded buffers -- and the gathered state,
    cut back to the shards, is the whole bank in rank order."""
    r = _run_bench("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1",
                   "--selftest-cpu", "--scaling", "strong", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["scaling"] == "strong" and line["tracks_total"] == 1001 and line["gather_ok"] is True
    assert line["tracks_rank0"] == -(-1001 // world)


def test_bench_refuses_more_ranks_than_gpus():
    """no silent 1-GPU run: asking for more GPUs than the node has is an error (this container has none)."""
    r = _run_bench("bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu")
    assert r.returncode != 0 and "GPU(s) are visible" in (r.stderr + r.stdout)


def test_bench_rejects_world_size_mismatch():
    r = _run_bench("bench.py", "--gpus", "4", "--selftest-cpu", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_c5_uses_the_same_launcher():
    """tools/bench_c5.py --gpus N: same rule as bench.py (self-spawn; never a silent 1-GPU run; WORLD_SIZE must agree)."""
    r = _run_bench("tools/bench_c5.py", "--gpus", "8")
    assert r.returncode != 0 and "GPU(s) are visible" in (r.stderr + r.stdout)
    r = _run_bench("tools/bench_c5.py", "--gpus", "4", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_forced_one_rank_group_runs_the_collectives():
    """--force-dist (VERDICT r2 missing 2): with ONE rank the process group is still created and the all-gather, the
    barrier and the max-over-ranks all-reduce run on it -- on the GPU box that is how the RCCL branch gets executed
    before the driver's 8-GPU run; here the same control flow on gloo."""
    r = _run_bench("bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--selftest-cpu", "--force-dist")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["gather_ok"] is True and line["collectives"] is True
    r = _run_bench("bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--selftest-cpu")
    assert r.returncode == 0 and _json_line(r.stdout)["collectives"] is False
