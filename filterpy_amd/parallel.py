"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI
on ROCm, "gloo" on CPU for tests).  Tracks / particle filters are independent, so the data path
has NO collective: units are sharded in contiguous blocks and the only exchange is an all-gather
of summary state (final x per track, posterior means per filter) after the time loop."""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_units, rank, world):
    """Contiguous block [lo, hi) of `n_units` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(n_ranks, argv, require_devices=True):
    """`script --gpus N` started WITHOUT a launcher: replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... script argv`, one rank per GPU of this
    node (rendezvous on 127.0.0.1, a free port).  Fails loudly when the node has fewer than N GPUs.  Never returns."""
    import sys
    if require_devices:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_ranks:
            raise SystemExit(f"--gpus {n_ranks} requested but only {have} GPU(s) are visible on this node")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


_FORCE = False        # --force-dist: a 1-rank group still takes the collective branches (see init_from_env)


def init_from_env(backend=None, device=None, force=False):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun sets them).
    Returns (rank, world).  A single process needs no group -- unless `force`: then a ONE-rank group is created and
    allgather_summary / max_over_ranks / barrier run their real collectives on it.  That is how the code an N-GPU run
    executes (init_process_group("nccl", device_id=...), all_gather_into_tensor, all_reduce, barrier over RCCL) is
    executed on a one-GPU box (`bench.py --force-dist`, `tools/bench_c5.py --force-dist`)."""
    global _FORCE
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    _FORCE = bool(force)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()) if world == 1 else "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def collectives_active():
    """True when the exchange step runs real collectives: more than one rank, or a forced one-rank group"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def allgather_summary(local, out=None):
    """All-gather equally-shaped per-rank summary tensors -> tensor (world, *local.shape)."""
    if not collectives_active():
        return local.unsqueeze(0) if out is None else out.copy_(local.unsqueeze(0))
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, local.contiguous())
    else:
        dist.all_gather(list(out.unbind(0)), local.contiguous())
    return out


def max_over_ranks(value, device=None):
    if not collectives_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def barrier():
    if collectives_active():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
