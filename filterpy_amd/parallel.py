"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI
on ROCm, "gloo" on CPU for tests).  Tracks / particle filters are independent, so the data path
has NO collective: units are sharded in contiguous blocks and the only exchange is an all-gather
of summary state (final x per track, posterior means per filter) after the time loop --
overlapped with the next step's launch by SummaryExchange (side stream, double-buffered)."""
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


class SummaryExchange:
    """The per-step all-gather of a rank's summary state, OVERLAPPED with the next step's kernel.

    The data path has no collective; the summary exchange is the only one, and it must never sit between two launches of
    the compute stream.  `depth` slots (default 2) double-buffer both sides:

        ex = SummaryExchange(like=x, depth=2)
        for k in range(steps):
            slot = k % ex.depth
            ex.acquire(slot)            # compute stream: the collective that last READ local[slot] is done (step k - depth)
            launch(kernel writing local[slot])
            ex.post(local[slot], slot)  # exchange stream: waits for the kernel by an event, gathers into ex.gathered[slot]
        ex.drain()                      # before the closing barrier

    On CUDA / ROCm tensors the collective is enqueued under a side stream that waits for the compute stream's event (backend
    "nccl" = RCCL: ProcessGroupNCCL orders its own communication stream behind the stream that is current at the call);
    the compute stream never waits for it -- only acquire() does, `depth` steps later, and by then it is long done.  On CPU
    tensors (backend gloo: the tests) the collective is issued with async_op=True and its work handle kept per slot.
    Without a process group (one rank) post() is a device copy, so that callers need not branch.
    `gather_ms()` = the median duration of the collectives on the exchange stream (events), reported apart from kernel time
    ("report it separately")."""

    def __init__(self, like, depth=2):
        self.depth = int(depth)
        self.world = dist.get_world_size() if collectives_active() else 1
        self.cuda = like.is_cuda
        self.gathered = [torch.empty((self.world,) + tuple(like.shape), dtype=like.dtype, device=like.device)
                         for _ in range(self.depth)]
        self._work = [None] * self.depth
        self._timed = []
        if self.cuda:
            self.stream = torch.cuda.Stream(device=like.device)
            self._ready = [torch.cuda.Event() for _ in range(self.depth)]       # kernel of the slot enqueued
            self._done = [None] * self.depth                                    # collective of the slot finished

    def acquire(self, slot):
        """Order the CURRENT stream behind the collective that last used `slot` (call before overwriting local[slot])."""
        if self.cuda:
            if self._done[slot] is not None:
                torch.cuda.current_stream().wait_event(self._done[slot])
        elif self._work[slot] is not None:
            self._work[slot].wait()
            self._work[slot] = None

    def post(self, local, slot, timed=False):
        """All-gather `local` into self.gathered[slot], behind everything enqueued so far on the current stream; returns at
        once.  `local` must stay untouched until acquire(slot) / wait(slot)."""
        out = self.gathered[slot]
        if self.cuda:
            self._ready[slot].record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(self._ready[slot])
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if not collectives_active():
                    out.copy_(local.unsqueeze(0))
                elif dist.get_backend() == "nccl":
                    dist.all_gather_into_tensor(out, local)
                else:
                    dist.all_gather(list(out.unbind(0)), local)
                if timed:
                    e1.record()
                    self._timed.append((e0, e1))
                done = torch.cuda.Event()
                done.record()
                self._done[slot] = done
        elif not collectives_active():
            out.copy_(local.unsqueeze(0))
        else:
            self._work[slot] = dist.all_gather(list(out.unbind(0)), local, async_op=True)

    def wait(self, slot):
        """Block the host until the collective of `slot` has finished; returns its (world, *shape) result."""
        if self.cuda:
            if self._done[slot] is not None:
                self._done[slot].synchronize()
        elif self._work[slot] is not None:
            self._work[slot].wait()
            self._work[slot] = None
        return self.gathered[slot]

    def drain(self):
        for slot in range(self.depth):
            self.wait(slot)

    def gather_ms(self):
        """median duration of the timed collectives on the exchange stream (None without any)"""
        if not self._timed:
            return None
        ts = sorted(a.elapsed_time(b) for a, b in self._timed)
        return float(ts[len(ts) // 2])


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
