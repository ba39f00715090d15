"""Device -> host downloads of the large history arrays at PCIe speed.

`tensor.cpu()` into fresh pageable memory runs at ~13.6 GB/s on the MI355X hosts (the driver's own staging, one thread): at
BASELINE configs[1] the 32 GB of histories took 2.35 s of a 3.25 s `batch_filter` call whose kernel is 5.7 ms.  Here the copy
is pipelined by hand:

    device slab --hipMemcpyAsync (DMA, its own stream)--> pinned staging buffer k --memcpy (worker thread, GIL released)--> result

with a few pinned buffers of 64 MiB in flight, so the DMA engine never waits for the host copy of the slab before and the
destination is written by several cores at once (its first touch -- page faults -- is the expensive half of a host memcpy).  The
result is an ordinary NumPy array owned by the caller (pageable memory: nothing pinned outlives the call but the staging buffers,
allocated once per process and device: NBUF x BUF_BYTES).  Values are copied bit for bit; nothing is transposed on the host.

PyTorch is plumbing here as everywhere: pinned allocations, streams, events.  CPU tensors (the CPU tests' stand-in engine) and
small arrays take `tensor.cpu().numpy()`."""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

BUF_BYTES = int(os.environ.get("FK_D2H_BUF_MB", "64")) << 20
NBUF = int(os.environ.get("FK_D2H_BUFS", "6"))
WORKERS = int(os.environ.get("FK_D2H_WORKERS", "6"))
MIN_BYTES = 32 << 20                 # below this the plain copy is as good

_pipes = {}
_pipes_lock = threading.Lock()


class _Pipe:
    """staging buffers, copy stream and worker pool of one device"""

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.bufs = [torch.empty(BUF_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(NBUF)]
        self.views = [b.numpy() for b in self.bufs]
        self.free = threading.Semaphore(NBUF)
        self.free_ids = list(range(NBUF))
        self.ids_lock = threading.Lock()
        self.pool = ThreadPoolExecutor(max_workers=WORKERS, thread_name_prefix="fk-d2h")
        self.lock = threading.Lock()          # one download at a time per device (the buffers are shared)

    def _take(self):
        self.free.acquire()
        with self.ids_lock:
            return self.free_ids.pop()

    def _give(self, i):
        with self.ids_lock:
            self.free_ids.append(i)
        self.free.release()

    def _drain(self, i, ev, dst, n):
        try:
            ev.synchronize()                                   # the slab has landed in staging buffer i
            np.copyto(dst, self.views[i][:n])                  # (contiguous bytes: NumPy releases the GIL)
        finally:
            self._give(i)

    def download(self, pairs, wait=True):
        """pairs: [(flat uint8 device tensor, flat uint8 NumPy destination)], equal lengths.  wait=False: returns the slabs' futures
        once all of them are submitted (the caller keeps source and destination alive and calls .result() on each before it
        touches either) -- the next download's slabs then queue right behind these, the pipeline never drains in between."""
        with self.lock:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))    # behind the kernel that wrote them
            futs = []
            for src, dst in pairs:
                total = src.numel()
                for a in range(0, total, BUF_BYTES):
                    n = min(BUF_BYTES, total - a)
                    i = self._take()
                    with torch.cuda.stream(self.stream):
                        self.bufs[i][:n].copy_(src[a:a + n], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    futs.append(self.pool.submit(self._drain, i, ev, dst[a:a + n], n))
        if not wait:
            return futs
        for f in futs:
            f.result()                                          # (re-raises a worker's exception)
        return []


def _upload(self, src, dst):
    """flat uint8 NumPy source -> flat uint8 device tensor.  Each worker takes one slab end to end: memcpy into a staging buffer
    (GIL released), the DMA on the copy stream, wait for it, give the buffer back; the caller's stream then waits for the copy
    stream."""
    with self.lock:
        total = src.size

        def slab(i, a, n):
            try:
                np.copyto(self.views[i][:n], src[a:a + n])
                with torch.cuda.stream(self.stream):
                    dst[a:a + n].copy_(self.bufs[i][:n], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                ev.synchronize()
            finally:
                self._give(i)
        futs = []
        for a in range(0, total, BUF_BYTES):
            n = min(BUF_BYTES, total - a)
            futs.append(self.pool.submit(slab, self._take(), a, n))
        for f in futs:
            f.result()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)


_Pipe.upload = _upload


def to_device(h, device):
    """contiguous NumPy array -> device tensor of the same shape / dtype, uploaded through the pinned pipeline when that pays
    (`torch.as_tensor(h, device=...)` from pageable memory runs at ~13 GB/s: 0.13 s for BASELINE configs[1]'s measurements)"""
    if h.nbytes < MIN_BYTES or os.environ.get("FK_D2H_PIPE", "1") == "0" or torch.device(device).type != "cuda":
        return torch.as_tensor(h, device=device)
    t = torch.empty(h.shape, dtype=getattr(torch, str(h.dtype)), device=device)
    _pipe(torch.device(device)).upload(h.reshape(-1).view(np.uint8), t.view(-1).view(torch.uint8))
    return t


def _pipe(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    with _pipes_lock:
        p = _pipes.get(key)
        if p is None:
            p = _pipes[key] = _Pipe(torch.device("cuda", key))
        return p


def to_host(tensors):
    """[device tensors] -> [NumPy arrays of the same shape / dtype], downloaded through the pinned pipeline when that pays.
    Non-contiguous tensors are made contiguous on the device first (one device copy: the interleaved-history views)."""
    out, pairs, dev = [], [], None
    big = sum(t.numel() * t.element_size() for t in tensors if t.is_cuda) >= MIN_BYTES and os.environ.get("FK_D2H_PIPE", "1") != "0"
    for t in tensors:
        if not t.is_cuda or not big or t.numel() == 0:
            out.append(t.cpu().numpy())
            continue
        tc = t if t.is_contiguous() else t.contiguous()
        h = np.empty(tuple(tc.shape), dtype=np.dtype(str(tc.dtype).replace("torch.", "")))
        pairs.append((tc.view(-1).view(torch.uint8), h.reshape(-1).view(np.uint8)))
        dev = tc.device
        out.append(h)
    if pairs:
        _pipe(dev).download(pairs)
    return out


def into_host(pairs, wait=True):
    """[(device tensor, NumPy array)] of equal shape / dtype, both contiguous: the tensor's bytes into the array (the caller owns
    it: a slice of a larger result).  Small or CPU tensors: a plain copy.  wait=False: futures to call .result() on (see
    _Pipe.download)."""
    big, dev = [], None
    piped = sum(t.numel() * t.element_size() for t, _ in pairs if t.is_cuda) >= MIN_BYTES and os.environ.get("FK_D2H_PIPE", "1") != "0"
    for t, h in pairs:
        assert tuple(t.shape) == tuple(h.shape) and t.is_contiguous() and h.flags.c_contiguous, (t.shape, h.shape)
        if not t.is_cuda or not piped or t.numel() == 0:
            np.copyto(h, t.cpu().numpy())
            continue
        big.append((t.view(-1).view(torch.uint8), h.reshape(-1).view(np.uint8)))
        dev = t.device
    if big:
        return _pipe(dev).download(big, wait)
    return []


def release():
    """free the staging buffers and stop the workers (they come back on the next large download)"""
    with _pipes_lock:
        for p in _pipes.values():
            p.pool.shutdown(wait=True)
        _pipes.clear()
