"""Placement of two large, concurrently written history arrays in HBM.

Measured on MI355X (tools/exp_alloc.py, exp_buffers.py, exp_regions.py; profiles/r03/placement/): the time of a kernel that
streams two large output arrays at once -- `batch_filter`'s prior and posterior covariance histories are 76 % of its bytes --
depends on WHICH physical memory the two arrays occupy.  Physical memory falls into a few large classes (tens of GiB each);
two write streams inside one class run 8-27 % slower than two streams in different classes, whatever their distance inside
the class; a single stream does not care.  The driver picks the backing at allocation time and user space cannot ask for a
class, so the only handle is to measure: reserve one arena, time the caller's own kernel for the two arrays on a grid of
offsets, keep the fastest pair.  Nothing here changes what a kernel computes or stores.
"""
import torch


def candidate_offsets(arena_bytes, nbytes, step):
    return [o for o in range(0, arena_bytes - nbytes + 1, step)]


def place_pair(nbytes, run_ms, device, arena_bytes=None, step=16 << 30, reserve=40 << 30, reps=2):
    """Two byte tensors of `nbytes` each, views of one arena, for which `run_ms(a, b)` -- the caller's kernel with its two big
    outputs in a and b, returning milliseconds -- was smallest over a grid of (offset_a, offset_b), `step` bytes apart.

    Returns (a, b, info).  The arena stays allocated as long as a or b lives.  `reserve` bytes of the free memory are left
    alone; with less than 2 * nbytes + step available the pair is simply allocated (info says so)."""
    free, _ = torch.cuda.mem_get_info(device)
    if arena_bytes is None:
        arena_bytes = min(176 << 30, free - reserve)
    arena_bytes = (arena_bytes // (2 << 20)) * (2 << 20)
    if arena_bytes < 2 * nbytes + step:
        a = torch.empty(nbytes, dtype=torch.uint8, device=device)
        b = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return a, b, {"method": "plain allocation (arena does not fit)", "free_GiB": free >> 30}
    arena = torch.empty(arena_bytes, dtype=torch.uint8, device=device)
    base = (-arena.data_ptr()) % (2 << 20)                        # views start on 2 MiB boundaries
    offs = [o for o in candidate_offsets(arena_bytes - base, nbytes, step)]

    def view(o):
        return arena[base + o:base + o + nbytes]

    def timed(oa, ob):
        best = None
        for r in range(reps + 1):
            ms = run_ms(view(oa), view(ob))
            if r and (best is None or ms < best):
                best = ms
        return best

    grid = {}
    for i, oa in enumerate(offs):
        for ob in offs[i + 1:]:
            if ob - oa >= nbytes:
                grid[(oa, ob)] = timed(oa, ob)
    (oa, ob), best = min(grid.items(), key=lambda kv: kv[1])
    vals = sorted(grid.values())
    info = {"method": "probe", "arena_GiB": arena_bytes >> 30, "step_GiB": step >> 30, "pairs": len(grid),
            "chosen_offsets_GiB": [oa >> 30, ob >> 30], "chosen_ms": round(best, 4), "median_ms": round(vals[len(vals) // 2], 4),
            "worst_ms": round(vals[-1], 4),
            "grid_ms": {f"{a >> 30},{b >> 30}": round(v, 3) for (a, b), v in grid.items()}}
    return view(oa), view(ob), info


# ---------------------------------------------------------------------------------------------------------------------------
# The product-side form (round 4; bounded in round 6).  `placed_pair` allocates SEPARATE candidate buffers of the history's
# size one at a time, times the caller's own kernel on every pair the newcomer forms, and stops as soon as the timings show
# two classes and a pair in the fast one (or at `max_chunks` buffers / `max_frac` of the free memory); it keeps the fastest
# pair and drops the rest.  The pair is remembered per (device, size) in a SMALL LRU (`MAX_REMEMBERED` shapes per device:
# callers whose T or N vary do not accumulate two histories per shape): a later call of the same shape gets the same two
# buffers back without a probe, provided nothing derived from them is still alive, else two plain buffers.
# KalmanFilterBank.batch_filter(device_outputs=True) (its default at dim_x <= 4 with histories of 256 MiB and more),
# placement="probe" and bench.py --placement auto / probe all come through here.
#
# What the outcome is NOT: reproducible.  Which physical memory the driver backs an allocation with differs from process to
# process and box to box (docs/PLACEMENT.md: on one box the first six candidates all shared a class -- 6.77 ms -- and the
# eighth opened a 5.5 ms pair), so nothing is persisted across processes; a pair is only as good as this process's draw.
import collections
import threading

MAX_REMEMBERED = 2                  # shapes per device
_PAIRS = collections.OrderedDict()  # (device index, nbytes) -> (a, b, (use counts at rest), info); guarded by _LOCK
_LOCK = threading.Lock()


def _use_count(t):
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


def _lend(ent):
    """fresh views of the remembered pair, made while _LOCK is held: the storages' use counts rise before anybody else can
    look, and fall back to their resting values when the last tensor derived from the views dies"""
    a, b, _, info = ent
    return a.view(torch.uint8), b.view(torch.uint8), dict(info, method="cached")


def placed_pair(nbytes, run_ms, device, max_chunks=11, reserve=24 << 30, reps=1, or_none=False, max_frac=0.5, early_stop=0.9):
    """Two uint8 tensors of `nbytes` for the two concurrently written history arrays of a kernel, chosen by measurement.

    run_ms(a, b): the caller's kernel with its two big outputs in the byte tensors a, b -> milliseconds.  Returns (a, b, info);
    info["method"]: "probe" (just measured), "cached" (the pair of an earlier call, free again), "plain allocation (...)".
    or_none=True: where the probe cannot run (no room for three candidates, the remembered pair still in use) return
    (None, None, info) instead of two plain buffers -- the caller has something better than the lottery to fall back on.

    Bounds: at most `max_chunks` candidates and `max_frac` of the memory that is free beyond `reserve` are ever allocated
    at once; candidates arrive one at a time and the probe ends early once the best pair is below `early_stop` x the worst
    one seen (two classes of physical memory have shown, and a pair that straddles them is in hand); the losers are
    released to torch's caching allocator, not to the driver (no process-wide empty_cache() here: forget_placed_pairs()
    does that).  Thread-safe: the remembered pair is handed out under a lock as fresh views, so two callers of one shape
    cannot both receive it."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else -1), int(nbytes))
    with _LOCK:
        ent = _PAIRS.get(key)
        if ent is not None:
            a, b, base, info = ent
            if _use_count(a) == base[0] and _use_count(b) == base[1]:
                _PAIRS.move_to_end(key)
                return _lend(ent)
            if or_none:
                return None, None, {"method": "not placed (the placed pair of this shape is still in use)"}
            return (torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(nbytes, dtype=torch.uint8, device=dev),
                    {"method": "plain allocation (the placed pair of this shape is still in use)"})
    free, _ = torch.cuda.mem_get_info(dev)
    room = int(max(0, free - reserve) * max_frac)
    k = int(min(max_chunks, room // max(1, nbytes)))
    if k < 3 and or_none:
        return None, None, {"method": "not placed (no room to probe)", "free_GiB": free >> 30}
    if k < 3:
        return (torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(nbytes, dtype=torch.uint8, device=dev),
                {"method": "plain allocation (no room to probe)", "free_GiB": free >> 30})

    def timed(a, b):
        best = None
        for r in range(reps + 1):                            # the first launch on a pair is a warm-up
            ms = run_ms(a, b)
            if r and (best is None or ms < best):
                best = ms
        return best

    chunks, grid, stopped = [], {}, None
    for j in range(k):
        chunks.append(torch.empty(nbytes, dtype=torch.uint8, device=dev))
        for i in range(j):
            grid[(i, j)] = timed(chunks[i], chunks[j])
        if j >= 2 and grid and min(grid.values()) < early_stop * max(grid.values()):
            stopped = j + 1
            break
    (i, j), best = min(grid.items(), key=lambda kv: kv[1])
    vals = sorted(grid.values())
    a, b = chunks[i], chunks[j]
    n_tried = len(chunks)
    del chunks                                              # the losers: back to torch's caching allocator
    info = {"method": "probe", "buffers_tried": n_tried, "buffers_max": k, "stopped_early": stopped is not None, "pairs": len(grid),
            "launches": len(grid) * (reps + 1), "chosen": [i, j], "chosen_ms": round(best, 4),
            "median_ms": round(vals[len(vals) // 2], 4), "worst_ms": round(vals[-1], 4), "first_pair_ms": round(grid[(0, 1)], 4),
            "grid_ms": {f"{p},{q}": round(v, 3) for (p, q), v in grid.items()}}
    with _LOCK:
        if key in _PAIRS:                                   # another thread probed the same shape meanwhile: keep ours private
            return a, b, info
        ent = (a, b, (_use_count(a), _use_count(b)), info)
        _PAIRS[key] = ent
        mine = [q for q in _PAIRS if q[0] == key[0]]
        for q in mine[:-MAX_REMEMBERED]:                    # LRU per device (their memory goes once the callers' tensors are gone)
            del _PAIRS[q]
        a2, b2, _ = _lend(ent)
        return a2, b2, info


def forget_placed_pairs():
    """drop the remembered pairs and hand every cached block -- the probes' losers too -- back to the driver (the pairs' own
    memory is freed once the callers' references are gone)"""
    with _LOCK:
        _PAIRS.clear()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
