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
