"""filterpy_amd.placement.placed_pair, the bookkeeping half (ADVICE r5 / VERDICT r5 next 9): a bounded probe (candidate count,
fraction of the free memory, early stop), a small LRU of remembered pairs, and the pair of a shape never in two callers' hands.
CPU tensors and a made-up timing function: no kernel runs here (the GPU test is tests/test_gpu_kf.py::
test_bank_placement_probe_places_by_measurement_and_remembers_the_pair)."""
import gc
import threading

import pytest
import torch

from filterpy_amd import placement


@pytest.fixture(autouse=True)
def _fresh(monkeypatch):
    placement.forget_placed_pairs()
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (200 << 30, 288 << 30))
    yield
    placement.forget_placed_pairs()


def _timer(fast_from=None, log=None):
    """a launch 'takes' 6.8 ms unless exactly one of its two buffers is a candidate numbered >= fast_from (then 5.4)"""
    order = {}

    def run_ms(a, b):
        for t in (a, b):
            order.setdefault(t.data_ptr(), len(order))
        if log is not None:
            log.append((order[a.data_ptr()], order[b.data_ptr()]))
        ia, ib = order[a.data_ptr()], order[b.data_ptr()]
        return 5.4 if fast_from is not None and (ia >= fast_from) != (ib >= fast_from) else 6.8
    return run_ms


def test_probe_stops_early_once_two_classes_show():
    log = []
    a, b, info = placement.placed_pair(1 << 20, _timer(fast_from=3, log=log), "cpu")
    assert info["method"] == "probe" and info["stopped_early"] and info["buffers_tried"] == 4, info
    assert info["chosen"][1] == 3 and info["chosen_ms"] == 5.4 and info["pairs"] == 6
    assert info["launches"] == 2 * info["pairs"] == len(log)          # one warm-up + one timed launch per pair


def test_probe_is_bounded_by_candidates_and_by_memory(monkeypatch):
    a, b, info = placement.placed_pair(1 << 20, _timer(), "cpu", max_chunks=5)
    assert info["buffers_tried"] == 5 and info["pairs"] == 10 and not info["stopped_early"]
    placement.forget_placed_pairs()
    # 24 GiB reserve + half of what is left: 40 GiB free -> 8 GiB of candidates -> four 2 GiB-sized requests (sizes are
    # only arithmetic here: the 'buffers' are 1 KiB)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (40 << 30, 288 << 30))
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda n, **kw: real_empty(1024, **kw))
    a, b, info = placement.placed_pair(2 << 30, _timer(), "cpu")
    assert info["buffers_tried"] == 4 and info["buffers_max"] == 4, info
    placement.forget_placed_pairs()
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (27 << 30, 288 << 30))
    assert placement.placed_pair(2 << 30, _timer(), "cpu", or_none=True)[0] is None
    a, b, info = placement.placed_pair(2 << 30, _timer(), "cpu")
    assert info["method"].startswith("plain allocation")


def test_remembered_pair_is_lent_to_one_caller_at_a_time():
    a, b, info = placement.placed_pair(4096, _timer(fast_from=2), "cpu")
    pa = a.data_ptr()
    view = a.view(torch.float64)                  # what a caller derives from it (and returns to its own caller)
    del a, b
    gc.collect()
    a2, b2, info2 = placement.placed_pair(4096, _timer(), "cpu")
    assert info2["method"].startswith("plain allocation") and a2.data_ptr() != pa        # still in use through `view`
    assert placement.placed_pair(4096, _timer(), "cpu", or_none=True)[0] is None
    del view, a2, b2
    gc.collect()
    a3, b3, info3 = placement.placed_pair(4096, _timer(), "cpu")
    assert info3["method"] == "cached" and a3.data_ptr() == pa
    # the hand-out itself marks the pair as taken: a second caller that arrives before the first has derived anything
    a4, b4, info4 = placement.placed_pair(4096, _timer(), "cpu", or_none=True)
    assert a4 is None and "still in use" in info4["method"]


def test_two_threads_of_one_shape_never_share_the_pair():
    placement.placed_pair(8192, _timer(fast_from=2), "cpu")
    gc.collect()
    got, barrier = [], threading.Barrier(8)

    def worker():
        barrier.wait()
        a, b, info = placement.placed_pair(8192, _timer(), "cpu")
        got.append((a, b, info["method"]))
    ts = [threading.Thread(target=worker) for _ in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert sum(m == "cached" for _, _, m in got) == 1
    ptrs = [a.data_ptr() for a, _, _ in got] + [b.data_ptr() for _, b, _ in got]
    assert len(set(ptrs)) == len(ptrs)


def test_lru_keeps_two_shapes_per_device():
    for nbytes in (1024, 2048, 4096):
        placement.placed_pair(nbytes, _timer(fast_from=2), "cpu")
    assert [k[1] for k in placement._PAIRS] == [2048, 4096]
    gc.collect()
    assert placement.placed_pair(2048, _timer(), "cpu")[2]["method"] == "cached"        # ... and touching one renews it
    gc.collect()
    placement.placed_pair(512, _timer(fast_from=2), "cpu")
    assert sorted(k[1] for k in placement._PAIRS) == [512, 2048]
