"""-m gpu parity tests: resampling kernels through the C ABI and the Python API -- indices must be
BIT-EXACT against the goldens frozen from the live reference (np.random.seed -> filterpy's own
functions) and against the oracle."""
import hashlib
import zlib

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _weights_for(N, seed, kind):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "make_goldens_w", os.path.join(os.path.dirname(__file__), "golden", "weights.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.weights_for(N, seed, kind)


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _golden_cases(prefix):
    g = golden("resample")
    keys = sorted(k[:-len("_wseed")] for k in g.files if k.startswith(prefix + "_N") and k.endswith("_wseed"))
    return g, keys


@pytest.mark.parametrize("name", ["sys", "strat", "multi", "resid"])
def test_python_api_vs_goldens(name):
    from filterpy_amd import monte_carlo as mc
    fn = dict(sys=mc.systematic_resample, strat=mc.stratified_resample, multi=mc.multinomial_resample,
              resid=mc.residual_resample)[name]
    g, keys = _golden_cases(name)
    assert keys
    for key in keys:
        N = int(key.split("_N")[1].split("_")[0])
        kind = key.split("_")[-1]
        if N > (1 << 20):
            continue
        w = _weights_for(N, int(g[key + "_wseed"]), kind.rstrip("01") if kind.startswith("rand") else kind)
        np.random.seed(int(g[key + "_useed"]))
        if key + "_indexerror" in g.files:
            with pytest.raises(IndexError):
                fn(w)
            continue
        if name == "resid" and not np.isfinite(w - np.floor(N * w)).all():
            continue
        try:
            idx = fn(w)
        except IndexError:
            raise AssertionError(f"{key}: unexpected IndexError")
        assert str(idx.dtype) == str(g[key + "_dtype"]), key
        if key + "_idx" in g.files:
            assert np.array_equal(idx, g[key + "_idx"]), (key, int(np.argmax(idx != g[key + "_idx"])))
        else:
            assert np.array_equal(idx[:1024], g[key + "_head"]) and np.array_equal(idx[-1024:], g[key + "_tail"]), key
            assert np.array_equal(_sha(idx), g[key + "_sha"]), key


@pytest.mark.parametrize("name", ["sys", "strat"])
def test_huge_filters_bit_exact(name):
    """N = 8e6: a blocked/pairwise scan flips indices here (recorded in the golden); ours must not."""
    from filterpy_amd import monte_carlo as mc
    fn = dict(sys=mc.systematic_resample, strat=mc.stratified_resample)[name]
    g = golden("resample")
    for si in range(2):
        key = f"{name}_N8000000_rand{si}"
        w = _weights_for(8000000, int(g[key + "_wseed"]), "rand")
        np.random.seed(int(g[key + "_useed"]))
        idx = fn(w)
        assert idx.dtype == np.int32
        assert np.array_equal(idx[:1024], g[key + "_head"]) and np.array_equal(idx[-1024:], g[key + "_tail"])
        fp = g[key + "_flip_pos"]
        assert np.array_equal(idx[fp], g[key + "_flip_idx"])
        assert np.array_equal(_sha(idx), g[key + "_sha"]), key


def test_bank_of_filters_vs_oracle():
    """C5 shape (reduced): many filters at once, every filter against the C oracle."""
    from filterpy_amd import monte_carlo as mc
    from oracle import resample_oracle as ro
    rs = np.random.RandomState(11)
    Fn, Np = 37, 8000
    w = rs.rand(Fn, Np)
    w /= w.sum(axis=1, keepdims=True)
    np.random.seed(5)
    got = mc.systematic_resample(w)
    np.random.seed(5)
    us = np.random.random(Fn)
    for f in range(Fn):
        ref, over = ro.systematic_c(w[f], us[f])
        assert over == 0 and np.array_equal(got[f], ref), f
    np.random.seed(6)
    got = mc.stratified_resample(w)
    np.random.seed(6)
    for f in range(Fn):
        ref, over = ro.stratified_c(w[f], np.random.random(Np))
        assert over == 0 and np.array_equal(got[f], ref), f


def test_residual_resample_bank_and_edges():
    """residual_resample on the device (fk_resample_residual_fill_f64 + _draw_f64; resampling.py:27-76): a bank draws
    random(N - k_f) per filter in filter order, so it equals the filters resampled one by one under the same seed -- and
    both equal the oracle's restatement of the reference's loops; weights that earn more than N copies raise IndexError"""
    import filterpy_amd.monte_carlo as mc
    from oracle import resample_oracle as ro
    rs = np.random.RandomState(404)
    for Np in (1, 7, 100, 2049, 8000):
        Fn = 6
        w = rs.rand(Fn, Np) ** rs.choice([1, 3], size=(Fn, 1))
        w /= w.sum(axis=1, keepdims=True)
        w[1] = np.full(Np, 1.0 / Np)                              # every weight earns exactly one copy: k = N, no draw
        if Np > 4:
            w[2, : Np // 2] = 0.0                                  # zero weights
            w[2] /= w[2].sum()
        np.random.seed(1234 + Np)
        bank = mc.residual_resample(w)
        np.random.seed(1234 + Np)
        single = np.stack([mc.residual_resample(w[f]) for f in range(Fn)])
        np.random.seed(1234 + Np)
        ref = np.stack([ro.residual_seeded(w[f]) for f in range(Fn)])
        assert bank.dtype == np.int32 and bank.shape == (Fn, Np)
        assert np.array_equal(bank, single) and np.array_equal(bank, ref), Np
    with pytest.raises(IndexError):
        mc.residual_resample(np.array([0.9, 0.9, 0.9]))            # floor(3 * .9) = 2 copies each: 6 > 3


def test_exact_cumsum_kernel_bitwise():
    import torch
    from filterpy_amd import _engine as E
    rs = np.random.RandomState(3)
    for N in (1, 5, 2048, 2049, 100003, 1 << 20):
        w = np.stack([rs.rand(N), np.exp(rs.randn(N) * 5), rs.rand(N) * (rs.rand(N) < 0.02),
                      np.concatenate([np.zeros(N // 2), rs.rand(N - N // 2)])])
        dw = E.dev(w)
        cs = torch.empty_like(dw)
        E.cumsum_exact(w.shape[0], N, dw, cs)
        got = cs.cpu().numpy()
        ref = np.cumsum(w, axis=1)
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), N


HARD = {
    "ties": lambda rs, N: rs.randint(0, 3, N) * 2.0 ** -53 + (rs.rand(N) < 0.01),       # exact half-ulp ties
    "ties_everywhere": lambda rs, N: np.full(N, 2.0 ** -12 + 2.0 ** -54),               # every add is a tie in [1/2, 1)
    "rare_tie": lambda rs, N: np.where(np.arange(N) % 5000 == 4999, 2.0 ** -54, 0.0) + rs.randint(1, 1000, N) * 2.0 ** -40,
    "tiny": lambda rs, N: rs.rand(N) * 1e-310,                                          # subnormal sums
    "growing": lambda rs, N: 1.5 ** (np.arange(N) % 900),                               # a binade crossing on most adds
    "huge_then_small": lambda rs, N: np.concatenate([[1e300], rs.rand(N - 1)]),
    "onehot": lambda rs, N: np.eye(1, N, N // 3)[0],
    "with_negative": lambda rs, N: rs.randn(N),                                         # not valid weights, still exact
    "heavy_tail": lambda rs, N: np.exp(rs.randn(N) * 6.0),
}


@pytest.mark.parametrize("name", sorted(HARD))
def test_exact_cumsum_kernel_hard_cases(name):
    """the tie-free double-increment fast path and its fall-back to the Mono scan, bit for bit against
    numpy.cumsum (the same hard inputs as tests/test_hostcheck_exact_scan.py)"""
    import torch
    from filterpy_amd import _engine as E
    for N in (7, 2048, 2049, 5000, 100003, 300007):
        rs = np.random.RandomState(len(name) * 31 + N)
        w = np.asarray(HARD[name](rs, N), dtype=np.float64)[None, :N]
        dw = E.dev(w)
        cs = torch.empty_like(dw)
        E.cumsum_exact(1, N, dw, cs)
        got, ref = cs.cpu().numpy(), np.cumsum(w, axis=1)
        nz = ref != 0                       # zeros may differ in sign only (-0.0 + 0.0)
        assert np.array_equal(got.view(np.uint64)[nz], ref.view(np.uint64)[nz]) and np.array_equal(got == 0, ref == 0), (name, N)


def test_overrun_raises_indexerror():
    """weights summing to < the last position: the reference raises IndexError."""
    from filterpy_amd import monte_carlo as mc
    w = np.full(1000, 0.5e-3)          # sums to 0.5
    np.random.seed(0)
    with pytest.raises(IndexError):
        mc.systematic_resample(w)


def test_chunk_parallel_path_equals_serial_path(monkeypatch):
    """Long vectors take the chunk-parallel path (many workgroups per filter); it must give the same
    indices as the one-workgroup-per-filter path and as the oracle, incl. hard inputs."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    rs = np.random.RandomState(21)
    Np = 300007
    ws = np.stack([rs.rand(Np), np.exp(rs.randn(Np) * 5), rs.rand(Np) * (rs.rand(Np) < 0.02),
                   np.concatenate([np.zeros(Np // 3), rs.rand(Np - Np // 3)]), np.eye(1, Np, 777)[0] + 1e-12])
    ws /= ws.sum(axis=1, keepdims=True)
    Fn = ws.shape[0]
    u = rs.rand(Fn)
    us = rs.rand(Fn, Np)
    dw, du, dus = E.dev(ws), E.dev(u), E.dev(us)
    out = {}
    for mode in ("parallel", "serial"):
        if mode == "serial":
            monkeypatch.setenv("FK_RESAMPLE_SERIAL", "1")
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dw.device)
        st = torch.zeros(Fn, dtype=torch.int32, device=dw.device)
        E.resample_systematic(Fn, Np, dw, du, idx, st)
        a = idx.cpu().numpy().copy()
        E.resample_stratified(Fn, Np, dw, dus, idx, st)
        out[mode] = (a, idx.cpu().numpy().copy(), st.cpu().numpy().copy())
    assert np.array_equal(out["parallel"][0], out["serial"][0]) and np.array_equal(out["parallel"][1], out["serial"][1])
    for f in range(Fn):
        ref, over = ro.systematic_c(ws[f], u[f])
        if over == 0:
            assert np.array_equal(out["parallel"][0][f], ref), f
        ref, over = ro.stratified_c(ws[f], us[f])
        if over == 0:
            assert np.array_equal(out["parallel"][1][f], ref), f


# ---- the one-pass path (resample_onepass.hip) ---------------------------------------------------------------------
_FAMILIES = ("uniform", "heavy_tail", "zeros", "leading_zeros", "one_heavy", "ties", "sum_half", "unnormalised",
             "all_zero_filter", "negative", "nan")


def _family(kind, Fn, Np, rs):
    """weight families that drive every route of the kernel: the quick route, binade crossings and half-ulp ties
    (general scan), runs of zero weights, one weight owning many windows of slots, positions past cumsum[-1]
    (IndexError), and garbage (negative / NaN: the reference's loop, literally)."""
    w = rs.rand(Fn, Np)
    if kind == "heavy_tail":
        w = w ** 12
    elif kind == "zeros":
        w = np.where(rs.rand(Fn, Np) < 0.7, 0.0, w)
    elif kind == "leading_zeros":
        w[:, : (Np * 3) // 10] = 0.0
    elif kind == "one_heavy":
        w[:, Np // 3] = 1e4
    elif kind == "ties":
        w = np.floor(w * 2 ** 20) * 2.0 ** -40
    elif kind == "dyadic":                 # an exact half-ulp tie at every add once the running sum is in [0.5, 1)
        w = np.full((Fn, Np), 3 * 2.0 ** -54)
        w[:, 0] = 0.75
        return np.ascontiguousarray(w)
    elif kind == "tiny":                   # running sums below 2^-900
        return np.ascontiguousarray(w * 1e-300)
    w = w / w.sum(axis=1, keepdims=True)
    if kind == "sum_half":
        w = w * 0.5
    elif kind == "unnormalised":
        w = w * 1e6
    elif kind == "all_zero_filter":
        w[0] = 0.0
    elif kind == "negative":
        w[-1, Np // 2] = -0.25 / Np
        w[0, 0] = -1e-9
    elif kind == "nan":
        w[-1, Np // 5] = np.nan
    return np.ascontiguousarray(w)


def _onepass_mode(monkeypatch, mode):
    """spec: the defaults (round 6: systematic calls on resample_onepass2_kernel, its binade predictions on); round3: FK_OP_V2=0,
    round 3's kernel with speculation; nopredict: round 6's kernel without the predictions (every chunk in round 3's order);
    two-stage: FK_OP_SPEC=0; tickets: FK_OP_SPEC=0 FK_OP_STATIC=0 (round 2)"""
    if mode == "round3":
        monkeypatch.setenv("FK_OP_V2", "0")
        return
    if mode in ("waves5", "waves6", "waves7"):     # the instantiation budgeted for that many workgroups per CU (FK_OP_WAVES)
        monkeypatch.setenv("FK_OP_WAVES", mode[-1])
        return
    if mode in ("lb0", "lb3", "lb7"):          # FK_OP_LB: how look-backs wait and when the general scan of a chunk is prepared (the
        monkeypatch.setenv("FK_OP_LB", mode[2:])   # launcher picks 3 or 7 by the number of filters; 0 = before round 6's last leases)
        return
    if mode == "nopredict":
        monkeypatch.setenv("FK_OP_PRED_BACK", "0")
        return
    if mode == "predict-near":                 # predictions from a chunk that is usually NOT finished yet, and from the nearest one
        monkeypatch.setenv("FK_OP_PRED_BACK", "1")
        return
    if mode != "spec":
        monkeypatch.setenv("FK_OP_SPEC", "0")
    if mode == "tickets":
        monkeypatch.setenv("FK_OP_STATIC", "0")


def _check_against_merge_loop(Fn, Np, kinds, filters, monkeypatch, force):
    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    if force:
        monkeypatch.setenv("FK_RESAMPLE_PATH", "onepass")
    for strat in (0, 1):
        for kind in kinds:
            rs = np.random.RandomState(zlib.crc32(f"{Np}-{kind}-{strat}".encode()))
            w = _family(kind, Fn, Np, rs)
            u = rs.rand(Fn, Np) if strat else rs.rand(Fn)
            dw, du = E.dev(w), E.dev(u)
            idx = torch.full((Fn, Np), -7, dtype=torch.int32, device=dw.device)
            st = torch.zeros(Fn, dtype=torch.int32, device=dw.device)
            (E.resample_stratified if strat else E.resample_systematic)(Fn, Np, dw, du, idx, st)
            got, sth = idx.cpu().numpy(), st.cpu().numpy()
            for f in filters:
                ref, over = (ro.stratified_c if strat else ro.systematic_c)(w[f], u[f])
                ok = ref < Np                      # the slots the reference fills before it raises IndexError
                assert np.array_equal(got[f][ok], ref[ok]), (Np, kind, strat, f, int(np.argmax(got[f][ok] != ref[ok])))
                assert bool(sth[f] & 4) == (over > 0) and not (sth[f] & 8), (Np, kind, strat, f, int(sth[f]))


@pytest.mark.parametrize("mode", ["spec", "round3", "nopredict", "predict-near", "waves5", "waves6", "waves7", "lb0", "lb3", "lb7", "two-stage", "tickets"])
@pytest.mark.parametrize("Np", [1, 2, 100, 2049, 65536])
def test_onepass_every_route_small(Np, mode, monkeypatch):
    """FK_RESAMPLE_PATH=onepass forces short vectors through the one-pass kernel: every weight family, every filter,
    against the reference's merge loop (C restatement), systematic and stratified."""
    kinds = [k for k in _FAMILIES if not (k in ("negative", "nan") and Np < 8)]
    _onepass_mode(monkeypatch, mode)
    _check_against_merge_loop(5, Np, kinds, range(5), monkeypatch, force=True)


@pytest.mark.parametrize("mode", ["spec", "round3", "nopredict", "predict-near", "waves5", "waves6", "waves7", "lb0", "lb3", "lb7", "two-stage", "tickets"])
def test_onepass_every_route_long(mode, monkeypatch):
    """default dispatch on a long ragged vector (not a multiple of the chunk, odd address alignment per filter), in the three
    protocols of the one-pass kernel: speculation + static chunk assignment (round 3's default), round 2's two stages on
    static assignment, and round 2's atomic tickets"""
    _onepass_mode(monkeypatch, mode)
    _check_against_merge_loop(6, 1000003, _FAMILIES + ("dyadic", "tiny"), (0, 3, 5), monkeypatch, force=False)


@pytest.mark.parametrize("mode", ["spec", "round3", "tickets"])
def test_onepass_repairs_itself_when_a_hand_off_times_out(mode, monkeypatch):
    """a bounded spin of the one-pass kernel that times out sets the abort word (FK_STATUS_INTERNAL).  The
    call must not report that, it must repair it: the repair pass (resample_local_kernel behind the one-pass kernel, one
    workgroup per filter, no workgroup waits for another) recomputes every filter.  FK_OP_FORCE_ABORT=1 presets the abort
    word, so every chunk that has to wait for a predecessor gives up: the indices must still be the merge loop's, bit for
    bit, and the status must not carry FK_STATUS_INTERNAL (it does carry the IndexError bit of the sum_half family)."""
    _onepass_mode(monkeypatch, mode)
    monkeypatch.setenv("FK_OP_FORCE_ABORT", "1")
    _check_against_merge_loop(6, 200003, _FAMILIES + ("dyadic", "tiny"), range(6), monkeypatch, force=False)
    _check_against_merge_loop(3, 40000, ("uniform", "heavy_tail", "sum_half"), range(3), monkeypatch, force=False)


@pytest.mark.parametrize("mode", ["spec", "round3", "predict-near", "waves7"])
def test_onepass_one_long_vector_walks_binade_segments(mode, monkeypatch):
    """ONE filter of 3e6 weights: the chunks of every binade segment wait for the carry-out of the segment before it (the
    serial part of a call with few filters: ~12 binade crossings, each resolved by the general scan of its chunk)"""
    _onepass_mode(monkeypatch, mode)
    _check_against_merge_loop(1, 3000017, ("uniform", "heavy_tail", "zeros"), (0,), monkeypatch, force=False)
    _check_against_merge_loop(40, 70001, ("uniform",), (0, 17, 39), monkeypatch, force=True)


@pytest.mark.parametrize("mode", ["spec", "round3", "waves7"])
def test_onepass_many_filters_of_a_hundred_thousand(mode, monkeypatch):
    """1000 x 100 000: more concurrent chains than workgroup slots; speculation on ordinary and on skewed weights (whose
    guesses mostly miss and fall back to the two stages)"""
    _onepass_mode(monkeypatch, mode)
    _check_against_merge_loop(1000, 100000, ("uniform", "heavy_tail", "zeros"), (0, 1, 499, 998, 999), monkeypatch, force=False)


def test_short_vectors_with_garbage_weights_follow_the_reference_loop(monkeypatch):
    """default dispatch, short vectors (one workgroup per filter): filters with a negative / NaN weight are redone
    by the literal merge loop"""
    _check_against_merge_loop(4, 8000, ("uniform", "negative", "nan", "sum_half"), range(4), monkeypatch, force=False)


@pytest.mark.parametrize("Np", [1, 2, 7, 100, 2047, 2048, 2049, 8000, 8192, 20000, 32767])
def test_local_kernel_every_route(Np, monkeypatch):
    """resample_local_kernel (one workgroup per filter, chunks in sequence, segmented exact scan at the vector start /
    binade crossings, literal loop for garbage) -- the default between 8193 and 32767 weights, forced below: every
    weight family, every filter, against the reference's merge loop (C restatement), systematic and stratified."""
    kinds = [k for k in _FAMILIES if not (k in ("negative", "nan") and Np < 8)]
    monkeypatch.setenv("FK_RESAMPLE_PATH", "local")
    _check_against_merge_loop(5, Np, kinds, range(5), monkeypatch, force=False)


@pytest.mark.parametrize("exact", ["0", "1"])
@pytest.mark.parametrize("Np", [1, 2, 3, 7, 100, 2047, 2048, 2049, 4095, 4096, 4097, 8000, 8189, 8190, 8191, 8192])
def test_whole_vector_kernel_every_route(Np, exact, monkeypatch):
    """resample_whole_kernel (round 3; the default up to 8192 weights: one workgroup takes the whole vector in one
    round): every weight family -- plus vectors the round declines (exact half-ulp ties by the hundred, running sums
    below 2^-900) and must hand to the literal loop --, every filter of a 5-filter call (odd Np: every filter at another
    16-byte phase, so the shifted window, the unshifted 4-byte-store fallback at Np > 8189 and the scalar weight loads
    all run), systematic and stratified, against the
    reference's merge loop (C restatement).  exact = "0": the kernel as dispatched -- boundaries from the plain prefix
    sums, the exact round only for a vector with an estimate inside the error band; "1" (FK_WHOLE_EXACT): the exact
    round for every vector."""
    monkeypatch.setenv("FK_WHOLE_EXACT", exact)
    kinds = [k for k in _FAMILIES + ("dyadic", "tiny") if not (k in ("negative", "nan") and Np < 8)]
    _check_against_merge_loop(5, Np, kinds, range(5), monkeypatch, force=False)


@pytest.mark.parametrize("quick", ["1", "0"])
@pytest.mark.parametrize("Np", [7, 2049, 4097, 8000, 8191])
def test_whole_vector_kernel_quick_path_and_its_tail(Np, quick, monkeypatch):
    """Round 5: short vectors run resample_whole_quick_kernel -- the common path alone within 64 VGPRs (two 1024-thread
    workgroups per CU), the whole algorithm (exact round, literal loop) as an unlikely tail behind it for a vector whose
    boundary estimates touch the error band, for garbage weights and for FK_WHOLE_EXACT=1.  37 filters, finished-by-the-common-
    path and tail ones mixed within a call (the `negative` / `nan` families: first and last filter garbage; `dyadic` / `tiny`:
    every filter declined down to the literal loop), every filter against the reference's merge loop, systematic and
    stratified; odd Np: every filter at another 16-byte phase.  quick = "0" (FK_WHOLE_QUICK): the one-code-path kernel of
    rounds 3 / 4."""
    monkeypatch.setenv("FK_WHOLE_QUICK", quick)
    kinds = [k for k in _FAMILIES + ("dyadic", "tiny") if not (k in ("negative", "nan") and Np < 8)]
    _check_against_merge_loop(37, Np, kinds, range(37), monkeypatch, force=False)
    monkeypatch.setenv("FK_WHOLE_EXACT", "1")
    _check_against_merge_loop(37, Np, ("uniform", "negative", "zeros"), range(37), monkeypatch, force=False)


def test_whole_vector_kernel_positions_on_cumulative_sums(monkeypatch):
    """u chosen so that a position lands on / next to a cumulative sum (0 ... 2^-20 slots on either side): the estimates
    inside the error band send their vector to the exact round, the others are decided by the plain prefix sums -- both
    must give the merge loop's indices (the same construction runs on the host emulation, test_hostcheck_resample_math.py)"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    rs = np.random.RandomState(2024)
    for Np in (8000, 8192, 4096, 1000):
        Fn = 66
        w = rs.rand(Fn, Np) ** 1.5
        w /= w.sum(axis=1, keepdims=True)
        deltas = [0.0] + [s * 2.0 ** -k for k in (48, 40, 33, 26, 20) for s in (1, -1)]
        u = np.empty(Fn)
        us = rs.rand(Fn, Np)
        for f in range(Fn):
            cs = np.cumsum(w[f])
            j = rs.randint(Np // 4, Np - 1)
            t = Np * cs[j]
            i = int(np.floor(t))
            v = (t - i) + deltas[f % len(deltas)]
            u[f] = v if 0.0 <= v < 1.0 else 0.5
            us[f, i] = u[f]
        for strat, uu in ((0, u), (1, us)):
            dw, du = E.dev(w), E.dev(uu)
            idx = torch.full((Fn, Np), -7, dtype=torch.int32, device=dw.device)
            st = torch.zeros(Fn, dtype=torch.int32, device=dw.device)
            (E.resample_stratified if strat else E.resample_systematic)(Fn, Np, dw, du, idx, st)
            got = idx.cpu().numpy()
            for f in range(Fn):
                ref, over = (ro.stratified_c if strat else ro.systematic_c)(w[f], uu[f])
                ok = ref < Np
                assert np.array_equal(got[f][ok], ref[ok]), (Np, strat, f)


def test_whole_vector_kernel_many_filters(monkeypatch):
    """the C5 shape: 1000 x 8000 and 125 x 8000 in one launch each (more workgroups than CUs; two filters per CU),
    sampled filters bit-exact"""
    _check_against_merge_loop(1000, 8000, ("uniform",), range(1000), monkeypatch, force=False)       # (every filter)
    _check_against_merge_loop(1000, 8000, ("heavy_tail", "negative"), (0, 1, 255, 256, 511, 767, 768, 999), monkeypatch, force=False)
    _check_against_merge_loop(125, 8000, ("uniform", "zeros"), (0, 1, 63, 124), monkeypatch, force=False)
    _check_against_merge_loop(3000, 2000, ("uniform",), range(0, 3000, 7), monkeypatch, force=False)   # 256-thread workgroups, eight per CU


def test_local_kernel_on_a_long_vector_and_many_filters(monkeypatch):
    """FK_RESAMPLE_PATH=local forces a long ragged vector through the same kernel (hundreds of chunks per workgroup, a
    crossing every few); and the C5 shape's filter count in one launch (more workgroups than the chip holds at once)"""
    monkeypatch.setenv("FK_RESAMPLE_PATH", "local")
    _check_against_merge_loop(3, 300007, ("uniform", "heavy_tail", "zeros", "ties", "one_heavy"), range(3), monkeypatch, force=False)
    _check_against_merge_loop(1000, 8000, ("uniform", "heavy_tail"), (0, 1, 511, 767, 768, 999), monkeypatch, force=False)


def test_c5_multi_filter_8e6_particles():
    """BASELINE configs[4] read as 8e6 particles PER filter: several such filters in one call (one GPU's share is
    125), indices bit-exact against the merge loop on the first, a middle and the last filter, plus the fused
    posterior mean (the summary state the ranks all-gather) against numpy on those filters."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    Fn, Np, d = 6, 8_000_000, 4
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    w = torch.rand((Fn, Np), generator=g, device=dev, dtype=torch.float64)
    w /= w.sum(dim=1, keepdim=True)
    u = E.dev(np.random.RandomState(5).rand(Fn))
    particles = torch.randn((Fn, Np, d), generator=g, device=dev, dtype=torch.float64)
    idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
    st = torch.zeros(Fn, dtype=torch.int32, device=dev)
    mean = torch.empty((Fn, d), dtype=torch.float64, device=dev)
    E.resample_systematic(Fn, Np, w, u, idx, st)
    E.resample_gather_mean(Fn, Np, d, particles, idx, mean)
    assert not st.any()
    for f in (0, 3, Fn - 1):
        ref, over = ro.systematic_c(w[f].cpu().numpy(), float(u[f]))
        assert over == 0 and np.array_equal(idx[f].cpu().numpy(), ref), f
        want = particles[f].cpu().numpy()[ref].mean(axis=0)
        assert np.allclose(mean[f].cpu().numpy(), want, rtol=1e-11, atol=1e-13), f


def test_gather_mean_vs_numpy():
    """fk_resample_gather_mean_f64 (the fused "resample from index" + mean of BASELINE configs[4]):
    against numpy on the indices the resampler produced, ragged sizes, d = 1..8."""
    import torch
    from filterpy_amd import _engine as E
    rs = np.random.RandomState(12)
    for Fn, Np, d in ((1, 1, 1), (3, 1000, 4), (5, 16385, 3), (2, 70001, 8), (4, 8000, 6)):
        w = rs.rand(Fn, Np)
        w /= w.sum(axis=1, keepdims=True)
        u = rs.rand(Fn)
        parts = rs.randn(Fn, Np, d)
        dw, du, dp = E.dev(w), E.dev(u), E.dev(parts)
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dw.device)
        st = torch.zeros(Fn, dtype=torch.int32, device=dw.device)
        E.resample_systematic(Fn, Np, dw, du, idx, st)
        mean = torch.full((Fn, d), float("nan"), dtype=torch.float64, device=dw.device)
        E.resample_gather_mean(Fn, Np, d, dp, idx, mean)
        ih = idx.cpu().numpy()
        ref = np.stack([parts[f][ih[f]].mean(axis=0) for f in range(Fn)])
        assert np.allclose(mean.cpu().numpy(), ref, rtol=1e-11, atol=1e-13), (Fn, Np, d)
