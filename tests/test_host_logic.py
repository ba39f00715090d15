"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares,
the host shim validates inputs like the reference (test_kf.py:529-655 style), the product never
imports the oracle and fails loudly without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "filterhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_exactly_the_header():
    from filterpy_amd import _abi
    lib = _abi.lib()                      # raises if libfilterhip.so is missing
    declared = _header_functions()
    assert declared and set(declared) == set(_abi.SIGNATURES), (set(declared) ^ set(_abi.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _abi.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (fk_[a-z0-9_]+)", out)))
    assert exported == declared, set(exported) ^ set(declared)
    assert lib.fk_abi_version() == 4 and lib.fk_build_arch() == b"gfx950"


def test_argument_errors_without_gpu():
    """Entry points validate before launching anything: usable without a device."""
    import ctypes
    from filterpy_amd import _abi
    lib = _abi.lib()
    d = _abi.fk_kf_desc(n=0, m=1, nu=0, model_mode=0, N=1, T=1, layout=0, update_first=0, alpha_sq=1.0)
    assert lib.fk_kf_batch_filter_f64(ctypes.byref(d), *([None] * 16)) == -1
    d.n = 17
    one = ctypes.c_void_p(8)
    assert lib.fk_kf_predict_f64(ctypes.byref(d), one, one, None, None, one, one, None, None) == -2   # unsupported dim
    assert b"dim" in lib.fk_last_error()
    assert lib.fk_resample_systematic_f64(1, -1, None, None, None, None, None, 0, None) == -1
    assert lib.fk_ut_sigma_points_f64(33, 1, 0, 1.0, one, one, one, None, None) == -2


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "filterpy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                where = os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), where
                # ... nor loads, executes or names a file of it by path (ctypes.CDLL of the C restatement, subprocess, sys.path)
                assert not re.search(r"(CDLL|LoadLibrary|dlopen|Popen|subprocess|sys\.path|import_module|__import__)[^\n]*oracle", txt), where
                assert not re.search(r"oracle[/\\](_build|_ref|[a-z_]+\.(py|c|so))", txt), where


def test_no_cpu_fallback_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from filterpy_amd.kalman import KalmanFilter
    from filterpy_amd._abi import FilterHipError
    kf = KalmanFilter(2, 1)
    with pytest.raises(FilterHipError):
        kf.predict()
    from filterpy_amd.monte_carlo import systematic_resample
    with pytest.raises(FilterHipError):
        systematic_resample(np.ones(4) / 4)


def test_constructor_and_attributes_like_reference():
    """kalman_filter.py:387-434."""
    from filterpy_amd.kalman import KalmanFilter
    for bad in ((0, 1, 0), (1, 0, 0), (1, 1, -1)):
        with pytest.raises(ValueError):
            KalmanFilter(*bad)
    kf = KalmanFilter(3, 2, 1)
    assert kf.x.shape == (3, 1) and np.array_equal(kf.P, np.eye(3)) and np.array_equal(kf.Q, np.eye(3))
    assert kf.H.shape == (2, 3) and not kf.H.any() and np.array_equal(kf.R, np.eye(2)) and kf.B is None
    assert kf.K.shape == (3, 2) and kf.y.shape == (2, 1) and kf.S.shape == (2, 2) and kf.SI.shape == (2, 2)
    assert kf.z.shape == (2, 1) and kf.z[0, 0] is None and kf.alpha == 1.0 and kf.inv is np.linalg.inv
    kf.alpha = 1.02
    assert abs(kf._alpha_sq - 1.02 ** 2) < 1e-15
    with pytest.raises(ValueError):
        kf.alpha = 0.5


def test_reshape_z_rules():
    """filterpy/common/helpers.py:324-342 and the accept/reject matrix of test_kf.py:529-655."""
    from filterpy_amd.common import reshape_z
    assert reshape_z(3.0, 1, 2).shape == (1, 1) and reshape_z(3.0, 1, 1).shape == (1,) and reshape_z(3.0, 1, 0) == 3.0
    assert reshape_z([1, 2], 2, 2).shape == (2, 1) and reshape_z([[1, 2]], 2, 2).shape == (2, 1)
    assert reshape_z([[1], [2]], 2, 1).shape == (2,)
    for bad, dz in (([1, 2, 3], 2), ([[1, 2], [3, 4]], 2), ([1, 2], 1), ([[1, 2, 3]], 2)):
        with pytest.raises(ValueError):
            reshape_z(bad, dz, 2)


def test_update_none_is_bookkeeping_only():
    """kalman_filter.py:515-520: no arithmetic, so no GPU needed."""
    from filterpy_amd.kalman import KalmanFilter
    kf = KalmanFilter(2, 1)
    kf.x = np.array([1., 2.])
    kf.update(None)
    assert kf.z.shape == (1, 1) and kf.z[0, 0] is None and np.array_equal(kf.x_post, kf.x) and not kf.y.any()
    assert kf._log_likelihood is None and kf._mahalanobis is None


def test_merwe_weights_host_side():
    """sigma_points.py:180-192 (host-side scalars): compare with the golden from the live reference."""
    from conftest import golden
    from filterpy_amd.kalman import MerweScaledSigmaPoints, JulierSigmaPoints
    g = golden("ukf_merwe")
    for ci, (n, m, alpha, beta, kappa) in enumerate(g["cases"]):
        pts = MerweScaledSigmaPoints(int(n), alpha, beta, kappa)
        assert np.array_equal(pts.Wm, g[f"c{ci}_Wm"]) and np.array_equal(pts.Wc, g[f"c{ci}_Wc"])
        assert pts.num_sigmas() == 2 * int(n) + 1
    jp = JulierSigmaPoints(4, 0.5)
    assert np.array_equal(jp.Wm, g["jul_Wm"])
    # constructor hooks are kept like the reference keeps them (sigma_points.py:106-116)
    hp = MerweScaledSigmaPoints(2, .1, 2., 1., sqrt_method=np.linalg.cholesky)
    assert hp.sqrt is np.linalg.cholesky and hp.subtract is np.subtract and pts.subtract is np.subtract
    # without a callable the public attributes are scipy.linalg.cholesky like the reference's (sigma_points.py:106-109,
    # UKF.py:318-321): user code that calls points.sqrt(...) / ukf.msqrt(...) keeps working
    from scipy.linalg import cholesky
    from filterpy_amd.kalman import UnscentedKalmanFilter
    A = np.array([[4., 1.], [1., 3.]])
    assert np.array_equal(pts.sqrt(A), cholesky(A)) and np.array_equal(jp.sqrt(A), cholesky(A)) and pts._sqrt is None
    ukf = UnscentedKalmanFilter(2, 1, 1.0, hx=lambda x: x[:1], fx=lambda x, dt: x, points=hp)
    assert np.array_equal(ukf.msqrt(A), cholesky(A))
    assert UnscentedKalmanFilter(2, 1, 1.0, hx=None, fx=None, points=hp, sqrt_fn=np.linalg.cholesky).msqrt is np.linalg.cholesky


def test_chunk_plan_windows_tile_the_time_axis(monkeypatch):
    """fk_chunk_plan (csrc/fk_chunks.hpp through the C ABI, host arithmetic only): for every forced decomposition the
    H + 1 staggered windows of every track group tile [0, L) in order; the default policy only cuts calls whose last
    round of waves would be mostly idle."""
    import ctypes
    from filterpy_amd import _abi
    lib = _abi.lib()
    win = (ctypes.c_int64 * 130)()
    G, H = ctypes.c_int32(), ctypes.c_int32()

    def plan(N, L, tpw, slots, g):
        n = lib.fk_chunk_plan(N, L, tpw, slots, g, ctypes.addressof(win), ctypes.addressof(G), ctypes.addressof(H))
        return n, [(win[2 * i], win[2 * i + 1]) for i in range(max(n, 0))]
    for L in list(range(1, 40)) + [64, 99, 100, 128]:
        for g_req in (1, 2, 3, 4):
            for h_req in {1, 2, 3, 5, min(L, 17), min(L, 64)}:
                monkeypatch.setenv("FK_ML_CHUNKS", f"{g_req},{h_req}")
                for g in range(g_req):
                    n, w = plan(100000, L, 16, 2048, g)
                    if g_req == 1 and h_req == 1:
                        assert (n, w) == (1, [(0, L)])
                        continue
                    assert G.value == g_req and H.value == min(h_req, L) and 1 <= n <= H.value + 1, (L, g_req, h_req, g)
                    assert w[0][0] == 0 and w[-1][1] == L and all(a[1] == b[0] and a[0] < a[1] for a, b in zip(w, w[1:] + [(L, L + 1)]))
                assert plan(100000, L, 16, 2048, g_req)[0] == -1          # no such group
    monkeypatch.delenv("FK_ML_CHUNKS")
    # default policy: config 3 (6250 waves on 2048 slots: 3.05 rounds) is cut, 4e5 tracks (12.2 rounds: 20 % tail) too,
    # but not a bank that fills its last round, nor a short run, nor one that fits two rounds
    assert plan(100000, 100, 16, 2048, 0)[0] > 1 and (G.value, H.value) == (3, 4)
    assert plan(98304, 100, 16, 2048, 0) == (1, [(0, 100)])
    assert plan(100000 + 16 * 1024, 100, 16, 2048, 0) == (1, [(0, 100)])      # last round more than 40 % full
    assert plan(100000, 8, 16, 2048, 0) == (1, [(0, 8)])
    assert plan(60000, 100, 16, 2048, 0) == (1, [(0, 100)])
    assert plan(200000, 100, 64, 1024, 0)[0] > 1                               # kf_fast (8,4): 3125 waves of 64 on 1024 slots


def test_design_table_is_the_committed_evidence():
    """DESIGN.md section 5's per-kernel table and section 4's byte table are generated from profiles/r03 (tools/make_design_table.py):
    the document cannot drift from the measurement files"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "make_design_table.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fused_ukf_size_routing_without_gpu():
    """fk_ukf_linear_batch_f64 / fk_ukf_linear_rts_f64 decide which sizes they serve before anything is launched: the
    one-lane classes as before; dim_x 10..16 (the several-lane kernels, csrc/ukf_mlg.hip) with the pair-weight flag -- unless
    FK_UKF_MLG=0 takes them out (A/B; read once per process, so each setting gets its own interpreter).  fk_ukf_linear_supported
    must give the same answers without being handed a descriptor: the host side asks it before choosing a path."""
    code = r'''
import ctypes, sys
from filterpy_amd import _abi
lib = _abi.lib()
one = ctypes.c_void_p(8)
def fwd(n, m, flags, N=0):
    d = _abi.fk_ukf_desc(n=n, m=m, N=N, T=1, layout=0, flags=flags, scale=1.0)
    return lib.fk_ukf_linear_batch_f64(ctypes.byref(d), one, one, one, one, one, one, one, None, one, one, None, None, None, None)
def rts(n, flags, N=0):
    d = _abi.fk_ukf_desc(n=n, m=1, N=N, T=1, layout=0, flags=flags, scale=1.0)
    return lib.fk_ukf_linear_rts_f64(ctypes.byref(d), one, one, one, one, one, one, one, one, None, None, None)
PAIR = 1
out = [fwd(6, 3, 0), fwd(6, 4, 0), fwd(9, 4, PAIR), fwd(9, 5, PAIR), fwd(12, 3, PAIR), fwd(12, 3, 0), fwd(16, 8, PAIR), fwd(16, 9, PAIR), fwd(17, 2, PAIR),
       rts(9, PAIR), rts(12, PAIR), rts(12, 0), rts(17, PAIR)]
q = lib.fk_ukf_linear_supported
ask = [q(6, 3, 0, 0), q(6, 4, 0, 0), q(9, 4, PAIR, 0), q(9, 5, PAIR, 0), q(12, 3, PAIR, 0), q(12, 3, 0, 0), q(16, 8, PAIR, 0), q(16, 9, PAIR, 0), q(17, 2, PAIR, 0),
       q(9, 1, PAIR, 1), q(12, 1, PAIR, 1), q(12, 1, 0, 1), q(17, 1, PAIR, 1)]
assert [0 if v else -2 for v in ask] == out, (ask, out)
print(" ".join(str(v) for v in out))
'''
    def run(env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("FK_UKF_MLG")}
        e.update(env)
        return [int(v) for v in subprocess.check_output([sys.executable, "-c", code], text=True, cwd=ROOT, env=e).split()]
    OK, UNS = 0, -2
    #                  6x3  6x4  9x4  9x5  12x3p 12x3 16x8p 16x9 17x2 | r9  r12p r12  r17
    assert run({"FK_UKF_MLG": "0"}) == [OK, UNS, OK, UNS, UNS, UNS, UNS, UNS, UNS, OK, UNS, UNS, UNS]
    assert run({}) == [OK, UNS, OK, UNS, OK, UNS, OK, UNS, UNS, OK, OK, UNS, UNS]
    assert run({"FK_UKF_MLG_MIN_NX": "7", "FK_UKF_MLG_RTS_MIN_NX": "10"}) == [OK, UNS, OK, UNS, OK, UNS, OK, UNS, UNS, OK, OK, UNS, UNS]


def test_a_strided_mask_is_copied_before_its_pointer_is_handed_to_the_c_abi():
    """The C ABI reads a [T][N] byte mask with row stride N.  NumPy's mask[:, idx] is column-major memory and
    torch.as_tensor keeps the strides: _engine._mask_ptr copies such a view (found by a GPU probe that fed a kernel a transposed
    mask, profiles/r04/lease_q) and leaves a contiguous one alone."""
    import ctypes
    import torch
    from filterpy_amd import _engine as E
    base = np.arange(12, dtype=np.uint8).reshape(4, 3)
    view = base[:, np.array([2, 0, 1, 1, 0])]                   # shape (4, 5), strides (1, 4)
    assert not view.flags.c_contiguous
    t = torch.as_tensor(view)
    assert not t.is_contiguous()
    keep = []
    p = E._mask_ptr(t, keep)
    assert len(keep) == 1 and keep[0].is_contiguous() and p == keep[0].data_ptr() != t.data_ptr()
    got = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(20,))
    assert np.array_equal(got, np.ascontiguousarray(view).ravel())
    c = torch.as_tensor(np.ascontiguousarray(view))
    keep = []
    assert E._mask_ptr(c, keep) == c.data_ptr() and E._mask_ptr(None, keep) is None


def test_quick_resampler_spills_only_in_its_unlikely_tail():
    """resample_whole_quick_kernel (csrc/resample_whole.hip) is worth its name only while the COMMON path stays inside the 64
    VGPRs that let two 1024-thread workgroups share a CU, with no scratch access: the whole algorithm hangs behind the common
    path's `return` as an unlikely tail and may spill all it wants.  Read from the built object: 64 VGPRs at 1024 / 512 threads,
    and no scratch_ instruction in front of the kernel's first s_endpgm (= the end of the common path)."""
    import re
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    obj = os.path.join(ROOT, "filterpy_amd", "csrc", "build", "resample_whole.o")
    if not os.path.exists(obj):
        pytest.skip("library not built")
    with tempfile.TemporaryDirectory() as tmp:
        elf = isa_lint.device_elf(obj, tmp)
        info = isa_lint.kernels(elf)
        dis = subprocess.check_output([f"{isa_lint.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], text=True)
    quick = {k: v for k, v in info.items() if "resample_whole_quick_kernel" in k}
    assert len(quick) == 6, sorted(info)
    for name, v in quick.items():
        if "Li1024E" in name or "Li512E" in name:
            assert v["vgpr"] <= 64, (name, v)
        body = dis.split("<" + name + ">:")[1]
        first_exit = body.index("s_endpgm")
        assert "scratch_" not in body[:first_exit], name
        assert len(re.findall(r"s_barrier", body[:first_exit])) >= 4, name        # (1), (A), (6), (7): the common path is all there


def test_pmc_evidence_still_describes_the_built_kernel():
    """bench.py's `roofline.traffic` is read from COMMITTED rocprofv3 --pmc passes (profiles/pmc_traffic.json), not measured in
    the run: this test ties that evidence to the library as built -- the kernel the counters were collected
    on must still exist under that name, with the workgroup size, LDS block, scratch size and register count the counter rows
    recorded.  A kernel change that could move the traffic fails here until the PMC passes are re-run."""
    import csv
    import json
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    if not os.path.isdir(os.path.join(ROOT, "filterpy_amd", "csrc", "build")):
        pytest.skip("library not built here (the objects do not travel with the .so)")
    checked = 0
    for layout in ("aos", "soa", "aos_interleave"):
        for key in ("fetch_csv", "write_csv"):
            path = rec.get(layout, {}).get(key)
            if not path:
                continue
            rows = [r for r in csv.DictReader(open(os.path.join(ROOT, path))) if "kf_fast_kernel" in r["Kernel_Name"]]
            assert rows, path
            r = rows[-1]
            m = re.match(r"void fk::(fastv_\d+_\d+_v\d+)::kf_fast_kernel<([^>]*)>", r["Kernel_Name"])
            assert m, r["Kernel_Name"]
            targs = ",".join({"false": "0", "true": "1"}.get(t.strip(), t.strip()) for t in m.group(2).split(","))
            nx, nz = targs.split(",")[:2]
            objs = [f for f in os.listdir(os.path.join(ROOT, "filterpy_amd", "csrc", "build")) if f.startswith(f"inst_fast_{nx}_{nz}_{m.group(1)[-1]}_") and f.endswith(".o")]
            assert objs, (nx, nz)
            with tempfile.TemporaryDirectory() as tmp:
                info = isa_lint.kernels(isa_lint.device_elf(os.path.join(ROOT, "filterpy_amd", "csrc", "build", objs[0]), tmp))
            kinds = "iiibbbibbbb"                   # NX, NZ, LAYOUT, HAS_MASK, OUTS, SYM, MMODE, UF, CTRL, EX, IL (kf_fast.hip)
            want = "kf_fast_kernelI" + "".join(f"L{k}{t}E" for k, t in zip(kinds, targs.split(","))) + "E"
            hit = [v for k, v in info.items() if want in k and m.group(1) in k]
            assert len(hit) == 1, (want, [k for k in info if "kf_fast" in k][:4])
            v = hit[0]
            assert int(r["Workgroup_Size"]) == 256
            assert int(r["Scratch_Size"]) == v["scratch"], (r["Scratch_Size"], v)
            assert int(r["LDS_Block_Size"]) == -(-v["lds"] // 512) * 512, (r["LDS_Block_Size"], v)
            assert int(r["VGPR_Count"]) * 2 == -(-v["vgpr"] // 8) * 8, (r["VGPR_Count"], v)        # (rocprofv3 reports half of the unified file)
            checked += 1
    assert checked >= 2


def test_counted_waits_behind_lds_dma_requests_have_their_stores():
    """the kernels that fetch the next measurement by LDS-DMA wait for it with a hand-counted `s_waitcnt vmcnt(K)`
    (K = the store instructions a step issues behind the request, written in the source as a lower bound).  The compiler never
    sees the DMA, so nothing but this test ties K to what it actually emitted: in every built instantiation with such a loop
    (IMM banks, kf_fast at dim_x >= 7; the several-lane filters' time loops hold rolled copy-out loops, which a static count
    cannot price) the vector-memory instructions behind the request must number at least K -- fewer, and the wait returns with
    the DMA in flight and the step reads a stale image."""
    import glob
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    build = os.path.join(ROOT, "filterpy_amd", "csrc", "build")
    if not os.path.isdir(build):
        pytest.skip("library not built here (the objects do not travel with the .so)")
    objs = sorted(set(glob.glob(os.path.join(build, "inst_imm_2_1_*_p[1256].o")) + glob.glob(os.path.join(build, "inst_imm_4_2_*_p[1256].o"))
                      + glob.glob(os.path.join(build, "inst_imm_6_3_*_p[1256].o")) + glob.glob(os.path.join(build, "inst_imm_9_4_[234]_*_p[1256].o"))
                      + glob.glob(os.path.join(build, "inst_fast_[789]_*.o")) + glob.glob(os.path.join(build, "inst_mlg_1[06]_*.o"))
                      + glob.glob(os.path.join(build, "ukf_mlg_1[06].o"))))
    assert len(objs) > 40, len(objs)
    seen = 0
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            for name, (K, younger, checkable) in isa_lint.dma_wait_margins(isa_lint.device_elf(o, tmp)).items():
                if not checkable:        # (stores inside rolled copy-out loops: a static count sees them once, not per trip)
                    continue
                assert younger >= K, (os.path.basename(o), isa_lint.short(name), K, younger)
                seen += 1
    assert seen >= 30, seen


def test_no_specialised_instantiation_spills_under_a_launch_bound_of_its_own_making():
    """Round 5 (docs/KERNEL_NOTES.md, "Every one-lane shape measured"): the element-major (3,2) / (3,3) kernels ran at 0.34 / 0.15 of
    HBM because `FK_FAST_INST`'s occupancy target left them 80 VGPRs and the rest went to scratch -- unnoticed for three rounds,
    since neither shape was benchmarked.  What made them visible is one line of tools/isa_lint.py, so that line is a test: no
    kernel of the one-lane specialised family (kf_fast.hip) whose register budget is a launch bound of OURS (fewer than the
    256 VGPRs of two waves per SIMD) may carry more than a few words of scratch in the call shape the tables serve -- plain or
    masked, all four outputs, shared model.  (At 256 / 512 VGPRs scratch is the kernel's size, not the bound's: (6,3), (9,x) --
    listed in docs/KERNEL_NOTES.md, not asserted here.)"""
    import glob
    import re
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    objs = sorted(glob.glob(os.path.join(ROOT, "filterpy_amd", "csrc", "build", "inst_fast_*.o")))
    if not objs:
        pytest.skip("library not built")
    seen, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            for name, v in isa_lint.kernels(isa_lint.device_elf(obj, tmp)).items():
                m = re.search(r"kf_fast_kernel<([\d,]+)>", isa_lint.short(name))
                if not m:
                    continue
                t = [int(x) for x in m.group(1).split(",")]      # NX, NZ, LAYOUT, HAS_MASK, OUTS, SYM, MMODE, UF, CTRL, EX, IL
                if t[4] == 1 and t[6] == 0 and t[7] == 0 and t[8] == 0 and t[9] == 0:
                    seen += 1
                    if v["vgpr"] < 256 and v["scratch"] > 64:
                        bad.append((tuple(t), v["vgpr"], v["scratch"]))
    assert seen >= 120, seen                                      # 30 shapes x 2 record orders x {plain, masked} (+ the IL twins)
    assert not bad, bad


def test_two_stage_build_lists_are_consistent():
    """csrc/Makefile links the library twice in a cold build: stage 1 from every object except the slow unrolled IMM classes, with
    rolled stand-ins (build/quick/) in their place -- a complete library half-way through --, stage 2 as shipped.  The lists
    the two link lines are made of: every slow object has exactly one stand-in of the same name, no object is linked twice, and
    the two link lines differ in nothing else."""
    import subprocess
    csrc = os.path.join(ROOT, "filterpy_amd", "csrc")
    db = subprocess.run(["make", "-C", csrc, "-pnq"], capture_output=True, text=True).stdout

    def var(name):
        for line in db.splitlines():
            if line.startswith(name + " := ") or line.startswith(name + " = "):
                return line.split("=", 1)[1].split()
        raise AssertionError(name + " not in the Makefile's database")
    base, slow, quick, objs = var("BASE_OBJS"), var("SLOW_OBJS"), var("QUICK_OBJS"), var("OBJS")
    assert len(set(objs)) == len(objs) and set(objs) == set(base) | set(slow) and not set(base) & set(slow)
    assert sorted(q.replace("build/quick/", "build/") for q in quick) == sorted(slow) and len(slow) % 8 == 0
    # the slow classes are exactly the ones built with the general kernel only (all eight parts of each)
    classes = sorted({o.rsplit("_p", 1)[0] for o in slow})
    assert all(sum(o.startswith(c + "_p") for o in slow) == 8 for c in classes)
    # where the shipped library and the stage stamp both exist, the library is stage 2's (stage 1 dates its own before the stamp)
    lib, stamp = os.path.join(ROOT, "filterpy_amd", "libfilterhip.so"), os.path.join(csrc, "build", "stage1.stamp")
    if os.path.exists(lib) and os.path.exists(stamp) and os.path.getmtime(lib) < os.path.getmtime(stamp):
        import warnings
        warnings.warn("filterpy_amd/libfilterhip.so is a STAGE 1 library (rolled stand-ins for the slow IMM classes): the build "
                      "was cut short; `make -C filterpy_amd/csrc -j` finishes it")


def test_every_entry_point_refuses_null_and_nonsense_without_touching_a_device():
    """All 31 symbols of include/filterhip.h, every pointer NULL and every size 0 / -1 / 1 / 33 in turn: each call returns
    FK_OK (nothing to do), FK_ERR_BAD_ARG or FK_ERR_UNSUPPORTED -- none dereferences a NULL pointer, none needs a GPU to say
    no.  One child process for all 100-odd calls (a crash would take the test runner with it)."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
from filterpy_amd import _abi
lib = _abi.lib()
bad = []
for name in _abi.SIGNATURES:
    if name in ("fk_abi_version", "fk_build_arch", "fk_last_error"):
        continue
    fn = getattr(lib, name)
    for v in (0, -1, 1, 33):
        keep, args = [], []
        for t in fn.argtypes:
            if t is ctypes.c_void_p:
                args.append(None)
            elif t is ctypes.c_double:
                args.append(1.0)
            elif t in (ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t):
                args.append(v)
            else:
                st = t._type_()
                for f, ft in st._fields_:
                    setattr(st, f, 1.0 if ft is ctypes.c_double else v)
                keep.append(st)
                args.append(ctypes.byref(st))
        r = fn(*args)
        sized = name.endswith("workspace_bytes") or name in ("fk_ukf_linear_supported", "fk_chunk_plan")
        if not sized and r not in (0, -1, -2):
            bad.append((name, v, r))
print("CALLS_DONE", bad)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    assert "CALLS_DONE []" in r.stdout, r.stdout[-400:]


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/filterhip.h is the boundary a non-Python host binds: it compiles as strict C99 and as C++11 without
    a warning, and a plain C program links against libfilterhip.so and calls into it (no HIP header, no GPU needed for that)."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text('#include "filterhip.h"\n#include <string.h>\n'
                   'int main(void) { fk_kf_desc d; memset(&d, 0, sizeof d);\n'
                   '  if (fk_abi_version() != 4 || strcmp(fk_build_arch(), "gfx950")) return 1;\n'
                   '  return fk_kf_batch_filter_f64(&d, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == -1 ? 0 : 2; }\n')
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "filterpy_amd")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        subprocess.check_call([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-x", lang, "-c", str(src),
                               "-o", str(tmp_path / ("host_" + lang.replace("+", "x") + ".o"))])
    exe = tmp_path / "host"
    subprocess.check_call(["gcc", str(tmp_path / "host_c.o"), "-L", libdir, "-lfilterhip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    assert subprocess.run([str(exe)], timeout=120).returncode == 0


def test_ctypes_signatures_match_the_header_parameter_by_parameter():
    """filterpy_amd/_abi.py binds by hand: every prototype of include/filterhip.h is parsed and compared with the ctypes
    signature -- parameter count, and the kind of every parameter (pointer / int32 / int64 / size_t / double / which struct) and of
    the return value -- and the three structs field by field (name, type, order).  A transposed or mistyped argument in a
    binding would hand the kernels garbage without any error."""
    import ctypes
    from filterpy_amd import _abi
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "filterhip.h")).read(), flags=re.S)
    kinds = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "double": ctypes.c_double,
             "int": ctypes.c_int}
    structs = {"fk_kf_desc": _abi.fk_kf_desc, "fk_ukf_desc": _abi.fk_ukf_desc, "fk_imm_desc": _abi.fk_imm_desc,
               "fk_kf_extras": _abi.fk_kf_extras}
    same_size = lambda a, b: ctypes.sizeof(a) == ctypes.sizeof(b)  # noqa: E731

    def kind(decl):
        decl = decl.replace("const", " ").strip()
        if decl == "void":
            return None
        if "*" in decl:
            base = decl.split("*")[0].split()[0]
            return ctypes.POINTER(structs[base]) if base in structs else ctypes.c_void_p
        return kinds[decl.split()[0]]
    protos = re.findall(r"^\s*(int|size_t|const char \*)\s*(fk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S)
    assert len(protos) == len(_abi.SIGNATURES), (len(protos), len(_abi.SIGNATURES))
    for ret, name, params in protos:
        restype, argtypes = _abi.SIGNATURES[name]
        want = [kind(p) for p in params.replace("\n", " ").split(",")]
        want = [w for w in want if w is not None]
        assert len(want) == len(argtypes), (name, len(want), len(argtypes))
        for i, (w, a) in enumerate(zip(want, argtypes)):
            if w is ctypes.c_void_p or (isinstance(w, type) and issubclass(w, ctypes._Pointer)):
                ok = (a is ctypes.c_void_p) if w is ctypes.c_void_p else (a is w or a is ctypes.c_void_p and False)
                assert ok, (name, i, w, a)
            else:
                assert same_size(w, a) and (w is ctypes.c_double) == (a is ctypes.c_double), (name, i, w, a)
        if ret == "const char *":
            assert restype is ctypes.c_char_p, name
        else:
            assert same_size(kinds[ret], restype), (name, ret, restype)
    for sname, cls in structs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (sname, sname), src, flags=re.S).group(1)
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            typ = stmt.split()[0]
            for nm in stmt[len(typ):].split(","):
                nm = nm.strip()
                fields.append((nm.lstrip("*").strip(), ctypes.c_void_p if nm.startswith("*") else kinds[typ]))
        got = [(f, t) for f, t in cls._fields_]
        assert [f for f, _ in fields] == [f for f, _ in got], (sname, fields, got)
        for (f, w), (_, a) in zip(fields, got):
            assert same_size(w, a) and (w is ctypes.c_double) == (a is ctypes.c_double), (sname, f, w, a)


def test_engine_wrappers_hand_every_operand_to_the_parameter_of_its_name(monkeypatch):
    """filterpy_amd/_engine.py is the one place where Python operands become positional C arguments.  Every wrapper is called with
    a distinct tensor per operand and a recording stand-in for the library; each recorded pointer must sit at the position of the
    header parameter that carries the operand's NAME (five documented aliases), each descriptor field must hold the wrapper's
    value of that name.  (The GPU suite proves the same by results; this one says which operand went astray, on the CPU.)"""
    import ctypes
    import inspect
    import torch
    from filterpy_amd import _abi, _engine as E
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "filterhip.h")).read(), flags=re.S)
    protos = {n: [p.strip().split()[-1].lstrip("*") for p in ps.replace("\n", " ").split(",")]
              for _, n, ps in re.findall(r"^\s*(int|size_t|const char \*)\s*(fk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S)}
    alias = {"noise": "noise_cov", "sig_in": "in", "sig_out": "out", "convention": "index_convention", "y": "y_out"}
    calls = []

    class Lib:
        def __getattr__(self, name):
            def fn(*a):
                calls.append((name, a))
                return 64 if name.endswith("workspace_bytes") else (1 if name == "fk_ukf_linear_supported" else 0)
            return fn
    monkeypatch.setattr(_abi, "lib", lambda: Lib())
    monkeypatch.setattr(E, "_stream", lambda: 4321)
    ints = dict(n=3, m=2, k=7, N=5, T=4, Fn=6, Np=8, Nu=9, d=4, n_in=3, n_out=2, n_models=2, nu=1, phase=2, scale=1.5)
    desc_kw = dict(n=3, m=2, nu=1, model_mode=3, N=5, T=4, layout=1, update_first=1, alpha_sq=1.25, flags=2)
    wrappers = {"kf_batch_filter": "fk_kf_batch_filter_f64", "kf_batch_filter_ex": "fk_kf_batch_filter_ex_f64",
                "kf_predict": "fk_kf_predict_f64", "kf_update": "fk_kf_update_f64", "kf_rts": "fk_kf_rts_f64",
                "ut_sigma_points": "fk_ut_sigma_points_f64", "ut_transform": "fk_ut_transform_f64",
                "ut_cross_variance": "fk_ut_cross_variance_f64", "ut_linear_map": "fk_ut_linear_map_f64",
                "ukf_correct": "fk_ukf_correct_f64", "ukf_linear_batch": "fk_ukf_linear_batch_f64",
                "ukf_linear_rts": "fk_ukf_linear_rts_f64", "kf_steadystate": "fk_kf_steadystate_f64",
                "kf_update_correlated": "fk_kf_update_correlated_f64", "ukf_rts_correct": "fk_ukf_rts_correct_f64",
                "imm_batch": "fk_imm_batch_ex_f64", "resample_systematic": "fk_resample_systematic_f64",
                "resample_stratified": "fk_resample_stratified_f64", "resample_multinomial": "fk_resample_multinomial_f64",
                "resample_residual_fill": "fk_resample_residual_fill_f64", "resample_residual_draw": "fk_resample_residual_draw_f64",
                "resample_gather_mean": "fk_resample_gather_mean_f64", "cumsum_exact": "fk_cumsum_exact_f64"}
    checked = 0
    for wname, cname in wrappers.items():
        fn = getattr(E, wname)
        kw, tensors = {}, {}
        for p in inspect.signature(fn).parameters:
            if p == "desc_kw":
                kw[p] = dict(desc_kw)
            elif p == "extras":
                kw[p] = {k: torch.zeros(3, dtype=torch.float64) for k in ("y", "K", "S", "SI", "log_likelihood", "mahalanobis")}
            elif p == "layout":
                kw[p] = "soa"
            elif p in ints and not (p == "k" and wname.startswith("resample_residual")):       # (there k is the count array)
                kw[p] = ints[p]
            elif p in ("mmae", "force_last_one"):
                kw[p] = True
            elif p == "paired":
                kw[p] = True
            elif p in ("convention",):
                kw[p] = 1
            else:
                tensors[p] = kw[p] = torch.zeros(3, dtype=torch.uint8 if p in ("mask", "zmask") else torch.float64)
        calls.clear()
        fn(**kw)
        rec = [a for nm, a in calls if nm == cname]
        assert len(rec) == 1, (wname, [nm for nm, _ in calls])
        args, cparams = rec[0], protos[cname]
        assert len(args) == len(cparams), (wname, len(args), len(cparams))
        for pname, t in tensors.items():
            c = pname if pname in cparams else alias.get(pname, pname)
            assert c in cparams, (wname, pname, cparams)
            assert args[cparams.index(c)] == t.data_ptr(), (wname, pname, "went to", [cparams[i] for i, a in enumerate(args) if a == t.data_ptr()])
            checked += 1
        assert args[cparams.index("stream")] == 4321, wname
        for c, a in zip(cparams, args):
            if c in ints and c in kw and not isinstance(kw[c], torch.Tensor) and not isinstance(a, ctypes.Structure):
                assert a == kw[c], (wname, c, a)
            if c == "index_convention":
                assert a == 1, wname
        if "desc" in cparams:
            d = args[cparams.index("desc")]
            want = kw.get("desc_kw") or dict(n=3, m=2, N=5, T=4, layout=E.LAYOUTS["soa"], scale=1.5, n_models=2, phase=2)
            for f, _ in d._fields_:
                if f in want and not (wname == "ukf_linear_rts" and f == "m"):       # (the smoother has no dim_z: m = 1)
                    assert getattr(d, f) == want[f], (wname, "desc." + f, getattr(d, f), want[f])
            if wname == "imm_batch":
                assert d.flags == 1
            if wname.startswith("ukf_linear"):
                assert d.flags == _abi.FK_UKF_FLAG_PAIR_WEIGHTS
        if "extras" in cparams:
            ex = args[cparams.index("extras")]
            for k, t in kw["extras"].items():
                assert getattr(ex, k) == t.data_ptr(), (wname, "extras." + k)
    assert checked > 150


def test_committed_bench_line_keeps_the_drivers_contract():
    """profiles/r06/bench_default.json is a line bench.py printed on an MI355X: it carries every key the driver's contract names
    (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
    config.workload, the roofline object and the cpu_baseline object), its numbers are consistent with one another (value =
    units / time, roofline.achieved = algorithmic bytes / kernel time, frac = achieved / peak, kernel time <= step time), the
    metric is BASELINE.json's, and the source of bench.py still prints each of those keys."""
    import json
    with open(os.path.join(ROOT, "profiles", "r06", "bench_default.json")) as fh:
        d = json.load(fh)
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        base = json.load(fh)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert "track-steps" in d["metric"] and "track-steps" in json.dumps(base)
    r, c = d["roofline"], d["cpu_baseline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference")
    N, T = d["config"]["tracks_per_gpu"], d["config"]["T"]
    assert abs(d["value"] - N * T * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) <= 1e-9 * d["value"]
    alg = 8.0 * (2 + 2 * 4 + 2 * 16) * N * T + 2 * 8.0 * (4 + 16) * N               # 336 B per track-step + 320 B per track once
    assert abs(r["algorithmic_bytes_per_launch"] - alg) < 1.0
    assert abs(r["achieved"] - alg / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-9 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0 and 0.5 < r["frac"] < 1.0
    assert r["kernel_ms"] <= d["ms_per_step"] and (r["traffic"] is None or 0.95 * alg < r["traffic"] < 1.1 * alg)
    assert d["parity_max_rel_vs_oracle"] < 1e-10 and c["value"] > 0 and c["cores"] >= 1
    # round 6: BASELINE configs[2], [3], [4] ride in the same line -- every row consistent with itself
    rows = d["configs"]
    names = " | ".join(r_["name"] for r_ in rows)
    for want in ("C3 kf batch_filter (9,3)", "C3 rts_smoother", "C4 fused linear UKF (6,3)", "1000 filters x 8000", "125 filters x 8000000"):
        assert want in names, want
    for r_ in rows:
        assert set(("name", "kernel", "kernel_ms", "algorithmic_bytes", "frac", "launches_timed")) <= set(r_), r_
        assert r_["launches_timed"] >= 10 and 0.0 < r_["frac"] < 1.0
        assert abs(r_["frac"] - r_["algorithmic_bytes"] / (r_["kernel_ms"] * 1e-3) / 8e12) <= 1e-6
        assert r_.get("bit_exact") is True or r_["parity_max_rel"] < 1e-10, r_
    assert d["value_unplaced"] > 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"',
              '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"traffic"', '"frac"', '"kind"', '"sample"', '"cores"'):
        assert k in src, k


def test_kernels_above_the_instruction_cache_are_the_listed_ones():
    """VERDICT r5 next 7 asked for tools/isa_lint.py in a CPU test: the instruction cache of a CU pair holds 64 KB, and a kernel whose
    code is larger streams (part of) its time loop from L2 every step.  The BASELINE kernels -- kf_fast, kf_ml / rts_ml (9,3), the fused
    UKF up to dim_x 8, the one-pass and the local resamplers, every building block -- are below; the families known to be above are
    listed here with the bound each is held to (DESIGN section 9 / docs/KERNEL_NOTES.md say what would take them below), so that a
    kernel that GROWS past the cache, or a new family, fails this test instead of showing up as a slow row two rounds later.
    (Total code per kernel, not the loop alone: an upper bound, and the quick resampler is listed for its unlikely exact tail.)"""
    import glob
    import re
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    objs = sorted(glob.glob(os.path.join(ROOT, "filterpy_amd", "csrc", "build", "*.o")))
    if len(objs) < 100 or not os.path.exists(f"{isa_lint.LLVM}/llvm-readelf"):
        pytest.skip("no built objects here")
    allowed = [            # (family, what its template arguments must satisfy, bytes it is held to)
        (r"imm_kernel<(\d+),(\d+),(\d+),", lambda a: a == [6, 3, 3], 90_000),
        (r"imm_quad_kernel<(\d+),(\d+),", lambda a: a == [16, 8], 160_000),
        # (the EXTENDED instantiation only -- MMAE, missing measurements, control input, single-phase calls as run-time branches;
        #  the plain (9,4) kernels are 45 KB)
        (r"imm_lanes_kernel<(\d+),(\d+),(\d+),(\d+)>", lambda a: a[:2] == [9, 4] and a[3] == 1, 82_000),
        (r"kf_mlg_kernel<(\d+),(\d+),", lambda a: a[0] >= 12 and a[1] >= 4, 120_000),
        (r"rts_mlg_kernel<(\d+),", lambda a: a[0] >= 14, 112_000),
        (r"ukf_mlg_rts_kernel<(\d+),", lambda a: a[0] >= 13, 108_000),
        (r"ukf_correct_kernel<(\d+),(\d+),", lambda a: a == [16, 8], 104_000),
        (r"ukf_linear_rts_kernel<(\d+),", lambda a: a[0] == 9, 80_000),
        (r"resample_whole_quick_kernel<", lambda a: True, 80_000),
        (r"kf_kernel<(\d+),(\d+),", lambda a: a[0] >= 6, 98_000),
    ]
    seen, over = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            try:
                elf = isa_lint.device_elf(o, tmp)
            except RuntimeError:
                continue                                          # a host-only object
            for name, k in isa_lint.kernels(elf).items():
                seen += 1
                code = k.get("code", 0)
                if code <= 65536:
                    continue
                sh = isa_lint.short(name)
                for pat, ok, bound in allowed:
                    m = re.search(pat, sh)
                    if m and ok([int(g) for g in m.groups()]) and code <= bound:
                        break
                else:
                    over.append((sh, code, os.path.basename(o)))
            os.remove(elf)
    assert seen > 1000, seen
    assert not over, over[:10]

