"""-m gpu parity tests: the HIP Kalman kernels, called through the C ABI, against the goldens
frozen from the live reference and against the NumPy oracle on seeded random banks."""
import numpy as np
import pytest

from conftest import golden, rel_err_rows
from oracle import kf_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-10      # BASELINE.json north_star: x/P within 1e-10 rel fp64 (normwise per matrix)
DIMS = [tuple(d) for d in golden("kf_dims")["dims"]]
NTRK = 300       # two workgroups, the second partially filled


def _per_track(a):
    """(T,N,...) -> worst normwise error helper expects leading axis = samples"""
    return a.reshape((-1,) + a.shape[2:])


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", DIMS)
def test_batch_filter_goldens(n, m, layout):
    from gpu_util import run_kf_batch, tile_tracks
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    for variant in ("plain", "uf", "alpha", "miss", "ctrl"):
        kw = {}
        if variant == "uf":
            kw["update_first"] = True
        if variant == "alpha":
            kw["alpha_sq"] = 1.02 ** 2
        if variant == "miss":
            kw["mask"] = tile_tracks(g[p + "mask"], NTRK, 1)
        if variant == "ctrl":
            kw["B"] = g[p + "B"]
            kw["us"] = tile_tracks(g[p + "us"], NTRK, 1)
        mu, cov, mup, covp, xf, Pf, st = run_kf_batch(
            tile_tracks(g[p + "x0"], NTRK), tile_tracks(g[p + "P0"], NTRK), tile_tracks(g[p + "zs"], NTRK, 1),
            g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"], layout=layout, **kw)
        for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
            ref = g[p + variant + "_" + key]
            for trk in (0, 63, 64, 255, 256, NTRK - 1):
                assert rel_err_rows(got[:, trk], ref) < TOL, (variant, key, trk)
            # every track saw the same inputs -> identical bits
            assert np.array_equal(got, np.broadcast_to(got[:, :1], got.shape)), (variant, key)
        if variant != "miss":
            assert rel_err_rows(xf[:1], g[p + variant + "_xfinal"].reshape(1, -1)) < TOL
            assert rel_err_rows(Pf[:1], g[p + variant + "_Pfinal"][None]) < TOL


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (6, 3), (9, 3)])
def test_batch_filter_per_step_models(n, m, layout):
    """Fs/Qs/Hs/Rs lists (kalman_filter.py:941-952): PER_STEP (shared) and PER_TRACK_STEP."""
    from gpu_util import run_kf_batch, tile_tracks
    from filterpy_amd._abi import FK_MODEL_PER_STEP, FK_MODEL_PER_TRACK_STEP
    g = golden("kf_models")
    p = f"n{n}m{m}_"
    N = 130
    x0, P0, zs = tile_tracks(g[p + "x0"], N), tile_tracks(g[p + "P0"], N), tile_tracks(g[p + "zs"], N, 1)
    for mode in (FK_MODEL_PER_STEP, FK_MODEL_PER_TRACK_STEP):
        mods = [g[p + k] for k in ("Fs", "Qs", "Hs", "Rs")]
        if mode == FK_MODEL_PER_TRACK_STEP:
            mods = [tile_tracks(M, N, 1) for M in mods]
        mu, cov, mup, covp, *_ = run_kf_batch(x0, P0, zs, *mods, layout=layout, mode=mode)
        for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
            for trk in (0, 64, N - 1):
                assert rel_err_rows(got[:, trk], g[p + key]) < TOL, (mode, key, trk)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (3, 2), (6, 3)])
def test_batch_filter_random_bank_vs_oracle(n, m, layout):
    """Every track different (x0, P0, z, and per-track models) against the NumPy oracle."""
    from gpu_util import run_kf_batch
    from filterpy_amd._abi import FK_MODEL_PER_TRACK, FK_MODEL_SHARED
    rs = np.random.RandomState(100 * n + m)
    N, T = 777, 25

    def spd(k, s):
        A = rs.randn(N, k, k)
        return s * (A @ A.transpose(0, 2, 1) / k + 0.5 * np.eye(k))
    x0, P0 = rs.randn(N, n), spd(n, 5.0)
    zs = rs.randn(T, N, m) * 3
    F = np.eye(n) + 0.1 * rs.randn(N, n, n)
    Q, H, R = spd(n, 0.1), rs.randn(N, m, n), spd(m, 0.5)
    mask = rs.rand(T, N) > 0.2
    sample = [0, 1, 63, 64, 255, 256, 511, 512, N - 1]
    for mode in (FK_MODEL_SHARED, FK_MODEL_PER_TRACK):
        mods = (F, Q, H, R) if mode == FK_MODEL_PER_TRACK else (F[0], Q[0], H[0], R[0])
        got = run_kf_batch(x0, P0, zs, *mods, layout=layout, mode=mode, mask=mask)
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, *mods, tracks=sample,
                                               model_mode=1 if mode == FK_MODEL_PER_TRACK else 0, mask=mask)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (mode, k)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", DIMS)
def test_rts_goldens(n, m, layout):
    from gpu_util import run_rts, tile_tracks
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    N = 100
    Xs, Ps = tile_tracks(g[p + "plain_mu"], N, 1), tile_tracks(g[p + "plain_cov"], N, 1)
    out = run_rts(Xs, Ps, g[p + "F"], g[p + "Q"], layout=layout)
    for got, key in zip(out, ("rts_x", "rts_P", "rts_K", "rts_Pp")):
        for trk in (0, 64, N - 1):
            assert rel_err_rows(got[:, trk], g[p + key]) < 1e-10, (key, trk)


@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (9, 3)])
def test_rts_index_conventions(n, m):
    """class method uses F[k+1],Q[k+1] (kalman_filter.py:1067); module function F[k],Q[k] (:1851)."""
    from gpu_util import run_rts, tile_tracks
    from filterpy_amd._abi import FK_MODEL_PER_STEP
    g = golden("kf_models")
    p = f"n{n}m{m}_"
    N = 70
    for conv, src, pre in ((0, "", "rts_"), (1, "mod_", "rtsm_")):
        Xs, Ps = tile_tracks(g[p + src + "mu"], N, 1), tile_tracks(g[p + src + "cov"], N, 1)
        out = run_rts(Xs, Ps, g[p + "Fs"], g[p + "Qs"], mode=FK_MODEL_PER_STEP, convention=conv)
        for got, key in zip(out, ("x", "P", "K", "Pp")):
            assert rel_err_rows(got[:, 5], g[p + pre + key]) < 1e-10, (conv, key)


def test_status_flags_non_pd():
    """S not positive definite -> status bit instead of a fault (the reference raises LinAlgError)."""
    from gpu_util import run_kf_batch
    N, T = 70, 3
    x0, P0 = np.zeros((N, 2)), np.tile(np.eye(2), (N, 1, 1))
    P0[7] = -np.eye(2)           # makes S = H P H' + R negative for track 7
    zs = np.ones((T, N, 1))
    out = run_kf_batch(x0, P0, zs, np.eye(2), np.zeros((2, 2)), np.array([[1., 0.]]), np.array([[0.5]]),
                       check_status=False)
    st = out[-1]
    assert st[7] & 1 == 0        # 1x1 S = -0.5 is invertible: NOT flagged (numpy.linalg.inv works too)
    P0[7] = np.zeros((2, 2))
    out = run_kf_batch(x0, P0, zs[:1], np.eye(2), np.zeros((2, 2)), np.array([[1., 0.]]), np.array([[0.0]]),
                       check_status=False)
    st = out[-1]
    assert st[7] != 0 and not st[np.arange(N) != 7].any()


def test_c2_shape_sample_vs_oracle():
    """BASELINE config 2 model at reduced N: constant-velocity 2-D, 100 steps, sample vs oracle."""
    from gpu_util import run_kf_batch
    from bench import c2_model, c2_inputs
    N, T = 70001, 100
    F, Q, H, R = c2_model()
    x0, P0, zs = c2_inputs(N, T, seed=1)
    for layout in ("soa", "aos"):
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout)
        sample = [0, 1, 255, 256, 4095, 4096, 65535, 65536, N - 1]
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=sample)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (layout, k)


@pytest.mark.parametrize("n,m,variants", [(4, 2, (0, 1)), (6, 3, (0, 1)), (9, 3, (0,))])
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_fast_kernel_tuning_variants(n, m, variants, layout, monkeypatch):
    """Every compiled tuning variant of the shared-model kernel (occupancy / packed-symmetric /
    prefetch depth, csrc/fk_dims_fast.def), the XCD-contiguous block mapping and the generic kernel
    (FK_NO_FAST) must give the same results as the goldens."""
    from gpu_util import run_kf_batch, tile_tracks
    g = golden("kf_dims")
    p = f"n{n}m{m}_"
    N = 777
    args = (tile_tracks(g[p + "x0"], N), tile_tracks(g[p + "P0"], N), tile_tracks(g[p + "zs"], N, 1),
            g[p + "F"], g[p + "Q"], g[p + "H"], g[p + "R"])
    mask = tile_tracks(g[p + "mask"], N, 1)
    # FK_NO_ML: keep the one-lane-per-track variants of (9,3) covered now that kf_ml.hip takes that call
    settings = ([{"FK_FAST_VARIANT": str(v), "FK_NO_ML": "1"} for v in variants] +
                [{}, {"FK_ML_PAIRS": "0"}, {"FK_FAST_XCD": "1", "FK_NO_ML": "1"}, {"FK_NO_FAST": "1"}])
    for env in settings:
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            for kw, tag in (({}, "plain"), ({"mask": mask}, "miss")):
                mu, cov, mup, covp, *_ = run_kf_batch(*args, layout=layout, **kw)
                for got, key in ((mu, "mu"), (cov, "cov"), (mup, "mup"), (covp, "covp")):
                    for trk in (0, 255, 256, N - 1):
                        assert rel_err_rows(got[:, trk], g[p + tag + "_" + key]) < TOL, (env, tag, key, trk)


@pytest.mark.parametrize("family", ["m", "g"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("outputs", [True, False])
def test_multilane_9_3_vs_oracle(outputs, masked, family, monkeypatch):
    """kf_ml.hip (three lanes per track, quad-permute row exchange; FK_ML9=m) and the generic four-lane kernel
    instantiated at (9, 3) (FK_ML9=g): every track its own state and measurements, N not a multiple of the 64 tracks
    per workgroup, alpha != 1."""
    from gpu_util import run_kf_batch
    monkeypatch.setenv("FK_ML9", family)
    n, m = 9, 3
    rs = np.random.RandomState(93)
    N, T = 1000, 40
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 3
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    mask = (rs.rand(T, N) > 0.25) if masked else None
    if masked:
        zs[~mask] = np.nan              # the kernel must not look at a masked measurement
    sample = [0, 1, 15, 16, 63, 64, 255, 256, 959, 960, N - 1]
    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=sample, alpha_sq=1.02 ** 2, mask=mask)
    for layout in ("soa", "aos"):
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, alpha_sq=1.02 ** 2, outputs=outputs, mask=mask)
        if outputs:
            for k in range(4):
                assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (layout, k)
        assert rel_err_rows(got[4][sample], ref[0][-1]) < TOL and rel_err_rows(got[5][sample], ref[1][-1]) < TOL


@pytest.mark.parametrize("family,layout", [("m", "soa"), ("m", "aos"), ("g", "soa"), ("g", "aos")])
@pytest.mark.parametrize("N", [1000, 777])
def test_multilane_rts_9_vs_oracle(N, family, layout, monkeypatch):
    """dim_x = 9 through both smoother families in both layouts (FK_ML9; the default takes rts_ml for SOA, rts_mlg<9> for
    AOS): rts_ml_kernel (three lanes per track; N even: 16-byte pair loads/stores, N odd: 8-byte path) and the
    four-lane rts_mlg_kernel; every track its own filter output, both index conventions."""
    from gpu_util import run_rts
    monkeypatch.setenv("FK_ML9", family)
    n = 9
    rs = np.random.RandomState(99)
    T = 30
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    sample = [0, 1, 15, 16, 63, 64, 255, 256, N - 2, N - 1]
    for conv, name in ((0, "class"), (1, "module")):
        got = run_rts(Xs, Ps, F, Q, layout=layout, convention=conv)
        ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=sample, convention=name)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < 1e-10, (name, k)


@pytest.mark.parametrize("N", [1, 2, 3, 16, 17, 64, 65, 66, 130])
def test_multilane_small_shapes(N):
    """tiny and ragged banks through the three-lane kernels (pair path for even N, tail quads, T = 1..3)."""
    from gpu_util import run_kf_batch, run_rts
    n, m = 9, 3
    rs = np.random.RandomState(500 + N)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    for T in (1, 2, 3):
        A = rs.randn(N, n, n)
        x0, P0 = rs.randn(N, n), A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
        zs = rs.randn(T, N, m)
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout="soa")
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=range(N))
        for k in range(4):
            assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (T, k)
        assert rel_err_rows(got[4], ref[0][-1]) < TOL and rel_err_rows(got[5], ref[1][-1]) < TOL
        sm = run_rts(ref[0], ref[1], F, Q, layout="soa")
        rr = kf_oracle.rts_smoother_tracks(ref[0], ref[1], F, Q, tracks=range(N))
        for k in range(4):
            assert rel_err_rows(_per_track(sm[k]), _per_track(rr[k])) < 1e-10, (T, "rts", k)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(n, m) for n in range(1, 10) for m in range(1, min(n, 4) + 1)
                                 if (n, m) not in ((1, 1), (2, 1), (4, 2), (6, 3), (9, 3))])
def test_lean_fast_instantiations_vs_oracle(n, m, layout, monkeypatch):
    """the dims that only have the lean specialised instantiation (shared model, optional mask): against
    the oracle, and identical to what the generic kernel gives within the parity bar"""
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(10 * n + m)
    N, T = 500, 30
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 2
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))     # stable, like the goldens' models
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    mask = rs.rand(T, N) > 0.2
    sample = [0, 1, 63, 64, 255, 256, N - 1]
    for kw in ({}, {"mask": mask}):
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, **kw)
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=sample, **kw)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (kw.keys(), k)
    with monkeypatch.context() as mp:
        mp.setenv("FK_NO_FAST", "1")
        gen = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout)
    plain = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout)
    assert rel_err_rows(_per_track(plain[1][:, sample]), _per_track(gen[1][:, sample])) < TOL


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n", [3, 5, 7, 8])
def test_rts_more_exact_dims_vs_oracle(n, layout):
    from gpu_util import run_rts
    rs = np.random.RandomState(70 + n)
    N, T = 300, 20
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    sample = [0, 63, 64, 255, 256, N - 1]
    got = run_rts(Xs, Ps, F, Q, layout=layout)
    ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=sample)
    for k in range(4):
        assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < 1e-10, k


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(2, 1), (3, 1), (4, 2), (5, 3), (6, 3), (7, 4), (8, 4)])
def test_tail_shapes_kf_and_rts(n, m, layout):
    """banks smaller than a wave / with a partial first wave in the tail workgroup (the cooperative AOS
    paths clamp tail lanes and clip their stores; the model fill must not depend on the clamped index)"""
    from gpu_util import run_kf_batch, run_rts
    rs = np.random.RandomState(900 + 10 * n + m)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    T = 4
    for N in (1, 44, 257, 300):
        A = rs.randn(N, n, n)
        x0, P0 = rs.randn(N, n), A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
        zs = rs.randn(T, N, m)
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout)
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=range(N))
        for k in range(4):
            assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (N, k)
        sm = run_rts(ref[0], ref[1], F, Q, layout=layout)
        rr = kf_oracle.rts_smoother_tracks(ref[0], ref[1], F, Q, tracks=range(N))
        for k in range(4):
            assert rel_err_rows(_per_track(sm[k]), _per_track(rr[k])) < 1e-10, (N, "rts", k)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("N", [1000, 777])
def test_multilane_9_3_variants_vs_oracle(N, layout, monkeypatch):
    """The VAR family of kf_ml.hip (per-step shared models in double-buffered LDS, control input with a per-step B,
    update_first, mask -- run-time switches of one instantiation, every combination): every track its own state,
    measurements and controls; N even (16-byte pair stores) and odd; against the oracle at 1e-10, and against the
    one-lane kernels (FK_ML_VAR=0 sends the same call there)."""
    from gpu_util import run_kf_batch
    from filterpy_amd._abi import FK_MODEL_PER_STEP, FK_MODEL_SHARED
    n, m, nu = 9, 3, 2
    rs = np.random.RandomState(1000 + N)
    T = 12
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 3
    us = rs.randn(T, N, nu)

    def spd(k, s, cnt):
        G = rs.randn(cnt, k, k)
        return s * (G @ G.transpose(0, 2, 1) / k + 0.5 * np.eye(k))
    Fs = np.eye(n) + 0.1 * rs.randn(T, n, n)
    Qs, Hs, Rs = spd(n, 0.1, T), rs.randn(T, m, n), spd(m, 0.5, T)
    Bs = rs.randn(T, n, nu)
    mask = rs.rand(T, N) > 0.25
    sample = [0, 1, 15, 16, 63, 64, 255, 256, N - 2, N - 1]
    for per_step in (False, True):
        for ctrl in (False, True):
            for uf in (False, True):
                for masked in (False, True):
                    if not (per_step or ctrl or uf):
                        continue                                    # the plain call: test_multilane_9_3_vs_oracle
                    mods = (Fs, Qs, Hs, Rs) if per_step else (Fs[0], Qs[0], Hs[0], Rs[0])
                    B = None if not ctrl else (Bs if per_step else Bs[0])
                    z = zs.copy()
                    if masked:
                        z[~mask] = np.nan                           # a masked measurement must not be looked at
                    kw = dict(mode=FK_MODEL_PER_STEP if per_step else FK_MODEL_SHARED, B=B, us=us if ctrl else None,
                              update_first=uf, mask=mask if masked else None, alpha_sq=1.01 ** 2)
                    got = run_kf_batch(x0, P0, z, *mods, layout=layout, **kw)
                    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, z, *mods, tracks=sample, B=B, us=us if ctrl else None,
                                                           update_first=uf, mask=mask if masked else None, alpha_sq=1.01 ** 2)
                    tag = (per_step, ctrl, uf, masked)
                    for k in range(4):
                        assert np.isfinite(got[k]).all(), (tag, k)
                        assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (tag, k)
                    last = ref[2] if uf else ref[0]                  # update_first leaves the PRIOR as the state
                    lastP = ref[3] if uf else ref[1]
                    assert rel_err_rows(got[4][sample], last[-1]) < TOL and rel_err_rows(got[5][sample], lastP[-1]) < TOL, tag
    # the same call through the one-lane kernels agrees to the tolerance too (the dispatch switch itself)
    monkeypatch.setenv("FK_ML_VAR", "0")
    one = run_kf_batch(x0, P0, zs, Fs, Qs, Hs, Rs, layout=layout, mode=FK_MODEL_PER_STEP, B=Bs, us=us, update_first=True)
    monkeypatch.delenv("FK_ML_VAR")
    ml = run_kf_batch(x0, P0, zs, Fs, Qs, Hs, Rs, layout=layout, mode=FK_MODEL_PER_STEP, B=Bs, us=us, update_first=True)
    for k in range(4):
        assert rel_err_rows(_per_track(ml[k]), _per_track(one[k])) < TOL, k


@pytest.mark.parametrize("N,T", [(1, 1), (2, 1), (3, 2), (17, 3), (66, 2)])
def test_multilane_variants_small_shapes(N, T):
    """tiny / ragged banks and T = 1..3 through the VAR family (tail quads, deferred covariance stores at the edges)"""
    from gpu_util import run_kf_batch
    from filterpy_amd._abi import FK_MODEL_PER_STEP
    n, m, nu = 9, 3, 1
    rs = np.random.RandomState(7000 + 10 * N + T)
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    zs, us = rs.randn(T, N, m), rs.randn(T, N, nu)
    Fs = np.eye(n) + 0.05 * rs.randn(T, n, n)
    G = rs.randn(T, n, n)
    Qs = 0.1 * (G @ G.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    Hs, Rs, Bs = rs.randn(T, m, n), np.tile(0.5 * np.eye(m), (T, 1, 1)), rs.randn(T, n, nu)
    for layout in ("soa", "aos"):
        for uf in (False, True):
            got = run_kf_batch(x0, P0, zs, Fs, Qs, Hs, Rs, layout=layout, mode=FK_MODEL_PER_STEP, B=Bs, us=us, update_first=uf)
            ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, Fs, Qs, Hs, Rs, tracks=range(N), B=Bs, us=us, update_first=uf)
            for k in range(4):
                assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (layout, uf, k)
            last, lastP = (ref[2], ref[3]) if uf else (ref[0], ref[1])
            assert rel_err_rows(got[4], last[-1]) < TOL and rel_err_rows(got[5], lastP[-1]) < TOL


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(10, 1), (10, 4), (11, 2), (12, 3), (13, 1), (13, 4), (14, 2), (15, 3), (16, 1), (16, 4),
                                 (10, 5), (11, 8), (12, 6), (13, 7), (14, 5), (15, 6), (16, 8)])
def test_four_lane_kernel_dims_10_to_16_vs_oracle(n, m, layout, monkeypatch):
    """kf_mlg.hip (four lanes per track; the rows past n-1 of lane 3 clamped to row n-1, entering the quad's H P sum
    with coefficient 0): every track its own state and measurements, N not a multiple of the 64 tracks of a workgroup,
    alpha != 1, a mask with NaN behind it -- against the oracle, and against the padded one-lane kernel the same call
    ran on before (FK_NO_MLG=1)."""
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(100 * n + m)
    N, T = 333, 12
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 3
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    sample = [0, 1, 15, 16, 63, 64, 255, 256, N - 2, N - 1]
    for masked in (False, True):
        mask = (rs.rand(T, N) > 0.25) if masked else None
        z = zs.copy()
        if masked:
            z[~mask] = np.nan
        got = run_kf_batch(x0, P0, z, F, Q, H, R, layout=layout, alpha_sq=1.02 ** 2, mask=mask)
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, z, F, Q, H, R, tracks=sample, alpha_sq=1.02 ** 2, mask=mask)
        for k in range(4):
            assert np.isfinite(got[k]).all(), (masked, k)
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (masked, k)
        assert rel_err_rows(got[4][sample], ref[0][-1]) < TOL and rel_err_rows(got[5][sample], ref[1][-1]) < TOL
    monkeypatch.setenv("FK_NO_MLG", "1")
    old = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, alpha_sq=1.02 ** 2)
    monkeypatch.delenv("FK_NO_MLG")
    new = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, alpha_sq=1.02 ** 2)
    for k in range(6):
        assert rel_err_rows(new[k].reshape(-1, new[k].shape[-1]), old[k].reshape(-1, old[k].shape[-1])) < TOL, k


@pytest.mark.parametrize("N", [1, 2, 3, 17, 64, 65])
def test_four_lane_kernel_small_shapes(N):
    """tiny / ragged banks, T = 1..3, both layouts (tail quads duplicate the last track; the AOS slab of a partly filled wave)"""
    from gpu_util import run_kf_batch
    n, m = 11, 2
    rs = np.random.RandomState(900 + N)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H, R = rs.randn(m, n), 0.5 * np.eye(m)
    for T in (1, 2, 3):
        A = rs.randn(N, n, n)
        x0, P0 = rs.randn(N, n), A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
        zs = rs.randn(T, N, m)
        ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=range(N))
        for layout in ("soa", "aos"):
            got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout)
            for k in range(4):
                assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (T, layout, k)
            assert rel_err_rows(got[4], ref[0][-1]) < TOL and rel_err_rows(got[5], ref[1][-1]) < TOL


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n", [10, 11, 12, 13, 14, 15, 16])
def test_four_lane_smoother_dims_10_to_16_vs_oracle(n, layout, monkeypatch):
    """rts_mlg.hip: every track its own filter output, both index conventions, N ragged; against the oracle and against
    the padded one-lane smoother (FK_NO_MLG=1)."""
    from gpu_util import run_rts
    rs = np.random.RandomState(1600 + n)
    N, T = 333, 9
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    sample = [0, 1, 15, 16, 63, 64, 255, 256, N - 2, N - 1]
    for conv, name in ((0, "class"), (1, "module")):
        got = run_rts(Xs, Ps, F, Q, layout=layout, convention=conv)
        ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=sample, convention=name)
        for k in range(4):
            assert np.isfinite(got[k]).all(), (name, k)
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (name, k)
    monkeypatch.setenv("FK_NO_MLG", "1")
    old = run_rts(Xs, Ps, F, Q, layout=layout)
    monkeypatch.delenv("FK_NO_MLG")
    new = run_rts(Xs, Ps, F, Q, layout=layout)
    for k in range(4):
        assert rel_err_rows(_per_track(new[k]), _per_track(old[k])) < TOL, k


@pytest.mark.parametrize("lanes", ["4", "8"])
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n", [13, 14, 15, 16])
def test_smoother_both_organisations_dims_13_to_16(n, layout, lanes, monkeypatch):
    """dims 13..16 through BOTH smoothers (FK_RTS_LANES): four lanes + DPP (rts_mlg.hip) and eight lanes + LDS exchange
    (rts_mlx.hip); every track its own data, ragged N (a partly filled group / wave / workgroup), against the oracle."""
    from gpu_util import run_rts
    monkeypatch.setenv("FK_RTS_LANES", lanes)
    rs = np.random.RandomState(2600 + n)
    N, T = 203, 7
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    sample = [0, 1, 7, 8, 31, 32, 63, 64, N - 2, N - 1]
    got = run_rts(Xs, Ps, F, Q, layout=layout)
    ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=sample)
    for k in range(4):
        assert np.isfinite(got[k]).all(), k
        assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, k
    for (Ns, Ts) in ((1, 2), (9, 3), (33, 2)):
        ref = kf_oracle.rts_smoother_tracks(Xs[:Ts, :Ns], Ps[:Ts, :Ns], F, Q, tracks=range(Ns))
        got = run_rts(np.ascontiguousarray(Xs[:Ts, :Ns]), np.ascontiguousarray(Ps[:Ts, :Ns]), F, Q, layout=layout)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (Ns, Ts, k)


@pytest.mark.parametrize("family", ["m", ""])
def test_smoother_dim_8_numpy_order_both_kernels(family, monkeypatch):
    """n = 8 in NumPy order runs rts_mlg<8> by default and the one-lane rts_kernel with FK_ML9=m: both against the oracle"""
    from gpu_util import run_rts
    if family:
        monkeypatch.setenv("FK_ML9", family)
    n = 8
    rs = np.random.RandomState(808)
    N, T = 333, 9
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    Q = 0.05 * np.eye(n)
    sample = [0, 1, 15, 16, 63, 64, 255, 256, N - 2, N - 1]
    for conv, name in ((0, "class"), (1, "module")):
        got = run_rts(Xs, Ps, F, Q, layout="aos", convention=conv)
        ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=sample, convention=name)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (name, k)


@pytest.mark.parametrize("N,T", [(1, 2), (2, 3), (17, 2), (65, 4)])
def test_four_lane_smoother_small_shapes(N, T):
    from gpu_util import run_rts
    n = 13
    rs = np.random.RandomState(40 + N)
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    Q = 0.05 * np.eye(n)
    ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=range(N))
    for layout in ("soa", "aos"):
        got = run_rts(Xs, Ps, F, Q, layout=layout)
        for k in range(4):
            assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (layout, k)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(10, 2), (13, 4), (16, 3), (12, 6), (16, 8)])
def test_four_lane_kernel_variants_vs_oracle(n, m, layout):
    """kf_mlg.hip's VAR instantiations (per-step model lists incl. B, control input, update_first, mask: every
    combination) at dims above 9 -- calls that ran on the padded one-lane kernel before."""
    from gpu_util import run_kf_batch
    from filterpy_amd._abi import FK_MODEL_PER_STEP, FK_MODEL_SHARED
    nu = 3
    rs = np.random.RandomState(5000 + 10 * n + m)
    N, T = 203, 7
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs, us = rs.randn(T, N, m) * 3, rs.randn(T, N, nu)

    def spd(k, s, cnt):
        G = rs.randn(cnt, k, k)
        return s * (G @ G.transpose(0, 2, 1) / k + 0.5 * np.eye(k))
    Fs = np.eye(n) + 0.1 * rs.randn(T, n, n)
    Qs, Hs, Rs, Bs = spd(n, 0.1, T), rs.randn(T, m, n), spd(m, 0.5, T), rs.randn(T, n, nu)
    mask = rs.rand(T, N) > 0.25
    sample = [0, 1, 15, 16, 63, 64, 127, 128, N - 2, N - 1]
    for per_step in (False, True):
        for ctrl in (False, True):
            for uf in (False, True):
                for masked in (False, True):
                    if not (per_step or ctrl or uf):
                        continue
                    mods = (Fs, Qs, Hs, Rs) if per_step else (Fs[0], Qs[0], Hs[0], Rs[0])
                    B = None if not ctrl else (Bs if per_step else Bs[0])
                    z = zs.copy()
                    if masked:
                        z[~mask] = np.nan
                    kw = dict(B=B, us=us if ctrl else None, update_first=uf, mask=mask if masked else None, alpha_sq=1.01 ** 2)
                    got = run_kf_batch(x0, P0, z, *mods, layout=layout, mode=FK_MODEL_PER_STEP if per_step else FK_MODEL_SHARED, **kw)
                    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, z, *mods, tracks=sample, **kw)
                    tag = (per_step, ctrl, uf, masked)
                    for k in range(4):
                        assert np.isfinite(got[k]).all(), (tag, k)
                        assert rel_err_rows(_per_track(got[k][:, sample]), _per_track(ref[k])) < TOL, (tag, k)
                    last, lastP = (ref[2], ref[3]) if uf else (ref[0], ref[1])
                    assert rel_err_rows(got[4][sample], last[-1]) < TOL and rel_err_rows(got[5][sample], lastP[-1]) < TOL, tag


@pytest.mark.parametrize("n,m", [(9, 3), (12, 2)])
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("N", [1000, 777, 130])
def test_multilane_chunked_call_is_bit_identical(N, layout, n, m, monkeypatch):
    """kf_chunked_call: G track groups x H time chunks on G streams (tail filling) must give the single launch's
    bits -- outputs, final state and status -- for every decomposition, with a mask, ragged N, AOS slabs; kf_ml (9,3)
    and the four-lane kf_mlg (12,2)"""
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(4242 + N)
    T = 23
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 3
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    mask = rs.rand(T, N) > 0.2
    mask[0, 5] = True
    P0[5] = -np.eye(n)                     # one track whose S is not positive definite: the status bit must survive the chunks
    monkeypatch.setenv("FK_ML_CHUNKS", "1,1")
    one = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, mask=mask, check_status=False)
    assert one[6][5] != 0 and not one[6][[0, 1, 6]].any()
    for spec in ("2,5", "4,3", "3,23", "2,1", "1,4"):
        monkeypatch.setenv("FK_ML_CHUNKS", spec)
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, mask=mask, check_status=False)
        for k in range(7):
            assert np.array_equal(got[k], one[k], equal_nan=True), (spec, k)
    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=[0, 64, N - 1], mask=mask)
    for k in range(4):
        assert rel_err_rows(_per_track(one[k][:, [0, 64, N - 1]]), _per_track(ref[k])) < TOL, k


@pytest.mark.parametrize("family,n", [("m", 9), ("g", 9), ("g", 12)])
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("N", [1000, 777, 130])
def test_multilane_chunked_smoother_is_bit_identical(N, layout, family, n, monkeypatch):
    """rts_chunked_call (backward time windows that share their top step, G track groups on G streams) against the single
    launch: every output bit, both layouts, ragged N; rts_ml (FK_ML9=m) and the four-lane rts_mlg at n = 9 and 12"""
    from gpu_util import run_rts
    monkeypatch.setenv("FK_ML9", family)
    rs = np.random.RandomState(777 + N)
    T = 19
    A = rs.randn(T, N, n, n)
    Xs, Ps = rs.randn(T, N, n), A @ A.transpose(0, 1, 3, 2) / n + 0.5 * np.eye(n)
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    Q = 0.05 * np.eye(n)
    monkeypatch.setenv("FK_ML_CHUNKS", "1,1")
    one = run_rts(Xs, Ps, F, Q, layout=layout)
    for spec in ("2,4", "3,3", "4,18", "1,5", "2,1"):
        monkeypatch.setenv("FK_ML_CHUNKS", spec)
        got = run_rts(Xs, Ps, F, Q, layout=layout)
        for k in range(4):
            assert np.array_equal(got[k], one[k]), (spec, k)
    ref = kf_oracle.rts_smoother_tracks(Xs, Ps, F, Q, tracks=[0, 64, N - 1])
    for k in range(4):
        assert rel_err_rows(_per_track(one[k][:, [0, 64, N - 1]]), _per_track(ref[k])) < TOL, k


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(7, 4), (8, 3), (9, 2)])
def test_one_lane_chunked_call_is_bit_identical(n, m, layout, monkeypatch):
    """the one-wave-per-SIMD kf_fast instantiations (dim_x 7..9) under tail filling (kf_chunked_call in run_kf): every
    decomposition gives the single launch's bits, incl. the packed-triangle variants whose P goes through memory
    between chunks, a mask and the status word"""
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(90 * n + m)
    N, T = 777, 17
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 3
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    mask = rs.rand(T, N) > 0.2
    mask[0, 5] = True
    P0[5] = -np.eye(n)
    monkeypatch.setenv("FK_ML_CHUNKS", "1,1")
    one = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, mask=mask, check_status=False)
    assert one[6][5] != 0 and not one[6][[0, 1, 6]].any()
    for spec in ("2,4", "3,3", "3,17"):
        monkeypatch.setenv("FK_ML_CHUNKS", spec)
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, mask=mask, check_status=False)
        for k in range(7):
            assert np.array_equal(got[k], one[k], equal_nan=True), (spec, k)
    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=[0, 64, N - 1], mask=mask)
    for k in range(4):
        assert rel_err_rows(_per_track(one[k][:, [0, 64, N - 1]]), _per_track(ref[k])) < TOL, k


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(9, 3), (11, 2)])
def test_variant_calls_chunked_are_bit_identical(n, m, layout, monkeypatch):
    """per-step model lists (incl. B), control input, update_first and a mask under tail filling: the model / control
    pointers advance with the time chunks; every decomposition must reproduce the single launch bit for bit"""
    from gpu_util import run_kf_batch
    from filterpy_amd._abi import FK_MODEL_PER_STEP, FK_MODEL_SHARED
    nu = 2
    rs = np.random.RandomState(31 * n + m)
    N, T = 333, 13
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 4.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs, us = rs.randn(T, N, m) * 3, rs.randn(T, N, nu)

    def spd(k, s, cnt):
        G = rs.randn(cnt, k, k)
        return s * (G @ G.transpose(0, 2, 1) / k + 0.5 * np.eye(k))
    Fs = np.eye(n) + 0.1 * rs.randn(T, n, n)
    Qs, Hs, Rs, Bs = spd(n, 0.1, T), rs.randn(T, m, n), spd(m, 0.5, T), rs.randn(T, n, nu)
    mask = rs.rand(T, N) > 0.25
    for per_step, ctrl, uf in ((True, True, True), (True, False, False), (False, True, False), (False, False, True)):
        mods = (Fs, Qs, Hs, Rs) if per_step else (Fs[0], Qs[0], Hs[0], Rs[0])
        B = None if not ctrl else (Bs if per_step else Bs[0])
        kw = dict(layout=layout, mode=FK_MODEL_PER_STEP if per_step else FK_MODEL_SHARED, B=B, us=us if ctrl else None,
                  update_first=uf, mask=mask)
        monkeypatch.setenv("FK_ML_CHUNKS", "1,1")
        one = run_kf_batch(x0, P0, zs, *mods, **kw)
        for spec in ("2,3", "3,13"):
            monkeypatch.setenv("FK_ML_CHUNKS", spec)
            got = run_kf_batch(x0, P0, zs, *mods, **kw)
            for k in range(7):
                assert np.array_equal(got[k], one[k]), (per_step, ctrl, uf, spec, k)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m,nu", [(9, 3, 2), (12, 2, 3), (16, 4, 1)])
def test_every_batch_filter_argument_at_once_vs_live_reference(n, m, nu, layout):
    """the golden frozen from the LIVE reference with Fs / Qs / Hs / Rs / Bs lists, us, update_first and missing
    measurements in one call (tests/golden/make_kf_combo_golden.py), through the VAR instantiations of kf_ml / kf_mlg,
    and its rts_smoother with constant models through the several-lanes smoothers"""
    from gpu_util import run_kf_batch, tile_tracks
    from filterpy_amd._abi import FK_MODEL_PER_STEP
    g = golden("kf_combo")
    p = f"n{n}m{m}_"
    N = 130
    zs = g[p + "zs"]
    mask = ~np.isnan(zs).all(axis=1)
    T = len(zs)
    for uf, q in ((False, p + "pu_"), (True, p + "uf_")):
        got = run_kf_batch(tile_tracks(g[p + "x0"], N), tile_tracks(g[p + "P0"], N), tile_tracks(zs, N, 1),
                           g[p + "Fs"], g[p + "Qs"], g[p + "Hs"], g[p + "Rs"], layout=layout, mode=FK_MODEL_PER_STEP,
                           B=g[p + "Bs"], us=tile_tracks(g[p + "us"], N, 1), update_first=uf, mask=tile_tracks(mask, N, 1))
        for k, key in enumerate(("mu", "cov", "mup", "covp")):
            for trk in (0, 63, 64, N - 1):
                assert rel_err_rows(got[k][:, trk].reshape(T, -1), g[q + key].reshape(T, -1)) < TOL, (uf, key, trk)
        assert rel_err_rows(got[4][:1], g[q + "xfinal"][None]) < TOL and rel_err_rows(got[5][:1].reshape(1, -1), g[q + "Pfinal"].reshape(1, -1)) < TOL


def _run_ex(x0, P0, zs, F, Q, H, R, layout, mask=None, keys=("y", "K", "S", "SI", "log_likelihood", "mahalanobis")):
    """fk_kf_batch_filter_ex_f64 on host arrays -> (the four outputs, {extras histories})"""
    import torch
    from filterpy_amd import _engine as E
    T, N, m = zs.shape
    n = x0.shape[1]
    dx, dP, dz = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0), E.to_records(zs, layout, 1)
    dmask = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dx.device)
    outs = [E.alloc_records((T,), N, w, layout).fill_(float("nan")) for w in (n, n * n, n, n * n)]
    shapes = dict(y=(m,), K=(n, m), S=(m, m), SI=(m, m))
    ex = {k: (E.alloc_records((T,), N, int(np.prod(shapes[k])), layout) if k in shapes
              else torch.empty((T, N), dtype=torch.float64, device=dx.device)).fill_(float("nan")) for k in keys}
    st = torch.zeros(N, dtype=torch.int32, device=dx.device)
    E.kf_batch_filter_ex(dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0),
                         E.dev(F), E.dev(Q), E.dev(H), E.dev(R), dz, dx, dP, ex, mask=dmask,
                         means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
    torch.cuda.synchronize()
    assert not st.any()
    res = [E.from_records(outs[0], layout, 1, (n,)), E.from_records(outs[1], layout, 1, (n, n)),
           E.from_records(outs[2], layout, 1, (n,)), E.from_records(outs[3], layout, 1, (n, n))]
    hist = {k: (E.from_records(v, layout, 1, shapes[k]) if k in shapes else v.cpu().numpy()) for k, v in ex.items()}
    return res, hist


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(n, m) for n in range(1, 10) for m in range(1, min(n, 4) + 1)])
def test_saver_histories_from_the_specialised_kernel(n, m, layout, monkeypatch):
    """y / K / S / SI / log-likelihood / mahalanobis histories (fk_kf_batch_filter_ex_f64; kalman_filter.py:533-563, :1203-1240)
    are stored by kf_fast's extras instantiations: against the oracle on sample tracks, against the generic kernel
    (FK_NO_FAST_EX=1) on every track, with a mask (missing measurement: y = 0, K / S / SI keep their last values),
    with a subset of the histories, and with the four regular outputs unchanged."""
    rs = np.random.RandomState(100 * n + m)
    N, T = 333, 12
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 2
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    mask = rs.rand(T, N) > 0.25
    mask[0] = True                                   # (before the first measurement S = 0: compared in test_gpu_api)
    sample = [0, 63, 64, 255, 256, N - 1]
    for kw in ({}, {"mask": mask}):
        outs, hist = _run_ex(x0, P0, zs, F, Q, H, R, layout, **kw)
        with monkeypatch.context() as mp:
            mp.setenv("FK_NO_FAST_EX", "1")
            outs_g, hist_g = _run_ex(x0, P0, zs, F, Q, H, R, layout, **kw)
        for a, b in zip(outs, outs_g):
            assert rel_err_rows(_per_track(a), _per_track(b)) < TOL
        for k in hist:
            assert np.isfinite(hist[k]).all(), k
            assert np.allclose(hist[k], hist_g[k], rtol=1e-10, atol=1e-11), (k, kw.keys())
        for i in sample:
            x, P = x0[i].copy(), P0[i].copy()
            last = None
            for t in range(T):
                x, P = kf_oracle.kf_predict(x, P, F, Q)
                if "mask" in kw and not mask[t, i]:
                    y, (K, S, SI) = np.zeros(m), last
                    ll, mh = kf_oracle.log_likelihood(y, S), 0.0
                else:
                    x, P, y, K, S, SI = kf_oracle.kf_update(x, P, zs[t, i], R, H)
                    last = (K, S, SI)
                    ll, mh = kf_oracle.log_likelihood(y, S), kf_oracle.mahalanobis(y, SI)
                for key, ref in (("y", y), ("K", K), ("S", S), ("SI", SI)):
                    assert np.allclose(hist[key][t, i], ref, rtol=1e-10, atol=1e-11), (key, t, i)
                assert abs(hist["log_likelihood"][t, i] - ll) <= 1e-10 * max(1.0, abs(ll)), (t, i)
                assert abs(hist["mahalanobis"][t, i] - mh) <= 1e-10 * max(1.0, mh), (t, i)
    # a subset: only the likelihood history
    _, h1 = _run_ex(x0, P0, zs, F, Q, H, R, layout, keys=("log_likelihood",))
    assert np.array_equal(h1["log_likelihood"], _run_ex(x0, P0, zs, F, Q, H, R, layout)[1]["log_likelihood"])


@pytest.mark.parametrize("with_status", [True, False])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_three_lane_persistent_grid_is_bit_identical(layout, masked, with_status, monkeypatch):
    """Round 4: banks of more workgroups than the chip holds at once run on a PERSISTENT grid -- 512 workgroups draw tickets,
    one per (time chunk, group of 64 tracks), chunk-major; a chunk waits for its group's previous chunk through a completion
    word and picks the state up from x / P in place (kf_ml.hip, PERS).  Same arithmetic per track: every output, the final
    state and the status equal the single launch's (FK_ML_PERSIST=0, FK_ML_CHUNKS=1,1) bit for bit -- a ragged bank of
    33 003 tracks (516 groups, the last one partial), 70 steps in 4 chunks, with and without missing measurements, a
    non-positive-definite track -- and a forced decomposition into 7 chunks as well.  with_status=False passes status = NULL
    through the C ABI: the hand-over's only drain used to be the wait that belonged to the status load."""
    import torch
    from filterpy_amd import _engine as E
    n, m, N, T = 9, 3, 33_003, 70
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    rs = np.random.RandomState(3)
    F = np.eye(n) + 0.05 * np.triu(rs.randn(n, n), 1)
    A = rs.randn(n, n)
    Q = 0.05 * (A @ A.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    z = E.alloc_records((T,), N, m, layout)
    z.copy_(torch.randn(z.shape, generator=g, device=dev, dtype=torch.float64))
    x0 = E.alloc_records((), N, n, layout)
    x0.copy_(torch.randn(x0.shape, generator=g, device=dev, dtype=torch.float64))
    P0h = np.tile(4.0 * np.eye(n), (N, 1, 1))
    P0h[N // 2] = -np.eye(n)                                  # one track that is not positive definite: its status bit must survive the chunks
    P0 = E.to_records(P0h, layout, 0)
    mask = (torch.rand((T, N), generator=g, device=dev) > 0.2).to(torch.uint8) if masked else None
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    mods = [E.dev(M) for M in (F, Q, H, R)]

    def run():
        x, P = x0.clone(), P0.clone()
        outs = [E.alloc_records((T,), N, w, layout).fill_(float("nan")) for w in (n, n * n, n, n * n)]
        st = torch.zeros(N, dtype=torch.int32, device=dev)
        E.kf_batch_filter(desc, *mods, z, x, P, mask=mask, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3],
                          status=st if with_status else None)
        torch.cuda.synchronize()
        return outs + [x, P, st]

    with monkeypatch.context() as mp:
        mp.setenv("FK_ML_PERSIST", "0")
        mp.setenv("FK_ML_CHUNKS", "1,1")
        ref = run()
    if with_status:
        assert int(ref[6][N // 2]) != 0 and int((ref[6] != 0).sum()) == 1
    for env in ({}, {"FK_ML_PERSIST_H": "7"}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            got = run()
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b), (layout, masked, env, i)


@pytest.mark.parametrize("with_status", [True, False])
@pytest.mark.parametrize("layout,N", [("soa", 33_003), ("soa", 33_002), ("aos", 33_003)])
def test_three_lane_smoother_persistent_grid_is_bit_identical(layout, N, with_status, monkeypatch):
    """Round 6: the (9) smoother on the forward kernel's persistent grid (FK_RTS_PERSIST=1; rts_ml_kernel PERS: tickets = (time chunk counted from
    the end, group of 64 tracks), the smoothed state of a chunk's first step handed to the chunk before it in time through the
    element-major block, agent-scope).  Same arithmetic per track: xs, Ps, K, Pp and the status equal the single launch's
    (FK_RTS_PERSIST=0, FK_ML_CHUNKS=1,1) bit for bit -- an odd bank (8-byte stores), an even one (16-byte pair stores), NumPy
    order through FK_ML9=m (by default that layout runs on the four-lane kernel), 70 steps in 3 chunks and forced into 7 and
    2, one track whose predicted covariance is not positive definite (its status bit must survive the chunks)."""
    import torch
    from filterpy_amd import _engine as E
    n, T = 9, 70
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    rs = np.random.RandomState(5)
    F = np.eye(n) + 0.05 * np.triu(rs.randn(n, n), 1)
    A = rs.randn(n, n)
    Q = 0.05 * (A @ A.T / n + 0.5 * np.eye(n))
    Xs = E.alloc_records((T,), N, n, layout)
    Xs.copy_(torch.randn(Xs.shape, generator=g, device=dev, dtype=torch.float64))
    # filtered covariances: a random SPD matrix per (step, track) built on the device (B B' / n + I), symmetric bit for bit
    B = torch.randn((T, N, n, n), generator=g, device=dev, dtype=torch.float64)
    Pd = B @ B.transpose(-1, -2) / n + torch.eye(n, device=dev, dtype=torch.float64)
    Pd = 0.5 * (Pd + Pd.transpose(-1, -2))
    Pd[T // 2, N // 2] = -torch.eye(n, device=dev, dtype=torch.float64) * 50.0   # Pp = F P F' + Q not positive definite at one step of one track
    Ps = E.alloc_records((T,), N, n * n, layout)
    if layout == "aos":
        Ps.copy_(Pd.reshape(Ps.shape))
    else:
        Ps.copy_(Pd.reshape(T, N, n * n).permute(0, 2, 1).reshape(Ps.shape))
    del B, Pd
    desc = dict(n=n, m=3, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    dF, dQ = E.dev(F), E.dev(Q)
    if layout == "aos":
        monkeypatch.setenv("FK_ML9", "m")

    def run():
        outs = [E.alloc_records((T,), N, w, layout).fill_(float("nan")) for w in (n, n * n, n * n, n * n)]
        st = torch.zeros(N, dtype=torch.int32, device=dev)
        E.kf_rts(desc, dF, dQ, Xs, Ps, outs[0], outs[1], outs[2], outs[3], convention=1, status=st if with_status else None)
        torch.cuda.synchronize()
        return outs + [st]

    with monkeypatch.context() as mp:
        mp.setenv("FK_RTS_PERSIST", "0")
        mp.setenv("FK_ML_CHUNKS", "1,1")
        ref = run()
    if with_status:
        assert int(ref[4][N // 2]) != 0 and int((ref[4] != 0).sum()) == 1
    assert not bool(torch.isnan(ref[0]).any())
    for env in ({}, {"FK_RTS_PERSIST_H": "7"}, {"FK_RTS_PERSIST_H": "2"}):
        with monkeypatch.context() as mp:
            mp.setenv("FK_RTS_PERSIST", "1")                   # (off by default: measured no faster than the single launch)
            for k, v in env.items():
                mp.setenv(k, v)
            got = run()
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b), (layout, N, env, i)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("N", [1000, 777, 64, 3])
def test_three_lane_slab_outputs_equal_the_pair_store_build_bit_for_bit(N, layout, monkeypatch):
    """Round 4: kf_ml's element-major outputs leave through a wave-private LDS slab (SLAB instantiations: 16-byte units of two
    adjacent tracks, one predicate per copy-out) instead of DPP pair stores / 8-byte stores; in NumPy order the slab
    descriptors are scalar now (wave_index(): no waterfall loop around each store).  Only addresses and store shapes change:
    every output, the final state and the status equal the round-3 store paths' (FK_ML_SLAB=0) bit for bit -- even and odd
    banks, ragged last wave (an odd tail's last track leaves as 8 bytes), with and without a mask."""
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(7 + N)
    n, m, T = 9, 3, 13
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 2
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    mask = (rs.rand(T, N) > 0.3).astype(np.uint8)
    for kw in (dict(), dict(mask=mask)):
        got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, **kw)
        with monkeypatch.context() as mp:
            mp.setenv("FK_ML_SLAB", "0")
            ref = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, **kw)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (N, layout, list(kw))
        # ... and both against the oracle on sample tracks
        for i in sorted({0, N // 2, N - 1}):
            o = kf_oracle.kf_batch_filter_tracks(x0[[i]], P0[[i]], zs[:, [i]], F, Q, H, R, tracks=range(1),
                                                 **({"mask": mask[:, [i]]} if kw else {}))
            assert rel_err_rows(got[0][:, i], o[0][:, 0]) < TOL and rel_err_rows(got[1][:, i].reshape(T, -1), o[1][:, 0].reshape(T, -1)) < TOL


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(9, 3), (10, 1), (10, 2), (11, 3), (12, 3), (12, 5), (13, 4), (14, 2), (15, 3), (16, 4), (16, 8)])
def test_saver_histories_from_the_four_lane_kernel(n, m, layout, monkeypatch):
    """Round 4 (VERDICT r3 missing 4): the by-product histories at dim_x >= 10 -- and (9,3) -- come from kf_mlg's EX
    instantiations (four lanes per track; every history leaves through the wave's LDS tile) instead of the padded
    generic kernel: against the oracle on sample tracks, against the generic kernel (FK_NO_MLG_EX=1) on every track
    of a ragged bank, with a subset of the histories (absent ones are written through zero-track descriptors), with
    the four regular outputs equal to the plain call's BIT FOR BIT, and with a forced time-chunk decomposition
    (the histories advance with the chunk's window)."""
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(100 * n + m)
    N, T = 333, 9
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    zs = rs.randn(T, N, m) * 2
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    outs, hist = _run_ex(x0, P0, zs, F, Q, H, R, layout)
    with monkeypatch.context() as mp:
        mp.setenv("FK_NO_MLG_EX", "1")
        mp.setenv("FK_NO_FAST_EX", "1")
        outs_g, hist_g = _run_ex(x0, P0, zs, F, Q, H, R, layout)
    for a, b in zip(outs, outs_g):
        assert rel_err_rows(_per_track(a), _per_track(b)) < TOL
    plain = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout)
    for a, b in zip(outs, plain[:4]):
        if n >= 10:
            assert np.array_equal(a, b)           # the same arithmetic as the plain four-lane kernel
        else:
            assert rel_err_rows(_per_track(a), _per_track(b)) < TOL      # (9,3): the plain call runs on the three-lane kernel
    for k in hist:
        assert np.isfinite(hist[k]).all(), k
        assert np.allclose(hist[k], hist_g[k], rtol=1e-10, atol=1e-11), k
    for i in [0, 15, 16, 63, 64, 255, 256, N - 1]:
        x, P = x0[i].copy(), P0[i].copy()
        for t in range(T):
            x, P = kf_oracle.kf_predict(x, P, F, Q)
            x, P, y, K, S, SI = kf_oracle.kf_update(x, P, zs[t, i], R, H)
            ll, mh = kf_oracle.log_likelihood(y, S), kf_oracle.mahalanobis(y, SI)
            for key, ref in (("y", y), ("K", K), ("S", S), ("SI", SI)):
                assert rel_err_rows(hist[key][t, i].reshape(1, -1), np.asarray(ref).reshape(1, -1)) < TOL, (key, t, i)
            assert abs(hist["log_likelihood"][t, i] - ll) <= 1e-10 * max(1.0, abs(ll)), (t, i)
            assert abs(hist["mahalanobis"][t, i] - mh) <= 1e-10 * max(1.0, mh), (t, i)
    # subsets: the absent histories' stores are dropped, the present ones keep their bits
    for keys in (("log_likelihood",), ("K", "mahalanobis"), ("y", "SI")):
        _, h1 = _run_ex(x0, P0, zs, F, Q, H, R, layout, keys=keys)
        for k in keys:
            assert np.array_equal(h1[k], hist[k]), (keys, k)
    # a forced decomposition into track groups x time chunks: same bits
    with monkeypatch.context() as mp:
        mp.setenv("FK_ML_CHUNKS", "2,3")
        outs_c, hist_c = _run_ex(x0, P0, zs, F, Q, H, R, layout)
    for a, b in zip(outs, outs_c):
        assert np.array_equal(a, b)
    for k in hist:
        assert np.array_equal(hist[k], hist_c[k]), k


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (3, 2), (4, 2), (5, 3), (6, 3), (7, 2), (8, 4), (9, 4), (9, 3)])
def test_interleaved_covariance_histories_equal_two_arrays_bit_for_bit(n, m, layout):
    """FK_KF_FLAG_COV_INTERLEAVED (the kernel-side cure of the placement lottery): both covariance histories
    in ONE array -- a track's posterior and prior record side by side in NumPy order, the two slabs of a step adjacent in
    element-major order -- so that a step writes one front.  Nothing but addresses changes: every output of the interleaved
    call equals the two-array call bit for bit -- ragged last workgroup, missing measurements, update_first, control input,
    per-track models (the kf_fast instantiations that exist at the size; a call the specialised kernel does not serve
    is FK_ERR_UNSUPPORTED and checked as such)."""
    from filterpy_amd._abi import FilterHipError, FK_ERR_UNSUPPORTED, FK_MODEL_PER_TRACK
    from gpu_util import run_kf_batch
    rs = np.random.RandomState(31 * n + m)
    N, T = 333, 9
    F = np.eye(n) + 0.05 * np.triu(rs.randn(n, n), 1)
    Q = 0.02 * np.eye(n)
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    x0, P0 = rs.randn(N, n), np.tile(4.0 * np.eye(n), (N, 1, 1))
    zs = rs.randn(T, N, m)
    mask = (rs.rand(T, N) > 0.2).astype(np.uint8)
    served = 0
    for kw in (dict(), dict(mask=mask), dict(update_first=True), dict(B=rs.randn(n, 2), us=rs.randn(T, N, 2)),
               dict(mode=FK_MODEL_PER_TRACK)):
        args = [F, Q, H, R]
        if kw.get("mode") == FK_MODEL_PER_TRACK:
            args = [np.tile(M, (N, 1, 1)) for M in args]
        ref = run_kf_batch(x0, P0, zs, *args, layout=layout, **kw)
        try:
            got = run_kf_batch(x0, P0, zs, *args, layout=layout, interleave=True, **kw)
        except FilterHipError as exc:
            assert exc.code == FK_ERR_UNSUPPORTED, (kw.keys(), exc)
            continue
        served += 1
        for a, b in zip(ref, got):
            assert np.array_equal(a, b, equal_nan=True), (n, m, layout, list(kw))
    if n <= 6 or (layout == "soa" and n <= 8):       # (dim_x 9 runs on the three- / four-lane kernels: two arrays only)
        assert served >= 2, served           # at least the plain and the masked call run on the specialised kernel


def test_interleaved_flag_checks_its_arguments():
    """the two pointers must be the halves of one array; final-state-only calls have nothing to interleave"""
    import torch
    from filterpy_amd import _engine as E
    from filterpy_amd._abi import FilterHipError, FK_ERR_BAD_ARG, FK_ERR_UNSUPPORTED
    n, m, N, T = 4, 2, 100, 3
    d = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS["aos"], update_first=0, alpha_sq=1.0, flags=2)
    F, Q, H, R = E.dev(np.eye(n)), E.dev(np.eye(n)), E.dev(np.eye(m, n)), E.dev(np.eye(m))
    z, x, P = E.dev(np.zeros((T, N, m))), E.dev(np.zeros((N, n))), E.dev(np.tile(np.eye(n), (N, 1, 1)))
    mu, mup = E.alloc_records((T,), N, n, "aos"), E.alloc_records((T,), N, n, "aos")
    c, cp = E.alloc_records((T,), N, n * n, "aos"), E.alloc_records((T,), N, n * n, "aos")
    with pytest.raises(FilterHipError) as ei:
        E.kf_batch_filter(d, F, Q, H, R, z, x, P, means=mu, covs=c, means_p=mup, covs_p=cp)
    assert ei.value.code == FK_ERR_BAD_ARG
    with pytest.raises(FilterHipError) as ei:
        E.kf_batch_filter(d, F, Q, H, R, z, x, P)
    assert ei.value.code == FK_ERR_BAD_ARG
    d16 = dict(d, n=12, m=2)
    F, Q, H, R = E.dev(np.eye(12)), E.dev(np.eye(12)), E.dev(np.eye(2, 12)), E.dev(np.eye(2))
    z, x, P = E.dev(np.zeros((T, N, 2))), E.dev(np.zeros((N, 12))), E.dev(np.tile(np.eye(12), (N, 1, 1)))
    mu, mup = E.alloc_records((T,), N, 12, "aos"), E.alloc_records((T,), N, 12, "aos")
    _, c, cp = E.alloc_cov_pair(T, N, 12, "aos")
    with pytest.raises(FilterHipError) as ei:
        E.kf_batch_filter(d16, F, Q, H, R, z, x, P, means=mu, covs=c, means_p=mup, covs_p=cp)
    assert ei.value.code == FK_ERR_UNSUPPORTED
    torch.cuda.synchronize()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m", [(4, 2), (9, 3), (12, 3)])
def test_bank_device_outputs_are_views_of_one_array_and_equal_the_host_outputs(n, m, layout):
    """KalmanFilterBank.batch_filter(device_outputs=True): the covariance histories are strided views of ONE allocation where
    the one-lane specialised kernel is the kernel of the call, two plain arrays otherwise ((9,3), dim_x 12: the several-lanes-
    per-track kernels), and in every case equal the host outputs of the same call bit for bit."""
    from filterpy_amd import _engine as E
    from filterpy_amd.kalman import KalmanFilterBank
    rs = np.random.RandomState(n + m)
    N, T = 200, 7
    zs = rs.randn(T, N, m)

    def bank():
        b = KalmanFilterBank(n, m, N, layout=layout)
        b.F = np.eye(n) + 0.05 * np.triu(np.ones((n, n)), 1)
        b.Q, b.R = 0.02 * np.eye(n), 0.5 * np.eye(m)
        b.H = np.eye(m, n)
        b.x, b.P = np.zeros((N, n)), np.tile(3.0 * np.eye(n), (N, 1, 1))
        return b
    host = bank().batch_filter(zs)
    b = bank()
    dev = b.batch_filter(zs, device_outputs=True)
    one_array = dev[1].untyped_storage().data_ptr() == dev[3].untyped_storage().data_ptr()
    assert one_array == (n <= 8), (n, layout, one_array)
    shapes = [(n,), (n, n), (n,), (n, n)]
    for h, d, shp in zip(host, dev, shapes):
        assert np.array_equal(h, E.from_records(d, layout, 1, shp))
    two = bank().batch_filter(zs, device_outputs=True, cov_interleave=False)
    assert two[1].is_contiguous() and two[1].untyped_storage().data_ptr() != two[3].untyped_storage().data_ptr()
    for a, c in zip(dev, two):
        assert np.array_equal(a.cpu().numpy(), c.cpu().numpy())


def test_bank_placement_probe_places_by_measurement_and_remembers_the_pair():
    """KalmanFilterBank.batch_filter(device_outputs=True, placement="probe") (filterpy_amd/placement.py: placed_pair): the two
    covariance histories are two dense arrays chosen by timing this very launch on candidate buffers; the pair is remembered
    per shape and handed out again only when nothing derived from it is alive.  Results equal the interleaved call bit for bit.
    Round 5: at dim_x <= 4 with histories of 256 MiB and more this is what placement=None -- the default -- does by itself,
    falling back to the interleaved array (not to two plain ones) where the probe cannot run."""
    import gc
    import torch
    from filterpy_amd import placement
    from filterpy_amd.kalman import KalmanFilterBank
    placement.forget_placed_pairs()
    n, m, N, T = 4, 2, 200_000, 12                       # 307 MB per covariance history
    rs = np.random.RandomState(3)
    zs = torch.as_tensor(rs.randn(T, N, m), device="cuda")

    def bank():
        b = KalmanFilterBank(n, m, N, layout="aos")
        b.F = np.eye(n) + 0.05 * np.triu(np.ones((n, n)), 1)
        b.Q, b.R, b.H = 0.02 * np.eye(n), 0.5 * np.eye(m), np.eye(m, n)
        b.x, b.P = np.zeros((N, n)), np.tile(3.0 * np.eye(n), (N, 1, 1))
        return b
    b0 = bank()
    ref = [t.clone() for t in b0.batch_filter(zs, device_outputs=True, placement="interleave")]
    assert b0.placement_info["method"] == "interleave"
    b1 = bank()
    out1 = b1.batch_filter(zs, device_outputs=True, placement="probe")
    assert b1.placement_info["method"] == "probe" and b1.placement_info["pairs"] >= 3, b1.placement_info
    assert out1[1].is_contiguous() and out1[3].is_contiguous()
    for a, c in zip(ref, out1):
        assert torch.equal(a, c)
    b2 = bank()
    out2 = b2.batch_filter(zs, device_outputs=True, placement="probe")      # out1 still alive: the pair is taken
    assert b2.placement_info["method"].startswith("plain allocation"), b2.placement_info
    assert out2[1].data_ptr() != out1[1].data_ptr()
    for a, c in zip(ref, out2):
        assert torch.equal(a, c)
    p1 = out1[1].data_ptr()
    del out1, out2
    gc.collect()
    b3 = bank()
    out3 = b3.batch_filter(zs, device_outputs=True, placement="probe")
    assert b3.placement_info["method"] == "cached" and out3[1].data_ptr() == p1, b3.placement_info
    for a, c in zip(ref, out3):
        assert torch.equal(a, c)
    # the default: the same measurement, unasked; a second default call while the first one's arrays are alive gets the
    # interleaved array (strided views of one allocation), not the lottery of two plain ones
    del out3, a, c                           # (the loop variables above still name two of its tensors)
    gc.collect()
    b4 = bank()
    out4 = b4.batch_filter(zs, device_outputs=True)
    assert b4.placement_info["method"] == "cached" and out4[1].data_ptr() == p1 and out4[1].is_contiguous(), b4.placement_info
    b5 = bank()
    out5 = b5.batch_filter(zs, device_outputs=True)
    assert b5.placement_info["method"] == "interleave" and "still in use" in b5.placement_info["note"], b5.placement_info
    assert out5[1].untyped_storage().data_ptr() == out5[3].untyped_storage().data_ptr()
    for a, c, e in zip(ref, out4, out5):
        assert torch.equal(a, c) and torch.equal(a, e)
    del out4, out5, a, c, e
    gc.collect()
    placement.forget_placed_pairs()
    b6 = bank()
    out6 = b6.batch_filter(zs, device_outputs=True)
    assert b6.placement_info["method"] == "probe" and out6[1].is_contiguous(), b6.placement_info
    for a, c in zip(ref, out6):
        assert torch.equal(a, c)
    del out6
    placement.forget_placed_pairs()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("ci", range(13))
def test_smoother_margin_on_badly_conditioned_models(ci, layout):
    """VERDICT r4 weak 2 / next 5.  Every smoother class (rts_kernel dim_x <= 8, rts_ml 9, rts_mlg 10..12, rts_mlx / rts_mlg
    13..16) on models whose backward recursion is badly conditioned (tests/golden/make_rts_conditioning.py: the unstable random
    models of tools/bench_configs.py, where round 4 measured 6.9e-11 at dim_x 10, and integrator chains up to cond(Pp) = 1e9),
    a bank of 70 copies so that tail lanes and several waves take part: every track bit-equal to the first, and that one inside
    max(1e-10, 2 x the reference's own spread under one-ulp perturbations of the smoother's inputs) -- the gain inside
    max(that, 8 x the reference's own distance from the exactly rounded gain), against the reference and against that gain."""
    from gpu_util import run_rts, tile_tracks
    g = golden("rts_conditioning")
    p = f"c{ci}_"
    N = 70
    xs, Ps, K, Pp = run_rts(tile_tracks(g[p + "mu"], N, 1), tile_tracks(g[p + "cov"], N, 1), g[p + "F"], g[p + "Q"], layout=layout)
    for a in (xs, Ps, K, Pp):
        assert np.array_equal(a, np.repeat(a[:, :1], N, axis=1))
    sp = g[p + "spread"]
    bar = {k: max(TOL, 2.0 * float(v)) for k, v in zip(("xs", "Ps", "K", "Pp"), sp)}
    bar["K"] = max(bar["K"], 8.0 * float(g[p + "K_ref_err"]))
    got = dict(xs=xs[:, 0], Ps=Ps[:, 0], K=K[:-1, 0], Pp=Pp[:-1, 0])
    ref = dict(xs=g[p + "xs"], Ps=g[p + "Ps"], K=g[p + "K"][:-1], Pp=g[p + "Pp"][:-1])
    errs = {k: rel_err_rows(got[k], ref[k]) for k in got}
    errs["K_exact"] = rel_err_rows(got["K"], g[p + "K_exact"][:-1])
    assert all(errs[k] < bar[k] for k in got) and errs["K_exact"] < bar["K"], (ci, layout, errs, bar)


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("n,m,org", [(9, 3, "three lanes per track (kf_ml / rts_ml)"), (12, 3, "four lanes (kf_mlg / rts_mlg)"),
                                     (14, 4, "four lanes forward, eight backward (kf_mlg / rts_mlx)"),
                                     (16, 8, "four lanes forward at dim_z 8, eight backward")])
def test_every_track_of_a_ragged_bank_against_the_oracle(n, m, org, layout):
    """VERDICT r4 weak 11 / next 9.  The other tests of the several-lanes-per-track kernels sample tracks against the oracle (and
    hold all of them to bit-equality or to the one-lane kernel); here EVERY track of a ragged random bank -- N = 777: twelve full
    workgroups of 64 tracks and a tail of 9, every track its own state, measurements and missing-measurement pattern -- is held
    to the oracle at 1e-10, forward (all four outputs + the final state) and backward (xs, Ps, K, Pp), once per lane
    organisation: an off-by-one in a tail quad or in the last lanes of a wave has nowhere to hide."""
    from gpu_util import run_kf_batch, run_rts
    rs = np.random.RandomState(7700 + 10 * n + m)
    N, T = 777, 8
    A = rs.randn(N, n, n)
    x0, P0 = rs.randn(N, n), 3.0 * (A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    F = np.eye(n) + 0.08 * rs.randn(n, n)
    B = rs.randn(n, n)
    Q = 0.1 * (B @ B.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    C = rs.randn(m, m)
    R = 0.5 * (C @ C.T / m + 0.5 * np.eye(m))
    mask = rs.rand(T, N) > 0.3
    zs = rs.randn(T, N, m) * 2
    zs[~mask] = np.nan
    got = run_kf_batch(x0, P0, zs, F, Q, H, R, layout=layout, mask=mask)
    ref = kf_oracle.kf_batch_filter_tracks(x0, P0, zs, F, Q, H, R, tracks=range(N), mask=mask)
    for k in range(4):
        assert np.isfinite(got[k]).all(), k
        assert rel_err_rows(_per_track(got[k]), _per_track(ref[k])) < TOL, (org, k)
    assert rel_err_rows(got[4], ref[0][-1]) < TOL and rel_err_rows(got[5], ref[1][-1]) < TOL
    sm = run_rts(ref[0], ref[1], F, Q, layout=layout)
    rsm = kf_oracle.rts_smoother_tracks(ref[0], ref[1], F, Q, tracks=range(N))
    for k in range(4):
        assert np.isfinite(sm[k]).all(), k
        assert rel_err_rows(_per_track(sm[k]), _per_track(rsm[k])) < TOL, (org, "smoother", k)
