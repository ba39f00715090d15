"""KalmanFilter.inv = numpy.linalg.pinv and rts_smoother(inv=numpy.linalg.pinv) (kalman_filter.py:363, 434, 541, 995, 1069)
against outputs of the LIVE reference frozen by tests/golden/make_kf_inv_golden.py: a singular S (identical rows of H, R = 0)
in update / batch_filter, singular predicted covariances (F a rank-one projector, Q = 0) in the smoother -- cases the default
inverse cannot serve.  The callable runs on the host between two launches of fk_kf_update_f64 / fk_kf_rts_f64
(FK_KF_FLAG_S_ONLY -> FK_KF_FLAG_SI_GIVEN, FK_KF_FLAG_PP_ONLY -> FK_KF_FLAG_PPINV_GIVEN; csrc/kf_given_inv.hip).

The same checks run on the CPU with the kernels replaced by the stand-ins of tests/fake_kf_engine.py (the host half of the path)."""
import os

import numpy as np
import pytest

from conftest import rel_err_rows

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "kf_inv.npz"))
CASES = [tuple(int(v) for v in c) for c in G["cases"]]
TOL = 1e-10
# a pseudo-inverse of a SINGULAR matrix amplifies the last bit of S by its cut-off decision: the reference's own K moves by
# ~1e-9 relative when S moves by an ulp at dim_z 8; everything downstream is held to this looser bar there
TOL_PINV = 2e-8


def _make(ci):
    from filterpy_amd.kalman import KalmanFilter
    n, m, nd = CASES[ci]
    g = lambda k: G[f"c{ci}_{k}"]          # noqa: E731
    kf = KalmanFilter(n, m)
    kf.x = np.zeros((n, 1)) if nd == 2 else np.zeros(n)
    kf.P, kf.F, kf.Q, kf.H, kf.R = g("P0").copy(), g("F").copy(), g("Q").copy(), g("H").copy(), g("R").copy()
    kf.inv = np.linalg.pinv
    return kf, g


def _close(got, ref, what, tol):
    got, ref = np.asarray(got, dtype=float), np.asarray(ref, dtype=float)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = rel_err_rows(got.reshape(1, -1), ref.reshape(1, -1))
    assert err < tol, (what, err)


def _check_steps(ci):
    kf, g = _make(ci)
    for t, z in enumerate(g("zs")):
        kf.predict()
        kf.update(z)
        for a in ("x", "P", "K", "y", "S", "SI"):
            _close(getattr(kf, a), g("step_" + a)[t], f"case {ci} step {t} {a}", TOL if a in ("y", "S") else TOL_PINV)


def _check_batch(ci, uf):
    kf, g = _make(ci)
    res = kf.batch_filter(list(g("zs")), update_first=bool(uf))
    for name, arr in zip(("means", "covs", "means_p", "covs_p"), res):
        ref = g(f"batch{uf}_{name}")
        assert arr.shape == ref.shape
        for t in range(len(ref)):
            _close(arr[t], ref[t], f"case {ci} batch {name}[{t}]", TOL_PINV)
    _close(kf.x, g(f"batch{uf}_xf"), "final x", TOL_PINV)
    _close(kf.P, g(f"batch{uf}_Pf"), "final P", TOL_PINV)


def _check_rts(ci):
    from filterpy_amd.kalman import KalmanFilter
    n, m, nd = CASES[ci]
    g = lambda k: G[f"c{ci}_{k}"]          # noqa: E731
    kf = KalmanFilter(n, m)
    kf.F, kf.Q = g("rts_F").copy(), np.zeros((n, n))
    T = len(g("rts_Xs"))
    r0 = kf.rts_smoother(g("rts_Xs"), g("rts_Ps"), inv=np.linalg.pinv)
    r1 = kf.rts_smoother(g("rts_Xs"), g("rts_Ps"), Fs=list(g("rts_Fs")), Qs=[np.zeros((n, n))] * T, inv=np.linalg.pinv)
    for tag, r in (("rts0_", r0), ("rts1_", r1)):
        for name, arr in zip(("x", "P", "K", "Pp"), r):
            ref = g(tag + name)
            assert arr.shape == ref.shape
            for t in range(T):
                if name == "K" and t == T - 1:
                    assert not np.any(arr[t])
                    continue
                _close(arr[t], ref[t], f"case {ci} {tag}{name}[{t}]", TOL if name == "Pp" else TOL_PINV)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_custom_inv_update_sequence(ci):
    _check_steps(ci)


@pytest.mark.gpu
@pytest.mark.parametrize("uf", [0, 1])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_custom_inv_batch_filter(ci, uf):
    _check_batch(ci, uf)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_custom_inv_rts_smoother(ci):
    _check_rts(ci)


@pytest.mark.gpu
def test_gpu_default_inv_still_raises_on_the_singular_S():
    """without the override the fused solve meets the singular S and reports it the reference's way"""
    kf, g = _make(1)
    kf.inv = np.linalg.inv
    kf.predict()
    with pytest.raises(np.linalg.LinAlgError):
        kf.update(g("zs")[0])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_gpu_given_inverse_flags_through_the_c_abi_on_a_bank(layout):
    """the two-launch protocol on a ragged bank of 700 filters through _engine (i.e. the C ABI), per-track models, a mask: with
    SI = inv(S) supplied by the caller the results equal the fused update's (same arithmetic but the inverse: 1e-10)"""
    import torch
    from filterpy_amd import _engine as E, _abi
    rs = np.random.RandomState(5)
    n, m, N = 5, 3, 700
    x0, z = rs.randn(N, n), rs.randn(N, m)
    A = rs.randn(N, n, n)
    P0 = A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    H = rs.randn(N, m, n)
    B = rs.randn(N, m, m)
    R = B @ B.transpose(0, 2, 1) / m + 0.3 * np.eye(m)
    mask = (rs.rand(N) > 0.2).astype(np.uint8)
    rec = lambda a: E.to_records(a, layout, 0)          # noqa: E731
    desc = dict(n=n, m=m, nu=0, model_mode=_abi.FK_MODEL_PER_TRACK, N=N, T=1, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    dmask = torch.as_tensor(mask, device="cuda")

    def outs():
        return dict(y=E.alloc_records((), N, m, layout).zero_(), K=E.alloc_records((), N, n * m, layout).zero_(),
                    S=E.alloc_records((), N, m * m, layout).zero_(), SI=E.alloc_records((), N, m * m, layout).zero_())
    xa, Pa, oa = rec(x0), rec(P0), outs()
    E.kf_update(desc, rec(H), rec(R), rec(z), xa, Pa, mask=dmask, **oa)
    xb, Pb, ob = rec(x0), rec(P0), outs()
    E.kf_update(dict(desc, flags=_abi.FK_KF_FLAG_S_ONLY), rec(H), rec(R), rec(z), xb, Pb, mask=dmask, **ob)
    assert np.array_equal(E.from_records(xb, layout, 0, (n,)), x0) and np.array_equal(E.from_records(Pb, layout, 0, (n, n)), P0)
    S = E.from_records(ob["S"], layout, 0, (m, m))
    live = mask.astype(bool)
    SI = np.zeros((N, m, m))
    SI[live] = np.linalg.inv(S[live])
    ob["SI"] = rec(SI)
    st = torch.zeros(N, dtype=torch.int32, device="cuda")
    E.kf_update(dict(desc, flags=_abi.FK_KF_FLAG_SI_GIVEN), rec(H), rec(R), rec(z), xb, Pb, mask=dmask, status=st, **ob)
    assert not st.any()
    for name, a, b, shp in (("x", xa, xb, (n,)), ("P", Pa, Pb, (n, n)), ("y", oa["y"], ob["y"], (m,)),
                            ("K", oa["K"], ob["K"], (n, m)), ("S", oa["S"], ob["S"], (m, m))):
        ga, gb = E.from_records(a, layout, 0, shp), E.from_records(b, layout, 0, shp)
        assert rel_err_rows(gb[live].reshape(live.sum(), -1), ga[live].reshape(live.sum(), -1)) < 1e-10, name
        if name in ("x", "P"):
            assert np.array_equal(gb[~live], ga[~live]), name          # update(None): untouched


# ---- the host half on the CPU (stand-in kernels) ------------------------------------------------------------------------
@pytest.fixture
def fake(monkeypatch):
    import fake_kf_engine
    return fake_kf_engine.install(monkeypatch)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_host_custom_inv_against_the_golden(fake, ci):
    _check_steps(ci)
    _check_batch(ci, 0)
    _check_batch(ci, 1)
    _check_rts(ci)
