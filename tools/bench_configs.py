#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2..4] on ONE MI355X (bench.py is configs[1]):

  C3  1e5 tracks dim_x=9 dim_z=3 (CV-3D, dt=0.1) batch_filter + rts_smoother backward pass
  C4  UKF dim_x=6 dim_z=3, 1e5 tracks: sigma_points / unscented_transform standalone, fused linear UKF
  C5  systematic_resample: 1e3 filters x 8e3 particles (one GPU's view), plus a few 8e6-particle filters

Each line: kernel, units/s, achieved algorithmic GB/s and fraction of the 8 TB/s HBM peak,
plus a parity figure against the oracle on a sample.  Writes JSON lines to stdout.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("FK_BENCH_LIB"):      # A/B of a variant build (csrc/exp_build/*.so) in one lease: this tool only, not the package
    from filterpy_amd import _abi as _abi_for_variant
    _abi_for_variant.LIB_PATH = os.path.abspath(os.environ["FK_BENCH_LIB"])
PEAK = 8000.0


# bench.py collects the rows of BASELINE configs[2..4] through these hooks (its "configs" key): ROWS = a list to append to
# instead of printing, REPS = event-timed launches per row (None: each call site's own count)
ROWS = None
REPS = None


def timeit(fn, warm=2, reps=5, pre=None):
    """median milliseconds of `fn` between two events on torch's current stream (the stream the C ABI launches on);
    `pre` (state reset) runs in front of every call, outside the event pair"""
    import torch
    reps = REPS or reps
    for _ in range(warm):
        if pre:
            pre()
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        if pre:
            pre()
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def emit(name, units, unit_name, ms, bytes_per_unit, **extra):
    gbs = units * bytes_per_unit / (ms * 1e-3) / 1e9
    row = dict(kernel=name, units=units, unit=unit_name, ms=ms, units_per_s=units / (ms * 1e-3),
               alg_bytes_per_unit=bytes_per_unit, achieved_GBs=gbs, frac_of_8TBs=gbs / PEAK, **extra)
    if ROWS is not None:
        ROWS.append(row)
    else:
        print(json.dumps(row), flush=True)


def cv3d_model(dt=0.1):
    F2 = np.array([[1, dt, dt * dt / 2], [0, 1, dt], [0, 0, 1.]])
    F = np.kron(np.eye(3), F2)
    H = np.zeros((3, 9))
    H[0, 0] = H[1, 3] = H[2, 6] = 1.0
    q = np.array([[dt ** 4 / 4, dt ** 3 / 2, dt ** 2 / 2], [dt ** 3 / 2, dt ** 2, dt], [dt ** 2 / 2, dt, 1.]]) * 0.01
    return F, np.kron(np.eye(3), q), H, 0.25 * np.eye(3)


def rel(a, b):
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    s = np.max(np.abs(b), axis=1)
    s[s == 0] = 1
    return float(np.max(np.max(np.abs(a - b), axis=1) / s))


def config3(layout, N, T):
    import torch
    from filterpy_amd import _engine as E
    from oracle import kf_oracle
    n, m = 9, 3
    F, Q, H, R = cv3d_model()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.zeros((N, n) if layout == "aos" else (n, N), dtype=torch.float64, device=dev)
    P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    x, P = x0.clone(), P0.clone()
    outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
            E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    dF, dQ, dH, dR = (E.dev(M) for M in (F, Q, H, R))
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)

    def reset():
        x.copy_(x0)
        P.copy_(P0)

    def fwd():
        E.kf_batch_filter(desc, dF, dQ, dH, dR, z, x, P, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
    ms = timeit(fwd, pre=reset)
    assert not st.any()
    sample = [0, 255, 256, N - 1]
    zs_h = (z[:, sample] if layout == "aos" else z[:, :, sample].permute(0, 2, 1)).cpu().numpy()
    ref = kf_oracle.kf_batch_filter_tracks(np.zeros((4, n)), np.tile(10 * np.eye(n), (4, 1, 1)), zs_h, F, Q, H, R, tracks=range(4))

    def pick(t, rec):                     # the sampled tracks of a [T][N][E] / [T][E][N] history, as a host array [T][4]+rec
        h = (t[:, sample] if layout == "aos" else t[:, :, sample].permute(0, 2, 1)).cpu().numpy()
        return h.reshape(T, len(sample), *rec)
    par = max(rel(pick(outs[k], r).reshape(-1, int(np.prod(r))), ref[k].reshape(-1, int(np.prod(r))))
              for k, r in ((0, (n,)), (1, (n, n)), (2, (n,)), (3, (n, n))))
    emit(f"C3 kf batch_filter (9,3) {layout}", N * T, "track-steps", ms, 8 * (m + 2 * n + 2 * n * n), parity_max_rel=par,
         kernel_fn="fk::kf_ml_kernel<9,3> (three lanes per track, persistent grid)", config="configs[2]")

    so = [E.alloc_records((T,), N, n, layout)] + [E.alloc_records((T,), N, n * n, layout) for _ in range(3)]

    def bwd():
        E.kf_rts(desc, dF, dQ, outs[0], outs[1], so[0], so[1], so[2], so[3], convention=0, status=st)
    ms = timeit(bwd)
    assert not st.any()
    sm = kf_oracle.rts_smoother_tracks(ref[0], ref[1], F, Q, tracks=range(4))
    par = max(rel(pick(so[k], r).reshape(-1, int(np.prod(r))), sm[k].reshape(-1, int(np.prod(r))))
              for k, r in ((0, (n,)), (1, (n, n)), (2, (n, n)), (3, (n, n))))
    emit(f"C3 rts_smoother n=9 {layout}", N * T, "track-steps", ms, 8 * (2 * n + 4 * n * n), parity_max_rel=par,
         kernel_fn="fk::rts_ml_kernel<9> (three lanes per track)", config="configs[2]")


def config_kf(layout, n, m, N, T):
    """plain batch_filter at other (dim_x, dim_z): random stable model, shared"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import kf_oracle
    rs = np.random.RandomState(n * 10 + m)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    A = rs.randn(n, n)
    Q = 0.1 * (A @ A.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    B = rs.randn(m, m)
    R = 0.5 * (B @ B.T / m + 0.5 * np.eye(m))
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.zeros((N, n) if layout == "aos" else (n, N), dtype=torch.float64, device=dev)
    P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    x, P = x0.clone(), P0.clone()
    outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
            E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    dF, dQ, dH, dR = (E.dev(M) for M in (F, Q, H, R))
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)

    def fwd():
        x.copy_(x0)
        P.copy_(P0)
        E.kf_batch_filter(desc, dF, dQ, dH, dR, z, x, P, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
    ms = timeit(fwd)
    assert not st.any()
    sample = [0, 255, 256, N - 1]
    zs_h = (z[:, sample] if layout == "aos" else z[:, :, sample].permute(0, 2, 1)).cpu().numpy()
    ref = kf_oracle.kf_batch_filter_tracks(np.zeros((4, n)), np.tile(10 * np.eye(n), (4, 1, 1)), zs_h, F, Q, H, R, tracks=range(4))
    cov = E.from_records(outs[1], layout, 1, (n, n))[:, sample]
    par = rel(cov.reshape(-1, n * n), ref[1].reshape(-1, n * n))
    emit(f"KF batch_filter ({n},{m}) N={N} {layout}", N * T, "track-steps", ms, 8 * (m + 2 * n + 2 * n * n), parity_max_rel=par)
    if os.environ.get("KF_NO_RTS"):      # sweeps of the forward kernels only
        return
    so = [E.alloc_records((T,), N, n, layout)] + [E.alloc_records((T,), N, n * n, layout) for _ in range(3)]
    ms = timeit(lambda: E.kf_rts(desc, dF, dQ, outs[0], outs[1], so[0], so[1], so[2], so[3], convention=0, status=st))
    assert not st.any()
    sm = kf_oracle.rts_smoother_tracks(ref[0], ref[1], F, Q, tracks=range(4))
    Ps = E.from_records(so[1], layout, 1, (n, n))[:, sample]
    emit(f"RTS smoother n={n} N={N} {layout}", N * T, "track-steps", ms, 8 * (2 * n + 4 * n * n),
         parity_max_rel=rel(Ps.reshape(-1, n * n), sm[1].reshape(-1, n * n)))


def config_generic(layout, N, T):
    """The generic kernel (kf_kernels.hip) on the C2 model: per-track models, per-step models,
    control input, update_first -- the calls the specialised kernel does not take."""
    import torch
    from filterpy_amd import _engine as E
    sys.path.insert(0, ROOT)
    from bench import c2_model, c2_inputs_device
    n, m = 4, 2
    F, Q, H, R = c2_model()
    dev = torch.device("cuda")
    x0, P0, z = c2_inputs_device(N, T, layout, 7, dev)
    x, P = x0.clone(), P0.clone()
    outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
            E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
    st = torch.zeros(N, dtype=torch.int32, device=dev)

    def rep(M, lead):
        t = torch.as_tensor(M, dtype=torch.float64, device=dev)
        t = t.reshape(1, -1).repeat(N, 1) if lead == 0 else t.reshape(1, 1, -1).repeat(T, N, 1)
        if layout == "soa":
            t = t.transpose(-1, -2).contiguous()
        return t
    cases = {
        "shared, update_first": (0, [E.dev(M) for M in (F, Q, H, R)], dict(update_first=1), 0),
        "shared, control input dim_u=2": (0, [E.dev(M) for M in (F, Q, H, R)], dict(nu=2), 16),
        "per-track models": (1, [rep(M, 0) for M in (F, Q, H, R)], {}, 8 * (2 * n * n + m * n + m * m) / T),
        "per-step shared models": (3, [E.dev(np.tile(M, (T, 1, 1))) for M in (F, Q, H, R)], {}, 0),
        "per-track-per-step models": (2, [rep(M, 1) for M in (F, Q, H, R)], {}, 8 * (2 * n * n + m * n + m * m)),
    }
    for name, (mode, mods, kw, extra) in cases.items():
        desc = dict(n=n, m=m, nu=0, model_mode=mode, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
        desc.update(kw)

        ctrl = {}
        if desc["nu"]:
            g2 = torch.Generator(device=dev)
            g2.manual_seed(11)
            ctrl = dict(B=E.dev(np.array([[0.5, 0.0], [1.0, 0.0], [0.0, 0.5], [0.0, 1.0]])),
                        u=torch.randn((T, N, 2) if layout == "aos" else (T, 2, N), generator=g2, device=dev, dtype=torch.float64))

        def run():
            x.copy_(x0)
            P.copy_(P0)
            E.kf_batch_filter(desc, *mods, z, x, P, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st, **ctrl)
        ms = timeit(run, warm=1, reps=3)
        assert not st.any()
        emit(f"KF (4,2) {name} N={N} {layout}", N * T, "track-steps", ms, 8 * (m + 2 * n + 2 * n * n) + extra)


def config_extras(layout, n, m, N, T):
    """fk_kf_batch_filter_ex_f64 with all six histories (a Saver's y / K / S / SI / log-likelihood /
    mahalanobis): kf_fast's extras instantiation against the generic kernel (FK_NO_FAST_EX=1), same buffers."""
    import torch
    from filterpy_amd import _engine as E
    rs = np.random.RandomState(5)
    dev = torch.device("cuda")
    F = np.eye(n) + 0.02 * rs.randn(n, n)
    A = rs.randn(n, n)
    Q = 0.1 * (A @ A.T / n + 0.5 * np.eye(n))
    H = rs.randn(m, n)
    R = 0.5 * np.eye(m)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    z = E.alloc_records((T,), N, m, layout)
    z.copy_(torch.randn(z.shape, generator=g, device=dev, dtype=torch.float64))
    x0 = E.alloc_records((), N, n, layout).zero_()
    P0 = E.to_records(np.tile(10.0 * np.eye(n), (N, 1, 1)), layout, 0)
    x, P = x0.clone(), P0.clone()
    outs = [E.alloc_records((T,), N, w, layout) for w in (n, n * n, n, n * n)]
    ex = {k: E.alloc_records((T,), N, w, layout) for k, w in (("y", m), ("K", n * m), ("S", m * m), ("SI", m * m))}
    ex["log_likelihood"] = torch.empty((T, N), dtype=torch.float64, device=dev)
    ex["mahalanobis"] = torch.empty((T, N), dtype=torch.float64, device=dev)
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    mods = [E.dev(M) for M in (F, Q, H, R)]

    def run():
        x.copy_(x0)
        P.copy_(P0)
        E.kf_batch_filter_ex(desc, *mods, z, x, P, ex, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
    nbytes = 8 * (m + 2 * n + 2 * n * n + m + n * m + 2 * m * m + 2)
    # (dim_x >= 9: the four-lane kernel's EX instantiations by default; FK_NO_MLG_EX=1 -> kf_fast's extras / generic)
    mlg = n >= 9
    rows = ([(None, None, "four-lane EX")] if mlg else []) + ([("1", None, "kf_fast extras")] if n <= 9 else []) + [("1", "1", "generic kernel")]
    for no_mlg, no_fast, name in rows:
        if no_mlg:
            os.environ["FK_NO_MLG_EX"] = no_mlg
        if no_fast:
            os.environ["FK_NO_FAST_EX"] = no_fast
        ms = timeit(run, warm=1, reps=3 if name != "generic kernel" or n < 10 else 1)
        os.environ.pop("FK_NO_FAST_EX", None)
        os.environ.pop("FK_NO_MLG_EX", None)
        assert not st.any()
        emit(f"KF ({n},{m}) batch_filter + six Saver histories, {name}, N={N} {layout}", N * T, "track-steps", ms, nbytes)


def config_imm(layout, n, m, nm, N, T):
    """N banks of nm filters, T x {IMM predict; update} in one launch.
    Algorithmic bytes per bank-step: z in, combined x, P and mu out."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import imm_oracle
    rs = np.random.RandomState(100 * n + 10 * m + nm)
    Fs = np.array([np.eye(n) + 0.03 * (j + 1) * rs.randn(n, n) for j in range(nm)])
    Qs = np.array([0.05 * (j + 1) * np.eye(n) for j in range(nm)])
    Hs = np.array([np.eye(m, n)] * nm)
    Rs = np.array([0.5 * np.eye(m)] * nm)
    M = np.full((nm, nm), 0.05 / (nm - 1)) + (0.95 - 0.05 / (nm - 1)) * np.eye(nm)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    xs0 = torch.zeros((N, nm * n) if layout == "aos" else (nm * n, N), dtype=torch.float64, device=dev)
    Ps0 = (4.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, nm)
    Ps0 = Ps0.contiguous() if layout == "aos" else Ps0.T.contiguous()
    mu0 = torch.full((N, nm) if layout == "aos" else (nm, N), 1.0 / nm, dtype=torch.float64, device=dev)
    xs, Ps, mu = xs0.clone(), Ps0.clone(), mu0.clone()
    out = dict(x_out=E.alloc_records((T,), N, n, layout), P_out=E.alloc_records((T,), N, n * n, layout),
               mu_out=E.alloc_records((T,), N, nm, layout))
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    d = [E.dev(a) for a in (Fs, Qs, Hs, Rs, M)]

    def run():
        xs.copy_(xs0)
        Ps.copy_(Ps0)
        mu.copy_(mu0)
        E.imm_batch(n, m, nm, N, T, layout, *d, z, xs, Ps, mu, status=st, **out)
    ms = timeit(run)
    assert not st.any()
    sample = [0, 255, 256, N - 1]
    zs_h = (z[:, sample] if layout == "aos" else z[:, :, sample].permute(0, 2, 1)).cpu().numpy()
    P_got = E.from_records(out["P_out"], layout, 1, (n, n))[:, sample]
    mu_got = E.from_records(out["mu_out"], layout, 1, (nm,))[:, sample]
    par, parmu = 0.0, 0.0
    for k in range(4):
        r = imm_oracle.imm_batch(np.zeros((nm, n)), np.tile(4 * np.eye(n), (nm, 1, 1)), np.full(nm, 1.0 / nm), M,
                                 zs_h[:, k], Fs, Qs, Hs, Rs)
        par = max(par, rel(P_got[:, k].reshape(-1, n * n), r[1].reshape(-1, n * n)))
        parmu = max(parmu, float(np.max(np.abs(mu_got[:, k] - r[2]))))
    emit(f"IMM ({n},{m}) x{nm} models N={N} {layout}", N * T, "bank-steps", ms, 8 * (m + n + n * n + nm),
         parity_max_rel=par, mu_max_abs=parmu, filter_steps_per_s=N * T * nm / (ms * 1e-3))


def config_steady(layout, n, m, N, T):
    """steady-state filter (fixed gain), T x {x = Fx; y = z - Hx; x += Ky} in one launch.
    Algorithmic bytes per track-step: z in, prior and posterior x out."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import kf_oracle
    rs = np.random.RandomState(n + m)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    H = np.eye(m, n)
    K = 0.3 * rs.rand(n, m)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.zeros((N, n) if layout == "aos" else (n, N), dtype=torch.float64, device=dev)
    x = x0.clone()
    means, means_p = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n, layout)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    d = [E.dev(a) for a in (F, H, K)]

    def run():
        x.copy_(x0)
        E.kf_steadystate(desc, *d, z, x, means=means, means_p=means_p)
    ms = timeit(run)
    sample = [0, 255, N - 1]
    zs_h = (z[:, sample] if layout == "aos" else z[:, :, sample].permute(0, 2, 1)).cpu().numpy()
    got = E.from_records(means, layout, 1, (n,))[:, sample]
    par = max(rel(got[:, k], kf_oracle.steadystate_filter(np.zeros(n), list(zs_h[:, k]), F, H, K)[0]) for k in range(3))
    emit(f"steady-state KF ({n},{m}) N={N} {layout}", N * T, "track-steps", ms, 8 * (m + 2 * n), parity_max_rel=par)


def config_ukf(layout, n, m, N, T):
    """The fused linear UKF (filter + smoother) at any size the library fuses -- dim_x 10..16 once the several-lane kernels
    are on (csrc/ukf_mlg.hip; FK_UKF_MLG) -- on a dense model, last track against the oracle."""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    alpha, beta, kappa = .5, 2., 3. - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    if not E.ukf_linear_supported(n, m, E.pair_weights(Wm, Wc, n)):
        return
    rs = np.random.RandomState(n * 10 + m)
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), 0.01 * np.eye(n), 0.5 * np.eye(m)
    sc = alpha ** 2 * (n + kappa)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn((N, n) if layout == "aos" else (n, N), generator=g, device=dev, dtype=torch.float64)
    P0 = (5.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    x, P = x0.clone(), P0.clone()
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]

    def fwd():
        x.copy_(x0)
        P.copy_(P0)
        E.ukf_linear_batch(n, m, N, T, layout, sc, *dd, z, x, P, means=means, covs=covs, status=st, paired=True)
    ms = timeit(fwd)
    assert not st.any()
    trk = N - 1
    zs_h = (z[:, trk] if layout == "aos" else z[:, :, trk]).cpu().numpy()
    x0h = (x0[trk] if layout == "aos" else x0[:, trk]).cpu().numpy()
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0h, 5 * np.eye(n), list(zs_h), lambda s, d: F @ s, lambda s: H @ s, 1.0, Q, R, alpha, beta, kappa)
    mu, cov = E.from_records(means, layout, 1, (n,))[:, trk], E.from_records(covs, layout, 1, (n, n))[:, trk]
    par = max(rel(mu, mu_ref), rel(cov.reshape(T, -1), cov_ref.reshape(T, -1)))
    emit(f"fused linear UKF ({n},{m}) N={N} {layout}", N * T, "track-steps", ms, 8 * (m + n + n * n), parity_max_rel=par)
    if not E.ukf_linear_rts_supported(n, True):
        return
    xs, ps, Ks = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout), E.alloc_records((T,), N, n * n, layout)
    ms = timeit(lambda: E.ukf_linear_rts(n, N, T, layout, sc, dd[0], dd[2], dd[4], dd[5], means, covs, xs, ps, K=Ks, status=st, paired=True))
    assert not st.any()
    xr, Pr, Kr = ukf_oracle.ukf_rts_smoother(mu_ref, cov_ref, lambda s, d: F @ s, 1.0, Q, alpha, beta, kappa)
    gx, gp = E.from_records(xs, layout, 1, (n,))[:, trk], E.from_records(ps, layout, 1, (n, n))[:, trk]
    par = max(rel(gx, xr), rel(gp.reshape(T, -1), Pr.reshape(T, -1)))
    emit(f"fused linear UKF smoother n={n} N={N} {layout}", N * T, "track-steps", ms, 8 * (2 * n + 3 * n * n), parity_max_rel=par)


def config4(layout, N, T, sizes=(1, 10, 50)):
    """sizes: the standalone sigma_points / unscented_transform launches run at N x each of these (bench.py: (1,))"""
    import torch
    from filterpy_amd import _engine as E
    from oracle import ukf_oracle
    n, m, k = 6, 3, 13
    alpha, beta, kappa = .1, 2., -3.
    lam = alpha ** 2 * (n + kappa) - n
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    dt = 0.1
    F = np.eye(n)
    for i in range(3):
        F[i, i + 3] = dt
    H = np.zeros((m, n))
    H[0, 0] = H[1, 1] = H[2, 2] = 1.0
    Q, R = 0.01 * np.eye(n), 0.5 * np.eye(m)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    for NN in [N * k_ for k_ in sizes]:     # 50N = 5e6 tracks: the (2n+1)n record slab stays below the 4 GiB addressing limit
        x = torch.randn((NN, n) if layout == "aos" else (n, NN), generator=g, device=dev, dtype=torch.float64)
        P = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(NN, 1)
        P = P.contiguous() if layout == "aos" else P.T.contiguous()
        sig = E.alloc_records((), NN, k * n, layout)
        xo, Po = E.alloc_records((), NN, n, layout), E.alloc_records((), NN, n * n, layout)
        dWm, dWc, dQ = E.dev(Wm), E.dev(Wc), E.dev(Q)
        ms = timeit(lambda: E.ut_sigma_points(n, NN, layout, lam + n, x, P, sig))
        par = None
        if NN == N:                         # the standalone blocks against the oracle on a sample (every track: tests/test_gpu_baseline_configs.py)
            smp = [0, 63, 64, NN - 1]
            xh = (x[smp] if layout == "aos" else x[:, smp].T).cpu().numpy()
            sg = (sig[smp] if layout == "aos" else sig[:, smp].T).cpu().numpy().reshape(len(smp), k, n)
            par = max(rel(sg[i], ukf_oracle.merwe_sigma_points(xh[i], 10.0 * np.eye(n), alpha, kappa)) for i in range(len(smp)))
        emit(f"C4 sigma_points n=6 N={NN} {layout}", NN, "tracks", ms, 8 * (n + n * n + k * n), kernel_fn="fk::sigma_kernel<6>",
             config="configs[3]", **({"parity_max_rel": par} if par is not None else {}))
        ms = timeit(lambda: E.ut_transform(n, k, NN, layout, sig, dWm, dWc, dQ, xo, Po))
        if NN == N:
            xg = (xo[smp] if layout == "aos" else xo[:, smp].T).cpu().numpy()
            Pg = (Po[smp] if layout == "aos" else Po[:, smp].T).cpu().numpy().reshape(len(smp), n, n)
            par = 0.0
            for i in range(len(smp)):
                xr, Pr = ukf_oracle.unscented_transform(sg[i], Wm, Wc, Q)
                par = max(par, rel(xg[i][None], xr[None]), rel(Pg[i].reshape(1, -1), Pr.reshape(1, -1)))
        emit(f"C4 unscented_transform n=6 N={NN} {layout}", NN, "tracks", ms, 8 * (k * n + n + n * n), kernel_fn="fk::ut_reg_kernel<6>",
             config="configs[3]", **({"parity_max_rel": par} if NN == N else {}))
    # fused linear UKF over T steps
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn((N, n) if layout == "aos" else (n, N), generator=g, device=dev, dtype=torch.float64)
    P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    x, P = x0.clone(), P0.clone()
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]

    paired = E.pair_weights(Wm, Wc, n)      # looked at once, outside the timed calls (Merwe's weights: True)

    def reset():
        x.copy_(x0)
        P.copy_(P0)

    def run():
        E.ukf_linear_batch(n, m, N, T, layout, lam + n, *dd, z, x, P, means=means, covs=covs, status=st, paired=paired)
    ms = timeit(run, pre=reset)
    assert not st.any()
    trk = 7
    zs_h = (z[:, trk] if layout == "aos" else z[:, :, trk]).cpu().numpy()
    x0h = (x0[trk] if layout == "aos" else x0[:, trk]).cpu().numpy()
    mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0h, 10 * np.eye(n), list(zs_h), lambda s, d: F @ s, lambda s: H @ s,
                                                  dt, Q, R, alpha, beta, kappa)
    mu = E.from_records(means, layout, 1, (n,))[:, trk]
    cov = E.from_records(covs, layout, 1, (n, n))[:, trk]
    par = max(rel(mu, mu_ref), rel(cov.reshape(T, -1), cov_ref.reshape(T, -1)))
    emit(f"C4 fused linear UKF (6,3) {layout}", N * T, "track-steps", ms, 8 * (m + n + n * n), parity_max_rel=par,
         kernel_fn="fk::ukf_linear_kernel<6,3> (pair-regrouped sums)", config="configs[3]")


def config5(shapes=((1000, 8000), (125, 8000), (8, 8_000_000), (1, 8_000_000)), stratified=True):
    import torch
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    dev = torch.device("cuda")
    for Fn, Np in shapes:
        rs = np.random.RandomState(5)
        w = rs.rand(min(Fn, 8), Np)
        w /= w.sum(axis=1, keepdims=True)
        wd = E.dev(np.tile(w, (Fn // w.shape[0] + 1, 1))[:Fn])
        u = E.dev(rs.rand(Fn))
        idx = torch.empty((Fn, Np), dtype=torch.int32, device=dev)
        st = torch.zeros(Fn, dtype=torch.int32, device=dev)
        ms = timeit(lambda: E.resample_systematic(Fn, Np, wd, u, idx, st), warm=1, reps=3)
        flagged = int((st != 0).sum())      # (a position >= cumsum[-1]: the reference's IndexError; the index stored is N - 1)
        # bit-exact against the merge loop (C restatement of resampling.py:139-149): the first, one in the middle and the last
        # filter of the call, every index
        exact = True
        uh = u.cpu().numpy()
        for f in sorted({0, Fn // 2, Fn - 1}):
            ref = np.minimum(ro.systematic_np(w[f % w.shape[0]], float(uh[f])), Np - 1)
            exact = exact and bool(np.array_equal(idx[f].cpu().numpy(), ref))
        emit(f"C5 systematic_resample {Fn} filters x {Np} particles", Fn * Np, "particles", ms, 12, bit_exact=exact, status_flagged=flagged,
             kernel_fn="fk::resample_whole_quick_kernel" if Np <= 8192 else "fk::resample_onepass_kernel (+ its two guard launches)",
             config="configs[4]")
        if Np <= 8000 and stratified:
            us = E.dev(rs.rand(Fn, Np))
            ms = timeit(lambda: E.resample_stratified(Fn, Np, wd, us, idx, st), warm=1, reps=3)
            emit(f"C5 stratified_resample {Fn} filters x {Np} particles", Fn * Np, "particles", ms, 20)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--layouts", default="soa,aos")
    ap.add_argument("--N", type=int, default=100_000)
    ap.add_argument("--T", type=int, default=100)
    a = ap.parse_args()
    if os.environ.get("FK_LIB"):      # an experimental build of the library (A/B of compile-time choices in one lease); tools only
        from filterpy_amd import _abi
        _abi.LIB_PATH = os.path.abspath(os.environ["FK_LIB"])
    for lay in a.layouts.split(","):
        if "3" in a.configs:
            config3(lay, a.N, a.T)
        if "4" in a.configs:
            config4(lay, a.N, a.T)
        if "7" in a.configs:
            config_generic(lay, 200_000, a.T)
        if "a" in a.configs and os.environ.get("KF_DIMS"):      # KF_DIMS="3x2,3x3": any instantiation, ~7.2e6 / dim_x^2 tracks
            for nm in os.environ["KF_DIMS"].split(","):
                n, m = (int(v) for v in nm.split("x"))
                config_kf(lay, n, m, max(20_000, 18_000_000 // (n * n)) // 1000 * 1000, a.T)
        elif "a" in a.configs:      # dims served by the lean specialised instantiations
            config_kf(lay, 3, 1, 2_000_000, a.T)
            config_kf(lay, 5, 2, 500_000, a.T)
            config_kf(lay, 6, 2, 400_000, a.T)
            config_kf(lay, 7, 4, 200_000, a.T)
            config_kf(lay, 8, 4, 200_000, a.T)
        if "b" in a.configs:      # dims above 9: four lanes per track (kf_mlg.hip) vs the padded one-lane kernels (FK_NO_MLG=1)
            for (n, m, N) in ((10, 2, 100_000), (12, 3, 100_000), (14, 4, 80_000), (16, 4, 60_000)):
                config_kf(lay, n, m, N, a.T)
                if os.environ.get("BENCH_PADDED"):     # the padded kernels take ~1 s per launch at this size
                    os.environ["FK_NO_MLG"] = "1"
                    config_kf(lay, n, m, N // 8, a.T)
                    del os.environ["FK_NO_MLG"]
        if "e" in a.configs:      # Saver histories from the specialised kernel (EXTRAS_DIMS="5x3,6x4": other instantiations)
            dims = os.environ.get("EXTRAS_DIMS")
            if dims:
                for nm in dims.split(","):
                    n, m = (int(v) for v in nm.split("x"))
                    config_extras(lay, n, m, max(20_000, 7_200_000 // (n * n)) // 1000 * 1000, a.T)
            else:
                config_extras(lay, 4, 2, 500_000, a.T)
                config_extras(lay, 6, 3, 200_000, a.T)
                config_extras(lay, 9, 3, 100_000, a.T)
                config_extras(lay, 12, 3, 100_000, a.T)
                config_extras(lay, 16, 4, 60_000, a.T)
        if "6" in a.configs:
            config_kf(lay, 6, 3, 300_000, a.T)
            config_kf(lay, 4, 2, 500_000, a.T)
            config_kf(lay, 2, 1, 2_000_000, a.T)
        if "9" in a.configs:
            config_steady(lay, 4, 2, 4_000_000, a.T)
            config_steady(lay, 9, 3, 1_000_000, a.T)
        if "8" in a.configs:
            config_imm(lay, 4, 2, 2, 500_000, a.T)
            config_imm(lay, 4, 2, 3, 300_000, a.T)
            config_imm(lay, 6, 3, 2, 200_000, a.T)
            config_imm(lay, 2, 1, 2, 1_000_000, a.T)
        if "r" in a.configs:      # the rolled IMM classes (banks in scratch memory): (9,4) x {2, 4, 8}, (16,8) x 2 -- VERDICT r3 next 8
            config_imm(lay, 9, 4, 2, 100_000, 20)
            config_imm(lay, 9, 3, 4, 100_000, 20)
            config_imm(lay, 9, 4, 5, 50_000, 20)
            config_imm(lay, 9, 4, 8, 50_000, 20)
            config_imm(lay, 16, 8, 2, 50_000, 20)
            # round 6: banks the one-lane-per-filter kernel took over from the padded (9,4) class (csrc/imm_lanes.hip)
            config_imm(lay, 4, 2, 4, 200_000, 20)
            config_imm(lay, 6, 3, 4, 100_000, 20)
            config_imm(lay, 9, 4, 16, 25_000, 20)
        if "u" in a.configs:      # the fused linear UKF above dim_x 9 (several lanes per track; rows appear once FK_UKF_MLG is on)
            for (n, m, N) in ((10, 2, 100_000), (12, 3, 100_000), (14, 4, 80_000), (16, 4, 60_000), (16, 8, 60_000)):
                config_ukf(lay, n, m, N, 50)
        if "S" in a.configs:      # STEADY_DIMS="16x8,14x6": the steady-state filter at any shape, 8e6 / dim_x tracks
            for nm in os.environ.get("STEADY_DIMS", "16x8").split(","):
                n, m = (int(v) for v in nm.split("x"))
                config_steady(lay, n, m, 8_000_000 // n // 1000 * 1000, a.T)
        if "s" in a.configs:      # steady-state / IMM above (9,4) (round 4's padded classes)
            config_steady(lay, 16, 8, 500_000, a.T)
    if "5" in a.configs:
        config5()
