#!/usr/bin/env python3
"""A/B timing of the (9,3) kernels on one GPU: three lanes per track (FK_ML_PAIRS=1|0: 16- or 8-byte accesses) vs one lane per track
(FK_NO_ML=1), with and without the four per-step outputs.  SOA, shared model, N tracks x T steps."""
import json
import os
import sys


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from filterpy_amd import _engine as E
    from tools.bench_configs import cv3d_model, timeit
    N, T, n, m = int(os.environ.get("ML_N", 100000)), 100, 9, 3
    F, Q, H, R = cv3d_model()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    z = torch.randn((T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.zeros((n, N), dtype=torch.float64, device=dev)
    P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1).T.contiguous()
    x, P = x0.clone(), P0.clone()
    outs = [E.alloc_records((T,), N, n, "soa"), E.alloc_records((T,), N, n * n, "soa"),
            E.alloc_records((T,), N, n, "soa"), E.alloc_records((T,), N, n * n, "soa")]
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    d = [E.dev(a) for a in (F, Q, H, R)]
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS["soa"], update_first=0, alpha_sq=1.0)
    for env in ({}, {"FK_ML_PAIRS": "0"}, {"FK_NO_ML": "1"}):
        for k in ("FK_ML_WAVES", "FK_NO_ML", "FK_ML_PAIRS", "FK_ML_VAR"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for with_out in (True, False):
            o = outs if with_out else [None] * 4

            def run():
                x.copy_(x0)
                P.copy_(P0)
                E.kf_batch_filter(desc, *d, z, x, P, means=o[0], covs=o[1], means_p=o[2], covs_p=o[3], status=st)
            ms = timeit(run, warm=2, reps=5)
            print(json.dumps(dict(env=env, outputs=with_out, N=N, ms=ms, track_steps_per_s=N * T / ms * 1e3,
                                  frac=N * T * 1464 / (ms * 1e-3) / 8e12 if with_out else None)), flush=True)


def rts():
    import torch
    from filterpy_amd import _engine as E
    from tools.bench_configs import cv3d_model, timeit
    N, T, n = int(os.environ.get("ML_N", 100000)), 100, 9
    F, Q, H, R = cv3d_model()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    Xs = torch.randn((T, n, N), generator=g, device=dev, dtype=torch.float64)
    A = torch.randn((n, n), generator=g, device=dev, dtype=torch.float64)
    P1 = (A @ A.T / n + torch.eye(n, device=dev, dtype=torch.float64)).reshape(1, n * n, 1)
    Ps = P1.repeat(T, 1, N).contiguous()
    o = [E.alloc_records((T,), N, n, "soa")] + [E.alloc_records((T,), N, n * n, "soa") for _ in range(3)]
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    desc = dict(n=n, m=1, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS["soa"], update_first=0, alpha_sq=1.0)
    dF, dQ = E.dev(F), E.dev(Q)
    for env in ({}, {"FK_ML_PAIRS": "0"}, {"FK_NO_ML": "1"}):
        for k in ("FK_ML_WAVES", "FK_NO_ML", "FK_ML_VAR", "FK_ML_PAIRS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = timeit(lambda: E.kf_rts(desc, dF, dQ, Xs, Ps, o[0], o[1], o[2], o[3], convention=0, status=st), warm=2, reps=5)
        print(json.dumps(dict(kernel="rts n=9", env=env, N=N, ms=ms, track_steps_per_s=N * T / ms * 1e3,
                              frac=N * T * 2736 / (ms * 1e-3) / 8e12)), flush=True)


def variants():
    """batch_filter's other arguments at (9, 3): per-step model lists, control input, update_first -- the VAR family
    of kf_ml.hip against the one-lane kernels the same calls ran on before (FK_ML_VAR=0)."""
    import numpy as np
    import torch
    from filterpy_amd import _engine as E
    from tools.bench_configs import cv3d_model, timeit
    N, T, n, m, nu = int(os.environ.get("ML_N", 100000)), 100, 9, 3, 2
    F, Q, H, R = cv3d_model()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    rs = np.random.RandomState(5)
    Bm = 0.01 * rs.randn(n, nu)
    for layout in ("soa", "aos"):
        zshape = (T, m, N) if layout == "soa" else (T, N, m)
        ushape = (T, nu, N) if layout == "soa" else (T, N, nu)
        z = torch.randn(zshape, generator=g, device=dev, dtype=torch.float64)
        u = torch.randn(ushape, generator=g, device=dev, dtype=torch.float64)
        x0 = E.to_records(np.zeros((N, n)), layout, 0)
        P0 = E.to_records(np.tile(10.0 * np.eye(n), (N, 1, 1)), layout, 0)
        x, P = x0.clone(), P0.clone()
        outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
                E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
        st = torch.zeros(N, dtype=torch.int32, device=dev)
        shared = [E.dev(a) for a in (F, Q, H, R)]
        stepped = [E.dev(np.tile(a, (T, 1, 1))) for a in (F, Q, H, R)]
        for name, per_step, ctrl, uf in (("per_step", 1, 0, 0), ("control", 0, 1, 0), ("update_first", 0, 0, 1),
                                         ("per_step+control+update_first", 1, 1, 1)):
            desc = dict(n=n, m=m, nu=nu if ctrl else 0, model_mode=3 if per_step else 0, N=N, T=T, layout=E.LAYOUTS[layout],
                        update_first=uf, alpha_sq=1.0)
            d = stepped if per_step else shared
            B = None if not ctrl else E.dev(np.tile(Bm, (T, 1, 1)) if per_step else Bm)
            for env in ({}, {"FK_ML_VAR": "0"}):
                os.environ.pop("FK_ML_VAR", None)
                os.environ.update(env)

                def run():
                    x.copy_(x0)
                    P.copy_(P0)
                    E.kf_batch_filter(desc, *d, z, x, P, B=B, u=u if ctrl else None, means=outs[0], covs=outs[1],
                                      means_p=outs[2], covs_p=outs[3], status=st)
                ms = timeit(run, warm=2, reps=5)
                assert not st.any()
                by = 1464 + (8 * nu if ctrl else 0)
                print(json.dumps(dict(call=name, layout=layout, kernel="one lane per track" if env else "kf_ml VAR", N=N, ms=ms,
                                      track_steps_per_s=N * T / ms * 1e3, frac=N * T * by / (ms * 1e-3) / 8e12)), flush=True)
    os.environ.pop("FK_ML_VAR", None)


def chunks():
    """tail filling (launch_kf_ml_chunked): sweep FK_ML_CHUNKS = "G,H" at (9,3), SOA and AOS, four outputs"""
    import numpy as np
    import torch
    from filterpy_amd import _engine as E
    from tools.bench_configs import cv3d_model, timeit
    N, T, n, m = int(os.environ.get("ML_N", 100000)), 100, 9, 3
    F, Q, H, R = cv3d_model()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for layout in ("soa", "aos"):
        z = torch.randn((T, m, N) if layout == "soa" else (T, N, m), generator=g, device=dev, dtype=torch.float64)
        x0 = E.to_records(np.zeros((N, n)), layout, 0)
        P0 = E.to_records(np.tile(10.0 * np.eye(n), (N, 1, 1)), layout, 0)
        x, P = x0.clone(), P0.clone()
        outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
                E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
        st = torch.zeros(N, dtype=torch.int32, device=dev)
        d = [E.dev(a) for a in (F, Q, H, R)]
        desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
        for spec in os.environ.get("ML_SPECS", "1,1;2,2;2,4;2,5;2,10;3,5;4,4;4,10;default").split(";"):
            os.environ.pop("FK_ML_CHUNKS", None)
            if spec != "default":
                os.environ["FK_ML_CHUNKS"] = spec

            def run():
                x.copy_(x0)
                P.copy_(P0)
                E.kf_batch_filter(desc, *d, z, x, P, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
            ms = timeit(run, warm=2, reps=5)
            print(json.dumps(dict(chunks=spec, layout=layout, N=N, ms=ms, frac=N * T * 1464 / (ms * 1e-3) / 8e12)), flush=True)
    os.environ.pop("FK_ML_CHUNKS", None)


if __name__ == "__main__":
    if os.environ.get("ML_WHAT", "kf") == "chunks":
        chunks()
    elif os.environ.get("ML_WHAT", "kf") == "var":
        variants()
    elif os.environ.get("ML_WHAT", "kf") == "rts":
        rts()
    else:
        main()
