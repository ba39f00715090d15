"""EXPERIMENTAL (written at the end of round 1 without GPU time left; never run on a GPU yet).
Builds the fused linear UKF kernel with -DFK_UKF_V2 (filterpy_amd/csrc/ukf_kernels.hip: same arithmetic,
reorganised to fit two waves per SIMD) into build/libfk_exp_ukf.so and compares it with the shipped kernel on the
same inputs: max relative difference of means / covariances (both must also meet the oracle), then times both.

    python tools/exp_ukf2.py --build                 # here (hipcc cross-compiles)
    python tools/exp_ukf2.py --run                   # on the GPU box
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "filterpy_amd", "csrc")
LIB = os.path.join(CSRC, "exp_build", "libfk_exp_ukf.so")


def build():
    os.makedirs(os.path.join(CSRC, "exp_build"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DFK_UKF_V2", "-o", LIB, os.path.join(CSRC, "ukf_kernels.hip"), "-x", "hip",
                           os.path.join(CSRC, "fk_host.cpp")], cwd=CSRC)
    print("built", LIB)


def run(N, T, iters):
    import torch
    from filterpy_amd import _abi, _engine as E
    from oracle import ukf_oracle
    exp = ctypes.CDLL(LIB)
    res, args = _abi.SIGNATURES["fk_ukf_linear_batch_f64"]
    exp.fk_ukf_linear_batch_f64.restype, exp.fk_ukf_linear_batch_f64.argtypes = res, args
    dev = torch.device("cuda")
    ok = True
    for (n, m) in ((6, 3), (4, 2), (2, 2), (5, 2)):
        k = 2 * n + 1
        alpha, beta, kappa = .1, 2., 3. - n
        lam = alpha ** 2 * (n + kappa) - n
        Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
        r = np.random.default_rng(n)
        F = np.eye(n) + 0.05 * np.triu(r.standard_normal((n, n)), 1)
        H = np.eye(m, n)
        Q, R = 0.01 * np.eye(n), 0.5 * np.eye(m)
        for layout in ("soa", "aos"):
            g = torch.Generator(device=dev)
            g.manual_seed(4)
            z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
            x0 = torch.randn((N, n) if layout == "aos" else (n, N), generator=g, device=dev, dtype=torch.float64)
            P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
            P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
            dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]
            out = {}
            for name in ("shipped", "v2"):
                x, P = x0.clone(), P0.clone()
                means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
                st = torch.zeros(N, dtype=torch.int32, device=dev)

                def go():
                    x.copy_(x0)
                    P.copy_(P0)
                    if name == "shipped":
                        E.ukf_linear_batch(n, m, N, T, layout, lam + n, *dd, z, x, P, means=means, covs=covs, status=st)
                    else:
                        d = _abi.fk_ukf_desc(n=n, m=m, N=N, T=T, layout=E.LAYOUTS[layout], reserved=0, scale=float(lam + n))
                        p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
                        rc = exp.fk_ukf_linear_batch_f64(d, *[p(t) for t in dd], p(z), p(None), p(x), p(P), p(means),
                                                         p(covs), p(st), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                        assert rc == 0, rc
                go()
                torch.cuda.synchronize()
                assert not st.any()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(iters):
                    go()
                t1.record()
                torch.cuda.synchronize()
                out[name] = dict(ms=t0.elapsed_time(t1) / iters, mu=E.from_records(means, layout, 1, (n,)),
                                 cov=E.from_records(covs, layout, 1, (n, n)))
            rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))  # noqa: E731
            dmu, dcov = rel(out["v2"]["mu"], out["shipped"]["mu"]), rel(out["v2"]["cov"], out["shipped"]["cov"])
            trk = 7
            zs_h = (z[:, trk] if layout == "aos" else z[:, :, trk]).cpu().numpy()
            x0h = (x0[trk] if layout == "aos" else x0[:, trk]).cpu().numpy()
            mu_ref, cov_ref = ukf_oracle.ukf_batch_filter(x0h, 10 * np.eye(n), list(zs_h), lambda s, d: F @ s,
                                                          lambda s: H @ s, 0.1, Q, R, alpha, beta, kappa)
            par = max(rel(out["v2"]["mu"][:, trk], mu_ref), rel(out["v2"]["cov"][:, trk], cov_ref))
            good = dmu < 1e-12 and dcov < 1e-12 and par < 1e-9
            ok &= good
            print(json.dumps({"dims": [n, m], "layout": layout, "N": N, "T": T, "v2_vs_shipped_mu": dmu,
                              "v2_vs_shipped_cov": dcov, "v2_vs_oracle": par, "ok": good,
                              "shipped_ms": round(out["shipped"]["ms"], 4), "v2_ms": round(out["v2"]["ms"], 4)}))
    print("V2 AGREES" if ok else "V2 DIFFERS")
    return 0 if ok else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--tracks", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    if a.build:
        build()
    rc = run(a.tracks, a.steps, a.iters) if a.run else 0
    if not (a.build or a.run):
        ap.print_help()
        rc = 2
    sys.exit(rc)
