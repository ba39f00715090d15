"""Round-2 GPU debugging: (1) which (step, track, element) of the C3 SOA forward covariances differ from track 0's
(P is data-independent: every track must hold the same sequence); (2) where the alpha = 1e-3 UKF predict picks up 1e-10."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from filterpy_amd import _engine as E  # noqa: E402


def c3(N, T, layout, env=None):
    from test_gpu_baseline_configs import ca3d_model
    for k in ("FK_ML_PAIRS", "FK_NO_ML"):
        os.environ.pop(k, None)
    os.environ.update(env or {})
    n, m = 9, 3
    F, Q, H, R = ca3d_model()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    dx = torch.zeros((N, n) if layout == "aos" else (n, N), dtype=torch.float64, device=dev)
    P0 = (10.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    dP = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
            E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
    for o in outs:
        o.fill_(float("nan"))
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    desc = dict(n=n, m=m, nu=0, model_mode=0, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
    E.kf_batch_filter(desc, E.dev(F), E.dev(Q), E.dev(H), E.dev(R), z, dx, dP, means=outs[0], covs=outs[1],
                      means_p=outs[2], covs_p=outs[3], status=st)
    torch.cuda.synchronize()
    for name, o in (("covs", outs[1]), ("covs_p", outs[3])):
        c = o if layout == "soa" else o.transpose(1, 2)          # [T][E][N]
        ref = c[:, :, :1]
        d = ((c - ref).abs() / ref.abs().amax(dim=1, keepdim=True)).amax(dim=1)      # [T][N]
        bad = (d > 1e-12) | ~torch.isfinite(d)
        nb = int(bad.sum())
        print(f"C3 {layout} N={N} T={T} env={env}: {name}: {nb} bad (step, track) pairs, worst {float(torch.nan_to_num(d, nan=9.0).max()):.3e}")
        if nb:
            idx = bad.nonzero()
            ts, trk = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
            print("   steps:", np.unique(ts)[:20], "... tracks:", np.unique(trk)[:24], "n_tracks", len(np.unique(trk)),
                  "track%256:", np.unique(trk % 256)[:20], "track%64:", np.unique(trk % 64)[:20])
            t0, k0 = int(ts[0]), int(trk[0])
            e = (c[t0, :, k0] - c[t0, :, 0]).abs()
            print("   first bad:", t0, k0, "elements", e.nonzero().flatten().cpu().numpy()[:20])
            # is it another step's value?
            for dt in (-1, 1):
                if 0 <= t0 + dt < T:
                    print("   equals step", t0 + dt, "of track 0:", bool(torch.equal(c[t0, :, k0], c[t0 + dt, :, 0])))
            # which dword is wrong, and where does the wrong one come from?
            cb = c.cpu().numpy() if c.numel() < 3e8 else None
            import struct
            for (tt, kk) in list(zip(ts[:4], trk[:4])):
                ee = int((c[tt, :, kk] != c[tt, :, 0]).nonzero().flatten()[0])
                bad_v, good_v = float(c[tt, ee, kk]), float(c[tt, ee, 0])
                bb, gg = struct.unpack("<Q", struct.pack("<d", bad_v))[0], struct.unpack("<Q", struct.pack("<d", good_v))[0]
                print(f"   (t={tt}, trk={kk}, e={ee}): bad {bb:016x} good {gg:016x}  lo differs {bb & 0xffffffff != gg & 0xffffffff} hi differs {bb >> 32 != gg >> 32}")
                cand = {}
                for nm, arr in (("covs", outs[1]), ("covs_p", outs[3])):
                    for d2 in (-1, 0, 1):
                        if 0 <= tt + d2 < T:
                            vals = (arr if layout == "soa" else arr.transpose(1, 2))[tt + d2, :, 0].cpu().numpy()
                            for e2, v2 in enumerate(vals):
                                q = struct.unpack("<Q", struct.pack("<d", float(v2)))[0]
                                if (q & 0xffffffff) == (bb & 0xffffffff):
                                    cand[(nm, d2, e2)] = f"{q:016x}"
                print("      same low dword found in (array, step offset, element):", cand)
            other = outs[3] if name == "covs" else outs[1]
            oc = other if layout == "soa" else other.transpose(1, 2)
            print("   equals the other output's value at the step:", bool(torch.equal(c[t0, :, k0], oc[t0, :, 0])),
                  "next step's:", bool(t0 + 1 < T and torch.equal(c[t0, :, k0], oc[t0 + 1, :, 0])))


def ukf_c2():
    from conftest import golden
    from oracle import ukf_oracle
    g = golden("ukf_merwe")
    ci = 2
    p = f"c{ci}_"
    n, m, alpha, beta, kappa = int(g["cases"][ci][0]), int(g["cases"][ci][1]), *[float(v) for v in g["cases"][ci][2:5]]
    F, Q, x0, P0 = g[p + "F"], g[p + "Q"], g[p + "x0"], g[p + "P0"]
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    sig_ref = ukf_oracle.merwe_sigma_points(x0, P0, alpha, kappa)
    from filterpy_amd.kalman import MerweScaledSigmaPoints, unscented_transform
    pts = MerweScaledSigmaPoints(n, alpha, beta, kappa)
    sig = pts.sigma_points(x0, P0)
    print("UKF c2: sigma points GPU vs oracle: max abs diff", float(np.max(np.abs(sig - sig_ref))), "n differing", int((sig != sig_ref).sum()))
    sf = np.array([F @ s for s in sig_ref])
    xr, Pr = ukf_oracle.unscented_transform(sf, Wm, Wc, Q)
    for lay in ("soa", "aos"):
        x, P = unscented_transform(sf, Wm, Wc, Q, layout=lay)
        print(f"   UT GPU ({lay}) on the oracle's sf: x rel {np.max(np.abs(x - xr)) / np.max(np.abs(xr)):.3e}  P rel {np.max(np.abs(P - Pr)) / np.max(np.abs(Pr)):.3e}")
    x, P = unscented_transform(np.array([F @ s for s in sig]), Wm, Wc, Q)
    print(f"   UT GPU on the GPU's sf: x rel {np.max(np.abs(x - xr)) / np.max(np.abs(xr)):.3e}  P rel {np.max(np.abs(P - Pr)) / np.max(np.abs(Pr)):.3e}")
    print("   golden s1_Pp vs oracle:", float(np.max(np.abs(Pr - g[p + 's1_Pp']))), " Wm equal:", np.array_equal(pts.Wm, Wm), np.array_equal(pts.Wc, Wc))


if __name__ == "__main__":
    c3(100_000, 30, "soa")
    c3(100_000, 100, "aos")
    c3(400_000, 20, "soa")

