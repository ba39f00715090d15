#!/usr/bin/env python3
"""Second probe of the four-lane UKF kernels: every dim_x 7..16, both layouts, several dim_z, ragged bank sizes, a mask, every
track of the bank against the oracle's history of its source filter -- until the time budget (argv[1] seconds) is spent; then
first timings at (12,3) and (16,4).  FK_UKF_MLG=1 (and FK_UKF_MLG_MIN_NX=7 for the small dims) must be set."""
import os
import sys
import time

t0 = time.time()
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from filterpy_amd import _engine as E   # noqa: E402
from oracle import ukf_oracle          # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def spd(rs, n, s, batch=()):
    A = rs.randn(*batch, n, n)
    return s * (A @ np.swapaxes(A, -1, -2) / n + 0.5 * np.eye(n))


def one(n, m, layout, N, T=4, nb=3, masked=False):
    rs = np.random.RandomState(100 * n + m + N)
    alpha, beta, kappa = .5, 2., 3. - n
    F = np.eye(n) + 0.1 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), spd(rs, n, 0.05), spd(rs, m, 0.5)
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    sc = alpha ** 2 * (n + kappa)
    # nb distinct tracks dealt out over the bank: every track of the bank is checked against its source's oracle run
    bx, bP, bz = rs.randn(nb, n), spd(rs, n, 2.0, (nb,)), rs.randn(T, nb, m)
    bm = np.ones((T, nb), dtype=np.uint8)
    if masked:
        bm[1, 0] = 0
        bm[2, nb - 1] = 0
    pick = rs.randint(0, nb, size=N)
    x0, P0, zs, mk = bx[pick], bP[pick], bz[:, pick], np.ascontiguousarray(bm[:, pick])   # (bm[:, pick] alone is column-major)
    refs = []
    for b in range(nb):
        zl = [bz[t, b] if bm[t, b] else None for t in range(T)]
        mu, cov = ukf_oracle.ukf_batch_filter(bx[b], bP[b], zl, lambda x, dt: F @ x, lambda x: H @ x, 1.0, Q, R, alpha, beta, kappa)
        refs.append((mu, cov) + tuple(ukf_oracle.ukf_rts_smoother(mu, cov, lambda x, dt: F @ x, 1.0, Q, alpha, beta, kappa)))
    dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.full((N,), -1, dtype=torch.int32, device=dx.device)
    dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]
    zz = np.where(mk[..., None] != 0, zs, np.nan)
    E.ukf_linear_batch(n, m, N, T, layout, sc, *dd, E.to_records(zz, layout, 1), dx, dP,
                       mask=torch.as_tensor(mk, device=dx.device) if masked else None, means=means, covs=covs, status=st, paired=True)
    mu, cov = E.from_records(means, layout, 1, (n,)), E.from_records(covs, layout, 1, (n, n))
    s1 = int(st.cpu().numpy().any())
    rmu = np.stack([refs[p][0] for p in pick], 1)
    rcov = np.stack([refs[p][1] for p in pick], 1)
    ef = max(max(rel(mu[:, i], rmu[:, i]), rel(cov[:, i], rcov[:, i])) for i in range(N))
    fin = max(rel(E.from_records(dx, layout, 0, (n,)), mu[-1]), rel(E.from_records(dP, layout, 0, (n, n)), cov[-1]))
    # the smoother on the ORACLE's filter output
    dX, dPs = E.to_records(rmu, layout, 1), E.to_records(rcov, layout, 1)
    xs, ps, Ks = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout), E.alloc_records((T,), N, n * n, layout)
    st.fill_(-1)
    E.ukf_linear_rts(n, N, T, layout, sc, dd[0], dd[2], dd[4], dd[5], dX, dPs, xs, ps, Ks, st, paired=True)
    s2 = int(st.cpu().numpy().any())
    gx, gp, gk = E.from_records(xs, layout, 1, (n,)), E.from_records(ps, layout, 1, (n, n)), E.from_records(Ks, layout, 1, (n, n))
    es = 0.0
    for i in range(N):
        r = refs[pick[i]]
        es = max(es, rel(gx[:, i], r[2]), rel(gp[:, i], r[3]), rel(gk[:-1, i], r[4][:-1]))
    top = float(np.abs(gk[-1]).max())
    return ef, fin, es, s1 | s2, top


small = os.environ.get("FK_UKF_MLG_MIN_NX") == "7"
cases = []
for n in (range(7, 17) if small else range(10, 17)):
    for layout in ("soa", "aos"):
        for (m, N, masked) in (((n % 4) + 1, 67, False), (4, 150, True)):
            cases.append((n, min(m, 4) if n < 10 else m, layout, N, masked))
for n, m in ((16, 8), (13, 5), (15, 7), (11, 6)):
    for layout in ("soa", "aos"):
        cases.append((n, m, layout, 33, True))
for N in (1, 2, 15, 16, 17, 1000):
    cases.append((14, 3, "soa", N, False))
    cases.append((14, 3, "aos", N, False))
bad = done = 0
worst = 0.0
for c in cases:
    if time.time() - t0 > BUDGET:
        break
    try:
        ef, fin, es, s, top = one(*c[:4], masked=c[4])
        done += 1
        worst = max(worst, ef, es)
        if not (ef < 1e-10 and es < 1e-10 and fin == 0.0 and s == 0 and top == 0.0):
            bad += 1
            print("BAD", c, "filter %.1e final %.1e smoother %.1e status %d K[T-1] %.1e" % (ef, fin, es, s, top), flush=True)
    except Exception as exc:      # noqa: BLE001
        bad += 1
        print("EXC", c, repr(exc)[:200], flush=True)
print("cases run %d of %d, bad %d, worst error %.1e, %.1fs" % (done, len(cases), bad, worst, time.time() - t0), flush=True)

# first timings (events around the call; 20000 tracks x 20 steps)
def timing(n, m, layout, N=20000, T=20):
    rs = np.random.RandomState(1)
    alpha, beta, kappa = .5, 2., 3. - n
    F = np.eye(n) + 0.05 * rs.randn(n, n)
    F /= max(1.0, 1.05 * np.max(np.abs(np.linalg.eigvals(F))))
    H, Q, R = rs.randn(m, n), 0.01 * np.eye(n), 0.5 * np.eye(m)
    Wm, Wc = ukf_oracle.merwe_weights(n, alpha, beta, kappa)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    z = torch.randn((T, N, m) if layout == "aos" else (T, m, N), generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn((N, n) if layout == "aos" else (n, N), generator=g, device=dev, dtype=torch.float64)
    P0 = (5.0 * torch.eye(n, dtype=torch.float64, device=dev)).reshape(1, n * n).repeat(N, 1)
    P0 = P0.contiguous() if layout == "aos" else P0.T.contiguous()
    means, covs = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)
    xs, ps, Ks = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout), E.alloc_records((T,), N, n * n, layout)
    st = torch.zeros(N, dtype=torch.int32, device=dev)
    dd = [E.dev(M) for M in (F, H, Q, R, Wm, Wc)]
    x, P = x0.clone(), P0.clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    res = []
    for rep in range(2):
        x.copy_(x0)
        P.copy_(P0)
        ev[0].record()
        E.ukf_linear_batch(n, m, N, T, layout, alpha ** 2 * (n + kappa), *dd, z, x, P, means=means, covs=covs, status=st, paired=True)
        ev[1].record()
        E.ukf_linear_rts(n, N, T, layout, alpha ** 2 * (n + kappa), dd[0], dd[2], dd[4], dd[5], means, covs, xs, ps, Ks, st, paired=True)
        ev[2].record()
        torch.cuda.synchronize()
        res = [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])]
    bf, bs = 8 * (m + n + n * n), 8 * (2 * n + 3 * n * n)
    print("timing (%d,%d) %s N=%d T=%d: filter %.3f ms = %.3f of 8 TB/s, smoother %.3f ms = %.3f; status %d" % (
        n, m, layout, N, T, res[0], N * T * bf / (res[0] * 1e-3) / 8e12, res[1], N * T * bs / (res[1] * 1e-3) / 8e12, int(st.any())), flush=True)


for c in ((12, 3, "soa"), (16, 4, "soa"), (16, 4, "aos")):
    if time.time() - t0 > BUDGET + 4:
        break
    try:
        timing(*c)
    except Exception as exc:      # noqa: BLE001
        print("EXC timing", c, repr(exc)[:200], flush=True)
print("total %.1fs" % (time.time() - t0), flush=True)
