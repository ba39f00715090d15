"""Helpers for the -m gpu parity tests: host arrays -> device records -> C ABI -> host arrays."""
import numpy as np
import torch

from filterpy_amd import _engine as E
from filterpy_amd._abi import (FK_MODEL_SHARED, FK_MODEL_PER_TRACK, FK_MODEL_PER_TRACK_STEP, FK_MODEL_PER_STEP)

LAYOUTS = ("soa", "aos")


def model_to_dev(M, mode, layout):
    """shared (a,b) | per-track (N,a,b) | per-track-step (T,N,a,b) | per-step (T,a,b)"""
    if M is None:
        return None
    if mode in (FK_MODEL_SHARED, FK_MODEL_PER_STEP):
        return E.dev(M)
    lead = 0 if mode == FK_MODEL_PER_TRACK else 1
    return E.to_records(M, layout, lead)


def run_kf_batch(x0, P0, zs, F, Q, H, R, *, layout="soa", mode=FK_MODEL_SHARED, mask=None, B=None, us=None,
                 alpha_sq=1.0, update_first=False, outputs=True, check_status=True, interleave=False):
    """x0 (N,n), P0 (N,n,n), zs (T,N,m) host arrays; returns host arrays
    (means (T,N,n), covs (T,N,n,n), means_p, covs_p, x_final (N,n), P_final (N,n,n), status (N,))."""
    T, N, m = zs.shape
    n = x0.shape[1]
    nu = 0 if us is None else us.shape[2]
    dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
    dz = E.to_records(zs, layout, 1)
    dF, dQ, dH, dR = (model_to_dev(M, mode, layout) for M in (F, Q, H, R))
    dB = model_to_dev(B, mode, layout)
    du = None if us is None else E.to_records(us, layout, 1)
    dmask = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dx.device)
    st = torch.zeros(N, dtype=torch.int32, device=dx.device)
    outs = [None] * 4
    if outputs:
        outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
                E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
        if interleave:          # both covariance histories in one array (FK_KF_FLAG_COV_INTERLEAVED)
            _, outs[1], outs[3] = E.alloc_cov_pair(T, N, n, layout)
        for o in outs:
            o.fill_(float("nan"))
    E.kf_batch_filter(dict(n=n, m=m, nu=nu, model_mode=mode, N=N, T=T, layout=E.LAYOUTS[layout],
                           update_first=int(update_first), alpha_sq=alpha_sq, flags=2 if interleave else 0),
                      dF, dQ, dH, dR, dz, dx, dP, B=dB, u=du, mask=dmask,
                      means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
    torch.cuda.synchronize()
    res = []
    if outputs:
        res = [E.from_records(outs[0], layout, 1, (n,)), E.from_records(outs[1], layout, 1, (n, n)),
               E.from_records(outs[2], layout, 1, (n,)), E.from_records(outs[3], layout, 1, (n, n))]
    else:
        res = [None] * 4
    res += [E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n)), st.cpu().numpy()]
    if check_status:
        assert not res[-1].any(), res[-1][res[-1] != 0][:8]
    return res


def run_rts(Xs, Ps, F, Q, *, layout="soa", mode=FK_MODEL_SHARED, convention=0):
    """Xs (T,N,n), Ps (T,N,n,n) -> xs, Ps_out, K, Pp host arrays."""
    T, N, n = Xs.shape
    dX, dPs = E.to_records(Xs, layout, 1), E.to_records(Ps, layout, 1)
    dF, dQ = model_to_dev(F, mode, layout), model_to_dev(Q, mode, layout)
    o = [E.alloc_records((T,), N, n, layout)] + [E.alloc_records((T,), N, n * n, layout) for _ in range(3)]
    for t in o:
        t.fill_(float("nan"))
    st = torch.zeros(N, dtype=torch.int32, device=dX.device)
    E.kf_rts(dict(n=n, m=1, nu=0, model_mode=mode, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0,
                  alpha_sq=1.0), dF, dQ, dX, dPs, o[0], o[1], o[2], o[3], convention=convention, status=st)
    torch.cuda.synchronize()
    assert not st.any()
    return (E.from_records(o[0], layout, 1, (n,)), E.from_records(o[1], layout, 1, (n, n)),
            E.from_records(o[2], layout, 1, (n, n)), E.from_records(o[3], layout, 1, (n, n)))


def tile_tracks(a, N, axis=0):
    """replicate a single-track array along a new track axis"""
    return np.repeat(np.expand_dims(a, axis), N, axis=axis)
