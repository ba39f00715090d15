"""Linear Kalman filter with filterpy's call surface, computed by the gfx950 kernels.

Mirrors rlabbe/filterpy v1.4.5 filterpy/kalman/kalman_filter.py for the hot path only:

    KalmanFilter.__init__ (:387-434)  predict (:437-482)  update (:485-561)
    batch_filter (:826-993)  rts_smoother (:995-1074; uses Fs[k+1], Qs[k+1])
    module-level predict (:1571-1621) update (:1401-1508)
    batch_filter (:1664-1788) rts_smoother (:1792-1858; uses Fs[k], Qs[k])

`KalmanFilter` is ONE filter, exactly like the reference (the GPU runs a bank of one).
`KalmanFilterBank` is the same interface for N independent filters stepped in
lock-step -- the shape the engine is built for (x (N,n), P (N,n,n), zs (T,N,m)).

There is no CPU path: every predict/update/batch_filter/rts_smoother call goes through
libfilterhip.so (filterpy_amd/_abi.py) and raises if no GPU / library is available.
S^-1 is applied by an in-lane LDL' (Cholesky) solve, so S must be symmetric positive
definite (the reference's numpy.linalg.inv also accepts indefinite S); a failed
factorisation raises numpy.linalg.LinAlgError like a singular S does in the reference.
"""
import os
import sys
from copy import deepcopy
from math import exp, log, sqrt

import numpy as np

from ..common.helpers import reshape_z, logpdf
from .. import _engine as E
from .. import _abi
from .._abi import FK_MODEL_SHARED, FK_MODEL_PER_TRACK, FK_MODEL_PER_STEP, FK_KF_FLAG_R_JOSEPH_DIAG

__all__ = ["KalmanFilter", "KalmanFilterBank", "predict", "update", "batch_filter", "rts_smoother",
           "predict_steadystate", "update_steadystate"]


# ----------------------------------------------------------------- helpers --
def _mat(M, rows, cols, name, scalar="eye"):
    """Attribute/kwarg -> (rows, cols) float64 matrix with the reference's broadcasting:
    a scalar *attribute* is used raw by numpy (`FPF' + q` adds q to every element,
    kalman_filter.py:478; `dot(F, p)` scales by p), which is what scalar= selects."""
    if np.isscalar(M) or np.ndim(M) == 0:
        v = float(M)
        if scalar == "full":
            return np.full((rows, cols), v)
        return np.eye(rows, cols) * v
    A = np.asarray(M, dtype=np.float64)
    if A.shape != (rows, cols):
        try:
            A = np.broadcast_to(A, (rows, cols))
        except ValueError:
            raise ValueError(f"{name} has shape {A.shape}, expected ({rows}, {cols})") from None
    return np.ascontiguousarray(A)


def _seq(v, n):
    """per-epoch list-like (length n) or None"""
    if v is None:
        return None
    if len(v) != n:
        raise ValueError(f"per-epoch list has length {len(v)}, expected {n}")
    return list(v)


class _Core:
    """Shared device plumbing for one bank of N filters (N = 1 for KalmanFilter)."""

    @staticmethod
    def batch(n, m, N, T, x0, P0, z, mask, F, Q, H, R, mode, B=None, us=None, nu=0,
              alpha_sq=1.0, update_first=False, layout="soa", want_outputs=True, device_outputs=False,
              extras=(), cov_interleave=True, placement=None, placement_out=None):
        """All inputs are host arrays shaped for `mode`:
        x0 (N,n) P0 (N,n,n) z (T,N,m) mask (T,N) or None;
        models: SHARED (a,b) | PER_TRACK (N,a,b) | PER_STEP (T,a,b) | PER_TRACK_STEP (T,N,a,b).
        Returns (means, covs, means_p, covs_p, x_final, P_final) as host arrays (T,N,...)
        or device tensors in `layout` when device_outputs.  placement_out: a dict of the CALLER's that receives how the covariance
        histories were placed (per call: no state shared between banks or threads)."""
        import torch
        E.require_gpu()

        def model(Mx):
            if Mx is None:
                return None
            if mode in (FK_MODEL_SHARED, FK_MODEL_PER_STEP):
                return E.dev(Mx)
            return E.to_records(Mx, layout, 0 if mode == FK_MODEL_PER_TRACK else 1)

        dx, dP = E.to_records(x0, layout, 0), E.to_records(P0, layout, 0)
        dz = z if isinstance(z, torch.Tensor) else E.to_records(z, layout, 1)
        dmask = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dx.device)
        du = None if us is None else E.to_records(us, layout, 1)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        outs = [None] * 4
        desc = dict(n=n, m=m, nu=nu, model_mode=mode, N=N, T=T, layout=E.LAYOUTS[layout],
                    update_first=int(bool(update_first)), alpha_sq=float(alpha_sq))
        # Host outputs of several GiB are STREAMED (round 6): the call is cut into time chunks of ~2 GiB of histories, each
        # launched into the same four device buffers and downloaded through the pinned pipeline straight into its rows of the
        # result arrays (every array of the path is time-major: a chunk is one contiguous block on both sides).  The histories
        # never exist in HBM as a whole -- no 144 GB allocation in front of a (9,3) x 1e6 call (that alone was ~1 s of 3.9), and
        # a host-output call is no longer limited by what fits next to its inputs on the device.  Same kernels, same arithmetic
        # per step: bit-identical to the single launch (a GPU test); the launches are ~1 % of the call.  FK_STREAM_OUTPUTS=0: off.
        # (FK_STREAM_MIN_BYTES / FK_STREAM_CHUNK_BYTES: the thresholds, for the tests)
        step_bytes = 2 * 8 * (n + n * n) * N
        if (want_outputs and not device_outputs and not extras and T >= 2 and N > 0
                and step_bytes * T >= int(os.environ.get("FK_STREAM_MIN_BYTES", 4 << 30))
                and os.environ.get("FK_STREAM_OUTPUTS", "1") != "0"):
            Tc = max(1, min(T, int(os.environ.get("FK_STREAM_CHUNK_BYTES", 2 << 30)) // step_bytes))
            shp = (lambda e: (T, N, e)) if layout == "aos" else (lambda e: (T, e, N))
            host = [np.empty(shp(e)) for e in (n, n * n, n, n * n)]
            # two sets of device buffers: chunk c + 1 is launched into the other set while chunk c is on its way out, and its slabs
            # queue right behind chunk c's (the pipeline never drains between chunks); a set is reused once its download is done
            bufs = [[E.alloc_records((Tc,), N, e, layout) for e in (n, n * n, n, n * n)] for _ in range(2)]
            pending = [[], []]
            per_t = mode not in (FK_MODEL_SHARED, FK_MODEL_PER_TRACK)
            mods = [model(Mx) for Mx in (F, Q, H, R, B)]
            st_all = torch.zeros_like(st)
            for c, t0 in enumerate(range(0, T, Tc)):
                t1 = min(T, t0 + Tc)
                cut = lambda v: None if v is None else v[t0:t1]              # noqa: E731
                mm = [cut(v) if per_t else v for v in mods]
                for f in pending[c & 1]:
                    f.result()
                o = [b[:t1 - t0] for b in bufs[c & 1]]
                E.kf_batch_filter(dict(desc, T=t1 - t0), mm[0], mm[1], mm[2], mm[3], dz[t0:t1], dx, dP, B=mm[4], u=cut(du),
                                  mask=cut(dmask), means=o[0], covs=o[1], means_p=o[2], covs_p=o[3], status=st)
                st_all |= st
                pending[c & 1] = E.download_into([(o[k], host[k][t0:t1]) for k in range(4)], wait=False) or []
            for fs in pending:
                for f in fs:
                    f.result()
            if placement_out is not None:
                placement_out.clear()
                placement_out.update({"method": "none", "note": "host outputs streamed in %d time chunks" % ((T + Tc - 1) // Tc)})
            E.raise_on_status(st_all, "batch_filter")
            res = [E.host_records(host[0], layout, 1, (n,)), E.host_records(host[1], layout, 1, (n, n)),
                   E.host_records(host[2], layout, 1, (n,)), E.host_records(host[3], layout, 1, (n, n))]
            return res + [E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n))]
        # Device-resident outputs: the two covariance histories (76 % of the bytes at dim_x = 4) live in ONE array, posterior
        # and prior record of a track side by side (FK_KF_FLAG_COV_INTERLEAVED): one write front instead of two that may
        # interfere (docs/PLACEMENT.md).  The caller gets strided views.  Where the specialised kernel does not serve the
        # call (FK_ERR_UNSUPPORTED) two plain arrays are used.
        # Asked for only where the library takes it (kf_dispatch.cpp, run_kf_window): the one-lane specialised kernel's calls -- dim_x
        # <= 8 (dim_x 9 runs on the three- / four-lane kernels); predict -> update without a control input; any model mode up to
        # dim_x 6, the shared constant model above.  (The except branch below stays as the safety net for the library's A/B
        # switches; it reports what it did.)
        inter = bool(want_outputs and device_outputs and not extras and cov_interleave and n <= 8
                     and m <= min(n, 4) and nu == 0 and not update_first and (mode == FK_MODEL_SHARED or n <= 6)
                     and 2 * N * n * n * 8 < 2 ** 32 and placement != "probe")
        # placement=None (round 5, the default): where the launch is HBM-bound and its two big streams are most of its bytes --
        # the one-lane specialised kernel at dim_x <= 4 -- and the histories are large (256 MiB each and more), place them by
        # measurement ("probe": 5.24-5.41 ms at BASELINE configs[1] on every box of rounds 4 / 5, where the interleaved array
        # ran 5.49-5.87 and two plain arrays 5.33-6.73); the interleaved array everywhere else, and where the probe cannot run
        # (no room for three candidates, the remembered pair of this shape still alive in the caller's hands).
        auto_probe = bool(placement is None and inter and n * n <= 16 and T * N * n * n * 8 >= (256 << 20))
        pinfo = {"method": "interleave" if inter else "none"}
        if want_outputs and device_outputs and not extras and (placement == "probe" or auto_probe) and T * N * n * n * 8 >= (256 << 20):
            # two dense arrays, placed in HBM by measuring this very launch on several candidate buffers (placement.py:
            # placed_pair; the pair is remembered per shape -- a two-entry LRU per device --, the losers go back to torch's
            # caching allocator).  Worth ~7 % over the interleaved array at BASELINE configs[1] (5.2 against 5.6 ms), costs a
            # few to ~100 launches once per shape (the probe stops as soon as a fast pair shows).
            from .. import placement as _pl
            dx0, dP0 = dx.clone(), dP.clone()
            shp = (T, N, n * n) if layout == "aos" else (T, n * n, N)
            as_rec = lambda b: b.view(torch.float64).view(shp)          # noqa: E731
            mu, mup = E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n, layout)

            def run_ms(a, b):
                dx.copy_(dx0)
                dP.copy_(dP0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                E.kf_batch_filter(desc, model(F), model(Q), model(H), model(R), dz, dx, dP, B=model(B), u=du, mask=dmask,
                                  means=mu, covs=as_rec(a), means_p=mup, covs_p=as_rec(b), status=st)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1)
            pa, pb, pinfo = _pl.placed_pair(T * N * n * n * 8, run_ms, dx.device, or_none=auto_probe)
            dx.copy_(dx0)
            dP.copy_(dP0)
            st.zero_()
            if pa is not None:
                outs = [mu, as_rec(pa), mup, as_rec(pb)]
                inter = False
            else:                            # (auto only) the interleaved array after all
                note = pinfo["method"]
                _, cpost, cprior = E.alloc_cov_pair(T, N, n, layout)
                outs = [mu, cpost, mup, cprior]
                pinfo = {"method": "interleave", "note": note}
        elif want_outputs:
            if inter:
                _, cpost, cprior = E.alloc_cov_pair(T, N, n, layout)
                outs = [E.alloc_records((T,), N, n, layout), cpost, E.alloc_records((T,), N, n, layout), cprior]
            else:
                outs = [E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout),
                        E.alloc_records((T,), N, n, layout), E.alloc_records((T,), N, n * n, layout)]
        ex = {}
        if extras:
            # per-step histories of the update's by-products
            shapes = dict(y=(m,), K=(n, m), S=(m, m), SI=(m, m))
            for k in extras:
                if k in shapes:
                    ex[k] = E.alloc_records((T,), N, int(np.prod(shapes[k])), layout).zero_()
                else:
                    ex[k] = torch.zeros((T, N), dtype=torch.float64, device=dx.device)
            E.kf_batch_filter_ex(desc, model(F), model(Q), model(H), model(R), dz, dx, dP, ex, B=model(B), u=du,
                                 mask=dmask, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], status=st)
        else:
            args = (model(F), model(Q), model(H), model(R), dz, dx, dP)
            kw = dict(B=model(B), u=du, mask=dmask, status=st)
            try:
                E.kf_batch_filter(dict(desc, flags=_abi.FK_KF_FLAG_COV_INTERLEAVED) if inter else desc, *args,
                                  means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], **kw)
            except _abi.FilterHipError as exc:
                if not (inter and exc.code == _abi.FK_ERR_UNSUPPORTED):
                    raise
                outs[1], outs[3] = E.alloc_records((T,), N, n * n, layout), E.alloc_records((T,), N, n * n, layout)
                pinfo = {"method": "none", "note": "interleaved histories declined by the library: " + str(exc)[:160]}
                E.kf_batch_filter(desc, *args, means=outs[0], covs=outs[1], means_p=outs[2], covs_p=outs[3], **kw)
        if placement_out is not None:
            placement_out.clear()
            placement_out.update(pinfo)
        E.raise_on_status(st, "batch_filter")
        if device_outputs:
            return outs + [dx, dP] + ([ex] if extras else [])
        res = [None] * 4
        if want_outputs:
            res = [E.from_records(outs[0], layout, 1, (n,)), E.from_records(outs[1], layout, 1, (n, n)),
                   E.from_records(outs[2], layout, 1, (n,)), E.from_records(outs[3], layout, 1, (n, n))]
        res += [E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n))]
        if extras:
            shapes = dict(y=(m,), K=(n, m), S=(m, m), SI=(m, m))
            res.append({k: (E.from_records(v, layout, 1, shapes[k]) if k in shapes else v.cpu().numpy())
                        for k, v in ex.items()})
        return res

    @staticmethod
    def rts(n, N, T, Xs, Ps, F, Q, mode, convention, layout="soa", inv=None):
        """inv: a caller-supplied inverse (rts_smoother(inv=...), kalman_filter.py:995, 1069) -- two launches with the callable
        applied on the host in between (include/filterhip.h: FK_KF_FLAG_PP_ONLY / FK_KF_FLAG_PPINV_GIVEN)."""
        import torch
        E.require_gpu()

        def model(Mx):
            if mode in (FK_MODEL_SHARED, FK_MODEL_PER_STEP):
                return E.dev(Mx)
            return E.to_records(Mx, layout, 0 if mode == FK_MODEL_PER_TRACK else 1)

        dX, dPs = E.to_records(Xs, layout, 1), E.to_records(Ps, layout, 1)
        o = [E.alloc_records((T,), N, n, layout)] + [E.alloc_records((T,), N, n * n, layout) for _ in range(3)]
        st = torch.zeros(N, dtype=torch.int32, device=dX.device)
        desc = dict(n=n, m=1, nu=0, model_mode=mode, N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0)
        if inv is not None:
            dF, dQ = model(F), model(Q)
            E.kf_rts(dict(desc, flags=_abi.FK_KF_FLAG_PP_ONLY), dF, dQ, dX, dPs, o[0], o[1], o[2], o[3],
                     convention=convention, status=st)
            Pp = E.from_records(o[3], layout, 1, (n, n))
            PpI = np.zeros((T, N, n, n))
            for k in range(T - 1):                        # the reference's own call, one matrix at a time (:1069)
                for i in range(N):
                    PpI[k, i] = np.asarray(inv(Pp[k, i]), dtype=np.float64).reshape(n, n)
            o[2] = E.to_records(PpI, layout, 1)          # inverses in, gains out
            E.kf_rts(dict(desc, flags=_abi.FK_KF_FLAG_PPINV_GIVEN), dF, dQ, dX, dPs, o[0], o[1], o[2], o[3],
                     convention=convention, status=st)
        else:
            E.kf_rts(desc, model(F), model(Q), dX, dPs, o[0], o[1], o[2], o[3], convention=convention, status=st)
        E.raise_on_status(st, "rts_smoother")
        return (E.from_records(o[0], layout, 1, (n,)), E.from_records(o[1], layout, 1, (n, n)),
                E.from_records(o[2], layout, 1, (n, n)), E.from_records(o[3], layout, 1, (n, n)))

    @staticmethod
    def predict(n, N, x, P, F, Q, mode, B=None, u=None, nu=0, alpha_sq=1.0, layout="soa"):
        import torch
        E.require_gpu()
        lead = 0

        def model(Mx):
            if Mx is None:
                return None
            return E.dev(Mx) if mode == FK_MODEL_SHARED else E.to_records(Mx, layout, lead)

        dx, dP = E.to_records(x, layout, 0), E.to_records(P, layout, 0)
        du = None if u is None else E.to_records(u, layout, 0)
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.kf_predict(dict(n=n, m=1, nu=nu, model_mode=mode, N=N, T=1, layout=E.LAYOUTS[layout], update_first=0,
                          alpha_sq=float(alpha_sq)), model(F), model(Q), dx, dP, B=model(B), u=du, status=st)
        E.raise_on_status(st, "predict")
        return E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n))

    @staticmethod
    def update(n, m, N, x, P, z, H, R, mode, mask=None, layout="soa", flags=0, inv=None):
        """inv: a caller-supplied inverse (KalmanFilter.inv, kalman_filter.py:363, 434, 541) -- y and S from one launch, the
        callable on the host, the correction from a second launch that takes its result (include/filterhip.h:
        FK_KF_FLAG_S_ONLY / FK_KF_FLAG_SI_GIVEN)."""
        import torch
        E.require_gpu()

        def model(Mx):
            return E.dev(Mx) if mode == FK_MODEL_SHARED else E.to_records(Mx, layout, 0)

        dx, dP, dz = E.to_records(x, layout, 0), E.to_records(P, layout, 0), E.to_records(z, layout, 0)
        dmask = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dx.device)
        y, K = E.alloc_records((), N, m, layout), E.alloc_records((), N, n * m, layout)
        S, SI = E.alloc_records((), N, m * m, layout), E.alloc_records((), N, m * m, layout)
        for t in (y, K, S, SI):
            t.zero_()
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        desc = dict(n=n, m=m, nu=0, model_mode=mode, N=N, T=1, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0, flags=flags)
        if inv is not None:
            dH, dR = model(H), model(R)
            E.kf_update(dict(desc, flags=flags | _abi.FK_KF_FLAG_S_ONLY), dH, dR, dz, dx, dP, mask=dmask, y=y, K=K, S=S, SI=SI,
                        status=st)
            Sh = E.from_records(S, layout, 0, (m, m))
            SIh = np.zeros((N, m, m))
            for i in range(N):
                if mask is None or np.ravel(mask)[i]:
                    SIh[i] = np.asarray(inv(Sh[i]), dtype=np.float64).reshape(m, m)
            SI = E.to_records(SIh, layout, 0)
            E.kf_update(dict(desc, flags=flags | _abi.FK_KF_FLAG_SI_GIVEN), dH, dR, dz, dx, dP, mask=dmask, y=y, K=K, S=S, SI=SI,
                        status=st)
        else:
            E.kf_update(desc, model(H), model(R), dz, dx, dP, mask=dmask, y=y, K=K, S=S, SI=SI, status=st)
        E.raise_on_status(st, "update")
        return (E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n)),
                E.from_records(y, layout, 0, (m,)), E.from_records(K, layout, 0, (n, m)),
                E.from_records(S, layout, 0, (m, m)), E.from_records(SI, layout, 0, (m, m)))



    @staticmethod
    def steadystate(n, m, N, T, x, F, H, K, z, mask=None, B=None, us=None, nu=0, k_per_track=False, layout="soa"):
        """fk_kf_steadystate_f64.  F None: update only; z None: predict only.
        Returns x_final (N,n), means (T,N,n) | None, means_p (T,N,n) | None, y (T,N,m) | None."""
        import torch
        E.require_gpu()
        dx = E.to_records(x, layout, 0)
        dz = None if z is None else E.to_records(z, layout, 1)
        dmask = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dx.device)
        dK = None if K is None else (E.to_records(K.reshape(N, n * m), layout, 0) if k_per_track else E.dev(K))
        du = None if us is None else E.to_records(us, layout, 1)
        means = None if z is None else E.alloc_records((T,), N, n, layout)
        means_p = None if F is None else E.alloc_records((T,), N, n, layout)
        y = None if z is None else E.alloc_records((T,), N, m, layout)
        E.kf_steadystate(dict(n=n, m=m, nu=nu, model_mode=FK_MODEL_PER_TRACK if k_per_track else FK_MODEL_SHARED,
                              N=N, T=T, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0),
                         None if F is None else E.dev(F), None if H is None else E.dev(H), dK, z=dz, x=dx,
                         B=None if B is None else E.dev(B), u=du, mask=dmask, means=means, means_p=means_p, y=y)
        back = lambda t, rec: None if t is None else E.from_records(t, layout, 1, rec)  # noqa: E731
        return E.from_records(dx, layout, 0, (n,)), back(means, (n,)), back(means_p, (n,)), back(y, (m,))

    @staticmethod
    def update_correlated(n, m, N, x, P, z, H, R, M, mask=None, m_per_track=False, layout="soa"):
        """fk_kf_update_correlated_f64 -> x, P, y, K, S, SI host arrays."""
        import torch
        E.require_gpu()
        dx, dP, dz = E.to_records(x, layout, 0), E.to_records(P, layout, 0), E.to_records(z, layout, 0)
        dmask = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dx.device)
        dM = E.to_records(M.reshape(N, n * m), layout, 0) if m_per_track else E.dev(M)
        y, K = E.alloc_records((), N, m, layout), E.alloc_records((), N, n * m, layout)
        S, SI = E.alloc_records((), N, m * m, layout), E.alloc_records((), N, m * m, layout)
        for t in (y, K, S, SI):
            t.zero_()
        st = torch.zeros(N, dtype=torch.int32, device=dx.device)
        E.kf_update_correlated(dict(n=n, m=m, nu=0, model_mode=FK_MODEL_PER_TRACK if m_per_track else FK_MODEL_SHARED,
                                    N=N, T=1, layout=E.LAYOUTS[layout], update_first=0, alpha_sq=1.0),
                               E.dev(H), E.dev(R), dM, dz, dx, dP, mask=dmask, y=y, K=K, S=S, SI=SI, status=st)
        E.raise_on_status(st, "update_correlated")
        return (E.from_records(dx, layout, 0, (n,)), E.from_records(dP, layout, 0, (n, n)),
                E.from_records(y, layout, 0, (m,)), E.from_records(K, layout, 0, (n, m)),
                E.from_records(S, layout, 0, (m, m)), E.from_records(SI, layout, 0, (m, m)))


# ------------------------------------------------------------ KalmanFilter --
class KalmanFilter(object):
    """One linear Kalman filter; filterpy.kalman.KalmanFilter's interface
    (kalman_filter.py:387-434 for the attributes), arithmetic on the GPU."""

    def __init__(self, dim_x, dim_z, dim_u=0):
        if dim_x < 1:
            raise ValueError('dim_x must be 1 or greater')
        if dim_z < 1:
            raise ValueError('dim_z must be 1 or greater')
        if dim_u < 0:
            raise ValueError('dim_u must be 0 or greater')
        self.dim_x, self.dim_z, self.dim_u = dim_x, dim_z, dim_u
        self.x = np.zeros((dim_x, 1))
        self.P = np.eye(dim_x)
        self.Q = np.eye(dim_x)
        self.B = None
        self.F = np.eye(dim_x)
        self.H = np.zeros((dim_z, dim_x))
        self.R = np.eye(dim_z)
        self._alpha_sq = 1.
        self.M = np.zeros((dim_x, dim_z))
        self.z = np.array([[None] * dim_z]).T
        self.K = np.zeros((dim_x, dim_z))
        self.y = np.zeros((dim_z, 1))
        self.S = np.zeros((dim_z, dim_z))
        self.SI = np.zeros((dim_z, dim_z))
        self._I = np.eye(dim_x)
        self.x_prior, self.P_prior = self.x.copy(), self.P.copy()
        self.x_post, self.P_post = self.x.copy(), self.P.copy()
        self._log_likelihood = log(sys.float_info.min)
        self._likelihood = sys.float_info.min
        self._mahalanobis = None
        # numpy.linalg.inv (the default): S^-1 is applied by the kernel's in-lane LDL' solve.  Anything else (the reference's
        # documented `kf.inv = np.linalg.pinv`, kalman_filter.py:363) is HONOURED: update() / batch_filter() then run the
        # step around the callable (_custom_inv below)
        self.inv = np.linalg.inv

    def _custom_inv(self):
        """the callable update() must apply to S, or None for numpy.linalg.inv (then the fused kernels' own solve runs)"""
        return None if self.inv is np.linalg.inv else self.inv

    # -- attribute normalisation ------------------------------------------------
    def _xP(self):
        x = np.asarray(self.x, dtype=np.float64)
        if x.size != self.dim_x:
            raise ValueError(f"x has {x.size} elements, expected dim_x = {self.dim_x}")
        return x.reshape(1, self.dim_x), _mat(self.P, self.dim_x, self.dim_x, "P")[None]

    def _set_x(self, xrow):
        self.x = xrow.reshape(np.shape(self.x)) if np.ndim(self.x) > 0 else float(xrow[0])

    def _R_eff(self, R):
        """The R *attribute* as update(z) uses it -> (matrix, desc flags).  A scalar attribute is taken raw by the
        reference (kalman_filter.py:522-523): `S = dot(H, PHT) + R` adds r to every element of S (:540) while
        `dot(dot(K, R), K.T)` is r K K' (:556) -- the kernel gets r * ones and FK_KF_FLAG_R_JOSEPH_DIAG."""
        m = self.dim_z
        if (np.isscalar(R) or np.ndim(R) == 0) and m > 1:
            return np.full((m, m), float(R)), FK_KF_FLAG_R_JOSEPH_DIAG
        return _mat(R, m, m, "R", scalar="full"), 0

    # -- predict ----------------------------------------------------------------
    def predict(self, u=None, B=None, F=None, Q=None):
        """kalman_filter.py:437-482.  x = Fx (+ Bu iff B and u are given); P = a^2 FPF' + Q."""
        n = self.dim_x
        if B is None:
            B = self.B
        F = self.F if F is None else F
        if Q is None:
            Qm = _mat(self.Q, n, n, "Q", scalar="full")       # attribute scalar is added to every element
        elif np.isscalar(Q):
            Qm = np.eye(n) * Q                                # kwarg scalar -> eye * Q  (:467-468)
        else:
            Qm = _mat(Q, n, n, "Q")
        x, P = self._xP()
        use_ctrl = B is not None and u is not None
        kw = {}
        if use_ctrl:
            uu = np.asarray(u, dtype=np.float64).reshape(1, -1)
            kw = dict(B=_mat(B, n, uu.shape[1], "B"), u=uu, nu=uu.shape[1])
        xn, Pn = _Core.predict(n, 1, x, P, _mat(F, n, n, "F"), Qm, FK_MODEL_SHARED,
                               alpha_sq=self._alpha_sq, **kw)
        self._set_x(xn[0])
        self.P = Pn[0]
        self.x_prior, self.P_prior = np.copy(self.x), self.P.copy()

    # -- update -----------------------------------------------------------------
    def update(self, z, R=None, H=None):
        """kalman_filter.py:485-561 (Joseph form)."""
        self._log_likelihood = None
        self._likelihood = None
        self._mahalanobis = None
        n, m = self.dim_x, self.dim_z
        if z is None:
            self.z = np.array([[None] * m]).T
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
            self.y = np.zeros((m, 1))
            return
        rflags = 0
        if R is None:
            Rm, rflags = self._R_eff(self.R)
        elif np.isscalar(R):
            Rm = np.eye(m) * R
        else:
            Rm = _mat(R, m, m, "R")
        x_ndim = np.ndim(self.x)
        if H is None:
            z = reshape_z(z, m, x_ndim)                       # raises ValueError on bad shapes
            H = self.H
            zz = np.asarray(z, dtype=np.float64)
            if zz.size != m:
                raise ValueError(f"z (shape {zz.shape}) does not hold dim_z = {m} values")
        else:
            # with an explicit H the reference skips reshape_z (kalman_filter.py:530-533) and numpy broadcasts
            # z against Hx: a scalar feeds every measurement, a z shaped like Hx is used as it is.  A z that
            # would broadcast the residual beyond Hx's shape turns the reference's STATE into a matrix (or
            # fails in a later dot); here that is a ValueError.
            zraw = np.asarray(z, dtype=np.float64)
            hx_shape = (m,) if x_ndim == 1 else (m, 1)
            try:
                y_shape = np.broadcast_shapes(zraw.shape, hx_shape)
            except ValueError:
                raise ValueError(f"z (shape {zraw.shape}) does not broadcast against Hx (shape {hx_shape})") from None
            if y_shape != hx_shape:
                raise ValueError(f"z (shape {zraw.shape}) would broadcast the residual to {y_shape}, "
                                 f"Hx has shape {hx_shape}")
            zz = np.broadcast_to(zraw, hx_shape)
        x, P = self._xP()
        xn, Pn, y, K, S, SI = _Core.update(n, m, 1, x, P, np.ascontiguousarray(zz).reshape(1, m), _mat(H, m, n, "H"),
                                           Rm, FK_MODEL_SHARED, flags=rflags, inv=self._custom_inv())
        self._set_x(xn[0])
        self.P = Pn[0]
        self.y = y[0].reshape(m, 1) if x_ndim == 2 else y[0]
        self.K, self.S, self.SI = K[0], S[0], SI[0]
        self.z = deepcopy(z)
        self.x_post, self.P_post = np.copy(self.x), self.P.copy()

    # -- get_prediction / get_update: the same kernels on a copy of the state ---------
    def get_prediction(self, u=None, B=None, F=None, Q=None):
        """kalman_filter.py:1076-1117: predict() without modifying the object; returns (x, P)."""
        keep = (np.copy(self.x), np.copy(self.P), np.copy(self.x_prior), np.copy(self.P_prior))
        try:
            self.predict(u=u, B=B, F=F, Q=Q)
            return np.copy(self.x), np.copy(self.P)
        finally:
            self.x, self.P, self.x_prior, self.P_prior = keep

    def get_update(self, z=None):
        """kalman_filter.py:1119-1173: update(z) without altering the filter; returns (x, P)."""
        if z is None:
            return self.x, self.P
        names = ("x", "P", "y", "K", "S", "SI", "z", "x_post", "P_post", "_log_likelihood", "_likelihood", "_mahalanobis")
        keep = {k: deepcopy(getattr(self, k)) for k in names}
        try:
            self.update(z)
            return np.copy(self.x), np.copy(self.P)
        finally:
            for k, v in keep.items():
                setattr(self, k, v)

    # -- steady state, correlated noise, sequential ---------------
    def predict_steadystate(self, u=0, B=None):
        """kalman_filter.py:563-593: x = Fx (+ Bu iff B is set); P is left unchanged."""
        n = self.dim_x
        if B is None:
            B = self.B
        x, _ = self._xP()
        kw = {}
        if B is not None:
            nu = np.shape(np.atleast_2d(B))[1] if np.ndim(B) else 1
            uu = np.broadcast_to(np.asarray(u, dtype=np.float64).reshape(-1), (nu,)).reshape(1, 1, nu).copy()
            kw = dict(B=_mat(B, n, nu, "B"), us=uu, nu=nu)
        xn, _, _, _ = _Core.steadystate(n, self.dim_z, 1, 1, x, _mat(self.F, n, n, "F"), None, None, None, **kw)
        self._set_x(xn[0])
        self.x_prior, self.P_prior = np.copy(self.x), np.copy(self.P)

    def update_steadystate(self, z):
        """kalman_filter.py:595-668: y = z - Hx ; x += K y with the stored gain; P, K, S untouched."""
        self._log_likelihood = None
        self._likelihood = None
        self._mahalanobis = None
        n, m = self.dim_x, self.dim_z
        if z is None:
            self.z = np.array([[None] * m]).T
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
            self.y = np.zeros((m, 1))
            return
        x_ndim = np.ndim(self.x)
        z = reshape_z(z, m, x_ndim)
        x, _ = self._xP()
        xn, _, _, y = _Core.steadystate(n, m, 1, 1, x, None, _mat(self.H, m, n, "H"), _mat(self.K, n, m, "K"),
                                        np.asarray(z, dtype=np.float64).reshape(1, 1, m))
        self._set_x(xn[0])
        self.y = y[0, 0].reshape(m, 1) if x_ndim == 2 else y[0, 0]
        self.z = deepcopy(z)
        self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)

    def update_correlated(self, z, R=None, H=None):
        """kalman_filter.py:670-752: update with process and measurement noise correlated through self.M."""
        if self._custom_inv() is not None:
            raise NotImplementedError("update_correlated applies S^-1 inside the kernel; a non-default `inv` is honoured by "
                                      "update(), batch_filter() and rts_smoother(inv=...) only")
        self._log_likelihood = None
        self._likelihood = None
        self._mahalanobis = None
        n, m = self.dim_x, self.dim_z
        if z is None:
            self.z = np.array([[None] * m]).T
            self.x_post, self.P_post = np.copy(self.x), np.copy(self.P)
            self.y = np.zeros((m, 1))
            return
        if R is None:
            Rm = self._R_eff(self.R)[0]       # R enters S only here (kalman_filter.py:735-745): a raw scalar is r * ones
        elif np.isscalar(R):
            Rm = np.eye(m) * R
        else:
            Rm = _mat(R, m, m, "R")
        x_ndim = np.ndim(self.x)
        if H is None:
            z = reshape_z(z, m, x_ndim)
            H = self.H
        zz = np.asarray(z, dtype=np.float64)
        if zz.size != m:
            raise ValueError(f"z (shape {zz.shape}) does not hold dim_z = {m} values")
        x, P = self._xP()
        xn, Pn, y, K, S, SI = _Core.update_correlated(n, m, 1, x, P, zz.reshape(1, m), _mat(H, m, n, "H"), Rm,
                                                      _mat(self.M, n, m, "M"))
        self._set_x(xn[0])
        self.P = Pn[0]
        self.y = y[0].reshape(m, 1) if x_ndim == 2 else y[0]
        self.K, self.S, self.SI = K[0], S[0], SI[0]
        self.z = deepcopy(z)
        self.x_post, self.P_post = np.copy(self.x), self.P.copy()

    def update_sequential(self, start, z_i, R_i=None, H_i=None):
        """kalman_filter.py:754-824: the ordinary (Joseph-form) update restricted to the measurement
        components start:start+len(z_i) -- the same kernel as update() on the sliced H, R."""
        n = self.dim_x
        length = 1 if np.isscalar(z_i) else len(z_i)
        z_i = np.reshape(np.asarray(z_i, dtype=np.float64), [length, 1])
        stop = start + length
        if R_i is None:
            R_i = _mat(self.R, self.dim_z, self.dim_z, "R")[start:stop, start:stop]
        elif np.isscalar(R_i):
            R_i = np.eye(length) * R_i
        if H_i is None:
            H_i = np.asarray(self.H, dtype=np.float64)[start:stop]
        H_i = np.reshape(np.asarray(H_i, dtype=np.float64), [length, n])
        x, P = self._xP()
        xn, Pn, y, K, S, SI = _Core.update(n, length, 1, x, P, z_i.reshape(1, length), H_i,
                                           _mat(R_i, length, length, "R_i"), FK_MODEL_SHARED)
        if np.shape(self.y) != (self.dim_z, 1):
            self.y = np.zeros((self.dim_z, 1))
        self.y[start:stop] = y[0].reshape(length, 1)
        self.K = np.array(self.K, dtype=np.float64)
        self.K[:, start:stop] = K[0]
        self._set_x(xn[0])
        self.P = Pn[0]
        if np.shape(self.z) != (self.dim_z, 1):
            self.z = np.array([[None] * self.dim_z]).T
        self.z[start:stop] = z_i
        self.x_post, self.P_post = np.copy(self.x), self.P.copy()

    # -- batch_filter -----------------------------------------------------------
    def batch_filter(self, zs, Fs=None, Qs=None, Hs=None, Rs=None, Bs=None, us=None,
                     update_first=False, saver=None):
        """kalman_filter.py:826-993.  Returns (means, covariances, means_p, covariances_p);
        the filter's own x, P end at the final state, like the reference."""
        n, m = self.dim_x, self.dim_z
        T = len(zs)
        if self._custom_inv() is not None:
            return self._batch_filter_epochwise(zs, Fs, Qs, Hs, Rs, Bs, us, update_first, saver)
        Fs, Qs, Hs, Rs, Bs = (_seq(v, T) for v in (Fs, Qs, Hs, Rs, Bs))
        want_hist = saver is not None
        x_ndim = np.ndim(self.x)
        # measurements: None = missing (:515-520).  batch_filter always passes H, so reshape_z is
        # skipped (:527-529): a z only needs to broadcast against H x
        z = np.zeros((T, 1, m))
        mask = np.ones((T, 1), dtype=np.uint8)
        for i, zi in enumerate(zs):
            if zi is None:
                mask[i, 0] = 0
                continue
            zi = np.asarray(zi, dtype=np.float64)
            if zi.size != m:
                raise ValueError(f"zs[{i}] has shape {zi.shape}, expected {m} values")
            if x_ndim == 2 and zi.ndim == 1 and m > 1:
                # the reference silently broadcasts y to (m, m) here and then fails storing x
                raise ValueError("with a column-vector state each z must be a (dim_z, 1) column")
            z[i, 0] = zi.reshape(m)
        per_step = any(v is not None for v in (Fs, Qs, Hs, Rs, Bs))
        # the attributes reach predict() / update() as KWARGS here (Qs = [self.Q] * n, Rs = [self.R] * n,
        # kalman_filter.py:944-947), and a scalar kwarg is eye * value (:467-468, :524-525) -- unlike the raw
        # attribute of a direct predict() / update() call
        Fm, Qm = _mat(self.F, n, n, "F"), _mat(self.Q, n, n, "Q", scalar="eye")
        Hm, Rm = _mat(self.H, m, n, "H"), _mat(self.R, m, m, "R", scalar="eye")
        kw = {}
        if us is not None and (Bs is not None or self.B is not None):
            U = np.asarray([np.ravel(np.asarray(u, dtype=np.float64)) for u in us])
            nu = U.shape[1]
            kw = dict(us=U.reshape(T, 1, nu), nu=nu)
            kw["B"] = (np.stack([_mat(b, n, nu, "B") for b in Bs]) if Bs is not None
                       else (np.broadcast_to(_mat(self.B, n, nu, "B"), (T, n, nu)).copy() if per_step
                             else _mat(self.B, n, nu, "B")))
        if per_step:
            def stack(lst, base, r, c, name, scalar="eye"):
                if lst is None:
                    return np.broadcast_to(base, (T, r, c)).copy()
                return np.stack([_mat(v, r, c, name, scalar) if not np.isscalar(v) else
                                 (np.eye(r, c) * v) for v in lst])
            Fm, Qm = stack(Fs, Fm, n, n, "F"), stack(Qs, Qm, n, n, "Q")
            Hm, Rm = stack(Hs, Hm, m, n, "H"), stack(Rs, Rm, m, m, "R")
            mode = FK_MODEL_PER_STEP
        else:
            mode = FK_MODEL_SHARED
        x, P = self._xP()
        res = _Core.batch(
            n, m, 1, T, x, P, z, None if mask.all() else mask, Fm, Qm, Hm, Rm, mode,
            alpha_sq=self._alpha_sq, update_first=update_first,
            extras=("y", "K", "S", "SI") if want_hist else (), **kw)
        mu, cov, mup, covp, xf, Pf = res[:6]
        self._set_x(xf[0])
        self.P = Pf[0]
        mu, cov, mup, covp = mu[:, 0], cov[:, 0], mup[:, 0], covp[:, 0]
        if T and not want_hist:
            # K, y, S, SI, z and the lazy likelihoods as the reference's per-epoch loop leaves them
            # (kalman_filter.py:511-561): z and y belong to the last epoch, K / S / SI to the last epoch that had a
            # measurement (update(None) leaves them alone) -- one single-step update launch on the state that
            # update started from reproduces them
            present = mask[:, 0].astype(bool)
            self.z = deepcopy(zs[T - 1]) if present[T - 1] else np.array([[None] * m]).T
            yv = None
            meas = np.flatnonzero(present)
            if meas.size:
                i = int(meas[-1])
                if update_first:
                    px, pP = (x, P) if i == 0 else (mup[i - 1][None], covp[i - 1][None])
                else:
                    px, pP = mup[i][None], covp[i][None]
                Hi, Ri = (Hm[i], Rm[i]) if per_step else (Hm, Rm)
                _, _, yv, Ki, Si, SIi = _Core.update(n, m, 1, np.ascontiguousarray(px), np.ascontiguousarray(pP),
                                                     z[i], Hi, Ri, FK_MODEL_SHARED)
                self.K, self.S, self.SI = Ki[0], Si[0], SIi[0]
                yv = yv[0]
            if present[T - 1]:
                self.y = yv.reshape(m, 1).copy() if x_ndim == 2 else yv.copy()
            else:
                self.y = np.zeros((m, 1))
            self._log_likelihood = self._likelihood = self._mahalanobis = None
        if x_ndim == 2:
            mu, mup = mu[..., None], mup[..., None]
        if want_hist:
            self._replay_for_saver(saver, zs, mask[:, 0], mu, cov, mup, covp, res[6], update_first)
        # bookkeeping of the last epoch, as the per-epoch loop leaves it
        if T:
            if update_first:
                self.x_post, self.P_post = mu[-1].copy(), cov[-1].copy()
                self.x_prior, self.P_prior = np.copy(self.x), self.P.copy()
            else:
                self.x_prior, self.P_prior = mup[-1].copy(), covp[-1].copy()
                self.x_post, self.P_post = np.copy(self.x), self.P.copy()
        return (mu, cov, mup, covp)

    def _batch_filter_epochwise(self, zs, Fs, Qs, Hs, Rs, Bs, us, update_first, saver):
        """batch_filter with a non-default `inv`: the reference's own per-epoch loop (kalman_filter.py:940-991) over this
        object's predict() / update(), every step on the GPU around the caller's callable -- one filter's escape hatch for a
        singular S, not the throughput path (that is the fused launch above)."""
        T = np.size(zs, 0)
        seq = lambda v, default: [default] * T if v is None else v          # noqa: E731
        Fs, Qs, Hs, Rs = seq(Fs, self.F), seq(Qs, self.Q), seq(Hs, self.H), seq(Rs, self.R)
        Bs = seq(Bs, self.B)
        us_ = [None] * T if us is None else us          # (without `us` the fused path ignores B too: see batch_filter)
        n = self.dim_x
        if np.ndim(self.x) == 1:
            means, means_p = np.zeros((T, n)), np.zeros((T, n))
        else:
            means, means_p = np.zeros((T, n, 1)), np.zeros((T, n, 1))
        covariances, covariances_p = np.zeros((T, n, n)), np.zeros((T, n, n))
        for i, (z, F, Q, H, R, B, u) in enumerate(zip(zs, Fs, Qs, Hs, Rs, Bs, us_)):
            if update_first:
                self.update(z, R=R, H=H)
                means[i, :], covariances[i, :, :] = self.x, self.P
                self.predict(u=u, B=B, F=F, Q=Q)
                means_p[i, :], covariances_p[i, :, :] = self.x, self.P
            else:
                self.predict(u=u, B=B, F=F, Q=Q)
                means_p[i, :], covariances_p[i, :, :] = self.x, self.P
                self.update(z, R=R, H=H)
                means[i, :], covariances[i, :, :] = self.x, self.P
            if saver is not None:
                saver.save()
        return (means, covariances, means_p, covariances_p)

    def _replay_for_saver(self, saver, zs, present, mu, cov, mup, covp, hist, update_first):
        """saver.save() reads the filter's attributes after every epoch (kalman_filter.py:990-991,
        filterpy/common/helpers.py:121-152).  The whole run was ONE kernel launch that also stored the
        per-epoch K, y, S, SI; here the attributes are set epoch by epoch from those histories and
        saver.save() is called -- bookkeeping only, no arithmetic."""
        x_ndim = np.ndim(self.x)
        m = self.dim_z
        for i in range(len(zs)):
            self.K, self.S, self.SI = hist["K"][i, 0].copy(), hist["S"][i, 0].copy(), hist["SI"][i, 0].copy()
            yv = hist["y"][i, 0]
            self.y = yv.reshape(m, 1).copy() if (x_ndim == 2 or not present[i]) else yv.copy()
            self.z = deepcopy(zs[i]) if present[i] else np.array([[None] * m]).T
            self._log_likelihood = self._likelihood = self._mahalanobis = None
            if update_first:
                self.x_post, self.P_post = mu[i].copy(), cov[i].copy()
                self.x_prior, self.P_prior = mup[i].copy(), covp[i].copy()
                self.x, self.P = mup[i].copy(), covp[i].copy()
            else:
                self.x_prior, self.P_prior = mup[i].copy(), covp[i].copy()
                self.x_post, self.P_post = mu[i].copy(), cov[i].copy()
                self.x, self.P = mu[i].copy(), cov[i].copy()
            saver.save()

    # -- rts_smoother -----------------------------------------------------------
    def rts_smoother(self, Xs, Ps, Fs=None, Qs=None, inv=None):
        """kalman_filter.py:995-1074 (class method: F[k+1], Q[k+1]).  Returns (x, P, K, Pp).
        inv=None / numpy.linalg.inv: Pp^-1 is applied by the kernel's in-lane LDL' solve (one launch).  Any other callable
        (the reference's documented use: numpy.linalg.pinv) is applied to every Pp[k] on the host between two launches."""
        if len(Xs) != len(Ps):
            raise ValueError('length of Xs and Ps must be the same')
        return _rts(np.asarray(Xs, dtype=np.float64), np.asarray(Ps, dtype=np.float64),
                    Fs if Fs is not None else self.F, Qs if Qs is not None else self.Q, convention=0,
                    inv=None if (inv is None or inv is np.linalg.inv) else inv)

    # -- small helpers the reference exposes -------------------------------------
    def residual_of(self, z):
        """kalman_filter.py:1175-1181."""
        z = reshape_z(z, self.dim_z, np.ndim(self.x))
        return z - np.dot(self.H, self.x_prior)

    def measurement_of_state(self, x):
        """kalman_filter.py:1183-1201."""
        return np.dot(self.H, x)

    @property
    def log_likelihood(self):
        """kalman_filter.py:1203-1211: lazily from y, S."""
        if self._log_likelihood is None:
            self._log_likelihood = logpdf(x=self.y, cov=self.S)
        return self._log_likelihood

    @property
    def likelihood(self):
        """kalman_filter.py:1213-1226."""
        if self._likelihood is None:
            self._likelihood = exp(self.log_likelihood)
            if self._likelihood == 0:
                self._likelihood = sys.float_info.min
        return self._likelihood

    @property
    def mahalanobis(self):
        """kalman_filter.py:1228-1240."""
        if self._mahalanobis is None:
            y = np.asarray(self.y, dtype=float).reshape(-1, 1)
            self._mahalanobis = sqrt(float(np.dot(np.dot(y.T, self.SI), y).item()))
        return self._mahalanobis

    def log_likelihood_of(self, z):
        """kalman_filter.py:1252-1260: log-density of ``z`` under N(Hx, S) with the S of the last update
        (host arithmetic on the kernel's outputs, like the reference's own lazily evaluated likelihoods)."""
        if z is None:
            return log(sys.float_info.min)
        return logpdf(z, np.dot(self.H, self.x), self.S)

    def test_matrix_dimensions(self, z=None, H=None, R=None, F=None, Q=None):
        """kalman_filter.py:1299-1398: assert that x, P, Q, F, H, R (the arguments override the attributes)
        and a measurement ``z`` have shapes the filter equations accept.  Raises AssertionError with the
        offending shape; returns None."""
        H = self.H if H is None else H
        R = self.R if R is None else R
        F = self.F if F is None else F
        Q = self.Q if Q is None else Q
        x, P, n = self.x, self.P, self.dim_x

        def need(ok, what, want, got):
            assert ok, "Shape of {} must be {}, but is {}".format(what, want, got)

        assert x.ndim in (1, 2), "x must have one or two dimensions, but has {}".format(x.ndim)
        need(x.shape == ((n,) if x.ndim == 1 else (n, 1)), "x", (n, 1), x.shape)
        need(P.shape == (n, n), "P", (n, n), P.shape)
        need(np.shape(Q) == (n, n), "Q", (n, n), np.shape(Q))
        need(np.shape(F) == (n, n), "F", (n, n), np.shape(F))
        need(np.ndim(H) == 2 and np.shape(H)[1] == P.shape[0], "H", ("dim_z", P.shape[0]), np.shape(H))
        mz = np.shape(H)[0]
        r_shape = np.shape(R)
        if mz == 1:      # a 1x1 innovation covariance: scalar, 1-element vector or 1x1 matrix
            assert r_shape in ((), (1,), (1, 1)), "R must be scalar or one element array, but is shaped {}".format(r_shape)
        else:
            need(r_shape == (mz, mz), "R", (mz, mz), r_shape)

        # z must be subtractable from Hx, whose shape follows x's (vector or column)
        z_shape = np.shape(z) if z is not None else (self.dim_z, 1)
        hx_shape = np.shape(np.dot(H, x))
        msg = "shape of z should be {}, not {} for the given H".format(hx_shape, z_shape)
        if z_shape == ():
            assert len(hx_shape) == 1 or hx_shape == (1, 1), msg
        elif hx_shape == (1,):
            assert z_shape[0] == 1, msg
        else:
            assert z_shape == hx_shape or (len(z_shape) == 1 and hx_shape == (z_shape[0], 1)), msg
        if len(hx_shape) > 1 and hx_shape != (1, 1):
            assert hx_shape == z_shape, msg

    @property
    def alpha(self):
        """kalman_filter.py:1242-1257."""
        return self._alpha_sq ** .5

    @alpha.setter
    def alpha(self, value):
        if not np.isscalar(value) or value < 1:
            raise ValueError('alpha must be a float greater than 1')
        self._alpha_sq = value ** 2

    def __repr__(self):
        return "\n".join(["KalmanFilter object (filterpy_amd, gfx950)"] +
                         [f"{k} = {getattr(self, k)!r}" for k in
                          ("dim_x", "dim_z", "dim_u", "x", "P", "F", "Q", "R", "H", "K", "y", "S", "alpha")])


def _rts(Xs, Ps, Fs, Qs, convention, inv=None):
    T = Xs.shape[0]
    n = Xs.shape[1]
    Xr = Xs.reshape(T, 1, n)
    Pr = np.asarray(Ps, dtype=np.float64).reshape(T, 1, n, n)
    per_step = (isinstance(Fs, (list, tuple)) or np.ndim(Fs) == 3 or isinstance(Qs, (list, tuple)) or np.ndim(Qs) == 3)
    if per_step:
        def stack(v, name, scalar):
            if isinstance(v, (list, tuple)) or np.ndim(v) == 3:
                if len(v) != T:
                    raise ValueError(f"{name} must hold one matrix per epoch")
                return np.stack([_mat(a, n, n, name, scalar) for a in v])
            return np.broadcast_to(_mat(v, n, n, name, scalar), (T, n, n)).copy()
        Fm, Qm, mode = stack(Fs, "F", "eye"), stack(Qs, "Q", "full"), FK_MODEL_PER_STEP
    else:
        Fm, Qm, mode = _mat(Fs, n, n, "F"), _mat(Qs, n, n, "Q", scalar="full"), FK_MODEL_SHARED
    if T == 0:
        return Xs.copy(), Ps.copy(), np.zeros((0, n, n)), Ps.copy()
    x, P, K, Pp = _Core.rts(n, 1, T, Xr, Pr, Fm, Qm, mode, convention, inv=inv)
    return x[:, 0].reshape(Xs.shape), P[:, 0], K[:, 0], Pp[:, 0]


# -------------------------------------------------------- KalmanFilterBank --
class KalmanFilterBank(object):
    """N independent linear Kalman filters stepped in lock-step on the GPU, with
    KalmanFilter's method names.  Shapes gain a track axis:

        x (N, dim_x)   P (N, dim_x, dim_x)   zs (T, N, dim_z)   [NaN row / mask = missing]
        F, Q, H, R: one matrix shared by all tracks, or (N, ., .) one per track

    batch_filter returns (means (T,N,n), covariances (T,N,n,n), means_p, covariances_p) as NumPy
    arrays -- for layout='soa' they are transposed *views* of the [T][e][N] device layout -- or,
    with device_outputs=True, the device tensors themselves (no PCIe copy).
    """

    def __init__(self, dim_x, dim_z, n_tracks, dim_u=0, layout="soa"):
        if dim_x < 1 or dim_z < 1 or dim_u < 0 or n_tracks < 1:
            raise ValueError("dim_x, dim_z, n_tracks must be >= 1 and dim_u >= 0")
        if layout not in E.LAYOUTS:
            raise ValueError("layout must be 'soa' or 'aos'")
        self.dim_x, self.dim_z, self.dim_u, self.n_tracks, self.layout = dim_x, dim_z, dim_u, n_tracks, layout
        self.x = np.zeros((n_tracks, dim_x))
        self.P = np.tile(np.eye(dim_x), (n_tracks, 1, 1))
        self.F, self.Q = np.eye(dim_x), np.eye(dim_x)
        self.H, self.R = np.zeros((dim_z, dim_x)), np.eye(dim_z)
        self.B = None
        self.K = np.zeros((dim_x, dim_z))
        self.M = np.zeros((dim_x, dim_z))
        self._alpha_sq = 1.0

    alpha = KalmanFilter.alpha

    def _models(self):
        n, m, N = self.dim_x, self.dim_z, self.n_tracks
        mats = dict(F=(self.F, n, n), Q=(self.Q, n, n), H=(self.H, m, n), R=(self.R, m, m))
        per_track = any(np.ndim(v[0]) == 3 for v in mats.values())
        out = {}
        for k, (v, r, c) in mats.items():
            v = np.asarray(v, dtype=np.float64)
            if v.ndim == 3:
                if v.shape != (N, r, c):
                    raise ValueError(f"{k} has shape {v.shape}, expected ({N}, {r}, {c})")
                out[k] = np.ascontiguousarray(v)
            else:
                mm = _mat(v, r, c, k)
                out[k] = np.broadcast_to(mm, (N, r, c)).copy() if per_track else mm
        return out, (FK_MODEL_PER_TRACK if per_track else FK_MODEL_SHARED)

    def _state(self):
        n, N = self.dim_x, self.n_tracks
        x = np.asarray(self.x, dtype=np.float64).reshape(N, n)
        P = np.asarray(self.P, dtype=np.float64)
        if P.shape != (N, n, n):
            P = np.broadcast_to(P, (N, n, n)).copy()
        return x, P

    def predict(self, u=None):
        mods, mode = self._models()
        x, P = self._state()
        kw = {}
        if u is not None and self.B is not None:
            uu = np.asarray(u, dtype=np.float64).reshape(self.n_tracks, -1)
            B = np.asarray(self.B, dtype=np.float64)
            if mode == FK_MODEL_PER_TRACK and B.ndim == 2:
                B = np.broadcast_to(B, (self.n_tracks,) + B.shape).copy()
            kw = dict(B=B, u=uu, nu=uu.shape[1])
        self.x, self.P = _Core.predict(self.dim_x, self.n_tracks, x, P, mods["F"], mods["Q"], mode,
                                       alpha_sq=self._alpha_sq, layout=self.layout, **kw)

    def update(self, z, mask=None):
        """z (N, dim_z); rows that are all-NaN (or mask == 0) are missing measurements."""
        mods, mode = self._models()
        x, P = self._state()
        z = np.asarray(z, dtype=np.float64).reshape(self.n_tracks, self.dim_z)
        nanrow = np.isnan(z).all(axis=1)
        if mask is None and nanrow.any():
            mask = ~nanrow
        if mask is not None:
            z = np.where(np.asarray(mask, dtype=bool)[:, None], z, 0.0)
        self.x, self.P, self.y, self.K, self.S, self.SI = _Core.update(
            self.dim_x, self.dim_z, self.n_tracks, x, P, z, mods["H"], mods["R"], mode, mask=mask, layout=self.layout)

    def batch_filter(self, zs, mask=None, update_first=False, store=True, device_outputs=False, extras=(),
                     cov_interleave=True, placement=None):
        """zs (T, N, dim_z) NumPy array or a device tensor already in self.layout.
        device_outputs=True: the four histories come back as device tensors in self.layout; the two covariance histories
        are then strided VIEWS of one array in which a track's posterior and prior record sit side by side (one write
        front: docs/PLACEMENT.md) -- `.contiguous()` gives a dense copy, cov_interleave=False two dense arrays.
        placement="probe" (device outputs of 256 MiB and more): two dense arrays placed in HBM by timing this very launch
        on candidate buffers (filterpy_amd/placement.py: candidates arrive one at a time, at most 11 and at most half of the
        free memory, and the probe stops as soon as a fast pair shows -- a few to ~100 launches once per shape; the pair is
        remembered -- the last two shapes per device -- and reused while no earlier result is alive; `self.placement_info`
        says what happened) -- the fastest arrangement measured, and what placement=None (the default) does by itself at
        dim_x <= 4 for histories of that size; placement="interleave" keeps the one-array form there too (no probe, no
        transient candidate buffers).  The probe's losing candidates stay in torch's caching allocator (reused by later
        allocations of this process); `filterpy_amd.placement.forget_placed_pairs()` drops the remembered pairs and hands
        every cached block back to the driver.  The outcome is this process's draw: nothing is kept across processes.
        extras: any of 'y', 'K', 'S', 'SI', 'log_likelihood', 'mahalanobis' -> also returns a dict of the
        per-step histories (T, N, ...) as a fifth element (what filterpy.common.Saver would record)."""
        import torch
        mods, mode = self._models()
        x, P = self._state()
        if isinstance(zs, torch.Tensor):
            T = zs.shape[0]
            z = zs
        else:
            z = np.asarray(zs, dtype=np.float64)
            T = z.shape[0]
            z = z.reshape(T, self.n_tracks, self.dim_z)
            # NaN rows = missing measurements.  (One reduction first: a sum is NaN whenever any element is -- 0.08 s for the
            # 1.6 GB of BASELINE configs[1] where isnan().all(axis=2) and its 200 MB temporary take 0.5 s; only a bank that
            # holds a NaN, or an inf - inf, pays for the row test.)
            if mask is None and z.size and np.isnan(np.sum(z)):
                nanrow = np.isnan(z).all(axis=2)
                if nanrow.any():
                    mask = ~nanrow
            if mask is not None:
                z = np.where(np.asarray(mask, dtype=bool)[..., None], z, 0.0)
        pinfo = {}
        out = _Core.batch(self.dim_x, self.dim_z, self.n_tracks, T, x, P, z, mask, mods["F"], mods["Q"],
                          mods["H"], mods["R"], mode, alpha_sq=self._alpha_sq, update_first=update_first,
                          layout=self.layout, want_outputs=store, device_outputs=device_outputs, extras=tuple(extras),
                          cov_interleave=cov_interleave, placement=placement, placement_out=pinfo)
        self.placement_info = pinfo
        if device_outputs:
            self.x = E.from_records(out[4], self.layout, 0, (self.dim_x,))
            self.P = E.from_records(out[5], self.layout, 0, (self.dim_x, self.dim_x))
        else:
            self.x, self.P = out[4], out[5]
        return tuple(out[:4]) + ((out[6],) if extras else ())

    # -- SURVEY §8f N4 on the bank ---------------------------------------------------
    def _gain(self):
        K = np.asarray(self.K, dtype=np.float64)
        per_track = K.ndim == 3
        if per_track and K.shape != (self.n_tracks, self.dim_x, self.dim_z):
            raise ValueError(f"K has shape {K.shape}, expected ({self.n_tracks}, {self.dim_x}, {self.dim_z})")
        return (np.ascontiguousarray(K) if per_track else _mat(K, self.dim_x, self.dim_z, "K")), per_track

    def _shared(self, name, r, c):
        v = np.asarray(getattr(self, name), dtype=np.float64)
        if v.ndim == 3:
            raise NotImplementedError(f"per-track {name} is not supported by the steady-state / correlated kernels")
        return _mat(v, r, c, name)

    def steadystate_filter(self, zs, mask=None, us=None):
        """T x { predict_steadystate(); update_steadystate(zs[t]) } (kalman_filter.py:563-668) in one
        launch with the fixed gain self.K ((n, m) shared or (N, n, m) per track).  zs (T, N, dim_z).
        Returns (means (T,N,n), means_p (T,N,n), residuals (T,N,m)); P is untouched."""
        n, m, N = self.dim_x, self.dim_z, self.n_tracks
        z = np.asarray(zs, dtype=np.float64)
        T = z.shape[0]
        z = z.reshape(T, N, m)
        nanrow = np.isnan(z).all(axis=2)
        if mask is None and nanrow.any():
            mask = ~nanrow
        if mask is not None:
            z = np.where(np.asarray(mask, dtype=bool)[..., None], z, 0.0)
        K, per_track = self._gain()
        x, _ = self._state()
        kw = {}
        if us is not None and self.B is not None:
            uu = np.asarray(us, dtype=np.float64).reshape(T, N, -1)
            kw = dict(B=_mat(self.B, n, uu.shape[2], "B"), us=uu, nu=uu.shape[2])
        self.x, means, means_p, y = _Core.steadystate(n, m, N, T, x, self._shared("F", n, n), self._shared("H", m, n),
                                                      K, z, mask=mask, k_per_track=per_track, layout=self.layout, **kw)
        return means, means_p, y

    def predict_steadystate(self, u=None):
        n, N = self.dim_x, self.n_tracks
        x, _ = self._state()
        kw = {}
        if u is not None and self.B is not None:
            uu = np.asarray(u, dtype=np.float64).reshape(1, N, -1)
            kw = dict(B=_mat(self.B, n, uu.shape[2], "B"), us=uu, nu=uu.shape[2])
        self.x = _Core.steadystate(n, self.dim_z, N, 1, x, self._shared("F", n, n), None, None, None,
                                   layout=self.layout, **kw)[0]

    def update_steadystate(self, z, mask=None):
        n, m, N = self.dim_x, self.dim_z, self.n_tracks
        K, per_track = self._gain()
        x, _ = self._state()
        z = np.asarray(z, dtype=np.float64).reshape(1, N, m)
        self.x, _, _, y = _Core.steadystate(n, m, N, 1, x, None, self._shared("H", m, n), K, z,
                                            mask=None if mask is None else np.asarray(mask).reshape(1, N),
                                            k_per_track=per_track, layout=self.layout)
        self.y = y[0]

    def update_correlated(self, z, mask=None):
        """kalman_filter.py:670-752 for every track; self.M is (n, m) shared or (N, n, m)."""
        n, m, N = self.dim_x, self.dim_z, self.n_tracks
        M = np.asarray(self.M, dtype=np.float64)
        per_track = M.ndim == 3
        x, P = self._state()
        z = np.asarray(z, dtype=np.float64).reshape(N, m)
        self.x, self.P, self.y, self.K, self.S, self.SI = _Core.update_correlated(
            n, m, N, x, P, z, self._shared("H", m, n), self._shared("R", m, m),
            np.ascontiguousarray(M) if per_track else _mat(M, n, m, "M"), mask=mask, m_per_track=per_track,
            layout=self.layout)

    def update_sequential(self, start, z_i):
        """kalman_filter.py:754-824 for every track: z_i (N, length) are measurement components
        start:start+length; H and R are sliced accordingly."""
        n, N = self.dim_x, self.n_tracks
        z_i = np.asarray(z_i, dtype=np.float64).reshape(N, -1)
        length = z_i.shape[1]
        stop = start + length
        H = self._shared("H", self.dim_z, n)[start:stop]
        R = self._shared("R", self.dim_z, self.dim_z)[start:stop, start:stop]
        x, P = self._state()
        self.x, self.P, y, K, _, _ = _Core.update(n, length, N, x, P, z_i, H, R, FK_MODEL_SHARED, layout=self.layout)
        return y, K

    def rts_smoother(self, Xs, Ps):
        """Xs (T,N,n), Ps (T,N,n,n) -> (x, P, K, Pp), class convention (kalman_filter.py:1067)."""
        mods, mode = self._models()
        Xs, Ps = np.asarray(Xs, dtype=np.float64), np.asarray(Ps, dtype=np.float64)
        if len(Xs) != len(Ps):
            raise ValueError('length of Xs and Ps must be the same')
        return _Core.rts(self.dim_x, self.n_tracks, Xs.shape[0], Xs, Ps, mods["F"], mods["Q"], mode, 0,
                         layout=self.layout)


# ------------------------------------------------- module-level functions --
def _as_state(x, P):
    """(x, P) possibly scalars -> (xrow (1,n), P (1,n,n), restore(xrow, P))"""
    scalar = np.isscalar(x) or np.ndim(x) == 0
    xa = np.atleast_1d(np.asarray(x, dtype=np.float64))
    n = xa.size
    shape = xa.shape
    Pm = _mat(P, n, n, "P")

    def restore(xr, Pr):
        if scalar:
            return float(xr[0]), (float(Pr[0, 0]) if np.ndim(P) == 0 else Pr)
        return xr.reshape(shape), Pr
    return xa.reshape(1, n), Pm[None], n, restore


def predict(x, P, F=1, Q=0, u=0, B=1, alpha=1.):
    """Module-level predict (kalman_filter.py:1571-1621): x = Fx + Bu; P = a^2 FPF' + Q."""
    xr, Pr, n, restore = _as_state(x, P)
    Fm = _mat(F, n, n, "F")
    Qm = _mat(Q, n, n, "Q", scalar="full")
    kw = {}
    bu = np.dot(B, u)
    if np.any(np.asarray(bu) != 0):
        # the reference adds dot(B, u) whatever its shape; fold it into a 1-column control input
        kw = dict(B=np.asarray(bu, dtype=np.float64).reshape(n, 1), u=np.ones((1, 1)), nu=1)
    xn, Pn = _Core.predict(n, 1, xr, Pr, Fm, Qm, FK_MODEL_SHARED, alpha_sq=alpha * alpha, **kw)
    return restore(xn[0], Pn[0])


def update(x, P, z, R, H=None, return_all=False):
    """Module-level update (kalman_filter.py:1401-1508).  z None -> unchanged."""
    if z is None:
        if return_all:
            return x, P, None, None, None, None
        return x, P
    xr, Pr, n, restore = _as_state(x, P)
    if H is None:
        H = np.array([1])
    Hm = np.atleast_2d(np.asarray(H, dtype=np.float64))
    if Hm.shape[1] != n:
        Hm = Hm.reshape(-1, n)
    m = Hm.shape[0]
    zz = reshape_z(z, m, np.ndim(x))
    # a scalar R is used as given in both places (:1477 adds it to every element of S, :1497 forms r K K')
    Rm = _mat(R, m, m, "R", scalar="full")
    rflags = FK_KF_FLAG_R_JOSEPH_DIAG if ((np.isscalar(R) or np.ndim(R) == 0) and m > 1) else 0
    xn, Pn, y, K, S, SI = _Core.update(n, m, 1, xr, Pr, np.asarray(zz, dtype=np.float64).reshape(1, m), Hm, Rm,
                                       FK_MODEL_SHARED, flags=rflags)
    xo, Po = restore(xn[0], Pn[0])
    if return_all:
        yy = y[0].reshape(m, 1) if np.ndim(x) == 2 else (y[0] if np.ndim(x) == 1 else float(y[0, 0]))
        # the reference evaluates the likelihood at the POSTERIOR state (:1506)
        ll = logpdf(zz, np.dot(Hm, np.asarray(xo, dtype=float).reshape(n, -1)), S[0])
        return xo, Po, yy, K[0], S[0], ll
    return xo, Po


def update_steadystate(x, z, K, H=None):
    """Module-level steady-state update (kalman_filter.py:1511-1568): x + K (z - Hx) with a given gain, no
    covariance anywhere.  z None -> x unchanged.  Runs fk_kf_steadystate_f64 on one track."""
    if z is None:
        return x
    scalar = np.isscalar(x) or np.ndim(x) == 0
    xa = np.atleast_1d(np.asarray(x, dtype=np.float64))
    n = xa.size
    Hm = np.atleast_2d(np.asarray(1.0 if H is None else H, dtype=np.float64))
    if Hm.shape[1] != n:
        Hm = Hm.reshape(-1, n)
    m = Hm.shape[0]
    zz = np.asarray(reshape_z(z, m, np.ndim(x)), dtype=np.float64).reshape(1, 1, m)
    # a scalar gain scales the residual (dot(K, y) = K * y); a scalar for a 1 x m gain multiplies every entry
    Km = np.eye(n) * float(K) if (np.ndim(K) == 0 and n == m) else _mat(K, n, m, "K", scalar="full")
    xn, _, _, _ = _Core.steadystate(n, m, 1, 1, xa.reshape(1, n), None, Hm, Km, zz)
    return float(xn[0, 0]) if scalar else xn[0].reshape(xa.shape)


def predict_steadystate(x, F=1, u=0, B=1):
    """Module-level steady-state predict (kalman_filter.py:1624-1660): Fx + Bu, no covariance.
    Runs fk_kf_steadystate_f64 on one track."""
    scalar = np.isscalar(x) or np.ndim(x) == 0
    xa = np.atleast_1d(np.asarray(x, dtype=np.float64))
    n = xa.size
    kw = {}
    bu = np.dot(B, u)
    if np.any(np.asarray(bu) != 0):
        # the reference adds dot(B, u) whatever its shape; fold it into a 1-column control input
        kw = dict(B=np.asarray(bu, dtype=np.float64).reshape(n, 1), us=np.ones((1, 1, 1)), nu=1)
    xn, _, _, _ = _Core.steadystate(n, 1, 1, 1, xa.reshape(1, n), _mat(F, n, n, "F"), None, None, None, **kw)
    return float(xn[0, 0]) if scalar else xn[0].reshape(xa.shape)


def batch_filter(x, P, zs, Fs, Qs, Hs, Rs, Bs=None, us=None, update_first=False, saver=None):
    """Module-level batch_filter (kalman_filter.py:1664-1788): per-epoch lists are required."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    m = np.atleast_2d(np.asarray(Hs[0], dtype=np.float64)).shape[0]
    kf = KalmanFilter(n, m)
    kf.x, kf.P = x.copy(), _mat(P, n, n, "P")
    kf.F, kf.Q, kf.H, kf.R = Fs[0], Qs[0], np.atleast_2d(Hs[0]), Rs[0]
    if us is None:
        Bs = None
    return kf.batch_filter(zs, Fs=Fs, Qs=Qs, Hs=[np.atleast_2d(h) for h in Hs], Rs=Rs, Bs=Bs, us=us,
                           update_first=update_first, saver=saver)


def rts_smoother(Xs, Ps, Fs, Qs):
    """Module-level rts_smoother (kalman_filter.py:1792-1858): uses Fs[k], Qs[k]."""
    if len(Xs) != len(Ps):
        raise ValueError('length of Xs and Ps must be the same')
    return _rts(np.asarray(Xs, dtype=np.float64), np.asarray(Ps, dtype=np.float64), Fs, Qs, convention=1)
