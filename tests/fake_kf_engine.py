"""CPU stand-ins for the linear-Kalman entry points of libfilterhip.so, for HOST-LOGIC tests only (like fake_ut_engine.py for the
unscented path): they let the whole Python layer above the C ABI -- filterpy_amd.kalman.KalmanFilter / KalmanFilterBank and the
module functions, their argument marshalling, record layouts, model modes, masks and flags -- run on CPU tensors.  Each stand-in
reads its operands exactly as include/filterhip.h lays them out (records in `layout`, models per `model_mode`, `mask` 0 =
missing, FK_KF_FLAG_R_JOSEPH_DIAG) and computes with the oracle (tests may use it).  The kernels themselves are tested under -m gpu;
what this buys is that the drop-in surface can be held against the LIVE reference on the CPU (tests/test_host_dropin_live.py)."""
import numpy as np
import torch

from oracle import kf_oracle

CPU = torch.device("cpu")
SHARED, PER_TRACK, PER_TRACK_STEP, PER_STEP = 0, 1, 2, 3
NOT_PD = 1
FLAG_R_JOSEPH_DIAG = 1


def _layout_codes():
    from filterpy_amd import _engine as E
    return {v: k for k, v in E.LAYOUTS.items()}


def get(rec, layout, lead, shape):
    """record tensor -> numpy lead + (N,) + shape"""
    a = rec.detach().numpy()
    E = int(np.prod(shape)) if shape else 1
    if layout == "soa":
        a = np.swapaxes(a.reshape(*a.shape[:lead], E, a.shape[-1]), -1, -2)
    return np.array(a.reshape(*a.shape[:lead + 1], *shape), dtype=float)


def put(rec, layout, lead, arr):
    """numpy lead + (N,) + shape -> into the record tensor (through views: interleaved histories are strided)"""
    if rec is None:
        return
    a = np.asarray(arr, dtype=float)
    a = a.reshape(*a.shape[:lead + 1], -1)
    if layout == "soa":
        a = np.swapaxes(a, -1, -2)
    rec.copy_(torch.as_tensor(np.ascontiguousarray(a)).reshape(rec.shape))


def model(Mx, mode, layout, shape, T, N):
    """-> function (t, i) -> matrix, from the operand as the ABI defines it for `mode`"""
    if Mx is None:
        return lambda t, i: None
    if mode == SHARED:
        a = Mx.detach().numpy().reshape(shape)
        return lambda t, i: a
    if mode == PER_STEP:
        a = Mx.detach().numpy().reshape(T, *shape)
        return lambda t, i: a[t]
    if mode == PER_TRACK:
        a = get(Mx, layout, 0, shape)
        return lambda t, i: a[i]
    a = get(Mx, layout, 1, shape)
    return lambda t, i: a[t, i]


def _update(x, P, z, R, H, rj_diag):
    if not rj_diag:
        return kf_oracle.kf_update(x, P, z, R, H)
    # kalman_filter.py:540, :556 with a scalar R attribute: S gets R as given (r on every element), the Joseph term r K K'
    y = z - H @ x
    PHT = P @ H.T
    S = H @ PHT + R
    SI = np.linalg.inv(S)
    K = PHT @ SI
    x = x + K @ y
    I_KH = np.eye(len(x)) - K @ H
    P = (I_KH @ P) @ I_KH.T + (K @ np.diag(np.diag(R))) @ K.T
    return x, P, y, K, S, SI


def install(monkeypatch):
    from filterpy_amd import _engine as E
    codes = _layout_codes()
    calls = []
    monkeypatch.setattr(E, "require_gpu", lambda: CPU)
    # host <-> device transfers COPY; on CPU tensors torch.as_tensor / .cpu().numpy() would alias the caller's arrays and a stand-in
    # that updates x in place (as the kernels do, on the device copy) would write into them
    real_dev, real_from = E.dev, E.from_records
    monkeypatch.setattr(E, "dev", lambda a, device=None: real_dev(a, device).clone())
    monkeypatch.setattr(E, "from_records", lambda t, layout, lead, rec_shape: real_from(t.clone(), layout, lead, rec_shape))

    def lay(desc):
        return codes[desc["layout"]]

    def kf_batch(desc, F, Q, H, R, z, x, P, extras=None, *, B=None, u=None, mask=None, means=None, covs=None, means_p=None,
                 covs_p=None, status=None):
        calls.append("batch")
        n, m, nu, mode, N, T, L = desc["n"], desc["m"], desc["nu"], desc["model_mode"], desc["N"], desc["T"], lay(desc)
        rj = bool(desc.get("flags", 0) & FLAG_R_JOSEPH_DIAG)
        fF, fQ, fH, fR = (model(M, mode, L, s, T, N) for M, s in ((F, (n, n)), (Q, (n, n)), (H, (m, n)), (R, (m, m))))
        fB = model(B, mode, L, (n, nu), T, N) if nu else (lambda t, i: None)
        zs = get(z, L, 1, (m,))
        us = get(u, L, 1, (nu,)) if (nu and u is not None) else None
        mk = None if mask is None else mask.detach().numpy().reshape(T, N)
        xs, Ps = get(x, L, 0, (n,)), get(P, L, 0, (n, n))
        out = {k: np.zeros((T, N) + s) for k, s in (("mu", (n,)), ("cov", (n, n)), ("mup", (n,)), ("covp", (n, n)), ("y", (m,)),
                                                       ("K", (n, m)), ("S", (m, m)), ("SI", (m, m)), ("ll", ()), ("maha", ()))}
        for i in range(N):
            xi, Pi = xs[i].copy(), Ps[i].copy()
            last = [np.zeros((n, m)), np.zeros((m, m)), np.zeros((m, m))]
            try:
                for t in range(T):
                    def pred():
                        return kf_oracle.kf_predict(xi, Pi, fF(t, i), fQ(t, i), fB(t, i), None if us is None else us[t, i],
                                                    desc["alpha_sq"])

                    def upd():
                        if mk is not None and not mk[t, i]:
                            out["K"][t, i], out["S"][t, i], out["SI"][t, i] = last
                            return xi, Pi
                        xn, Pn, y, K, S, SI = _update(xi, Pi, zs[t, i], fR(t, i), fH(t, i), rj)
                        out["y"][t, i], out["K"][t, i], out["S"][t, i], out["SI"][t, i] = y, K, S, SI
                        if extras and extras.get("log_likelihood") is not None:
                            out["ll"][t, i] = kf_oracle.log_likelihood(y, S)
                        if extras and extras.get("mahalanobis") is not None:
                            out["maha"][t, i] = kf_oracle.mahalanobis(y, SI)
                        last[:] = [K, S, SI]
                        return xn, Pn
                    if desc["update_first"]:
                        xi, Pi = upd()
                        out["mu"][t, i], out["cov"][t, i] = xi, Pi
                        xi, Pi = pred()
                        out["mup"][t, i], out["covp"][t, i] = xi, Pi
                    else:
                        xi, Pi = pred()
                        out["mup"][t, i], out["covp"][t, i] = xi, Pi
                        xi, Pi = upd()
                        out["mu"][t, i], out["cov"][t, i] = xi, Pi
            except np.linalg.LinAlgError:
                if status is not None:
                    status[i] |= NOT_PD
            xs[i], Ps[i] = xi, Pi
        put(x, L, 0, xs)
        put(P, L, 0, Ps)
        for rec, key in ((means, "mu"), (covs, "cov"), (means_p, "mup"), (covs_p, "covp")):
            put(rec, L, 1, out[key])
        if extras:
            for k, key in (("y", "y"), ("K", "K"), ("S", "S"), ("SI", "SI")):
                put(extras.get(k), L, 1, out[key])
            for k, key in (("log_likelihood", "ll"), ("mahalanobis", "maha")):
                if extras.get(k) is not None:
                    extras[k].copy_(torch.as_tensor(out[key]))

    def kf_batch_filter(desc, F, Q, H, R, z, x, P, **kw):
        return kf_batch(desc, F, Q, H, R, z, x, P, None, **kw)

    def kf_batch_filter_ex(desc, F, Q, H, R, z, x, P, extras, **kw):
        return kf_batch(desc, F, Q, H, R, z, x, P, extras, **kw)

    def kf_predict(desc, F, Q, x, P, *, B=None, u=None, status=None):
        calls.append("predict")
        n, nu, mode, N, L = desc["n"], desc["nu"], desc["model_mode"], desc["N"], lay(desc)
        fF, fQ = model(F, mode, L, (n, n), 1, N), model(Q, mode, L, (n, n), 1, N)
        fB = model(B, mode, L, (n, nu), 1, N) if nu else (lambda t, i: None)
        us = get(u, L, 0, (nu,)) if (nu and u is not None) else None
        xs, Ps = get(x, L, 0, (n,)), get(P, L, 0, (n, n))
        for i in range(N):
            xs[i], Ps[i] = kf_oracle.kf_predict(xs[i], Ps[i], fF(0, i), fQ(0, i), fB(0, i), None if us is None else us[i],
                                                desc["alpha_sq"])
        put(x, L, 0, xs)
        put(P, L, 0, Ps)

    def _one_update(fn, desc, H, R, z, x, P, mask, y, K, S, SI, status, extra=None):
        n, m, mode, N, L = desc["n"], desc["m"], desc["model_mode"], desc["N"], lay(desc)
        rj = bool(desc.get("flags", 0) & FLAG_R_JOSEPH_DIAG)
        zs = get(z, L, 0, (m,))
        mk = None if mask is None else mask.detach().numpy().reshape(N)
        xs, Ps = get(x, L, 0, (n,)), get(P, L, 0, (n, n))
        o = dict(y=np.zeros((N, m)), K=np.zeros((N, n, m)), S=np.zeros((N, m, m)), SI=np.zeros((N, m, m)))
        for i in range(N):
            if mk is not None and not mk[i]:
                continue
            try:
                xs[i], Ps[i], o["y"][i], o["K"][i], o["S"][i], o["SI"][i] = fn(i, xs[i], Ps[i], zs[i], rj)
            except np.linalg.LinAlgError:
                if status is not None:
                    status[i] |= NOT_PD
        put(x, L, 0, xs)
        put(P, L, 0, Ps)
        for rec, key in ((y, "y"), (K, "K"), (S, "S"), (SI, "SI")):
            put(rec, L, 0, o[key])

    def kf_update(desc, H, R, z, x, P, *, mask=None, y=None, K=None, S=None, SI=None, status=None):
        calls.append("update")
        n, m, mode, N, L = desc["n"], desc["m"], desc["model_mode"], desc["N"], lay(desc)
        fH, fR = model(H, mode, L, (m, n), 1, N), model(R, mode, L, (m, m), 1, N)
        given = desc.get("flags", 0) & (4 | 8)           # FK_KF_FLAG_S_ONLY / FK_KF_FLAG_SI_GIVEN (a caller-supplied inverse)
        if given:
            zs, xs, Ps = get(z, L, 0, (m,)), get(x, L, 0, (n,)), get(P, L, 0, (n, n))
            mk = None if mask is None else mask.detach().numpy().reshape(N)
            yo, So = get(y, L, 0, (m,)), get(S, L, 0, (m, m))
            Ko = get(K, L, 0, (n, m)) if K is not None else np.zeros((N, n, m))
            SIs = get(SI, L, 0, (m, m)) if SI is not None else None
            rj = bool(desc.get("flags", 0) & FLAG_R_JOSEPH_DIAG)
            for i in range(N):
                if mk is not None and not mk[i]:
                    continue
                Hm, Rm = fH(0, i), fR(0, i)
                yo[i] = zs[i] - Hm @ xs[i]
                PHT = Ps[i] @ Hm.T
                So[i] = Hm @ PHT + Rm
                if given == 8:
                    Ko[i] = PHT @ SIs[i]
                    xs[i] = xs[i] + Ko[i] @ yo[i]
                    I_KH = np.eye(n) - Ko[i] @ Hm
                    Ps[i] = (I_KH @ Ps[i]) @ I_KH.T + (Ko[i] @ (np.diag(np.diag(Rm)) if rj else Rm)) @ Ko[i].T
            put(y, L, 0, yo)
            put(S, L, 0, So)
            if given == 8:
                put(x, L, 0, xs)
                put(P, L, 0, Ps)
                if K is not None:
                    put(K, L, 0, Ko)
            return
        _one_update(lambda i, xi, Pi, zi, rj: _update(xi, Pi, zi, fR(0, i), fH(0, i), rj), desc, H, R, z, x, P, mask, y, K, S, SI,
                    status)

    def kf_update_correlated(desc, H, R, M, z, x, P, *, mask=None, y=None, K=None, S=None, SI=None, status=None):
        calls.append("update_correlated")
        n, m, mode, N, L = desc["n"], desc["m"], desc["model_mode"], desc["N"], lay(desc)
        Hm, Rm = H.detach().numpy().reshape(m, n), R.detach().numpy().reshape(m, m)
        fM = model(M, mode, L, (n, m), 1, N)
        _one_update(lambda i, xi, Pi, zi, rj: kf_oracle.update_correlated(xi, Pi, zi, Rm, Hm, fM(0, i)), desc, H, R, z, x, P, mask,
                    y, K, S, SI, status)

    def kf_rts(desc, F, Q, Xs, Ps, xs, Ps_out, K, Pp, *, convention=0, status=None):
        calls.append("rts")
        n, mode, N, T, L = desc["n"], desc["model_mode"], desc["N"], desc["T"], lay(desc)
        fF, fQ = model(F, mode, L, (n, n), T, N), model(Q, mode, L, (n, n), T, N)
        X, Pm = get(Xs, L, 1, (n,)), get(Ps, L, 1, (n, n))
        o = [np.zeros((T, N, n))] + [np.zeros((T, N, n, n)) for _ in range(3)]
        given = desc.get("flags", 0) & (16 | 32)         # FK_KF_FLAG_PP_ONLY / FK_KF_FLAG_PPINV_GIVEN
        if given:
            off = 0 if convention else 1
            Kio = get(K, L, 1, (n, n))
            o[0][:], o[1][:] = X, Pm
            for i in range(N):
                o[3][T - 1, i] = Pm[T - 1, i]
                for k in range(T - 2, -1, -1):
                    Fk, Qk = fF(k + off, i), fQ(k + off, i)
                    o[3][k, i] = Fk @ Pm[k, i] @ Fk.T + Qk
                    if given == 32:
                        Kk = Pm[k, i] @ Fk.T @ Kio[k, i]
                        o[2][k, i] = Kk
                        o[0][k, i] = X[k, i] + Kk @ (o[0][k + 1, i] - Fk @ X[k, i])
                        o[1][k, i] = Pm[k, i] + Kk @ (o[1][k + 1, i] - o[3][k, i]) @ Kk.T
            if given == 16:
                put(Pp, L, 1, o[3])
            else:
                for rec, arr in zip((xs, Ps_out, K, Pp), o):
                    put(rec, L, 1, arr)
            return
        for i in range(N):
            try:
                r = kf_oracle.rts_smoother(X[:, i], Pm[:, i], [fF(t, i) for t in range(T)], [fQ(t, i) for t in range(T)],
                                           "module" if convention else "class")
                for dst, src in zip(o, r):
                    dst[:, i] = src
            except np.linalg.LinAlgError:
                if status is not None:
                    status[i] |= NOT_PD
        for rec, arr in zip((xs, Ps_out, K, Pp), o):
            put(rec, L, 1, arr)

    def kf_steadystate(desc, F, H, K, z, x, *, B=None, u=None, mask=None, means=None, means_p=None, y=None):
        calls.append("steadystate")
        n, m, nu, N, T, L = desc["n"], desc["m"], desc["nu"], desc["N"], desc["T"], lay(desc)
        Fm = None if F is None else F.detach().numpy().reshape(n, n)
        Hm = None if H is None else H.detach().numpy().reshape(m, n)
        Bm = None if B is None else B.detach().numpy().reshape(n, nu)
        fK = (lambda i: None) if K is None else ((lambda i, a=get(K, L, 0, (n, m)): a[i]) if desc["model_mode"] == PER_TRACK
                                                 else (lambda i, a=K.detach().numpy().reshape(n, m): a))
        zs = None if z is None else get(z, L, 1, (m,))
        us = get(u, L, 1, (nu,)) if (nu and u is not None) else None
        mk = None if mask is None else mask.detach().numpy().reshape(T, N)
        xs = get(x, L, 0, (n,))
        mu, mup, ys = np.zeros((T, N, n)), np.zeros((T, N, n)), np.zeros((T, N, m))
        for i in range(N):
            xi = xs[i].copy()
            for t in range(T):
                if Fm is not None:
                    xi = Fm @ xi + (Bm @ us[t, i] if (Bm is not None and us is not None) else 0.0)
                    mup[t, i] = xi
                if zs is not None and (mk is None or mk[t, i]):
                    yy = zs[t, i] - Hm @ xi
                    xi = xi + fK(i) @ yy
                    ys[t, i] = yy
                mu[t, i] = xi
            xs[i] = xi
        put(x, L, 0, xs)
        put(means, L, 1, mu)
        put(means_p, L, 1, mup)
        put(y, L, 1, ys)

    for name, fn in (("kf_batch_filter", kf_batch_filter), ("kf_batch_filter_ex", kf_batch_filter_ex), ("kf_predict", kf_predict),
                     ("kf_update", kf_update), ("kf_update_correlated", kf_update_correlated), ("kf_rts", kf_rts),
                     ("kf_steadystate", kf_steadystate)):
        monkeypatch.setattr(E, name, fn)
    return calls


def install_imm(monkeypatch):
    """fk_imm_batch_ex_f64's stand-in (include/filterhip.h: phases, FK_IMM_FLAG_MMAE, zmask, the ll0 record, control input):
    IMM.py:160-249 / mmae.py:140-212 step by step on the bank records, with the oracle's pieces"""
    import sys
    from filterpy_amd import _engine as E
    from oracle import imm_oracle
    if getattr(E.require_gpu, "__name__", "") != "<lambda>":       # (install() not called yet: the transfers must copy here too)
        monkeypatch.setattr(E, "require_gpu", lambda: CPU)
        real_dev, real_from = E.dev, E.from_records
        monkeypatch.setattr(E, "dev", lambda a, device=None: real_dev(a, device).clone())
        monkeypatch.setattr(E, "from_records", lambda t, layout, lead, rec_shape: real_from(t.clone(), layout, lead, rec_shape))

    def imm_batch(n, m, nm, N, T, layout, F, Q, H, R, M, z, xs, Ps, mu, *, x_out=None, P_out=None, mu_out=None, x_prior_out=None,
                  P_prior_out=None, likelihood_out=None, status=None, phase=0, mmae=False, zmask=None, ll0=None, nu=0, B=None,
                  u=None):
        L = layout
        Fm, Qm = F.detach().numpy().reshape(nm, n, n), Q.detach().numpy().reshape(nm, n, n)
        Hm, Rm = H.detach().numpy().reshape(nm, m, n), R.detach().numpy().reshape(nm, m, m)
        Mm = None if M is None else M.detach().numpy().reshape(nm, nm)
        Bm = None if (B is None or not nu) else B.detach().numpy().reshape(nm, n, nu)
        steps = T if phase == 0 else 1
        zs = None if z is None else get(z, L, 1, (m,))
        us = None if (u is None or not nu) else get(u, L, 1, (nu,))
        mk = None if zmask is None else zmask.detach().numpy().reshape(steps, N)
        X, PP, MU = get(xs, L, 0, (nm, n)), get(Ps, L, 0, (nm, n, n)), get(mu, L, 0, (nm,))
        LL = None if ll0 is None else get(ll0, L, 0, (nm,))
        o = dict(x=np.zeros((steps, N, n)), P=np.zeros((steps, N, n, n)), mu=np.zeros((steps, N, nm)), xp=np.zeros((steps, N, n)),
                 Pp=np.zeros((steps, N, n, n)), L=np.zeros((steps, N, nm)))
        for i in range(N):
            x, P, p = [X[i, j].copy() for j in range(nm)], [PP[i, j].copy() for j in range(nm)], MU[i].copy()
            l0 = np.full(nm, -np.inf) if LL is None else LL[i].copy()
            for t in range(steps):
                if phase in (0, 1):
                    if mmae:
                        src_x, src_P = x, P
                    else:
                        _, omega = imm_oracle.mixing(p, Mm)
                        src_x, src_P = [], []
                        for j in range(nm):
                            xm = np.zeros(n)
                            for xi, wi in zip(x, omega[:, j]):
                                xm += xi * wi
                            Pm = np.zeros((n, n))
                            for xi, Pi, wi in zip(x, P, omega[:, j]):
                                yy = xi - xm
                                Pm += wi * (np.outer(yy, yy) + Pi)
                            src_x.append(xm)
                            src_P.append(Pm)
                    for j in range(nm):
                        x[j], P[j] = kf_oracle.kf_predict(src_x[j], src_P[j], Fm[j], Qm[j], None if Bm is None else Bm[j],
                                                          None if us is None else us[t, i])
                    if not mmae:
                        o["xp"][t, i], o["Pp"][t, i] = imm_oracle.state_estimate(x, P, p)
                if phase in (0, 2):
                    present = mk is None or bool(mk[t, i])
                    Lk = np.zeros(nm)
                    for j in range(nm):
                        if present:
                            x[j], P[j], y, K, S, SI = kf_oracle.kf_update(x[j], P[j], zs[t, i], Rm[j], Hm[j])
                            ll = kf_oracle.log_likelihood(y, S)
                            l0[j] = kf_oracle.log_likelihood(np.zeros(m), S)
                        else:
                            ll = l0[j]
                        Lk[j] = np.exp(ll)
                        if Lk[j] == 0:
                            Lk[j] = sys.float_info.min
                    if mmae:
                        p = p * Lk
                        p = p / sum(p)
                        xe = np.zeros(n)
                        for xj, pj in zip(x, p):
                            xe += np.dot(xj, pj)
                        Pe = np.zeros((n, n))
                        for xk, xj, Pj, pj in zip(xe, x, P, p):      # mmae.py:205-207: zips the COMPONENTS of x with the filters
                            yy = xj - xk
                            Pe += pj * (np.outer(yy, yy) + Pj)
                    else:
                        cbar, _ = imm_oracle.mixing(p, Mm)
                        p = cbar * Lk
                        p = p / np.sum(p)
                        xe, Pe = imm_oracle.state_estimate(x, P, p)
                    o["x"][t, i], o["P"][t, i], o["mu"][t, i], o["L"][t, i] = xe, Pe, p, Lk
            for j in range(nm):
                X[i, j], PP[i, j] = x[j], P[j]
            MU[i] = p
            if LL is not None:
                LL[i] = l0
        put(xs, L, 0, X)
        put(Ps, L, 0, PP)
        put(mu, L, 0, MU)
        if ll0 is not None:
            put(ll0, L, 0, LL)
        for rec, key in ((x_out, "x"), (P_out, "P"), (mu_out, "mu"), (x_prior_out, "xp"), (P_prior_out, "Pp"), (likelihood_out, "L")):
            put(rec, L, 1, o[key])

    monkeypatch.setattr(E, "imm_batch", imm_batch)


def install_resample(monkeypatch):
    """stand-ins for the resampling entry points (include/filterhip.h: one u per filter / N per filter, int32 / int64 indices,
    FK_STATUS_OVERRUN where the reference's merge loop would run off the end, the residual fill / draw pair): the oracle's loops"""
    from filterpy_amd import _engine as E
    from oracle import resample_oracle as ro
    OVERRUN = 4
    monkeypatch.setattr(E, "require_gpu", lambda: CPU)

    def merge(name, Fn, Np, w, u, idx, status):
        wh, uh = w.detach().numpy().reshape(Fn, Np), u.detach().numpy()
        for f in range(Fn):
            got, over = (ro.systematic_c(wh[f], float(uh.reshape(-1)[f])) if name == "sys" else ro.stratified_c(wh[f], uh.reshape(Fn, Np)[f]))
            idx[f] = torch.as_tensor(np.minimum(got, Np - 1))
            if over and status is not None:
                status[f] |= OVERRUN

    def resample_multinomial(Fn, Np, Nu, w, u, idx):
        wh, uh = w.detach().numpy().reshape(Fn, Np), u.detach().numpy().reshape(Fn, Nu)
        for f in range(Fn):
            idx[f] = torch.as_tensor(ro.multinomial(wh[f], uh[f]).astype(np.int64))

    def resample_residual_fill(Fn, Np, w, idx, k, cs, status):
        wh = w.detach().numpy().reshape(Fn, Np)
        for f in range(Fn):
            copies, kk, c = ro.residual_parts(wh[f])
            fill = np.repeat(np.arange(Np), np.maximum(copies, 0))[:Np]
            idx[f, :len(fill)] = torch.as_tensor(fill.astype(np.int32))
            k[f] = int(kk)
            cs[f] = torch.as_tensor(c)

    def resample_residual_draw(Fn, Np, cs, k, uoff, u, idx):
        ch, kh, uh, off = cs.detach().numpy(), k.detach().numpy(), u.detach().numpy(), uoff.detach().numpy()
        for f in range(Fn):
            cnt = Np - int(kh[f])
            if cnt > 0:
                idx[f, int(kh[f]):] = torch.as_tensor(np.searchsorted(ch[f], uh[off[f]:off[f] + cnt]).astype(np.int32))

    monkeypatch.setattr(E, "resample_systematic", lambda Fn, Np, w, u, idx, status=None: merge("sys", Fn, Np, w, u, idx, status))
    monkeypatch.setattr(E, "resample_stratified", lambda Fn, Np, w, u, idx, status=None: merge("str", Fn, Np, w, u, idx, status))
    monkeypatch.setattr(E, "resample_multinomial", resample_multinomial)
    monkeypatch.setattr(E, "resample_residual_fill", resample_residual_fill)
    monkeypatch.setattr(E, "resample_residual_draw", resample_residual_draw)
