"""Thin Python layer over the C ABI: device buffers (torch tensors as plain HBM allocations),
argument marshalling and status handling.  All arithmetic happens in libfilterhip.so.

Record layouts (see include/filterhip.h): 'aos' = NumPy C order [..][N][E], 'soa' =
lane-coalesced [..][E][N].  Every function here takes/returns torch CUDA tensors of dtype
float64 already in the requested layout; `to_records`/`from_records` convert NumPy arrays.
"""

import numpy as np
import torch

from . import _abi
from . import _transfer
from ._abi import (FK_LAYOUT_AOS, FK_LAYOUT_SOA, FK_MODEL_SHARED, FK_MODEL_PER_TRACK,
                   FK_MODEL_PER_TRACK_STEP, FK_MODEL_PER_STEP, fk_kf_desc, fk_ukf_desc)  # noqa: F401

LAYOUTS = {"aos": FK_LAYOUT_AOS, "soa": FK_LAYOUT_SOA}


def require_gpu():
    if not torch.cuda.is_available():
        raise _abi.FilterHipError("filterpy_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                                  "there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _ptr(t):
    return None if t is None else t.data_ptr()


def _mask_ptr(t, keep):
    """A [T][N] byte mask as the C ABI reads it: row stride N.  A strided view (NumPy's zs_mask[:, idx] hands out column-major
    memory, and torch.as_tensor keeps its strides) is copied; `keep` holds the copy for the duration of the call."""
    if t is None:
        return None
    if not t.is_contiguous():
        t = t.contiguous()
    keep.append(t)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def dev(a, device=None):
    """NumPy/torch -> contiguous float64 tensor on the current GPU."""
    device = device or require_gpu()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    h = np.ascontiguousarray(a, dtype=np.float64)
    if h.nbytes >= _transfer.MIN_BYTES:
        return _transfer.to_device(h, device)          # (large inputs: pinned, pipelined -- _transfer.py)
    if not h.flags.writeable:          # e.g. a broadcast view: torch refuses to wrap read-only memory silently
        h = h.copy()
    return torch.as_tensor(h, device=device)


def to_records(a, layout, lead):
    """Host array shaped lead + (N,) + rec  ->  device tensor in `layout`.

    `lead` = number of leading (time) axes (0 or 1).  For 'soa' the track axis is moved
    behind the flattened record axis."""
    t = dev(a)
    if layout == "aos":
        return t
    shp = t.shape
    N = shp[lead]
    rec = int(np.prod(shp[lead + 1:])) if len(shp) > lead + 1 else 1      # (explicit: -1 is ambiguous for N = 0)
    t = t.reshape(*shp[:lead], N, rec)
    return t.transpose(-1, -2).contiguous()


def download_into(pairs, wait=True):
    """[(device tensor, NumPy array of the same shape and dtype, both contiguous)]: the tensors' bytes into the arrays, through the
    pinned pipeline where that pays (kalman_filter.py: the streamed host outputs of batch_filter).  wait=False: returns futures the
    caller must .result() before it touches the tensors or the arrays again."""
    return _transfer.into_host(pairs, wait)


def from_records(t, layout, lead, rec_shape):
    """Device tensor in `layout` -> host NumPy array lead + (N,) + rec_shape (zero-copy view of
    the downloaded buffer for 'soa': a transposed view, as the API docs describe)."""
    h = _transfer.to_host([t])[0]        # (large histories: pinned, pipelined, several host threads -- _transfer.py)
    return host_records(h, layout, lead, rec_shape)


def host_records(h, layout, lead, rec_shape):
    """the downloaded buffer (device order) as the array the API returns: see from_records"""
    rec = int(np.prod(rec_shape)) if rec_shape else 1                      # (explicit sizes: -1 is ambiguous for empty banks)
    if layout == "aos":
        return h.reshape(*h.shape[:lead + 1], *rec_shape) if rec_shape else h
    # [lead][E][N] -> [lead][N][E]
    h = np.swapaxes(h.reshape(*h.shape[:lead], rec, h.shape[-1]), -1, -2)
    return h.reshape(*h.shape[:lead + 1], *rec_shape)


def alloc_records(lead_shape, N, E, layout, device=None):
    device = device or require_gpu()
    shape = (*lead_shape, N, E) if layout == "aos" else (*lead_shape, E, N)
    return torch.empty(shape, dtype=torch.float64, device=device)


def alloc_cov_pair(T, N, n, layout, device=None):
    """Both covariance histories of a batch_filter call in ONE array (FK_KF_FLAG_COV_INTERLEAVED, include/filterhip.h):
    returns (cov2, covs, covs_p) with covs / covs_p strided views -- NumPy order cov2[T][N][2][n*n], element-major
    cov2[T][2][n*n][N].  A step's posterior and prior covariance then leave as one contiguous write front; two separate
    arrays are two fronts, and those interfere when the driver backed both with the same class of physical memory
    (docs/PLACEMENT.md: 5.2 .. 6.9 ms for the same launch)."""
    device = device or require_gpu()
    nn = n * n
    if layout == "aos":
        cov2 = torch.empty((T, N, 2, nn), dtype=torch.float64, device=device)
        return cov2, cov2[:, :, 0], cov2[:, :, 1]
    cov2 = torch.empty((T, 2, nn, N), dtype=torch.float64, device=device)
    return cov2, cov2[:, 0], cov2[:, 1]


def raise_on_status(status, what):
    """Map per-track status bits to the exception the reference would raise."""
    if status is None:
        return
    bad = status.nonzero()
    if bad.numel() == 0:
        return
    first = int(bad[0])
    bits = int(status[first])
    if bits & _abi.FK_STATUS_BAD_WEIGHTS:
        raise ValueError(f"{what}: FK_UKF_FLAG_PAIR_WEIGHTS given, but the weights of a +- pair of sigma points differ")
    if bits & _abi.FK_STATUS_NOT_PD:
        raise np.linalg.LinAlgError(
            f"{what}: matrix not positive definite / singular for {bad.numel()} track(s), first = {first}")
    if bits & _abi.FK_STATUS_NONFINITE:
        raise FloatingPointError(f"{what}: non-finite state for {bad.numel()} track(s), first = {first}")


def kf_batch_filter(desc_kw, F, Q, H, R, z, x, P, *, B=None, u=None, mask=None,
                    means=None, covs=None, means_p=None, covs_p=None, status=None):
    """fk_kf_batch_filter_f64.  desc_kw: n, m, nu, model_mode, N, T, layout, update_first, alpha_sq."""
    _keep = []
    d = fk_kf_desc(**desc_kw)
    rc = _abi.lib().fk_kf_batch_filter_f64(d, _ptr(F), _ptr(Q), _ptr(H), _ptr(R), _ptr(B), _ptr(u),
                                           _ptr(z), _mask_ptr(mask, _keep), _ptr(x), _ptr(P), _ptr(means), _ptr(covs),
                                           _ptr(means_p), _ptr(covs_p), _ptr(status), _stream())
    _abi.check(rc, "fk_kf_batch_filter_f64")


def kf_batch_filter_ex(desc_kw, F, Q, H, R, z, x, P, extras, *, B=None, u=None, mask=None,
                       means=None, covs=None, means_p=None, covs_p=None, status=None):
    """fk_kf_batch_filter_ex_f64: `extras` maps y/K/S/SI/log_likelihood/mahalanobis -> device tensor."""
    _keep = []
    d = fk_kf_desc(**desc_kw)
    ex = _abi.fk_kf_extras(**{k: _ptr(extras.get(k)) for k in ("y", "K", "S", "SI", "log_likelihood", "mahalanobis")})
    rc = _abi.lib().fk_kf_batch_filter_ex_f64(d, _ptr(F), _ptr(Q), _ptr(H), _ptr(R), _ptr(B), _ptr(u), _ptr(z),
                                              _mask_ptr(mask, _keep), _ptr(x), _ptr(P), _ptr(means), _ptr(covs), _ptr(means_p),
                                              _ptr(covs_p), ex, _ptr(status), _stream())
    _abi.check(rc, "fk_kf_batch_filter_ex_f64")


def kf_predict(desc_kw, F, Q, x, P, *, B=None, u=None, status=None):
    d = fk_kf_desc(**desc_kw)
    rc = _abi.lib().fk_kf_predict_f64(d, _ptr(F), _ptr(Q), _ptr(B), _ptr(u), _ptr(x), _ptr(P),
                                      _ptr(status), _stream())
    _abi.check(rc, "fk_kf_predict_f64")


def kf_update(desc_kw, H, R, z, x, P, *, mask=None, y=None, K=None, S=None, SI=None, status=None):
    _keep = []
    d = fk_kf_desc(**desc_kw)
    rc = _abi.lib().fk_kf_update_f64(d, _ptr(H), _ptr(R), _ptr(z), _mask_ptr(mask, _keep), _ptr(x), _ptr(P),
                                     _ptr(y), _ptr(K), _ptr(S), _ptr(SI), _ptr(status), _stream())
    _abi.check(rc, "fk_kf_update_f64")


def kf_rts(desc_kw, F, Q, Xs, Ps, xs, Ps_out, K, Pp, *, convention=0, status=None):
    d = fk_kf_desc(**desc_kw)
    rc = _abi.lib().fk_kf_rts_f64(d, _ptr(F), _ptr(Q), _ptr(Xs), _ptr(Ps), _ptr(xs), _ptr(Ps_out),
                                  _ptr(K), _ptr(Pp), int(convention), _ptr(status), _stream())
    _abi.check(rc, "fk_kf_rts_f64")


def ut_sigma_points(n, N, layout, scale, x, P, sigmas, status=None):
    rc = _abi.lib().fk_ut_sigma_points_f64(n, N, LAYOUTS[layout], float(scale), _ptr(x), _ptr(P),
                                           _ptr(sigmas), _ptr(status), _stream())
    _abi.check(rc, "fk_ut_sigma_points_f64")


def ut_transform(n, k, N, layout, sigmas, Wm, Wc, noise, x_out, P_out):
    rc = _abi.lib().fk_ut_transform_f64(n, k, N, LAYOUTS[layout], _ptr(sigmas), _ptr(Wm), _ptr(Wc),
                                        _ptr(noise), _ptr(x_out), _ptr(P_out), _stream())
    _abi.check(rc, "fk_ut_transform_f64")


def ut_cross_variance(n, m, k, N, layout, x, z, sigmas_f, sigmas_h, Wc, Pxz):
    rc = _abi.lib().fk_ut_cross_variance_f64(n, m, k, N, LAYOUTS[layout], _ptr(x), _ptr(z), _ptr(sigmas_f),
                                             _ptr(sigmas_h), _ptr(Wc), _ptr(Pxz), _stream())
    _abi.check(rc, "fk_ut_cross_variance_f64")


def ut_linear_map(n_in, n_out, k, N, layout, M, sig_in, sig_out):
    """fk_ut_linear_map_f64: out[i] = M in[i] for every sigma point of every track (a linear fx / hx given as a matrix)"""
    rc = _abi.lib().fk_ut_linear_map_f64(n_in, n_out, k, N, LAYOUTS[layout], _ptr(M), _ptr(sig_in), _ptr(sig_out), _stream())
    _abi.check(rc, "fk_ut_linear_map_f64")


def ukf_correct(n, m, N, layout, Pxz, zp, S, z, x, P, K=None, status=None):
    rc = _abi.lib().fk_ukf_correct_f64(n, m, N, LAYOUTS[layout], _ptr(Pxz), _ptr(zp), _ptr(S), _ptr(z), _ptr(x),
                                       _ptr(P), _ptr(K), _ptr(status), _stream())
    _abi.check(rc, "fk_ukf_correct_f64")


def ukf_linear_supported(n, m, paired=False):
    """will fk_ukf_linear_batch_f64 take this size?  The LIBRARY answers (fk_ukf_linear_supported: csrc/ukf_kernels.hip, dim_x <= 9
    one track per lane; 10..16: csrc/ukf_mlg.hip, several lanes per track, for weights equal within every +- pair), so the host
    and the library cannot disagree about the library's A/B switches"""
    return bool(_abi.lib().fk_ukf_linear_supported(int(n), int(m), _abi.FK_UKF_FLAG_PAIR_WEIGHTS if paired else 0, 0))


def ukf_linear_rts_supported(n, paired=False):
    """will fk_ukf_linear_rts_f64 take this dim_x?  (asked of the library like ukf_linear_supported)"""
    return bool(_abi.lib().fk_ukf_linear_supported(int(n), 1, _abi.FK_UKF_FLAG_PAIR_WEIGHTS if paired else 0, 1))


def pair_weights(Wm, Wc, n):
    """True where the sigma-point weights are equal within every +- pair (Wm[1+k] == Wm[1+n+k], likewise Wc): what
    MerweScaledSigmaPoints and JulierSigmaPoints produce (sigma_points.py:180-192, :358-372).  The fused kernels then form the
    unscented-transform sums over the n pairs (FK_UKF_FLAG_PAIR_WEIGHTS, include/filterhip.h); any other weight set keeps the
    reference's index-order sums.  Wm / Wc: NumPy arrays or device tensors (those are downloaded: 2 (2n+1) doubles)."""
    a = Wm.detach().cpu().numpy() if isinstance(Wm, torch.Tensor) else np.asarray(Wm)
    b = Wc.detach().cpu().numpy() if isinstance(Wc, torch.Tensor) else np.asarray(Wc)
    return (a.shape == (2 * n + 1,) and b.shape == (2 * n + 1,) and bool(np.array_equal(a[1:n + 1], a[n + 1:]))
            and bool(np.array_equal(b[1:n + 1], b[n + 1:])))


def _ukf_flags(Wm, Wc, n, paired):
    if paired is None:
        paired = pair_weights(Wm, Wc, n)
    return _abi.FK_UKF_FLAG_PAIR_WEIGHTS if paired else 0


def ukf_linear_batch(n, m, N, T, layout, scale, F, H, Q, R, Wm, Wc, z, x, P, *, mask=None,
                     means=None, covs=None, status=None, paired=None):
    """fk_ukf_linear_batch_f64.  paired: None = look at the weights (pair_weights), True / False = the caller knows."""
    _keep = []
    d = fk_ukf_desc(n=n, m=m, N=N, T=T, layout=LAYOUTS[layout], flags=_ukf_flags(Wm, Wc, n, paired), scale=float(scale))
    rc = _abi.lib().fk_ukf_linear_batch_f64(d, _ptr(F), _ptr(H), _ptr(Q), _ptr(R), _ptr(Wm), _ptr(Wc),
                                            _ptr(z), _mask_ptr(mask, _keep), _ptr(x), _ptr(P), _ptr(means), _ptr(covs),
                                            _ptr(status), _stream())
    _abi.check(rc, "fk_ukf_linear_batch_f64")


def ukf_linear_rts(n, N, T, layout, scale, F, Q, Wm, Wc, Xs, Ps, xs, Ps_out, K=None, status=None, paired=None):
    """fk_ukf_linear_rts_f64: the UKF smoother's whole backward pass for a linear fx, one launch."""
    d = fk_ukf_desc(n=n, m=1, N=N, T=T, layout=LAYOUTS[layout], flags=_ukf_flags(Wm, Wc, n, paired), scale=float(scale))
    rc = _abi.lib().fk_ukf_linear_rts_f64(d, _ptr(F), _ptr(Q), _ptr(Wm), _ptr(Wc), _ptr(Xs), _ptr(Ps), _ptr(xs),
                                          _ptr(Ps_out), _ptr(K), _ptr(status), _stream())
    _abi.check(rc, "fk_ukf_linear_rts_f64")


def kf_steadystate(desc_kw, F, H, K, z, x, *, B=None, u=None, mask=None, means=None, means_p=None, y=None):
    """fk_kf_steadystate_f64 (F None: update only; z None: predict only)."""
    _keep = []
    d = fk_kf_desc(**desc_kw)
    rc = _abi.lib().fk_kf_steadystate_f64(d, _ptr(F), _ptr(H), _ptr(K), _ptr(B), _ptr(u), _ptr(z), _mask_ptr(mask, _keep),
                                          _ptr(x), _ptr(means), _ptr(means_p), _ptr(y), _stream())
    _abi.check(rc, "fk_kf_steadystate_f64")


def kf_update_correlated(desc_kw, H, R, M, z, x, P, *, mask=None, y=None, K=None, S=None, SI=None, status=None):
    _keep = []
    d = fk_kf_desc(**desc_kw)
    rc = _abi.lib().fk_kf_update_correlated_f64(d, _ptr(H), _ptr(R), _ptr(M), _ptr(z), _mask_ptr(mask, _keep), _ptr(x), _ptr(P),
                                                _ptr(y), _ptr(K), _ptr(S), _ptr(SI), _ptr(status), _stream())
    _abi.check(rc, "fk_kf_update_correlated_f64")


def ukf_rts_correct(n, N, layout, Pxb, xb, Pb, xn, Pn, x, P, K=None, status=None):
    rc = _abi.lib().fk_ukf_rts_correct_f64(n, N, LAYOUTS[layout], _ptr(Pxb), _ptr(xb), _ptr(Pb), _ptr(xn), _ptr(Pn),
                                           _ptr(x), _ptr(P), _ptr(K), _ptr(status), _stream())
    _abi.check(rc, "fk_ukf_rts_correct_f64")


def imm_batch(n, m, n_models, N, T, layout, F, Q, H, R, M, z, xs, Ps, mu, *, x_out=None, P_out=None,
              mu_out=None, x_prior_out=None, P_prior_out=None, likelihood_out=None, status=None, phase=0, mmae=False,
              zmask=None, ll0=None, nu=0, B=None, u=None):
    """fk_imm_batch_ex_f64: T x { IMMEstimator.predict(); IMMEstimator.update(z or None) } for N banks."""
    _keep = []
    d = _abi.fk_imm_desc(n=n, m=m, n_models=n_models, layout=LAYOUTS[layout], N=N, T=T, phase=phase, flags=1 if mmae else 0)
    rc = _abi.lib().fk_imm_batch_ex_f64(d, _ptr(F), _ptr(Q), _ptr(H), _ptr(R), _ptr(M), _ptr(z), _mask_ptr(zmask, _keep), _ptr(ll0),
                                            int(nu), _ptr(B), _ptr(u),
                                            _ptr(xs), _ptr(Ps), _ptr(mu), _ptr(x_out), _ptr(P_out), _ptr(mu_out),
                                            _ptr(x_prior_out), _ptr(P_prior_out), _ptr(likelihood_out), _ptr(status), _stream())
    _abi.check(rc, "fk_imm_batch_ex_f64")


def resample_workspace_bytes(Fn, Np):
    return int(_abi.lib().fk_resample_workspace_bytes(Fn, Np))


def _resample(fn_name, Fn, Np, w, u, idx, status):
    nbytes = resample_workspace_bytes(Fn, Np)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=w.device)
    rc = getattr(_abi.lib(), fn_name)(Fn, Np, _ptr(w), _ptr(u), _ptr(idx), _ptr(status), _ptr(ws), nbytes,
                                      _stream())
    _abi.check(rc, fn_name)


def resample_systematic(Fn, Np, w, u, idx, status=None):
    _resample("fk_resample_systematic_f64", Fn, Np, w, u, idx, status)


def resample_stratified(Fn, Np, w, u, idx, status=None):
    _resample("fk_resample_stratified_f64", Fn, Np, w, u, idx, status)


def resample_multinomial(Fn, Np, Nu, w, u, idx):
    nbytes = int(_abi.lib().fk_multinomial_workspace_bytes(Fn, Np))
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=w.device)
    rc = _abi.lib().fk_resample_multinomial_f64(Fn, Np, Nu, _ptr(w), _ptr(u), _ptr(idx), _ptr(ws), nbytes,
                                                _stream())
    _abi.check(rc, "fk_resample_multinomial_f64")


def resample_residual_fill(Fn, Np, w, idx, k, cs, status):
    rc = _abi.lib().fk_resample_residual_fill_f64(Fn, Np, _ptr(w), _ptr(idx), _ptr(k), _ptr(cs), _ptr(status), _stream())
    _abi.check(rc, "fk_resample_residual_fill_f64")


def resample_residual_draw(Fn, Np, cs, k, uoff, u, idx):
    rc = _abi.lib().fk_resample_residual_draw_f64(Fn, Np, _ptr(cs), _ptr(k), _ptr(uoff), _ptr(u), _ptr(idx), _stream())
    _abi.check(rc, "fk_resample_residual_draw_f64")


def resample_gather_mean(Fn, Np, d, particles, idx, mean):
    """fk_resample_gather_mean_f64: mean[f] = particles[f][idx[f]].mean(axis=0) without the resampled copy."""
    rc = _abi.lib().fk_resample_gather_mean_f64(Fn, Np, d, _ptr(particles), _ptr(idx), _ptr(mean), _stream())
    _abi.check(rc, "fk_resample_gather_mean_f64")


def cumsum_exact(Fn, Np, w, cs, force_last_one=False):
    rc = _abi.lib().fk_cumsum_exact_f64(Fn, Np, _ptr(w), _ptr(cs), int(bool(force_last_one)), _stream())
    _abi.check(rc, "fk_cumsum_exact_f64")
