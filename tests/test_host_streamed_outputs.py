"""The host half of the streamed host outputs (kalman_filter.py, _Core.batch: time chunks launched into the same device buffers,
each downloaded into its rows of the result): on the stand-in engine of tests/fake_kf_engine.py the chunked call must return what
the single call returns, array for array -- slices of z / mask / per-epoch models / u, the status ORed over the chunks, the final
state, both record orders.  (The kernels' side of the same statement is tests/test_gpu_api.py.)"""
import numpy as np
import pytest

import fake_kf_engine


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("n,m", [(2, 1), (4, 2), (9, 3)])
def test_bank_streamed_equals_single_call(layout, n, m, monkeypatch):
    from filterpy_amd.kalman import KalmanFilterBank
    fake_kf_engine.install(monkeypatch)
    rs = np.random.RandomState(100 * n + m)
    N, T = 37, 23
    zs = rs.randn(T, N, m)
    mask = (rs.rand(T, N) > 0.2).astype(np.uint8)
    x0 = rs.randn(N, n)
    step = 2 * 8 * (n + n * n) * N
    res = []
    for env in ({"FK_STREAM_OUTPUTS": "0"}, {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": str(5 * step)},
                {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": "1"}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            bank = KalmanFilterBank(n, m, N, layout=layout)
            bank.F = np.eye(n) + 0.05 * np.triu(np.ones((n, n)), 1)
            bank.Q, bank.R, bank.H = 0.02 * np.eye(n), 0.5 * np.eye(m), np.eye(m, n)
            bank.x, bank.P = x0.copy(), np.tile(3.0 * np.eye(n), (N, 1, 1))
            info = {}
            res.append(bank.batch_filter(zs, mask=mask, update_first=bool(m == 2)) + (bank.x, bank.P))
            info = getattr(bank, "placement_info", None)
            if "FK_STREAM_MIN_BYTES" in env:
                assert info and "streamed" in info.get("note", ""), info
    for other in res[1:]:
        for u, v in zip(res[0], other):
            assert u.shape == v.shape and np.array_equal(u, v)


def test_filter_streamed_equals_single_call_with_per_epoch_models(monkeypatch):
    from filterpy_amd.kalman import KalmanFilter
    fake_kf_engine.install(monkeypatch)
    rs = np.random.RandomState(4)
    n, m, T = 4, 2, 31
    Fs = [np.eye(n) + 0.01 * (t + 1) * np.triu(np.ones((n, n)), 1) for t in range(T)]
    Qs = [0.01 * (1 + t % 3) * np.eye(n) for t in range(T)]
    Rs = [(0.2 + 0.01 * t) * np.eye(m) for t in range(T)]
    Bs = [0.1 * (t + 1) * np.ones((n, 1)) for t in range(T)]
    us = [np.array([0.5 * t]) for t in range(T)]
    zl = [None if t in (3, 17, T - 1) else rs.randn(m, 1) for t in range(T)]
    out = []
    for env in ({"FK_STREAM_OUTPUTS": "0"}, {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": str(4 * 2 * 8 * (n + n * n))}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            kf = KalmanFilter(n, m, dim_u=1)
            kf.H, kf.P = np.eye(m, n), 2.0 * np.eye(n)
            out.append(kf.batch_filter(zl, Fs=Fs, Qs=Qs, Rs=Rs, Bs=Bs, us=us) + (kf.x, kf.P))
    for u, v in zip(*out):
        assert np.array_equal(u, v)


def test_a_failing_track_raises_whichever_chunk_it_fails_in(monkeypatch):
    from filterpy_amd.kalman import KalmanFilterBank
    fake_kf_engine.install(monkeypatch)
    n, m, N, T = 2, 1, 5, 12
    zs = np.zeros((T, N, m))
    for env in ({"FK_STREAM_OUTPUTS": "0"}, {"FK_STREAM_MIN_BYTES": "1", "FK_STREAM_CHUNK_BYTES": "1"}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            bank = KalmanFilterBank(n, m, N)
            bank.H, bank.R = np.eye(m, n), 0.0 * np.eye(m)
            P = np.tile(np.eye(n), (N, 1, 1))
            P[3] = -np.eye(n)                              # S = H P H' + R < 0: not positive definite
            bank.P = P
            with pytest.raises(np.linalg.LinAlgError):
                bank.batch_filter(zs)
