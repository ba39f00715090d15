"""filterpy_amd.kalman -- the Kalman-family part of the hot path (filterpy/kalman/__init__.py:21-33
re-exports everything; here only what the engine implements)."""
from .kalman_filter import (KalmanFilter, KalmanFilterBank, predict, update, batch_filter,  # noqa: F401
                            rts_smoother, predict_steadystate, update_steadystate)
from .sigma_points import MerweScaledSigmaPoints, JulierSigmaPoints  # noqa: F401
from .unscented_transform import unscented_transform  # noqa: F401
from .UKF import UnscentedKalmanFilter  # noqa: F401
from .IMM import IMMEstimator  # noqa: F401
from .mmae import MMAEFilterBank  # noqa: F401
