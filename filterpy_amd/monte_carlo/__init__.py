"""filterpy_amd.monte_carlo -- resampling (filterpy/monte_carlo/__init__.py:22-24)."""
from .resampling import (residual_resample, stratified_resample, systematic_resample,  # noqa: F401
                         multinomial_resample)
