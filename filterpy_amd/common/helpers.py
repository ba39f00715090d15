"""Host-side helpers that are part of the drop-in boundary (filterpy/common/helpers.py,
filterpy/stats/stats.py): measurement shape rules and the lazy likelihood."""
import numpy as np


def reshape_z(z, dim_z, ndim):
    """Bring z to the shape the filter state uses (filterpy/common/helpers.py:324-342):
    (dim_z, 1) for a column-vector state, (dim_z,) for a 1-D state, scalar for ndim 0.
    Raises ValueError when z cannot be read as dim_z values."""
    z = np.atleast_2d(z)
    if z.shape[1] == dim_z:
        z = z.T
    if z.shape != (dim_z, 1):
        raise ValueError("z (shape {}) must be convertible to shape ({}, 1)".format(z.shape, dim_z))
    if ndim == 1:
        z = z[:, 0]
    if ndim == 0:
        z = z[0, 0]
    return z


def logpdf(x, mean=None, cov=1, allow_singular=True):
    """Log-density of N(mean, cov) at x (filterpy/stats/stats.py:131-154: a thin wrapper over
    scipy.stats.multivariate_normal.logpdf on flattened inputs)."""
    from scipy.stats import multivariate_normal
    flat_mean = None if mean is None else np.asarray(mean).flatten()
    flat_x = np.asarray(x).flatten()
    return multivariate_normal.logpdf(flat_x, flat_mean, cov, allow_singular)
