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


class Saver(object):
    """History recorder with the interface of filterpy.common.Saver (filterpy/common/helpers.py:27-219): every
    ``save()`` appends a deep copy of each attribute of the watched filter -- and the value of each of its
    properties, which forces the lazily evaluated ones (log_likelihood, likelihood, mahalanobis) -- to a list
    of the same name, readable as ``saver.x``, ``saver['x']`` or, after ``to_array()``, as one ndarray per name.

    ``KalmanFilter.batch_filter(zs, saver=s)`` runs the whole sequence in one launch and then replays the
    per-epoch attributes into the filter object, calling ``s.save()`` once per epoch (kalman_filter.py:990-991),
    so the histories are the ones the reference's epoch-by-epoch loop records."""

    def __init__(self, kf, save_current=False, skip_private=False, skip_callable=False, ignore=()):
        import inspect
        from collections import defaultdict
        self._kf = kf
        self._DL = defaultdict(list)
        self._skip_private = skip_private
        self._skip_callable = skip_callable
        self._ignore = ignore
        self._len = 0
        # (name, property object) pairs like the reference keeps them: callers index [0]
        self.properties = [(name, prop) for name, prop in
                           inspect.getmembers(type(kf), lambda member: isinstance(member, property))
                           if name not in ignore]
        if save_current:
            self.save()

    def _wanted(self, name, value):
        if self._skip_private and name.startswith("_"):
            return False
        if self._skip_callable and callable(value):
            return False
        return name not in self._ignore

    def save(self):
        import copy
        kf = self._kf
        for name, _ in self.properties:                     # properties first: their lists lead the key order
            self._DL[name].append(getattr(kf, name))
        for name, value in copy.deepcopy(kf.__dict__).items():
            if self._wanted(name, value):
                self._DL[name].append(value)
        self.__dict__.update(self._DL)
        self._len += 1

    def __getitem__(self, key):
        return self._DL[key]

    def __setitem__(self, key, newvalue):
        self._DL[key] = newvalue
        self.__dict__.update(self._DL)

    def __len__(self):
        return self._len

    @property
    def keys(self):
        return list(self._DL.keys())

    def to_array(self, flatten=False):
        """Every history becomes one ndarray (the lists stay available through ``[]``).  Raises ValueError,
        leaving the attributes as lists, if some attribute changed shape between epochs."""
        for key in self.keys:
            try:
                self.__dict__[key] = np.array(self._DL[key])
            except Exception:
                self.__dict__.update(self._DL)
                raise ValueError("could not convert {} into np.array".format(key)) from None
        if flatten:
            self.flatten()

    def flatten(self):
        """(T, n, 1) histories of column vectors become (T, n) -- and (T,) when n = 1.  Histories that are not
        at least 3-D arrays are left alone, like in the reference (a (T, 1) history stays (T, 1)).  One way."""
        for key in self.keys:
            arr = self.__dict__[key]
            if not isinstance(arr, np.ndarray) or arr.ndim < 3 or arr.shape[2] != 1:
                continue
            if arr.size != arr.shape[0] * arr.shape[1]:       # (T, n, 1, k...): not a column-vector history
                continue
            arr = arr.reshape(arr.shape[0], arr.shape[1])
            self.__dict__[key] = arr.ravel() if arr.shape[1] == 1 else arr

    def __repr__(self):
        return "<Saver object at {}\n  Keys: {}>".format(hex(id(self)), " ".join(self.keys))
