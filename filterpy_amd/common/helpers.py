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
        self._watched = kf
        self._hist = {}                      # name -> list of per-epoch values, in first-seen order
        self._epochs = 0
        self._drop_private, self._drop_callable, self._ignored = skip_private, skip_callable, tuple(ignore)
        # (name, property object) pairs, the form callers of the reference index with [0]
        self.properties = [pair for pair in inspect.getmembers(type(kf), lambda member: isinstance(member, property))
                           if pair[0] not in self._ignored]
        if save_current:
            self.save()

    def _keep(self, name, value):
        return not ((self._drop_private and name.startswith("_")) or (self._drop_callable and callable(value))
                    or name in self._ignored)

    def _publish(self):
        """the histories are also attributes of the saver (lists until to_array())"""
        self.__dict__.update(self._hist)

    def save(self):
        import copy
        snapshot = [(name, getattr(self._watched, name)) for name, _ in self.properties]   # properties lead the key order
        snapshot += [(name, value) for name, value in copy.deepcopy(self._watched.__dict__).items()
                     if self._keep(name, value)]
        for name, value in snapshot:
            self._hist.setdefault(name, []).append(value)
        self._publish()
        self._epochs += 1

    def __getitem__(self, key):
        return self._hist.setdefault(key, [])           # an unknown key reads as an empty history

    def __setitem__(self, key, newvalue):
        self._hist[key] = newvalue
        self._publish()

    def __len__(self):
        return self._epochs

    @property
    def keys(self):
        return list(self._hist)

    def to_array(self, flatten=False):
        """Every history becomes one ndarray (the lists stay available through ``[]``).  Raises ValueError,
        leaving the attributes as lists, if some attribute changed shape between epochs."""
        arrays = {}
        for name, values in self._hist.items():
            try:
                arrays[name] = np.array(values)
            except Exception:
                self._publish()
                raise ValueError("could not convert {} into np.array".format(name)) from None
        self.__dict__.update(arrays)
        if flatten:
            self.flatten()

    def flatten(self):
        """(T, n, 1) histories of column vectors become (T, n) -- and (T,) when n = 1.  Histories that are not
        at least 3-D arrays are left alone, like in the reference (a (T, 1) history stays (T, 1)).  One way."""
        for name in self._hist:
            arr = self.__dict__[name]
            if not isinstance(arr, np.ndarray) or arr.ndim < 3 or arr.shape[2] != 1:
                continue
            if arr.size != arr.shape[0] * arr.shape[1]:       # (T, n, 1, k...): not a column-vector history
                continue
            arr = arr.reshape(arr.shape[0], arr.shape[1])
            self.__dict__[name] = arr.ravel() if arr.shape[1] == 1 else arr

    def __repr__(self):
        return "<Saver object at {}\n  Keys: {}>".format(hex(id(self)), " ".join(self.keys))
