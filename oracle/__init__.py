"""CPU oracle for the filterpy hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in NumPy (and plain C for the resampling loops), the
arithmetic of the reference's hot path (rlabbe/filterpy v1.4.5) in the
reference's own operation order.  It exists to *check* the HIP kernels and to
be *timed* as the CPU baseline.  It is never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * nothing under ``filterpy_amd/`` imports it (tests/test_host_logic.py::test_product_never_imports_oracle
    enforces that).

Parity pin: every function here is checked against outputs of the live
reference (imported from /root/reference in the build container) that were
frozen into ``tests/golden/*.npz`` by ``tests/golden/make_goldens.py``.

Third-party arithmetic the reference delegates to (source not under
/root/reference; versions unpinned by the reference, pinned here to what the
goldens were generated with: NumPy 2.2.6, SciPy 1.15.3):
  numpy.dot (BLAS), numpy.linalg.inv (LAPACK dgesv), scipy.linalg.cholesky
  (LAPACK dpotrf, upper), numpy.cumsum (sequential fp64 adds),
  numpy.searchsorted, numpy.random (global MT19937).
The NumPy oracle calls the same routines; the C oracle (resample_oracle.c)
restates cumsum / the two-pointer merge / searchsorted as plain loops.
"""
