from .helpers import reshape_z, logpdf  # noqa: F401
