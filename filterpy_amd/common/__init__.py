from .helpers import reshape_z, logpdf, Saver  # noqa: F401
