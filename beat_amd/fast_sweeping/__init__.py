from . import fast_sweep, fast_sweep_ext  # noqa: F401
