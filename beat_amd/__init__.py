"""
beat_amd -- MI355X (gfx950) forward-model + likelihood engine for BEAT's SMC/PT inner
loop: fast-sweeping rupture times, Green's-function stacking and the covariance-weighted
Gaussian log-likelihood as hand-written HIP kernels behind a C ABI
(``include/beat_amd.h``), with Python adaptors that keep the reference's call
signatures (``beat.ffi``, ``beat.pytensorf.Sweeper``, ``beat.models.distributions``,
``beat.sampler``).
"""
from ._lib import BeatAmdError, LIB_PATH  # noqa: F401
from .engine import Context, get_context  # noqa: F401

__version__ = "0.1.0"
