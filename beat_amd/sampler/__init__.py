from .base import multivariate_proposals, step_tune  # noqa: F401
from .metropolis import BatchedMetropolis  # noqa: F401
from .pt import TemperingManager, pt_sample  # noqa: F401
from .smc import SMC, smc_sample  # noqa: F401
