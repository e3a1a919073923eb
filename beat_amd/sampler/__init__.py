from .base import available_proposals, choose_proposal  # noqa: F401
from .metropolis import BatchedMetropolis, Metropolis  # noqa: F401
from .pt import TemperingManager, pt_sample  # noqa: F401
from .smc import SMC, smc_sample  # noqa: F401
