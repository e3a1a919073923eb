from .problem import (FFIProblem, GeodeticData, LogpForwFunc, ParameterLayout,  # noqa: F401
                      SeismicWavemap, prior_logp_func)
from .distributions import get_hyper_name, multivariate_normal_chol  # noqa: F401,E402
from .geometry import GeodeticGeometryProblem, los_vectors  # noqa: F401,E402
