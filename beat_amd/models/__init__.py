from .problem import (FFIProblem, GeodeticData, LogpForwFunc, ParameterLayout,  # noqa: F401
                      SeismicWavemap, prior_logp_func)
