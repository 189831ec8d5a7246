"""transport_analysis_amd — MI355X-native time-correlation kernels behind the
transport-analysis API (VelocityAutocorr, ViscosityHelfand)."""
__version__ = "0.1.0"

from .velocityautocorr import VelocityAutocorr  # noqa: F401
from .viscosity import ViscosityHelfand  # noqa: F401
