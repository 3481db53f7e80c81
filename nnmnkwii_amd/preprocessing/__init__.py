from .generic import delta_features, trim_zeros_frames  # noqa: F401
from . import alignment  # noqa: F401
from .modspec import inv_modspec, modphase, modspec, modspec_smoothing  # noqa: F401
