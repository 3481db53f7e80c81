from .generic import delta_features, trim_zeros_frames  # noqa: F401
from . import alignment  # noqa: F401
