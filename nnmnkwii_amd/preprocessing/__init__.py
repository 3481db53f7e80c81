from .generic import trim_zeros_frames  # noqa: F401
from . import alignment  # noqa: F401
