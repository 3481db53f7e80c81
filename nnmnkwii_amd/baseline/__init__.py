"""GMM-based voice-conversion baseline: the direct caller of paramgen.mlpg and of the DTW
aligner.  Mirrors /root/reference/nnmnkwii/baseline/gmm.py."""
from . import gmm  # noqa: F401
