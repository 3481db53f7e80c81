"""nnmnkwii_amd -- MI355X-native MLPG parameter generation and DTW alignment
behind the Python signatures of r9y9/nnmnkwii's hot path.

    nnmnkwii_amd.paramgen       mlpg, mlpg_grad, unit_variance_mlpg_matrix, reshape_means, ...
    nnmnkwii_amd.autograd       mlpg / MLPG, unit_variance_mlpg / UnitVarianceMLPG (torch, ROCm)
    nnmnkwii_amd.preprocessing  trim_zeros_frames, delta_features, alignment.DTWAligner / IterativeDTWAligner,
                                modspec / inv_modspec / modspec_smoothing
    nnmnkwii_amd.baseline.gmm   MLPGBase, MLPG (GMM voice-conversion baseline; caller of paramgen.mlpg)

Everything numerical runs in hand-written HIP kernels (``csrc/``) behind the
C ABI of ``include/mlpg_hip.h``.  There is no CPU fallback: without the built
extension or without a GPU the calls raise ``HipExtensionError``.
"""
from ._hip import HipExtensionError  # noqa: F401

__version__ = "0.1.0"
