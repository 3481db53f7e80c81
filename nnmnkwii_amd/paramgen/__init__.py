from ._mlpg import (  # noqa: F401
    BandMat,
    build_win_mats,
    full_window_mat,
    mlpg,
    mlpg_batch,
    mlpg_grad,
    multi_stream_mlpg,
    reshape_means,
    unit_variance_mlpg_matrix,
)

__all__ = [
    "build_win_mats",
    "mlpg",
    "mlpg_grad",
    "unit_variance_mlpg_matrix",
    "reshape_means",
    "full_window_mat",
    "mlpg_batch",
    "multi_stream_mlpg",
]
