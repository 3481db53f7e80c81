from ._mlpg import MLPG, UnitVarianceMLPG, mlpg, unit_variance_mlpg  # noqa: F401
