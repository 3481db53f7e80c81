from ._mlpg import MLPG, UnitVarianceMLPG, mlpg, unit_variance_mlpg  # noqa: F401
from ._modspec import ModSpec, modspec  # noqa: F401
