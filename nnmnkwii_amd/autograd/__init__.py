from ._mlpg import (MLPG, UnitVarianceMLPG, UnitVarianceMLPGMSELoss, mlpg, unit_variance_mlpg,  # noqa: F401
                    unit_variance_mlpg_mse_loss)
from ._modspec import ModSpec, modspec  # noqa: F401
