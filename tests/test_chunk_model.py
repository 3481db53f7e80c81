"""CPU: the numpy model of the chunked kernel (tools/chunk_model.py, the executable specification of csrc/mlpg_chunk_impl.h:
interior rows of a chunk as the only pivots, the left-coupling columns riding along, block-tridiagonal solve over the
separators, second elimination + back-substitution) against the oracle."""
import os
import sys

import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import chunk_model as CM  # noqa: E402


@pytest.mark.parametrize("wname", ["wide3", "std3", "asym2", "static"])
@pytest.mark.parametrize("T", [1, 2, 3, 5, 19, 20, 21, 41, 97])
def test_chunk_model_vs_oracle(wname, T):
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    rng = np.random.RandomState(T)
    m = rng.randn(T, nw * 2)
    v = rng.rand(T, nw * 2) + 0.1
    yo = O.mlpg(m, v, windows)
    for I in (4, 16):
        if I < 2 * mw:
            continue
        y = CM.mlpg_model(m, v, windows, I=I)
        assert np.abs(y - yo).max() <= 1e-10 * max(1e-300, np.abs(yo).max()), (wname, T, I)
