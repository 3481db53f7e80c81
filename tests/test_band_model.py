"""The numpy blueprint of a strip scheme for window extents up to 2 (tools/band_model.py) against the oracle, on the CPU:
the three-level elimination with separators of q frames and q x q blocks is algebraically the solve the reference does, for
the shipped case (q = 2) and for the reference's 5-tap test windows (q = 4)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
from cases import WINDOW_SETS  # noqa: E402

import band_model as BM  # noqa: E402
from oracle import mlpg as O  # noqa: E402


@pytest.mark.parametrize("wname", ["std3", "std2", "wide3", "asym2"])
@pytest.mark.parametrize("T", [1, 5, 16, 17, 63, 64, 65, 130, 257])
def test_band_model_vs_oracle(wname, T):
    rng = np.random.RandomState(T * 7 + len(wname))
    win = WINDOW_SETS[wname]
    sd = 3
    m = rng.randn(T, len(win) * sd)
    v = rng.rand(T, len(win) * sd) + 0.1
    ref = O.mlpg(m, v, win)
    got = BM.mlpg_model(m, v, win)
    assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,W", [(12, 4), (16, 2), (24, 8)])
def test_band_model_chunk_and_strip_shapes(M, W):
    rng = np.random.RandomState(M + W)
    win = WINDOW_SETS["wide3"]
    T, sd = 200, 2
    m = rng.randn(T, 3 * sd)
    v = np.exp(rng.randn(T, 3 * sd))          # log-normal variances: long-range coupling
    ref = O.mlpg(m, v, win)
    got = BM.mlpg_model(m, v, win, M=M, W=W)
    assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())


def test_record_sizes():
    assert BM.record_doubles(2) == 14      # = kRec of the shipped strip kernel (mlpg_strip_impl.h)
    assert BM.record_doubles(4) == 44
