"""CPU: the numpy model of the FIR form of unit-variance MLPG (tools/fir_model.py, the executable specification of csrc/mlpg_fir.hip:
a 49-tap filter on the right-hand side in the interior, 24 table rows per end, the table from one reference solve on 160 frames)
against the oracle -- the truncation the kernel relies on, checked in float64 where nothing else contributes."""
import os
import sys

import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fir_model as FM  # noqa: E402

TRUNC = 2.0 ** -22      # relative to the trajectory's scale: 49 terms of at most 2^-26 each and the table's own tolerance


@pytest.mark.parametrize("wname", ["std3", "std2", "asym2", "wide3"])
@pytest.mark.parametrize("T", [48, 49, 95, 96, 97, 128, 161, 500])
def test_fir_model_forward_vs_oracle(wname, T):
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    rng = np.random.RandomState(T + nw)
    m = rng.randn(T, nw * 3)
    yo = O.mlpg(m, np.ones((T, nw * 3)), windows)
    y = FM.forward(m, windows, T)
    assert np.abs(y - yo).max() <= TRUNC * np.abs(yo).max(), (wname, T)


@pytest.mark.parametrize("wname", ["std3", "asym2", "wide3"])
@pytest.mark.parametrize("T", [96, 130, 333])
def test_fir_model_backward_vs_oracle_gradient(wname, T):
    """R^T g with R = P^-1 W~^T built densely from the oracle's window matrices (what the reference's backward multiplies by)."""
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    rng = np.random.RandomState(T)
    go = rng.randn(T, 2)
    mask = O._edge_mask(T, mw)
    Ws = [O.window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for (l, u, c) in windows]
    Wt = [W if w == 0 else mask[:, None] * W for w, W in enumerate(Ws)]
    P = sum(Wt[w].T @ Ws[w] for w in range(nw))
    z = np.linalg.solve(P, go)
    ref = np.concatenate([Wt[w] @ z for w in range(nw)], axis=1)
    g = FM.backward(go, windows, T)
    assert np.abs(g - ref).max() <= TRUNC * np.abs(ref).max(), (wname, T)


def test_fir_model_table_is_refused_when_the_inverse_does_not_decay():
    """A static window of weight 1e-3 against a second-difference window: P^-1 is smooth over hundreds of frames."""
    windows = [(0, 0, np.array([1e-3])), (1, 1, np.array([1.0, -2.0, 1.0]))]
    _, ok = FM.build_taps(windows)
    assert not ok
    for wname in ("std3", "std2", "asym2", "wide3"):
        assert FM.build_taps(WINDOW_SETS[wname])[1], wname
