"""util.apply_each2d_padded / apply_each2d_trim (reference util/__init__.py:19-66): the batched fast paths
for paramgen.mlpg and delta_features must equal the per-utterance loop the reference runs."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))
from cases import WINDOW_SETS  # noqa: E402

pytestmark = pytest.mark.gpu
STD3 = WINDOW_SETS["std3"]


def _batch(rng, N=4, T=60, D=6):
    X = rng.randn(N, T, D)
    lengths = np.array([60, 17, 33, 5])
    for n in range(N):
        X[n, lengths[n]:] = 0
    return X, lengths


def test_apply_each2d_padded_mlpg_fast_path_equals_loop_and_oracle():
    from nnmnkwii_amd import paramgen as G
    from nnmnkwii_amd.util import apply_each2d_padded, apply_each2d_trim
    from oracle import mlpg as O
    rng = np.random.RandomState(0)
    X, lengths = _batch(rng)
    var = rng.rand(6) + 0.1
    fast = apply_each2d_padded(G.mlpg, X, lengths, var, STD3)
    loop = apply_each2d_padded(lambda x, v, w: G.mlpg(x, v, w), X, lengths, var, STD3)   # not recognised: per utterance
    assert fast.shape == loop.shape == (4, 60, 2) and fast.dtype == np.float64
    np.testing.assert_array_equal(fast, loop)           # same kernel, same systems: bit for bit
    for n in range(4):
        ref = O.mlpg(X[n, :lengths[n]], var, STD3)
        np.testing.assert_allclose(fast[n, :lengths[n]], ref, rtol=1e-9, atol=1e-12)
        assert not fast[n, lengths[n]:].any()
    np.testing.assert_array_equal(apply_each2d_trim(G.mlpg, X, var, STD3), fast)


def test_apply_each2d_delta_features_and_generic_callable():
    from nnmnkwii_amd.preprocessing import delta_features
    from nnmnkwii_amd.util import apply_each2d_padded, apply_each2d_trim
    from oracle import mlpg as O
    rng = np.random.RandomState(1)
    X, lengths = _batch(rng, D=3)
    fast = apply_each2d_padded(delta_features, X, lengths, STD3)
    assert fast.shape == (4, 60, 9)
    for n in range(4):
        ref = O.delta_features(X[n, :lengths[n]], STD3)
        np.testing.assert_allclose(fast[n, :lengths[n]], ref, rtol=1e-12, atol=1e-14)
        assert not fast[n, lengths[n]:].any()
    np.testing.assert_array_equal(apply_each2d_trim(delta_features, X, STD3), fast)
    # an arbitrary callable runs per utterance, like the reference
    Y = apply_each2d_padded(lambda x, k: np.cumsum(x, axis=0)[:, :k], X, lengths, 2)
    assert Y.shape == (4, 60, 2)
    np.testing.assert_allclose(Y[1, :17], np.cumsum(X[1, :17], axis=0)[:, :2])
    assert not Y[1, 17:].any()
