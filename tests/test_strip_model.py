"""CPU: the numpy model of the strip kernel's three-level elimination (tools/strip_model.py, the
executable specification the HIP code in csrc/mlpg_strip.hip transliterates) against the oracle."""
import os
import sys

import numpy as np
import pytest

from cases import WINDOW_SETS, rand_case
from oracle import mlpg as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import strip_model as S  # noqa: E402


def rel_err(y, ref):
    scale = np.abs(ref).max(axis=0, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float((np.abs(y - ref) / scale).max())


@pytest.mark.parametrize("wname", ["std3", "std2", "static", "asym2", "zero2"])
@pytest.mark.parametrize("T", [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 66, 127, 130, 200, 257, 1000])
def test_model_vs_oracle(wname, T):
    m, v, vg = rand_case(wname, "f64", T, 3, salt=7)
    for W in (1, 2, 4):
        if T > 300 and W != 4:
            continue
        y, bad = S.mlpg_strip(m, v, WINDOW_SETS[wname], W=W)
        assert not bad.any()
        assert rel_err(y, O.mlpg(m, v, WINDOW_SETS[wname])) < 1e-11, (wname, T, W)
    y, _ = S.mlpg_strip(m, vg, WINDOW_SETS[wname], W=4)
    assert rel_err(y, O.mlpg(m, vg, WINDOW_SETS[wname])) < 1e-11


def test_model_one_sided_and_two_sided_level3_agree():
    m, v, _ = rand_case("std3", "f64", 1000, 2, salt=9)
    y1, _ = S.mlpg_strip(m, v, WINDOW_SETS["std3"], two_sided=False)
    y2, _ = S.mlpg_strip(m, v, WINDOW_SETS["std3"], two_sided=True)
    assert rel_err(y1, y2) < 1e-12
    assert rel_err(y2, O.mlpg(m, v, WINDOW_SETS["std3"])) < 1e-11


def test_model_ragged_length():
    m, v, _ = rand_case("std3", "f64", 150, 2, salt=1)
    for T in (149, 128, 113, 112, 97, 64, 50, 3):
        y, _ = S.mlpg_strip(m, v, WINDOW_SETS["std3"], W=4, T=T)
        yo = O.mlpg(m[:T], v[:T], WINDOW_SETS["std3"])
        assert rel_err(y[:T], yo) < 1e-11
        assert not y[T:].any()


def test_model_flags_indefinite():
    m, v, _ = rand_case("std3", "f64", 100, 2, salt=2)
    v = v.copy()
    v[40, 0] = -0.05
    _, bad = S.mlpg_strip(m, v, WINDOW_SETS["std3"])
    assert bad[0] and not bad[1]


def _long_range_case(T=1000, sd=2, seed=5):
    """Static variances of 1e14 over frames 200 .. 800: there only the dynamic features tie the trajectory down,
    so strips far apart stay coupled."""
    rng = np.random.RandomState(seed)
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.1
    v[200:800, :sd] = 1e14
    return m, v


def test_model_local_window_level3():
    """Level 3 restricted to strips r-2 .. r+2: accepted wherever the damping bound is below 1e-22, with the same
    numbers as the sweep over the whole utterance; weakly damped strips are recognised and fall back."""
    W3 = WINDOW_SETS["std3"]
    m, v, _ = rand_case("std3", "f64", 1000, 2, salt=11)
    stats = []
    y, bad = S.mlpg_strip(m, v, W3, local_k=2, stats=stats)
    assert not bad.any()
    assert all(d < 1e-22 for d, _ in stats)                 # every strip accepts the window
    assert max(e for _, e in stats) < 1e-13
    assert rel_err(y, O.mlpg(m, v, W3)) < 1e-11
    m, v = _long_range_case()
    stats = []
    y, bad = S.mlpg_strip(m, v, W3, local_k=2, stats=stats)
    assert not bad.any()
    rejected = [d >= 1e-22 for d, _ in stats]
    assert any(rejected) and not all(rejected)
    yf, _ = S.mlpg_strip(m, v, W3, local_k=0)
    assert rel_err(y, yf) < 1e-12                           # accepted windows changed nothing
    # the bound is what protects the result: forcing the window everywhere is visibly wrong
    yw, _ = S.mlpg_strip(m, v, W3, local_k=2, local_tol=np.inf)
    assert rel_err(yw, yf) > 1e-6


def test_model_wider_windows_for_tighter_dynamic_variances():
    """Dynamic variances 10 x / 100 x tighter than the static ones: the 5-strip window's bound (~1e-13) is not below
    1e-22, the 9-strip window's (~1e-27) is, and its result equals the full sweep's."""
    W3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(4)
    T, sd = 1100, 3
    m = rng.randn(T, 3 * sd)
    v = np.concatenate([rng.rand(T, sd) + 0.1, 0.1 * (rng.rand(T, sd) + 0.1), 0.01 * (rng.rand(T, sd) + 0.1)], axis=1)
    s2, s4 = [], []
    S.mlpg_strip(m, v, W3, local_k=2, stats=s2)
    y4, bad = S.mlpg_strip(m, v, W3, local_k=4, stats=s4)
    assert not bad.any()
    assert max(d for d, _ in s2) > 1e-22 and max(d for d, _ in s4) < 1e-22
    assert max(e for _, e in s4) < 1e-13
    assert rel_err(y4, O.mlpg(m, v, W3)) < 1e-10


def test_three_strip_window_and_the_ladder():
    W3 = WINDOW_SETS["std3"]
    """Round 5: the 3-strip level-3 window.  On the bench's data (variances of one order of magnitude) its rigorous bound sits
    around 1e-22 -- on either side of the 5-strip window's acceptance bound -- while its result equals the exact solve; with
    the kernel's bound of 2^-66 every strip accepts it.  With dynamic variances 10 x tighter it is rejected and the ladder
    goes on to the 5-strip window (or to the exact solve): the trajectory matches the oracle either way."""
    rng = np.random.RandomState(5)
    T, sd = 700, 7
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.1
    ref = O.mlpg(m, v, W3)
    stats = []
    y, bad = S.mlpg_strip(m, v, W3, local_k=(1, 2), local_tol=(2.0 ** -66, 1e-22), stats=stats)
    assert not bad.any() and np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    d1 = [d for d, e, k in stats if k == 1]
    assert len(d1) == len(stats) and max(d1) < 2.0 ** -66          # every strip took the 3-strip window ...
    assert 1e-25 < np.median(d1) < 1e-20                            # ... whose bound is nowhere near 1e-47 (the 5-strip one's)
    assert max(e for d, e, k in stats) <= 1e-15                     # and the separator values are the exact solve's
    v2 = v.copy()
    v2[:, sd:] *= 0.1                                               # dynamic features 10 x tighter: slower decay
    ref2 = O.mlpg(m, v2, W3)
    stats = []
    y2, bad = S.mlpg_strip(m, v2, W3, local_k=(1, 2), local_tol=(2.0 ** -66, 1e-22), stats=stats)
    assert not bad.any() and np.abs(y2 - ref2).max() <= 1e-12 * np.abs(ref2).max()
    assert any(k == 2 for d, e, k in stats)                         # the ladder was used
