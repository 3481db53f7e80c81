"""A few seconds of the random soaks (tools/dbg/dtw_soak.py, tools/dbg/mlpg_soak.py) on every GPU test run: the fastdtw kernel
against the C oracle, merged multi-stream launches against per-stream dense launches, the fused unit-variance step against the
forward + backward launches -- random shapes, fixed seeds (the tools' own runs cover other seeds for minutes at a time:
profiles/r03_notes.md sections 11 and 12)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "dbg"))


def test_fastdtw_random_soak():
    import dtw_soak
    batches, checked, bad = dtw_soak.soak(4.0, seed=7)
    assert bad is None, bad
    assert checked > 100


def test_mlpg_streams_and_fused_step_random_soak():
    import mlpg_soak
    n_streams, n_fused, bad = mlpg_soak.soak(5.0, seed=8)[:3]
    assert bad is None, bad
    assert n_streams + n_fused > 50
