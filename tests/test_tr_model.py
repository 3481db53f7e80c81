"""CPU tests of tools/tr_model.py -- the executable specification of the strip kernel's transposed form (what a lane is there: the
lane -> (utterance, dim) map, the per-lane offsets from the group's first utterance, the masked assembly with a lengths vector) --
against the oracle.  The GPU counterpart is tests/test_strip_tr_gpu.py."""
import os
import sys

import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import tr_model as TM  # noqa: E402

STD3 = WINDOW_SETS["std3"]


@pytest.mark.parametrize("B,T,sd", [(70, 9, 1), (33, 12, 5), (5, 20, 32), (3, 1, 7), (2, 2, 1), (14, 17, 25), (40, 6, 3)])
@pytest.mark.parametrize("vm", ["frame", "global", "unit"])
def test_lane_map_and_offsets_reproduce_every_utterance(B, T, sd, vm):
    rng = np.random.RandomState(B + 10 * T + sd)
    m = rng.randn(B, T, 3 * sd)
    v = rng.rand(B, T, 3 * sd) + 0.1 if vm == "frame" else (rng.rand(3 * sd) + 0.1 if vm == "global" else None)
    ref, _, rc = O.mlpg_batch(m, np.ones(3 * sd) if v is None else v, STD3)
    assert rc == 0
    out, touched = TM.forward(m, v, STD3)
    assert touched.all()                                  # every output element written exactly once (forward() asserts the "once")
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("B,T,sd", [(70, 9, 1), (33, 12, 5), (14, 17, 25), (40, 6, 3), (9, 30, 2)])
@pytest.mark.parametrize("vm", ["frame", "global", "unit"])
def test_ragged_lengths_with_anything_in_the_padding(B, T, sd, vm):
    """A group runs to its longest utterance; a lane's own dead frames are never read into the system (NaN in the padding), its
    rows beyond its length are identity rows: the live part equals the reference's solve of the utterance alone, the rest is 0."""
    rng = np.random.RandomState(3 * B + T + sd)
    lengths = rng.randint(0, T + 1, size=B)
    lengths[0], lengths[-1] = T, 0
    if B > 4:
        lengths[1], lengths[2], lengths[3] = 1, 2, 3
    m = rng.randn(B, T, 3 * sd)
    v = rng.rand(B, T, 3 * sd) + 0.1 if vm == "frame" else (rng.rand(3 * sd) + 0.1 if vm == "global" else None)
    m_ref = m.copy()
    for b in range(B):
        m_ref[b, lengths[b]:] = 0
    ref, _, rc = O.mlpg_batch(m_ref, np.ones(3 * sd) if v is None else v, STD3, lengths)
    assert rc == 0
    for b in range(B):
        m[b, lengths[b]:] = np.nan
        if vm == "frame":
            v[b, lengths[b]:] = np.nan
    out, touched = TM.forward(m, v, STD3, lengths)
    assert touched.all() and np.isfinite(out).all()
    for b in range(B):
        assert not out[b, lengths[b]:].any()
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


def test_a_piece_of_a_stream_inside_a_wide_batch():
    """What mlpg_hip_forward_streams leaves over: dims [3, 5) of a 5-dim stream (window pitch 5, not 2) that sits at column 183 of
    198 x 3-wide rows, written to columns [64, 66) of a 66-wide output; only those output columns are touched."""
    rng = np.random.RandomState(1)
    B, T = 37, 10
    ld_in, ld_out = 3 * 66, 66
    m = rng.randn(B, T, ld_in)
    v = rng.rand(B, T, ld_in) + 0.1
    bap_col, pitch, d_first, nd = 183, 5, 3, 2
    out, touched = TM.forward(m, v, STD3, in_col=bap_col + d_first, sd=nd, pitch=pitch, out_col=64, ld_out=ld_out)
    cols = [bap_col + w * pitch + d_first + d for w in range(3) for d in range(nd)]
    ref, _, rc = O.mlpg_batch(m[:, :, cols], v[:, :, cols], STD3)
    assert rc == 0
    assert touched[:, :, 64:66].all() and not touched[:, :, :64].any()
    assert np.abs(out[:, :, 64:66] - ref).max() <= 1e-11 * np.abs(ref).max()
    # the same with a global variance vector over the wide row (its columns carry no utterance offset)
    vg = rng.rand(ld_in) + 0.1
    out, _ = TM.forward(m, vg, STD3, in_col=bap_col + d_first, sd=nd, pitch=pitch, out_col=64, ld_out=ld_out)
    ref, _, rc = O.mlpg_batch(m[:, :, cols], vg[cols], STD3)
    assert rc == 0 and np.abs(out[:, :, 64:66] - ref).max() <= 1e-11 * np.abs(ref).max()


def test_the_2_gb_window_rule():
    assert TM.fits(512, 2000, 1, 594, 198) and TM.fits(512, 2000, 5, 594, 198)
    assert not TM.fits(512, 16384, 1, 4096, 4096)          # 64 utterances x 16384 frames x 32 KB rows: beyond the descriptor's window
    assert not TM.fits(1, 100, 1, 3, 1) and not TM.fits(8, 100, 33, 99, 33)
