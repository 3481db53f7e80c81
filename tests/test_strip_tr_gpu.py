"""GPU parity tests (-m gpu) of the strip kernel's TRANSPOSED form (round 5): a narrow stream -- 1 .. 32 static dims -- whose lanes run
over 64 / sd consecutive utterances x its dims (csrc/common.h StreamMap::tr_u; forward, three windows of extent <= 1; per-frame,
global (D,) or unit variances; with or without a lengths vector).  Through the C ABI against the CPU oracle; MLPG_HIP_ALGO_STRIP takes the form wherever it
applies, MLPG_HIP_ALGO_AUTO where it is preferred (launch counter kind 9)."""
import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

pytestmark = pytest.mark.gpu
KIND_TR = 9


def _count():
    from nnmnkwii_amd import _hip
    return int(_hip.lib().mlpg_hip_launch_count(KIND_TR))


@pytest.mark.parametrize("B,T,sd", [(70, 700, 1), (33, 300, 5), (64, 1100, 2), (10, 130, 25), (5, 64, 32), (3, 1, 7), (2, 2, 1),
                                     (130, 65, 3), (9, 2049, 16), (40, 500, 21)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_transposed_form_against_the_oracle(B, T, sd, dt):
    """Every lane group shape: one dim per utterance (64 utterances a group), dims that do not divide 64 (idle lanes), a last block
    with fewer utterances than a group holds, one frame, one strip, many strips."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(B * 1000 + T + sd)
    m = rng.randn(B, T, 3 * sd).astype(dt)
    v = (rng.rand(B, T, 3 * sd) + 0.1).astype(dt)
    ref, _, rc = O.mlpg_batch(m, v, STD3)
    assert rc == 0
    mg, vg = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda()
    n0 = _count()
    out, st = _hip.forward(mg, vg, STD3, algo=_hip.ALGO_STRIP)
    assert _count() == n0 + 1
    out2, _ = _hip.forward(mg, vg, STD3, algo=_hip.ALGO_STRIP)
    assert int(st.abs().max()) == 0 and torch.equal(out, out2)
    out = out.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert (np.abs(out - ref) / scale).max() <= (1e-9 if dt == np.float64 else 5e-6)
    # a lengths vector of full lengths: the same form (per-lane frame counts), the same numbers
    L = torch.full((B,), T, dtype=torch.int32, device="cuda")
    out3, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    assert _count() == n0 + 3
    assert float((out3.double().cpu() - torch.from_numpy(out)).abs().max()) <= (1e-9 if dt == np.float64 else 5e-6) * float(np.abs(ref).max())


@pytest.mark.parametrize("B,T,sd", [(70, 700, 1), (33, 300, 5), (64, 1100, 2), (10, 130, 25), (130, 65, 3), (9, 2049, 16), (7, 3, 2)])
@pytest.mark.parametrize("vm", ["frame", "global", "unit"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_transposed_form_with_ragged_lengths(B, T, sd, vm, dt):
    """A lengths vector: the lanes of a wavefront belong to utterances of different lengths.  The group runs to its longest
    utterance; a lane's own dead frames enter by per-lane selects, so whatever the padding holds (NaN here) never reaches a
    live frame; padding rows of the output are zero.  Lengths 0, 1, 2 and T included."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(B * 100 + T + sd + 3)
    lengths = rng.randint(0, T + 1, size=B).astype(np.int32)
    lengths[rng.randint(B)] = T
    for k, val in enumerate((0, 1, 2, T - 1, 3)):
        if k < B - 1:
            lengths[(k * 7 + 1) % B] = max(0, min(T, val))
    m = rng.randn(B, T, 3 * sd).astype(dt)
    v = (rng.rand(B, T, 3 * sd) + 0.1).astype(dt) if vm == "frame" else ((rng.rand(3 * sd) + 0.1).astype(dt) if vm == "global" else np.ones(3 * sd, dtype=dt))
    m_ref = m.copy()
    for b in range(B):
        m_ref[b, lengths[b]:] = 0
    ref, _, rc = O.mlpg_batch(m_ref, np.where(np.isfinite(v), v, 1.0) if vm != "frame" else v, STD3, lengths)
    assert rc == 0
    for b in range(B):                    # padding: NaN in the means and (per-frame mode) in the variances
        m[b, lengths[b]:] = np.nan
        if vm == "frame":
            v[b, lengths[b]:] = np.nan
    mg = torch.from_numpy(m).cuda()
    vg = None if vm == "unit" else torch.from_numpy(v).cuda()
    L = torch.from_numpy(lengths).cuda()
    n0 = _count()
    out, st = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    assert _count() == n0 + 1 and int(st.abs().max()) == 0
    out2, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    assert torch.equal(out, out2)
    out = out.cpu().numpy().astype(np.float64)
    assert np.isfinite(out).all()
    for b in range(B):
        assert not out[b, lengths[b]:].any(), b
    scale = np.abs(ref).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert (np.abs(out - ref) / scale).max() <= (1e-9 if dt == np.float64 else 5e-6)


def test_transposed_form_ragged_failing_pivots():
    """Negative variances in live frames fail their systems, in the padding they do not: status = the natural-order kernel's."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(19)
    B, T, sd = 90, 500, 4
    lengths = rng.randint(50, T + 1, size=B).astype(np.int32)
    m = torch.from_numpy(rng.randn(B, T, 3 * sd)).cuda()
    v = torch.from_numpy(rng.rand(B, T, 3 * sd) + 0.1).cuda()
    for b in range(0, B, 7):
        v[b, int(lengths[b]) - 3, b % sd] = -1e-3                    # live: fails
        if lengths[b] < T:
            v[b + 1, int(lengths[b + 1]):, (b + 1) % sd] = -1e-3       # padding only: no failure
    L = torch.from_numpy(lengths).cuda()
    out, st = _hip.forward(m, v, STD3, L, algo=_hip.ALGO_STRIP)
    ref, st_ref = _hip.forward(m, v, STD3, L, algo=_hip.ALGO_GENERIC)
    assert torch.equal(st, st_ref) and int((st != 0).sum()) == len(range(0, B, 7))
    bad = (st.view(B, sd) != 0)[:, None, :].expand_as(out)
    assert not bool((out != 0)[bad].any())
    assert float((out - ref)[~bad].abs().max()) <= 1e-9 * float(ref[~bad].abs().max())


@pytest.mark.parametrize("B,T,sd", [(70, 700, 1), (33, 300, 5), (64, 1100, 2), (10, 130, 25), (3, 1, 7), (130, 65, 3)])
@pytest.mark.parametrize("vm", ["global", "unit"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_transposed_form_with_global_and_unit_variances(B, T, sd, vm, dt):
    """The same lane map with a global (D,) variance vector (no utterance offset in ITS columns) and with unit variances."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(B * 1000 + T + sd + 7)
    m = rng.randn(B, T, 3 * sd).astype(dt)
    v = (rng.rand(3 * sd) + 0.1).astype(dt) if vm == "global" else np.ones(3 * sd, dtype=dt)
    ref, _, rc = O.mlpg_batch(m, v, STD3)
    assert rc == 0
    mg = torch.from_numpy(m).cuda()
    vg = torch.from_numpy(v).cuda() if vm == "global" else None
    n0 = _count()
    out, st = _hip.forward(mg, vg, STD3, algo=_hip.ALGO_STRIP)
    assert _count() == n0 + 1 and int(st.abs().max()) == 0
    out = out.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert (np.abs(out - ref) / scale).max() <= (1e-9 if dt == np.float64 else 5e-6)


def test_transposed_form_negative_global_variance_gives_the_reference_verdict():
    """A negative entry of the global variance vector fails that dim in EVERY utterance: status per (utterance, dim) = the
    natural-order kernel's, zero columns, the other dims untouched."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(4)
    B, T, sd = 45, 400, 5
    m = torch.from_numpy(rng.randn(B, T, 3 * sd)).cuda()
    v = torch.from_numpy(rng.rand(3 * sd) + 0.1).cuda()
    v[3] = -1e-3
    out, st = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
    ref, st_ref = _hip.forward(m, v, STD3, algo=_hip.ALGO_GENERIC)
    assert torch.equal(st, st_ref) and int((st.view(B, sd)[:, 3] != 0).sum()) == B and int((st != 0).sum()) == B
    assert not bool(out[:, :, 3].any())
    keep = [0, 1, 2, 4]
    assert float((out[:, :, keep] - ref[:, :, keep]).abs().max()) <= 1e-9 * float(ref.abs().max())


def test_transposed_form_on_a_column_slice_of_a_wide_batch():
    """A stream consumed in place: lf0 (1 dim) and bap (5 dims) of a (B, T, 198 x 3) Merlin-style batch through
    mlpg_hip_forward_streams, each as a call of its own -- the lane offsets carry the utterance stride of the PARENT rows."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(5)
    B, T = 70, 600
    dims = [60, 1, 5]
    D = 3 * sum(dims)
    m = rng.randn(B, T, D)
    v = rng.rand(B, T, D) + 0.1
    mg, vg = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda()
    col = 0
    for sd in dims:
        n0 = _count()
        out, st = _hip.forward_streams(mg, vg, [(col, sd, STD3)], algo=_hip.ALGO_STRIP)
        assert _count() == n0 + (1 if sd <= 32 else 0), sd
        ref, _, rc = O.mlpg_batch(m[:, :, col:col + 3 * sd], v[:, :, col:col + 3 * sd], STD3)
        assert rc == 0 and int(st.abs().max()) == 0 and tuple(out.shape) == (B, T, sd)
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
        col += 3 * sd


def test_transposed_form_failing_pivots_get_the_reference_verdict():
    """Negative variances in some (utterance, dim) systems of a narrow stream: status = the natural-order kernel's (the reference's
    first failing pivot), an all-zero column for those systems, every other system untouched -- utterances that share a lane group
    with a failing one included."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(9)
    B, T, sd = 150, 900, 3
    m = torch.from_numpy(rng.randn(B, T, 3 * sd)).cuda()
    v = torch.from_numpy(rng.rand(B, T, 3 * sd) + 0.1).cuda()
    for b, t, c in ((0, 5, 0), (7, 450, 1), (20, 899, 2), (21, 100, 3 + 1), (63, 640, 0), (64, 3, 2 * 3 + 2), (149, 700, 1), (149, 10, 2)):
        v[b, t, c] = -1e-3
    n0 = _count()
    out, st = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
    assert _count() == n0 + 1
    ref, st_ref = _hip.forward(m, v, STD3, algo=_hip.ALGO_GENERIC)
    assert torch.equal(st, st_ref) and int((st != 0).sum()) >= 7
    bad = (st.view(B, sd) != 0)[:, None, :].expand_as(out)
    assert not bool((out != 0)[bad].any())
    assert float((out - ref)[~bad].abs().max()) <= 1e-9 * float(ref[~bad].abs().max())
    clean, st2 = _hip.forward(m, v.abs(), STD3, algo=_hip.ALGO_STRIP)      # the control words were left clean
    ref2, _ = _hip.forward(m, v.abs(), STD3, algo=_hip.ALGO_GENERIC)
    assert int(st2.abs().max()) == 0 and float((clean - ref2).abs().max()) <= 1e-9 * float(ref2.abs().max())


def test_merged_launch_with_its_piece_on_the_transposed_form():
    """mlpg_hip_forward_streams on a Merlin-style batch WITHOUT lengths: mgc + lf0 + three dims of bap fill the 64 lanes of the merged
    launch, the two dims left over run as a piece -- on the transposed form (32 utterances x 2 dims a group; window pitch 5, not 2).
    Per stream against the oracle."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(12)
    B, T = 128, 1100
    dims = [60, 1, 5]
    D = 3 * sum(dims)
    m = rng.randn(B, T, D)
    v = rng.rand(B, T, D) + 0.1
    mg, vg = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda()
    streams, col = [], 0
    for sd in dims:
        streams.append((col, sd, STD3))
        col += 3 * sd
    n0, m0 = _count(), int(_hip.lib().mlpg_hip_launch_count(3))
    out, st = _hip.forward_streams(mg, vg, streams)
    assert _count() == n0 + 1 and int(_hip.lib().mlpg_hip_launch_count(3)) == m0 + 1
    assert int(st.abs().max()) == 0
    col = ocol = 0
    for sd in dims:
        ref, _, rc = O.mlpg_batch(m[:, :, col:col + 3 * sd], v[:, :, col:col + 3 * sd], STD3)
        got = out[:, :, ocol:ocol + sd].cpu().numpy()
        assert rc == 0 and np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max(), sd
        col += 3 * sd
        ocol += sd
    out2, _ = _hip.forward_streams(mg, vg, streams)
    assert torch.equal(out, out2)
    # with a lengths vector: the same routes (per-lane frame counts), same numbers
    L = torch.full((B,), T, dtype=torch.int32, device="cuda")
    n1 = _count()
    out3, _ = _hip.forward_streams(mg, vg, streams, L)
    assert _count() == n1 + 1 and float((out3 - out).abs().max()) <= 1e-9 * float(out.abs().max())


def test_auto_takes_the_transposed_form_only_where_it_is_preferred():
    """AUTO: full lane groups and enough (group, strip) items -- 256 in float64, 64 beyond 1024 frames (csrc/mlpg_strip.hip
    strip_tr_preferred); otherwise the wave-per-system kernel as before."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(3)
    for (B, T, sd), want in (((256, 1000, 1), False), ((256, 1000, 5), True), ((8, 1000, 1), False), ((256, 100, 1), False),
                             ((256, 1000, 40), False), ((512, 2000, 1), True), ((80, 1100, 3), True), ((16, 1100, 3), False)):
        m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g)
        v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g) + 0.1
        n0 = _count()
        a, st = _hip.forward(m, v, STD3)
        assert (_count() == n0 + 1) == want, (B, T, sd)
        b, _ = _hip.forward(m, v, STD3, algo=_hip.ALGO_GENERIC)
        assert int(st.abs().max()) == 0 and float((a - b).abs().max()) <= 1e-9 * float(b.abs().max())
