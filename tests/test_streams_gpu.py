"""multi-stream MLPG on the padded (N, Tmax, D) batch, streams consumed in place
(mlpg_hip_forward_streams).  Checked against the reference's per-stream paramgen.mlpg on its own example
data (golden), against the oracle per stream on random ragged batches, and against the dense entry point."""
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


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "mlpg_golden.npz"))


def test_merlin_example_utterance_matches_reference(golden):
    from nnmnkwii_amd import paramgen as G
    feats, gvar = golden["merlin/feats"], golden["merlin/var"]
    sizes, dyn = [180, 3, 1, 3], [True, True, False, True]
    y = G.multi_stream_mlpg(feats, gvar, STD3, sizes, dyn)
    ref = golden["merlin/y"]
    assert y.shape == ref.shape == (feats.shape[0], 60 + 1 + 1 + 1) and y.dtype == np.float32
    scale = np.abs(ref).max(axis=0)
    assert (np.abs(y - ref).max(axis=0) <= 2e-6 * scale + 1e-7).all()       # float32 outputs of float64 solves
    np.testing.assert_array_equal(y[:, 61], feats[:, 183])                  # vuv passes through untouched
    # per-frame float64 variances
    y64 = G.multi_stream_mlpg(feats.astype(np.float64), golden["merlin/v64"], STD3, sizes, dyn)
    r64 = golden["merlin/y64"]
    assert (np.abs(y64 - r64).max(axis=0) <= 1e-10 * np.abs(r64).max(axis=0) + 1e-14).all()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("vmode", ["frame", "global", "unit"])
def test_streams_vs_oracle_ragged_batch(dt, vmode):
    """C5 layout mgc 180 | lf0 3 | bap 15 (+ a pass-through column), ragged lengths, both kernels' paths."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    from oracle import mlpg as O
    rng = np.random.RandomState(11)
    B, T = 5, 300
    sizes, dyn = [180, 3, 1, 15], [True, True, False, True]
    D = sum(sizes)
    m = rng.randn(B, T, D).astype(dt)
    lengths = np.array([300, 1, 257, 64, 2], dtype=np.int32)
    for b in range(B):
        m[b, lengths[b]:] = 0
    if vmode == "frame":
        v = (rng.rand(B, T, D) + 0.1).astype(dt)
    elif vmode == "global":
        v = (rng.rand(D) + 0.1).astype(dt)
    else:
        v = None
    y = G.multi_stream_mlpg(m, v, STD3, sizes, dyn, lengths=lengths)
    assert y.shape == (B, T, 60 + 1 + 1 + 5) and y.dtype == dt
    c0, o0 = 0, 0
    tol = 1e-9 if dt == np.float64 else 2e-6
    for size, d_ in zip(sizes, dyn):
        if d_:
            sd = size // 3
            vs = None if v is None else (v[c0:c0 + size] if v.ndim == 1 else np.ascontiguousarray(v[:, :, c0:c0 + size]))
            if vs is None:
                vs = np.ones(size, dtype=dt)
            ref, st, rc = O.mlpg_batch(np.ascontiguousarray(m[:, :, c0:c0 + size]), vs, STD3, lengths)
            assert rc == 0
            got = y[:, :, o0:o0 + sd]
            assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max())
            o0 += sd
        else:
            ref = m[:, :, c0:c0 + size].copy()
            np.testing.assert_array_equal(y[:, :, o0:o0 + size], ref)
            o0 += size
        c0 += size
    for b in range(B):
        assert not y[b, lengths[b]:].any()           # padding frames zero-filled
    # the in-place stream slice equals the dense entry point on a contiguous copy (bit for bit: same kernel)
    md = torch.from_numpy(m).cuda()
    vd = None if v is None else torch.from_numpy(v).cuda()
    Ld = torch.from_numpy(lengths).cuda()
    out, st = _hip.forward_streams(md, vd, [(0, 60, STD3), (180, 1, STD3), (183, 1, None), (184, 5, STD3)], Ld)
    assert int(st.abs().sum().item()) == 0
    dense_v = None if vd is None else (vd[:180].contiguous() if vd.dim() == 1 else vd[:, :, :180].contiguous())
    dense, _ = _hip.forward(md[:, :, :180].contiguous(), dense_v, STD3, Ld)
    assert torch.equal(out[:, :, :60], dense)


def test_streams_status_and_errors():
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(5)
    m = rng.randn(2, 40, 9)
    v = rng.rand(2, 40, 9) + 0.1
    v[1, 7, 6] = -1e-3                                 # stream 1 = cols 6..8 (sd = 1): static variance of frame 7 negative
    sizes, dyn = [6, 3], [True, True]
    with pytest.raises(np.linalg.LinAlgError, match="leading minor not positive definite"):
        G.multi_stream_mlpg(m, v, STD3, sizes, dyn)
    out, st = _hip.forward_streams(torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), [(0, 2, STD3), (6, 1, STD3)])
    st = st.cpu().numpy()
    assert st.shape == (2, 3) and (st[0] == 0).all() and (st[1, :2] == 0).all() and st[1, 2] > 0
    assert not out[1, :, 2].cpu().numpy().any()       # the failing system is zero-filled
    # per-stream window lists, one of them wide (generic kernel) and a 2-D (T, D) input
    x = rng.randn(50, 4 + 6)
    y = G.multi_stream_mlpg(x, None, [WINDOW_SETS["std2"], WINDOW_SETS["wide3"]], [4, 6], [True, True])
    from oracle import mlpg as O
    r0 = O.mlpg(np.ascontiguousarray(x[:, :4]), np.ones(4), WINDOW_SETS["std2"])
    r1 = O.mlpg(np.ascontiguousarray(x[:, 4:]), np.ones(6), WINDOW_SETS["wide3"])
    np.testing.assert_allclose(y, np.concatenate([r0, r1], axis=1), rtol=1e-9, atol=1e-11)
    with pytest.raises(_hip.HipExtensionError):
        _hip.forward_streams(torch.zeros(1, 4, 6, dtype=torch.float64).cuda(), None, [(4, 1, STD3)])   # does not fit


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("T,algo", [(300, "strip"), (1500, "auto"), (2100, "auto")])   # 2100: the piece runs on the natural-order kernel
def test_streams_merged_into_one_strip_launch(dt, T, algo):
    """Streams that share their three windows (per-frame variances) run as ONE strip-kernel launch, their static dims
    side by side on the lanes (66 = 60 + 1 + 5: the 64 lanes take mgc, three dims of bap and lf0; bap's last two dims
    run as a piece of their own): every stream what the strip kernel gives on a dense copy of its columns, status in
    the stream's own columns, failing systems zeroed."""
    import torch
    from nnmnkwii_amd import _hip
    rng = np.random.RandomState(17)
    B = 6
    D = 180 + 3 + 1 + 15
    m = rng.randn(B, T, D).astype(dt)
    v = (rng.rand(B, T, D) + 0.1).astype(dt)
    lengths = np.array([T, 1, T - 43, 64, 2, T // 2], dtype=np.int32)
    v[2, 11, 184 + 2] = -1e-3          # bap dim 2 of utterance 2: a negative static variance
    v[5, 3, 180] = -1e-3               # lf0 of utterance 5 likewise
    for b in range(B):
        m[b, lengths[b]:] = 0
    md, vd, Ld = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lengths).cuda()
    streams = [(0, 60, STD3), (180, 1, STD3), (183, 1, None), (184, 5, STD3)]
    a = _hip.ALGO_STRIP if algo == "strip" else _hip.ALGO_AUTO
    out, st = _hip.forward_streams(md, vd, streams, Ld, algo=a)
    o0 = 0
    for in_col, sd, win in streams:
        if win is None:
            assert torch.equal(out[:, :, o0:o0 + sd], md[:, :, in_col:in_col + sd] * (torch.arange(T, device="cuda")[None, :, None] < Ld[:, None, None]))
        else:
            dense, dst = _hip.forward(md[:, :, in_col:in_col + 3 * sd].contiguous(), vd[:, :, in_col:in_col + 3 * sd].contiguous(),
                                      STD3, Ld, algo=_hip.ALGO_STRIP)
            if sd == 60:    # entirely inside the merged launch: the same kernel, the same bits
                assert torch.equal(out[:, :, o0:o0 + sd], dense), (in_col, sd)
            else:           # may be cut between the merged launch and a wave-kernel launch of its last dims
                tol = 1e-9 if dt == np.float64 else 2e-6
                assert float((out[:, :, o0:o0 + sd] - dense).abs().max()) <= tol * max(1.0, float(dense.abs().max())), (in_col, sd)
            assert torch.equal(st[:, o0:o0 + sd] != 0, dst.reshape(B, sd) != 0), (in_col, sd)
        o0 += sd
    st = st.cpu().numpy()
    assert st[2, 62 + 2] > 0 and st[5, 60] > 0 and (st != 0).sum() == 2
    assert not out[2, :, 62 + 2].cpu().numpy().any() and not out[5, :, 60].cpu().numpy().any()


def test_streams_merged_launch_really_is_one_kernel():
    """rocprof-free check: the merged path leaves the strip scratch's generation alone and is faster than the sum of
    its parts on a config-5 shaped batch is bench material; here only that streams with DIFFERENT windows are not
    merged (they would be wrong) and two wide streams are."""
    import torch
    from nnmnkwii_amd import _hip
    from oracle import mlpg as O
    rng = np.random.RandomState(3)
    B, T = 2, 1100
    m = rng.randn(B, T, 6 + 6)
    v = rng.rand(B, T, 6 + 6) + 0.1
    other = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.25, 0.0, 0.25])), (1, 1, np.array([1.0, -2.0, 1.0]))]
    out, st = _hip.forward_streams(torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), [(0, 2, STD3), (6, 2, other)],
                                   algo=_hip.ALGO_STRIP)
    assert int(st.abs().sum().item()) == 0
    out = out.cpu().numpy()
    for k, (c0, win) in enumerate(((0, STD3), (6, other))):
        ref, _, rc = O.mlpg_batch(np.ascontiguousarray(m[:, :, c0:c0 + 6]), np.ascontiguousarray(v[:, :, c0:c0 + 6]), win, None)
        assert rc == 0
        assert np.abs(out[:, :, 2 * k:2 * k + 2] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_merged_stream_launch_against_the_oracle_directly(dt):
    """The merged multi-stream launch (strip_kernel<..., MULTI>) against the CPU oracle per stream -- not against
    another HIP launch: 40 x 1100 x 198 ragged (>= 512 strips: AUTO merges), and the same batch with ALGO_STRIP forced.
    mlpg_hip_launch_count shows that the merged instantiation really ran."""
    import torch
    from nnmnkwii_amd import _hip
    from oracle import mlpg as O
    rng = np.random.RandomState(23)
    B, T = 40, 1100
    D = 180 + 3 + 15
    m = rng.randn(B, T, D).astype(dt)
    v = (rng.rand(B, T, D) + 0.1).astype(dt)
    lengths = rng.randint(1, T + 1, B).astype(np.int32)
    lengths[:3] = (T, 1, 65)
    for b in range(B):
        m[b, lengths[b]:] = 0
    md, vd, Ld = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lengths).cuda()
    streams = [(0, 60, STD3), (180, 1, STD3), (183, 5, STD3)]
    tol = 1e-9 if dt == np.float64 else 2e-6
    sel = [0, 1, 2, 7, 21, 39]
    for algo in (_hip.ALGO_AUTO, _hip.ALGO_STRIP):
        n0 = _hip.lib().mlpg_hip_launch_count(3)
        out, st = _hip.forward_streams(md, vd, streams, Ld, algo=algo)
        torch.cuda.synchronize()
        assert _hip.lib().mlpg_hip_launch_count(3) == n0 + 1, "the streams were not merged into one strip launch"
        assert int(st.abs().max()) == 0
        out = out.cpu().numpy()
        o0 = 0
        for in_col, sd, _ in streams:
            ref, _, rc = O.mlpg_batch(np.ascontiguousarray(m[sel][:, :, in_col:in_col + 3 * sd]),
                                      np.ascontiguousarray(v[sel][:, :, in_col:in_col + 3 * sd]), STD3, lengths[sel])
            assert rc == 0
            got = out[sel][:, :, o0:o0 + sd]
            scale = np.abs(ref).max()
            assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= tol * scale, (algo, in_col)
            for i, b in enumerate(sel):
                assert not got[i, lengths[b]:].any()
            o0 += sd


def test_merged_stream_launch_against_the_reference_itself_config5_utterance():
    """One config-5 shaped utterance (T = 2000, mgc 180 | lf0 3 | bap 15 of 198 columns, float64) inside a batch large
    enough to merge: every stream against the reference's own paramgen.mlpg (compiled into oracle/_ref), or against the
    oracle restatement (pinned on the reference's goldens) where /root/reference was not available to build it."""
    import torch
    from nnmnkwii_amd import _hip
    from oracle import mlpg as O
    from oracle import ref as R
    rng = np.random.RandomState(29)
    B, T, D = 20, 2000, 198
    m = rng.randn(B, T, D)
    v = rng.rand(B, T, D) + 0.1
    streams = [(0, 60, STD3), (180, 1, STD3), (183, 5, STD3)]
    n0 = _hip.lib().mlpg_hip_launch_count(3)
    out, st = _hip.forward_streams(torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), streams)
    torch.cuda.synchronize()
    assert _hip.lib().mlpg_hip_launch_count(3) == n0 + 1
    out = out.cpu().numpy()
    ref_mlpg = R.load().mlpg if R.available() else O.mlpg
    o0 = 0
    for in_col, sd, _ in streams:
        for b in (0, 13):
            y = ref_mlpg(np.ascontiguousarray(m[b, :, in_col:in_col + 3 * sd]), np.ascontiguousarray(v[b, :, in_col:in_col + 3 * sd]), STD3)
            assert np.abs(out[b, :, o0:o0 + sd] - y).max() <= 1e-9 * np.abs(y).max(), (in_col, b)
        o0 += sd


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("vmode", ["global", "unit"])
def test_merged_stream_launch_global_and_unit_variances_against_the_oracle(dt, vmode):
    """Round 5: with global (D,) or unit variances the streams of a Merlin-style row (mgc 60 | lf0 1 | bap 5; the parent
    row also holds a pass-through vuv column) share ONE constant-coefficient launch (cst::stream_kernel<..., MULTI>: 64 lanes
    = 60 + 1 + the first 3 bap dims; the other 2 bap dims run as a piece on the wave-per-system kernel).  Against the CPU
    oracle per stream, ragged lengths, AUTO (the batch has a sequence per CU) and ALGO_CONST forced on a small batch;
    mlpg_hip_launch_count(8) shows that the merged instantiation ran."""
    import torch
    from nnmnkwii_amd import _hip
    from oracle import mlpg as O
    rng = np.random.RandomState(41)
    D = 180 + 3 + 1 + 15
    streams = [(0, 60, STD3), (180, 1, STD3), (183, 1, None), (184, 5, STD3)]
    tol = 1e-9 if dt == np.float64 else 5e-6
    for B, T, algo in ((200, 300, _hip.ALGO_AUTO), (6, 531, _hip.ALGO_CONST)):
        m = rng.randn(B, T, D).astype(dt)
        vg = (rng.rand(D) + 0.1).astype(dt)
        lengths = rng.randint(1, T + 1, B).astype(np.int32)
        lengths[:4] = (T, 1, 2, 129)
        for b in range(B):
            m[b, lengths[b]:] = 0
        md, Ld = torch.from_numpy(m).cuda(), torch.from_numpy(lengths).cuda()
        vd = torch.from_numpy(vg).cuda() if vmode == "global" else None
        n0 = _hip.lib().mlpg_hip_launch_count(8)
        out, st = _hip.forward_streams(md, vd, streams, Ld, algo=algo)
        torch.cuda.synchronize()
        assert _hip.lib().mlpg_hip_launch_count(8) == n0 + 1, "the streams were not merged into one constant-coefficient launch"
        assert int(st.abs().max()) == 0
        out = out.cpu().numpy()
        sel = [0, 1, 2, 3, B // 2, B - 1]
        o0 = 0
        for in_col, sd, win in streams:
            got = out[sel][:, :, o0:o0 + sd]
            if win is None:
                assert np.array_equal(got, m[sel][:, :, in_col:in_col + sd])
            else:
                var = np.ascontiguousarray(vg[in_col:in_col + 3 * sd]) if vmode == "global" else np.ones(3 * sd, dtype=dt)
                ref, _, rc = O.mlpg_batch(np.ascontiguousarray(m[sel][:, :, in_col:in_col + 3 * sd]), var, STD3, lengths[sel])
                assert rc == 0
                scale = np.abs(ref).max()
                assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= tol * scale, (B, algo, in_col, vmode)
                for i, b in enumerate(sel):
                    assert not got[i, lengths[b]:].any()
            o0 += sd


def test_merged_stream_launch_global_variances_reports_a_failing_pivot_like_the_reference():
    """A negative global variance in ONE stream of the merged launch: that stream's systems get the reference's verdict
    (k-th leading minor, zero column), the other streams' trajectories are untouched."""
    import torch
    from nnmnkwii_amd import _hip
    from oracle import mlpg as O
    rng = np.random.RandomState(43)
    B, T, D = 200, 200, 198
    m = rng.randn(B, T, D)
    vg = rng.rand(D) + 0.1
    vg[180] = -0.5                      # lf0's static variance
    streams = [(0, 60, STD3), (180, 1, STD3), (183, 5, STD3)]
    n0 = _hip.lib().mlpg_hip_launch_count(8)
    out, st = _hip.forward_streams(torch.from_numpy(m).cuda(), torch.from_numpy(vg).cuda(), streams)
    torch.cuda.synchronize()
    assert _hip.lib().mlpg_hip_launch_count(8) == n0 + 1
    out, st = out.cpu().numpy(), st.cpu().numpy()
    ref, stat, rc = O.mlpg_batch(np.ascontiguousarray(m[:2, :, 180:183]), np.ascontiguousarray(vg[180:183]), STD3, None)
    assert int(stat.ravel()[0]) > 0
    assert (st[:, 60] == int(stat.ravel()[0])).all() and not out[:, :, 60].any()
    assert not st[:, :60].any() and not st[:, 61:].any()
    ref, _, rc = O.mlpg_batch(np.ascontiguousarray(m[:2, :, :180]), np.ascontiguousarray(vg[:180]), STD3, None)
    assert rc == 0 and np.abs(out[:2, :, :60] - ref).max() <= 1e-9 * np.abs(ref).max()
