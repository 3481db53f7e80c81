"""GPU parity tests (-m gpu) of the constant-coefficient MLPG kernel (algo = MLPG_HIP_ALGO_CONST: global (D,) and unit
variances, the matrix factorised once, the solves as constant-coefficient recurrences, one workgroup walking one
utterance with a one-super-step lag or two sweeps), through the C ABI, against the CPU oracle."""
import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

pytestmark = pytest.mark.gpu

TOL64 = 1e-9
TOL32 = 5e-6
CONST_WINDOWS = ("std3", "std2", "asym2")   # extents <= 1, at least one dynamic window


def rel_err(y, ref):
    scale = np.abs(ref).max(axis=0, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float((np.abs(y.astype(np.float64) - ref.astype(np.float64)) / scale).max())


def _fwd(M_, var, windows, lengths, algo=None):
    import torch
    from nnmnkwii_amd import _hip
    m = torch.from_numpy(M_).cuda()
    v = None if var is None else torch.from_numpy(var).cuda()
    L = None if lengths is None else torch.from_numpy(lengths).cuda()
    y, st = _hip.forward(m, v, windows, L, algo=_hip.ALGO_CONST if algo is None else algo)
    return y.cpu().numpy(), st.cpu().numpy()


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 34, 35, 63, 64, 65, 66, 67, 127, 128, 129, 130, 131, 200,
                               257, 513, 1000, 1025, 2048, 4100])
def test_const_all_lengths(T):
    """Every chunk / super-step count around the boundaries (16-frame chunks, 128-frame super-steps), ragged lengths down
    to 1 frame, 70 static dims in two dim groups, global and unit variances, float64 and float32, forward and backward."""
    import torch
    from nnmnkwii_amd import _hip
    for wname in ("std3", "asym2") if T > 600 else CONST_WINDOWS:
        windows = WINDOW_SETS[wname]
        nw = len(windows)
        B, sd = 4, 70 if T <= 300 else 20
        rng = np.random.RandomState(T + nw)
        M_ = rng.randn(B, T, nw * sd)
        vg = rng.rand(nw * sd) + 0.1
        lengths = np.array([T, max(1, T - 1), max(1, T // 2), max(1, T - 2)], dtype=np.int32)
        for var in (vg, None):
            ys, sts = _fwd(M_, var, windows, lengths)
            assert int(np.abs(sts).max()) == 0
            yo, _, rc = O.mlpg_batch(M_, np.ones(nw * sd) if var is None else var, windows, lengths)
            assert rc == 0
            assert rel_err(ys.reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64, (wname, T, var is None)
            for b in range(B):
                assert not ys[b, lengths[b]:].any()
        # backward against the natural-order kernel (== the reference's mlpg_grad: tests/test_parity_r2_gpu.py)
        go = torch.from_numpy(rng.randn(B, T, sd)).cuda()
        L = torch.from_numpy(lengths).cuda()
        for var in (torch.from_numpy(vg).cuda(), None):
            gs, st = _hip.backward(var, go, windows, nw * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_CONST)
            gg, _ = _hip.backward(var, go, windows, nw * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_GENERIC)
            assert int(st.abs().max()) == 0
            scale = float(gg.abs().max()) + 1e-300
            assert float((gs - gg).abs().max()) <= 1e-10 * scale, (wname, T)
        # float32
        M32 = M_.astype(np.float32)
        v32 = vg.astype(np.float32)
        a, _ = _fwd(M32, v32, windows, lengths)
        ao, _, _ = O.mlpg_batch(M32, v32, windows, lengths)
        assert rel_err(a.reshape(-1, sd), ao.reshape(-1, sd)) <= TOL32, (wname, T)
        a, _ = _fwd(M32, None, windows, lengths)
        ao, _, _ = O.mlpg_batch(M32, np.ones(nw * sd, dtype=np.float32), windows, lengths)
        assert rel_err(a.reshape(-1, sd), ao.reshape(-1, sd)) <= TOL32, (wname, T)


def test_const_large_shape_many_strips():
    """Config-2 sized lanes, more sequences than CUs on small devices, ragged."""
    windows = WINDOW_SETS["std3"]
    B, T, sd = 160, 1000, 60
    rng = np.random.RandomState(7)
    M_ = rng.randn(B, T, 3 * sd)
    vg = rng.rand(3 * sd) + 0.1
    lengths = rng.randint(1, T + 1, B).astype(np.int32)
    lengths[:4] = (T, T - 1, T - 2, 129)
    for var in (vg, None):
        ys, sts = _fwd(M_, var, windows, lengths)
        assert int(np.abs(sts).max()) == 0
        sel = [0, 1, 2, 3, 17, 99, 159]
        yo, _, rc = O.mlpg_batch(M_[sel], np.ones(3 * sd) if var is None else var, windows, lengths[sel])
        assert rc == 0
        assert rel_err(ys[sel].reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64
        for b in range(B):
            assert not ys[b, lengths[b]:].any()
    # same results as the wave-per-system kernel everywhere
    yw, _ = _fwd(M_, vg, windows, lengths, algo=2)
    yc, _ = _fwd(M_, vg, windows, lengths)
    assert rel_err(yc.reshape(-1, sd), yw.reshape(-1, sd)) <= TOL64


def test_const_repeat_launches_bitwise_equal():
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(8)
    M_ = rng.randn(64, 700, 180)
    vg = rng.rand(180) + 0.1
    y0, _ = _fwd(M_, vg, windows, None)
    for _ in range(3):
        y1, _ = _fwd(M_, vg, windows, None)
        assert np.array_equal(y0, y1)


def test_const_slow_decay_variances():
    """Dynamic features 100x / 10000x tighter than the static ones: the factor converges over hundreds of rows, a
    super-step does not damp what comes up from below: the exact two-sweep path."""
    windows = WINDOW_SETS["std3"]
    sd, T, B = 20, 1500, 6
    rng = np.random.RandomState(9)
    M_ = rng.randn(B, T, 3 * sd)
    vg = np.concatenate([rng.rand(sd) + 0.5, (rng.rand(sd) + 0.5) * 1e-2, (rng.rand(sd) + 0.5) * 1e-4])
    lengths = np.array([T, T - 1, 700, 130, 3, 1], dtype=np.int32)
    ys, sts = _fwd(M_, vg, windows, lengths)
    assert int(np.abs(sts).max()) == 0
    yo, _, rc = O.mlpg_batch(M_, vg, windows, lengths)
    assert rc == 0 and rel_err(ys.reshape(-1, sd), yo.reshape(-1, sd)) <= 1e-7


def test_const_negative_global_variance_gives_the_reference_verdict():
    windows = WINDOW_SETS["std3"]
    sd, T, B = 24, 300, 3
    rng = np.random.RandomState(10)
    M_ = rng.randn(B, T, 3 * sd)
    vg = rng.rand(3 * sd) + 0.1
    vg[5] = -0.3            # static variance of dim 5
    vg[sd + 9] = -2.0       # delta variance of dim 9
    lengths = np.array([T, 40, 2], dtype=np.int32)
    ys, sts = _fwd(M_, vg, windows, lengths)
    sts = sts.reshape(B, sd)
    # the oracle (like the reference) stops at the first failing system: one dim at a time
    nbad = 0
    for d in range(sd):
        cols = [d, sd + d, 2 * sd + d]
        yo, so, rc = O.mlpg_batch(np.ascontiguousarray(M_[:, :, cols]), vg[cols], windows, lengths)
        for b in range(B):
            yb, sb, _ = O.mlpg_batch(np.ascontiguousarray(M_[b:b + 1, :, cols]), vg[cols], windows, lengths[b:b + 1])
            assert sts[b, d] == sb[0, 0], (b, d, sts[b, d], sb[0, 0])
            if sb[0, 0]:
                nbad += 1
                assert not ys[b, :, d].any()
            else:
                assert rel_err(ys[b, :, d:d + 1], yb[0]) <= TOL64
    assert nbad >= 3


def test_const_is_the_auto_choice_for_global_and_unit_variances():
    """AUTO == CONST bit for bit on a wide stream with global variances when the launch has a sequence per CU; a small
    batch stays with the wave-per-system kernel."""
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(11)
    M_ = rng.randn(200, 150, 180)
    vg = rng.rand(180) + 0.1
    ya, _ = _fwd(M_, vg, windows, None, algo=0)
    yc, _ = _fwd(M_, vg, windows, None)
    assert np.array_equal(ya, yc)
    ya, _ = _fwd(M_[:8], vg, windows, None, algo=0)
    yw, _ = _fwd(M_[:8], vg, windows, None, algo=2)
    assert np.array_equal(ya, yw)


def test_const_unit_table_is_reused_and_invalidated_correctly():
    """Unit variances: the host skips the table launch when the stream ran the same windows and shape last.  Interleaved with
    global-variance launches (which rewrite the stream's table), other windows and another length, every result still equals
    the oracle."""
    rng = np.random.RandomState(12)
    B, sd = 200, 40
    seq = [("std3", 150, None), ("std3", 150, None), ("std3", 150, "g"), ("std3", 150, None), ("asym2", 150, None),
           ("std3", 150, None), ("std3", 97, None), ("std3", 150, None), ("std3", 150, "g"), ("std3", 150, "g"), ("std3", 150, None)]
    for wname, T, mode in seq:
        windows = WINDOW_SETS[wname]
        nw = len(windows)
        M_ = rng.randn(B, T, nw * sd)
        var = (rng.rand(nw * sd) + 0.1) if mode == "g" else None
        ys, sts = _fwd(M_, var, windows, None)
        assert int(np.abs(sts).max()) == 0
        sel = [0, 1, 77, 199]
        yo, _, rc = O.mlpg_batch(M_[sel], np.ones(nw * sd) if var is None else var, windows)
        assert rc == 0 and rel_err(ys[sel].reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64, (wname, T, mode)


def test_const_unit_table_survives_graph_replays_on_the_same_stream():
    """a captured launch bakes the capture stream's scratch table into its graph and rebuilds it on every
    replay without passing through the host; an eager unit-variance call on that stream afterwards must not trust the
    host-side 'same as last time' shortcut.  Sequence: eager A, capture B, eager A, replay B, eager A -- all against the
    oracle."""
    import torch
    from nnmnkwii_amd import _hip
    rng = np.random.RandomState(31)
    sd = 40
    wA, wB = WINDOW_SETS["std3"], WINDOW_SETS["asym2"]
    MA = rng.randn(200, 150, len(wA) * sd)
    MB = rng.randn(200, 97, len(wB) * sd)
    refA, _, rcA = O.mlpg_batch(MA[:3], np.ones(len(wA) * sd), wA)
    refB, _, rcB = O.mlpg_batch(MB[:3], np.ones(len(wB) * sd), wB)
    assert rcA == 0 and rcB == 0
    s = torch.cuda.Stream()
    mA, mB = torch.from_numpy(MA).cuda(), torch.from_numpy(MB).cuda()
    torch.cuda.synchronize()

    def eager_a():
        with torch.cuda.stream(s):
            y, _ = _hip.forward(mA, None, wA, None, algo=_hip.ALGO_CONST)
        s.synchronize()
        return y[:3].cpu().numpy()

    assert rel_err(eager_a().reshape(-1, sd), refA.reshape(-1, sd)) <= TOL64
    with torch.cuda.stream(s):
        _hip.forward(mB, None, wB, None, algo=_hip.ALGO_CONST, want_status=False)   # warm-up: scratch grown outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        yB, _ = _hip.forward(mB, None, wB, None, algo=_hip.ALGO_CONST, want_status=False)
    assert rel_err(eager_a().reshape(-1, sd), refA.reshape(-1, sd)) <= TOL64
    for _ in range(2):
        g.replay()
        torch.cuda.synchronize()
        assert rel_err(yB[:3].cpu().numpy().reshape(-1, sd), refB.reshape(-1, sd)) <= TOL64
        assert rel_err(eager_a().reshape(-1, sd), refA.reshape(-1, sd)) <= TOL64
