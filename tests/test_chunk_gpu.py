"""GPU parity tests (-m gpu) of the chunked MLPG kernel (algo = MLPG_HIP_ALGO_CHUNK: window extents up to 2 -- the reference's own
5-tap test windows, tests/test_paramgen.py:21-26 -- chunks of 16 + 4 frames eliminated twice around a block-tridiagonal solve over
their separators), through the C ABI, against the CPU oracle."""
import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

pytestmark = pytest.mark.gpu

TOL64 = 1e-9
TOL32 = 5e-6


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
    y, st = _hip.forward(m, v, windows, L, algo=_hip.ALGO_CHUNK if algo is None else algo)
    return y.cpu().numpy(), st.cpu().numpy()


@pytest.mark.parametrize("wname", ["wide3", "std3", "asym2", "std2"])
@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 15, 16, 17, 19, 20, 21, 22, 39, 40, 41, 59, 60, 61, 100, 257, 1000, 2051])
def test_chunk_all_lengths(wname, T):
    """Every chunk count around the boundaries (20-frame chunks for extents of 2, 16-frame chunks for extents of 1), ragged
    lengths down to 1 frame, 70 static dims in two dim groups; per-frame, global and unit variances; float64 and float32."""
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    B, sd = 4, 70 if T <= 300 else 9
    rng = np.random.RandomState(T + nw)
    M_ = rng.randn(B, T, nw * sd)
    V_ = rng.rand(B, T, nw * sd) + 0.1
    lengths = np.array([T, max(1, T - 1), max(1, T // 2), max(1, T - 3)], dtype=np.int32)
    for var in (V_, V_[0, 0].copy(), None):
        ys, sts = _fwd(M_, var, windows, lengths)
        assert int(np.abs(sts).max()) == 0
        yo, _, rc = O.mlpg_batch(M_, np.ones(nw * sd) if var is None else var, windows, lengths)
        assert rc == 0
        assert rel_err(ys.reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64, (wname, T, None if var is None else var.ndim)
        for b in range(B):
            assert not ys[b, lengths[b]:].any()
    M32, V32 = M_.astype(np.float32), V_.astype(np.float32)
    a, _ = _fwd(M32, V32, windows, lengths)
    ao, _, _ = O.mlpg_batch(M32, V32, windows, lengths)
    assert a.dtype == np.float32 and rel_err(a.reshape(-1, sd), ao.reshape(-1, sd)) <= TOL32, (wname, T)


def test_chunk_config2_shape_wide_windows_against_the_other_kernels():
    """256 x 1000 x 60 with the 5-tap windows: the chunked kernel against the natural-order kernel everywhere and the oracle on a
    few utterances; AUTO takes it; repeat launches bitwise equal."""
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["wide3"]
    B, T, sd = 256, 1000, 60
    rng = np.random.RandomState(17)
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    lengths = rng.randint(1, T + 1, B).astype(np.int32)
    lengths[:3] = (T, T - 1, 20)
    yc, st = _fwd(M_, V_, windows, lengths)
    assert int(np.abs(st).max()) == 0
    yg, _ = _fwd(M_, V_, windows, lengths, algo=_hip.ALGO_GENERIC)
    assert rel_err(yc.reshape(-1, sd), yg.reshape(-1, sd)) <= TOL64
    sel = [0, 1, 2, 100, 255]
    yo, _, rc = O.mlpg_batch(M_[sel], V_[sel], windows, lengths[sel])
    assert rc == 0 and rel_err(yc[sel].reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64
    n0 = _hip.lib().mlpg_hip_launch_count(6)
    ya, _ = _fwd(M_, V_, windows, lengths, algo=_hip.ALGO_AUTO)
    assert _hip.lib().mlpg_hip_launch_count(6) > n0 and np.array_equal(ya, yc)
    yc2, _ = _fwd(M_, V_, windows, lengths)
    assert np.array_equal(yc, yc2)


def test_chunk_negative_variance_gives_the_reference_verdict():
    windows = WINDOW_SETS["wide3"]
    sd, T, B = 12, 130, 3
    rng = np.random.RandomState(10)
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    V_[1, 77, 5] = -1e-3          # static variance of dim 5, utterance 1
    V_[2, 30, sd + 9] = -1e-3     # delta variance of dim 9, utterance 2
    lengths = np.array([T, T, 100], dtype=np.int32)
    ys, sts = _fwd(M_, V_, windows, lengths)
    sts = sts.reshape(B, sd)
    nbad = 0
    for b in range(B):
        for d in range(sd):
            cols = [d, sd + d, 2 * sd + d]
            yb, sb, _ = O.mlpg_batch(np.ascontiguousarray(M_[b:b + 1, :, cols]), np.ascontiguousarray(V_[b:b + 1, :, cols]), windows,
                                     lengths[b:b + 1])
            assert sts[b, d] == sb[0, 0], (b, d, sts[b, d], sb[0, 0])
            if sb[0, 0]:
                nbad += 1
                assert not ys[b, :, d].any()
            else:
                assert rel_err(ys[b, :, d:d + 1], yb[0]) <= TOL64
    assert nbad >= 2


def test_chunk_ill_conditioned_variances_as_the_other_kernels():
    """Log-normal variances (sigma = 2, 4): the chunked elimination order deviates from the oracle's natural-order Cholesky no more
    than the natural-order kernel does."""
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["wide3"]
    rng = np.random.RandomState(12)
    B, T, sd = 3, 400, 8
    M_ = rng.randn(B, T, 3 * sd)
    for sigma, tol in ((2.0, 1e-11), (4.0, 1e-7)):
        V_ = np.exp(sigma * rng.randn(B, T, 3 * sd))
        yc, st = _fwd(M_, V_, windows, None)
        yg, _ = _fwd(M_, V_, windows, None, algo=_hip.ALGO_GENERIC)
        yo, _, rc = O.mlpg_batch(M_, V_, windows)
        assert rc == 0 and int(np.abs(st).max()) == 0
        ec, eg = rel_err(yc.reshape(-1, sd), yo.reshape(-1, sd)), rel_err(yg.reshape(-1, sd), yo.reshape(-1, sd))
        assert ec <= max(tol, 20 * eg), (sigma, ec, eg)


def test_chunk_through_forward_streams_column_slices():
    """Streams of one padded (B, T, ld) batch with the 5-tap windows (column slices read in place: row stride != stream width):
    every stream takes the chunked kernel and equals the oracle on a dense copy of its slice."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["wide3"]
    rng = np.random.RandomState(23)
    B, T = 5, 300
    sds = [20, 3, 70]
    ld = sum(3 * s for s in sds) + 4            # 4 unused columns between the streams and at the end
    Y = rng.randn(B, T, ld)
    V = rng.rand(B, T, ld) + 0.1
    lengths = np.array([T, T - 1, 150, 21, 1], dtype=np.int32)
    cols, c0 = [], 0
    for s in sds:
        cols.append(c0)
        c0 += 3 * s + 1
    n0 = _hip.lib().mlpg_hip_launch_count(6)
    out, status = _hip.forward_streams(torch.from_numpy(Y).cuda(), torch.from_numpy(V).cuda(),
                                       [(c, s, windows) for c, s in zip(cols, sds)], torch.from_numpy(lengths).cuda())
    assert _hip.lib().mlpg_hip_launch_count(6) >= n0 + 2      # (a 3-dim stream may go to the natural-order kernel)
    out = out.cpu().numpy()
    assert int(status.abs().max()) == 0
    o0 = 0
    for c, s in zip(cols, sds):
        yo, _, rc = O.mlpg_batch(np.ascontiguousarray(Y[:, :, c:c + 3 * s]), np.ascontiguousarray(V[:, :, c:c + 3 * s]), windows, lengths)
        assert rc == 0
        assert rel_err(out[:, :, o0:o0 + s].reshape(-1, s), yo.reshape(-1, s)) <= TOL64
        o0 += s


@pytest.mark.parametrize("wname", ["wide3", "std3", "asym2"])
@pytest.mark.parametrize("T", [1, 2, 3, 5, 17, 19, 20, 21, 22, 23, 40, 41, 42, 61, 100, 257, 1000])
def test_chunk_backward_all_lengths(wname, T):
    """mlpg_hip_backward on the chunked kernel (the gradient rows of a chunk are written by the chunk to their right, the utterance's
    last chunk writes its own): against the natural-order kernel (== the reference's mlpg_grad: tests/test_parity_r2_gpu.py) for
    every chunk count around the boundaries, ragged lengths, two dim groups, the three variance modes, float64 and float32; small T
    also against the oracle's dense mlpg_grad."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    B, sd = 4, 70 if T <= 300 else 9
    rng = np.random.RandomState(T + 3 * nw)
    V_ = rng.rand(B, T, nw * sd) + 0.1
    go = rng.randn(B, T, sd)
    lengths = np.array([T, max(1, T - 1), max(1, T // 2), max(1, T - 3)], dtype=np.int32)
    L = torch.from_numpy(lengths).cuda()
    for dt, tol in ((np.float64, 1e-10), (np.float32, 3e-6)):
        g = torch.from_numpy(go.astype(dt)).cuda()
        for var in (V_, V_[0, 0].copy(), None):
            v = None if var is None else torch.from_numpy(var.astype(dt)).cuda()
            gc, st = _hip.backward(v, g, windows, nw * sd, L, out_dtype=g.dtype, algo=_hip.ALGO_CHUNK)
            gg, _ = _hip.backward(v, g, windows, nw * sd, L, out_dtype=g.dtype, algo=_hip.ALGO_GENERIC)
            assert int(st.abs().max()) == 0
            scale = float(gg.abs().max()) + 1e-300
            assert float((gc - gg).abs().max()) <= tol * scale, (wname, T, dt.__name__, None if var is None else var.ndim)
            for b in range(B):
                assert not bool(gc[b, int(lengths[b]):].any())
    if T <= 41:
        gc, _ = _hip.backward(torch.from_numpy(V_).cuda(), torch.from_numpy(go).cuda(), windows, nw * sd, L, out_dtype=torch.float64,
                              algo=_hip.ALGO_CHUNK)
        gc = gc.cpu().numpy()
        for b in range(B):
            Tb = int(lengths[b])
            gr = O.mlpg_grad(np.zeros((Tb, nw * sd)), V_[b, :Tb], windows, go[b, :Tb])      # float32, as the reference
            assert np.abs(gc[b, :Tb] - gr).max() <= 5e-7 * max(1.0, np.abs(gr).max()), (wname, T, b)


def test_chunk_backward_config2_shape_is_the_auto_choice_for_5_tap_windows():
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["wide3"]
    B, T, sd = 64, 1000, 60
    gen = torch.Generator(device="cuda").manual_seed(3)
    v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen) + 0.1
    g = torch.randn(B, T, sd, dtype=torch.float64, device="cuda", generator=gen)
    n0 = _hip.lib().mlpg_hip_launch_count(6)
    ga, st = _hip.backward(v, g, windows, 3 * sd, out_dtype=torch.float64)
    assert _hip.lib().mlpg_hip_launch_count(6) == n0 + 1 and int(st.abs().max()) == 0
    gg, _ = _hip.backward(v, g, windows, 3 * sd, out_dtype=torch.float64, algo=_hip.ALGO_GENERIC)
    assert float((ga - gg).abs().max()) <= 1e-10 * float(gg.abs().max())
    # adjointness with the forward pass of the same kernel: <forward(m), g> == <m, backward(g)>
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen)
    y, _ = _hip.forward(m, v, windows, algo=_hip.ALGO_CHUNK)
    lhs, rhs = float((y * g).sum()), float((m * ga).sum())
    assert abs(lhs - rhs) <= 1e-9 * abs(lhs)
