"""GPU parity tests (-m gpu) of the FIR form of unit-variance MLPG on float32 tensors (algo = MLPG_HIP_ALGO_FIR: a 49-tap filter in the
interior, 24 table rows per end, every 32-frame tile independent), through the C ABI, against the CPU oracle (forward) and the
natural-order kernel / the oracle's dense mlpg_grad (backward)."""
import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

pytestmark = pytest.mark.gpu

TOL32 = 5e-6


def rel_err(y, ref):
    scale = np.abs(ref).max(axis=0, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float((np.abs(y.astype(np.float64) - ref.astype(np.float64)) / scale).max())


@pytest.mark.parametrize("wname", ["std3", "std2", "asym2", "wide3"])
@pytest.mark.parametrize("T", [96, 97, 100, 127, 128, 129, 160, 161, 333, 500, 1000])
def test_fir_forward_and_backward(wname, T):
    """Every tile count around the 32-frame boundaries, the shortest utterance it takes (96 frames: the two end tables meet), 70 static
    dims in two dim groups: forward against the oracle, backward against the natural-order kernel (== the reference's mlpg_grad)."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    B, sd = 3, 70 if T <= 200 else 12
    rng = np.random.RandomState(T + nw)
    M_ = rng.randn(B, T, nw * sd).astype(np.float32)
    y, st = _hip.forward(torch.from_numpy(M_).cuda(), None, windows, None, algo=_hip.ALGO_FIR)
    yo, _, rc = O.mlpg_batch(M_, np.ones(nw * sd, dtype=np.float32), windows)
    assert rc == 0 and int(st.abs().max()) == 0
    assert rel_err(y.cpu().numpy().reshape(-1, sd), yo.reshape(-1, sd)) <= TOL32, (wname, T)
    g = torch.from_numpy(rng.randn(B, T, sd).astype(np.float32)).cuda()
    gf, _ = _hip.backward(None, g, windows, nw * sd, out_dtype=torch.float32, algo=_hip.ALGO_FIR)
    gg, _ = _hip.backward(None, g, windows, nw * sd, out_dtype=torch.float32, algo=_hip.ALGO_GENERIC)
    assert float((gf - gg).abs().max()) <= TOL32 * float(gg.abs().max()), (wname, T)


def test_fir_is_the_auto_choice_for_config3_and_matches_the_dense_definition():
    """64 x 500 x 180 float32, unit variances: AUTO takes the FIR kernel forward and backward; y against the dense float64 R mu of the
    reference's definition (paramgen/_mlpg.py:297-373 without the float32 cast), the gradient against R^T g."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    B, T, sd = 64, 500, 60
    rng = np.random.RandomState(1234)
    M_ = rng.rand(B, T, 3 * sd).astype(np.float32)
    go = rng.randn(B, T, sd).astype(np.float32)
    n0 = _hip.lib().mlpg_hip_launch_count(7)
    y, _ = _hip.forward(torch.from_numpy(M_).cuda(), None, windows)
    g, _ = _hip.backward(None, torch.from_numpy(go).cuda(), windows, 3 * sd, out_dtype=torch.float32)
    assert _hip.lib().mlpg_hip_launch_count(7) == n0 + 2
    mask = O._edge_mask(T, 1)
    Ws = [O.window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for (l, u, c) in windows]
    Wt = [W if w == 0 else mask[:, None] * W for w, W in enumerate(Ws)]
    P = sum(Wt[w].T @ Ws[w] for w in range(3))
    R = np.linalg.solve(P, np.concatenate([Wt[w].T for w in range(3)], axis=1))       # (T, 3 T)
    sel = [0, 31, 63]
    mu = M_[sel].astype(np.float64).reshape(len(sel), T, 3, sd).transpose(0, 2, 1, 3).reshape(len(sel), 3 * T, sd)
    yd = np.einsum("tk,bkd->btd", R, mu)
    assert rel_err(y.cpu().numpy()[sel].reshape(-1, sd), yd.reshape(-1, sd)) <= TOL32
    gd = np.einsum("tk,btd->bkd", R, go[sel].astype(np.float64)).reshape(len(sel), 3, T, sd).transpose(0, 2, 1, 3).reshape(len(sel), T, 3 * sd)
    assert np.abs(g.cpu().numpy()[sel] - gd).max() <= TOL32 * np.abs(gd).max()


def test_fir_leaves_what_it_does_not_take_to_the_other_kernels():
    """Ragged batches, float64, short utterances, global variances: AUTO does not route them here, the explicit algo is refused."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    m = torch.rand(4, 200, 18, dtype=torch.float32, device="cuda")
    n0 = _hip.lib().mlpg_hip_launch_count(7)
    L = torch.tensor([200, 150, 96, 10], dtype=torch.int32, device="cuda")
    _hip.forward(m, None, windows, L)
    _hip.forward(m.double(), None, windows)
    _hip.forward(m[:, :64].contiguous(), None, windows)
    _hip.forward(m, torch.rand(18, device="cuda") + 0.1, windows)
    assert _hip.lib().mlpg_hip_launch_count(7) == n0
    with pytest.raises(_hip.HipExtensionError):
        _hip.forward(m, None, windows, L, algo=_hip.ALGO_FIR)
    _hip.forward(m, None, windows)
    assert _hip.lib().mlpg_hip_launch_count(7) == n0 + 1


def _dense_R(windows, T):
    """R = P^-1 [W~_w^T] in float64 from the oracle's window matrices (paramgen/_mlpg.py:297-373 without the float32 cast): (T, nw T)."""
    nw = len(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    mask = O._edge_mask(T, mw)
    Ws = [O.window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for (l, u, c) in windows]
    Wt = [W if w == 0 else mask[:, None] * W for w, W in enumerate(Ws)]
    P = sum(Wt[w].T @ Ws[w] for w in range(nw))
    return np.linalg.solve(P, np.concatenate([Wt[w].T for w in range(nw)], axis=1))


@pytest.mark.parametrize("wname", ["std3", "wide3"])
@pytest.mark.parametrize("shape", [(64, 500, 60), (3, 1000, 4), (2, 200, 25), (3, 97, 70), (2, 1500, 5)])
def test_fir_training_step_against_the_dense_float64_definition(shape, wname):
    """mlpg_hip_unit_mse_step on float32 batches without lengths runs in the FIR form (two launches; also for window extents of 2 and
    T > 1024, which the one-launch kernel does not take): loss, gradient and trajectory against y = R mu, loss = mean((y - target)^2),
    d loss / d mu = R^T 2 (y - target) / N in float64; the loss is bitwise repeatable; y is optional."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS[wname]
    nw = len(windows)
    B, T, sd = shape
    rng = np.random.RandomState(B + T + sd)
    M_ = rng.rand(B, T, nw * sd).astype(np.float32)
    tg = rng.rand(B, T, sd).astype(np.float32)
    means, target = torch.from_numpy(M_).cuda(), torch.from_numpy(tg).cuda()
    n0 = _hip.lib().mlpg_hip_launch_count(7)
    loss, grad, y, st = _hip.unit_mse_step(means, target, windows, want_y=True, want_status=True)
    assert _hip.lib().mlpg_hip_launch_count(7) == n0 + 2
    assert int(st.abs().max()) == 0
    sel = sorted({0, B // 2, B - 1})
    R = _dense_R(windows, T)
    mu = M_[sel].astype(np.float64).reshape(len(sel), T, nw, sd).transpose(0, 2, 1, 3).reshape(len(sel), nw * T, sd)
    yd = np.einsum("tk,bkd->btd", R, mu)
    assert rel_err(y.cpu().numpy()[sel].reshape(-1, sd), yd.reshape(-1, sd)) <= TOL32
    e = yd - tg[sel]
    gd = np.einsum("tk,btd->bkd", R, 2.0 * e / (B * T * sd)).reshape(len(sel), nw, T, sd).transpose(0, 2, 1, 3).reshape(len(sel), T, nw * sd)
    assert np.abs(grad.cpu().numpy()[sel] - gd).max() <= 4 * TOL32 * np.abs(gd).max()
    # the loss over the whole batch: from the kernel's own y (checked above on the selected utterances) in float64
    ref_loss = float(((y.cpu().numpy().astype(np.float64) - tg) ** 2).mean())
    assert abs(float(loss) - ref_loss) <= 1e-6 * ref_loss
    loss2, grad2, y2, _ = _hip.unit_mse_step(means, target, windows)
    assert y2 is None and float(loss2) == float(loss) and torch.equal(grad2, grad)


def test_fir_training_step_leaves_ragged_and_float64_batches_to_the_one_launch_kernel():
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    means = torch.rand(4, 200, 18, device="cuda")
    target = torch.rand(4, 200, 6, device="cuda")
    n0 = _hip.lib().mlpg_hip_launch_count(7)
    L = torch.tensor([200, 150, 96, 10], dtype=torch.int32, device="cuda")
    _hip.unit_mse_step(means, target, windows, lengths=L)
    _hip.unit_mse_step(means.double(), target.double(), windows)
    _hip.unit_mse_step(means[:, :64].contiguous(), target[:, :64].contiguous(), windows)
    assert _hip.lib().mlpg_hip_launch_count(7) == n0
    la, ga, _, _ = _hip.unit_mse_step(means, target, windows)
    assert _hip.lib().mlpg_hip_launch_count(7) == n0 + 2
    lb, gb, _, _ = _hip.unit_mse_step(means.double(), target.double(), windows)
    assert abs(float(la) - float(lb)) <= 2e-6 * float(lb)
    assert float((ga.double() - gb).abs().max()) <= 4 * TOL32 * float(gb.abs().max())
