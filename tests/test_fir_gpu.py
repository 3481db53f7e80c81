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
