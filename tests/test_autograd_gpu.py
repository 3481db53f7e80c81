"""GPU tests of the torch.autograd wrappers (-m gpu), modelled on the
reference's tests/test_autograd.py:39-218 plus golden vectors from it."""
import numpy as np
import pytest
import torch

from cases import WINDOW_SETS

pytestmark = pytest.mark.gpu


def _windows(name):
    return WINDOW_SETS[name]


def test_autograd_goldens(golden):
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    for k in golden.files:
        if k.startswith("autograd_uv/") and k.endswith("/y"):
            base = k[:-2]
            wname, B, T, sd = k.split("/")[1].split("-")
            T = int(T[1:])
            windows = _windows(wname)
            R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T))
            for dev in ("cuda", "cpu"):
                means = torch.tensor(golden[base + "/means"], device=dev, requires_grad=True)
                target = torch.tensor(golden[base + "/target"], device=dev)
                y = AF.unit_variance_mlpg(R.to(dev), means)
                assert y.device.type == dev and y.dtype == torch.float32
                torch.nn.MSELoss()(y, target).backward()
                assert np.abs(y.detach().cpu().numpy() - golden[base + "/y"]).max() <= 2e-5
                assert np.abs(means.grad.cpu().numpy() - golden[base + "/grad"]).max() <= 2e-6
        if k.startswith("autograd_mlpg/") and k.endswith("/y"):
            base = k[:-2]
            wname = k.split("/")[1].split("-")[0]
            windows = _windows(wname)
            for dev in ("cuda", "cpu"):
                means = torch.tensor(golden[base + "/means"], device=dev, requires_grad=True)
                v = torch.tensor(golden[base + "/vars"], device=dev)
                target = torch.tensor(golden[base + "/target"], device=dev)
                y = AF.mlpg(means, v, windows)
                assert y.dtype == torch.float32 and y.device.type == dev
                torch.nn.MSELoss()(y, target).backward()
                assert np.abs(y.detach().cpu().numpy() - golden[base + "/y"]).max() <= 2e-5
                assert np.abs(means.grad.cpu().numpy() - golden[base + "/grad"]).max() <= 2e-6


def test_functional_mlpg_matches_paramgen():
    # reference tests/test_autograd.py:39-72
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    static_dim, T = 2, 10
    torch.manual_seed(1234)
    for wname in ("static", "std2", "std3", "wide3"):
        windows = _windows(wname)
        nw = len(windows)
        means = torch.rand(T, static_dim * nw, requires_grad=True, device="cuda")
        variances = torch.ones(static_dim * nw, device="cuda")
        y = G.mlpg(means.detach().cpu().numpy(), variances.cpu().numpy(), windows)
        y_hat = AF.mlpg(means, variances, windows)
        assert np.allclose(y_hat.detach().cpu().numpy(), y, atol=1e-6)
        R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T)).cuda()
        y_hat2 = AF.unit_variance_mlpg(R, means)
        assert np.allclose(y_hat2.detach().cpu().numpy(), y, atol=1e-5)
        # reshaped means
        rm = torch.from_numpy(G.reshape_means(means.detach().cpu().numpy(), static_dim)).cuda().requires_grad_()
        y_hat3 = AF.unit_variance_mlpg(R, rm)
        assert np.allclose(y_hat3.detach().cpu().numpy(), y, atol=1e-5)
        y_hat3.sum().backward()
        assert rm.grad.shape == rm.shape
        # class-style call keeps working (reference tests call .apply directly)
        y4 = AF.UnitVarianceMLPG.apply(means, R)
        assert torch.allclose(y4, y_hat2)
        y5 = AF.MLPG.apply(means, variances.expand(T, static_dim * nw), windows)
        assert torch.allclose(y5, y_hat)


def test_gradcheck_float32_reference_settings():
    # reference tests/test_autograd.py:108-113,180-204: eps=1e-3, atol=1e-3 on float32
    from torch.autograd import gradcheck
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    static_dim, T = 2, 5
    torch.manual_seed(1234)
    for wname in ("std2", "std3", "wide3"):
        windows = _windows(wname)
        nw = len(windows)
        means = torch.rand(T, static_dim * nw, requires_grad=True, device="cuda")
        R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T)).cuda()
        assert gradcheck(AF.UnitVarianceMLPG.apply, (means, R), eps=1e-3, atol=1e-3)
        for variances in (torch.ones(T, static_dim * nw, device="cuda"),
                          torch.rand(T, static_dim * nw, device="cuda") + 0.5):
            assert gradcheck(lambda m: AF.MLPG.apply(m, variances, windows), (means,), eps=1e-3, atol=1e-3)


def test_batched_unit_variance():
    # reference tests/test_autograd.py:142-177: batch grads equal per-item grads
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    windows = _windows("std3")
    B, T, sd = 4, 50, 6
    torch.manual_seed(1234)
    R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T)).cuda()
    means = torch.rand(B, T, 3 * sd, device="cuda", requires_grad=True)
    y = AF.unit_variance_mlpg(R, means)
    assert y.shape == (B, T, sd)
    y.pow(2).sum().backward()
    for b in range(B):
        mb = means[b].detach().clone().requires_grad_()
        yb = AF.unit_variance_mlpg(R, mb)
        assert torch.allclose(yb, y[b], atol=1e-6)
        yb.pow(2).sum().backward()
        assert torch.allclose(mb.grad, means.grad[b], atol=1e-6)


def test_foreign_R_is_multiplied_densely():
    """An R that did not come from unit_variance_mlpg_matrix (here: random, and a modified genuine one) is
    just a matrix: y = R @ reshape(means), grads = R^T @ g, exactly the reference's dense definition
    (autograd/_impl/mlpg.py:138,158), for all four input layouts."""
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    torch.manual_seed(3)
    T, nw, sd, B = 7, 3, 2, 2
    WINDOWS = WINDOW_SETS["std3"]
    Rg = torch.from_numpy(G.unit_variance_mlpg_matrix(WINDOWS, T)).cuda()
    for R in (torch.rand(T, nw * T, device="cuda"), Rg * 1.5):
        for batched in (False, True):
            for reshaped in (False, True):
                shape = ((B,) if batched else ()) + ((nw * T, sd) if reshaped else (T, nw * sd))
                means = torch.rand(*shape, device="cuda", requires_grad=True)
                y = AF.unit_variance_mlpg(R, means)
                m3 = means.detach().reshape((B if batched else 1,) + shape[-2:])
                rm = m3 if reshaped else m3.view(-1, T, nw, sd).transpose(1, 2).reshape(-1, nw * T, sd)
                ref = torch.matmul(R, rm)
                assert y.shape == ((B, T, sd) if batched else (T, sd))
                assert torch.allclose(y.reshape(ref.shape), ref, atol=1e-5)
                g = torch.rand_like(y)
                y.backward(g)
                gr = torch.matmul(R.t(), g.reshape(ref.shape))                   # (b, nw*T, sd)
                if not reshaped:
                    gr = gr.view(-1, nw, T, sd).transpose(1, 2).reshape(-1, T, nw * sd)
                assert torch.allclose(means.grad.reshape(gr.shape), gr, atol=1e-5)
    # the genuine matrix still takes the banded kernels and agrees with its own dense product
    means = torch.rand(B, T, nw * sd, device="cuda")
    dense = torch.matmul(Rg, means.view(B, T, nw, sd).transpose(1, 2).reshape(B, nw * T, sd))
    assert torch.allclose(AF.unit_variance_mlpg(Rg, means), dense, atol=1e-5)


@pytest.mark.parametrize("dt", ["float32", "float64"])
@pytest.mark.parametrize("shape", [(64, 500, 60), (5, 37, 7), (3, 1000, 4), (2, 200, 25)])
def test_fused_unit_variance_mse_step_equals_two_node_form(shape, dt):
    """autograd.unit_variance_mlpg_mse_loss (one fused launch: mlpg_hip_unit_mse_step) against
    mse_loss(unit_variance_mlpg(R, means), target): same loss, same gradient; config-3 size included.  The float64 case
    is also checked against the dense float64 definition y = R mu."""
    import torch
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    B, T, sd = shape
    dtype = getattr(torch, dt)
    windows = WINDOW_SETS["std3"]
    torch.manual_seed(B * T)
    means = torch.rand(B, T, 3 * sd, dtype=dtype, device="cuda", requires_grad=True)
    target = torch.rand(B, T, sd, dtype=dtype, device="cuda")
    R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T)).cuda()
    loss_a = torch.nn.functional.mse_loss(AF.unit_variance_mlpg(R, means), target)
    loss_a.backward()
    ga = means.grad.clone()
    means.grad = None
    loss_b = AF.unit_variance_mlpg_mse_loss(R, means, target)
    loss_b.backward()
    gb = means.grad.clone()
    tol = 2e-6 if dt == "float32" else 1e-12
    assert abs(float(loss_a) - float(loss_b)) <= tol * abs(float(loss_a))
    assert float((ga - gb).abs().max()) <= tol * float(ga.abs().max()) + 1e-30
    # the window list instead of R, a scaled loss, repeatability
    means.grad = None
    (3.0 * AF.unit_variance_mlpg_mse_loss(windows, means, target)).backward()
    assert float((means.grad - 3.0 * gb).abs().max()) <= 1e-6 * float(gb.abs().max()) + 1e-30
    l2 = AF.unit_variance_mlpg_mse_loss(R, means, target)
    assert float(l2) == float(loss_b)
    if dt == "float64" and T <= 500:
        Rd = R.double().view(T, 3, T)
        y = torch.einsum("tws,bswd->btd", Rd, means.detach().view(B, T, 3, sd))
        ref = ((y - target) ** 2).mean()
        assert abs(float(ref) - float(loss_b)) <= 1e-6 * float(ref)     # R itself is a float32 matrix


@pytest.mark.parametrize("shape", [(64, 500, 60), (3, 211, 7)])
def test_fused_step_gradient_against_the_dense_float64_definition(shape):
    """mlpg_hip_unit_mse_step's GRADIENT against the definition itself, not against the two-launch HIP form:
    y = R mu (R = the float64 matrix P^-1 W~^T built densely on the CPU from the oracle's window matrices),
    loss = mean((y - target)^2), d loss / d mu = R^T 2 (y - target) / N, all in float64."""
    import torch
    from nnmnkwii_amd import autograd as AF
    from oracle import mlpg as O
    B, T, sd = shape
    windows = WINDOW_SETS["std3"]
    nw = len(windows)
    torch.manual_seed(7 * T)
    means = torch.rand(B, T, nw * sd, dtype=torch.float64, device="cuda", requires_grad=True)
    target = torch.rand(B, T, sd, dtype=torch.float64, device="cuda")
    loss = AF.unit_variance_mlpg_mse_loss(windows, means, target)
    loss.backward()
    # the dense definition on the CPU (paramgen/_mlpg.py:297-373 without the float32 cast)
    mask = O._edge_mask(T, 1)
    Ws = [O.window_matrix(l, u, c, T) for (l, u, c) in windows]
    Wt = [W if w == 0 else mask[:, None] * W for w, W in enumerate(Ws)]
    P = sum(Wt[w].T @ Ws[w] for w in range(nw))
    Rd = np.linalg.solve(P, np.concatenate([Wt[w].T for w in range(nw)], axis=1))       # (T, nw*T)
    mu = means.detach().cpu().numpy().reshape(B, T, nw, sd).transpose(0, 2, 1, 3).reshape(B, nw * T, sd)
    y = np.einsum("tk,bkd->btd", Rd, mu)
    tg = target.cpu().numpy()
    N = B * T * sd
    ref_loss = ((y - tg) ** 2).sum() / N
    gy = 2.0 * (y - tg) / N
    gmu = np.einsum("tk,btd->bkd", Rd, gy).reshape(B, nw, T, sd).transpose(0, 2, 1, 3).reshape(B, T, nw * sd)
    assert abs(float(loss) - ref_loss) <= 1e-11 * ref_loss
    g = means.grad.cpu().numpy()
    assert np.abs(g - gmu).max() <= 1e-10 * np.abs(gmu).max()


def test_fused_mse_loss_falls_back_where_the_fused_kernel_does_not_apply():
    """unit_variance_mlpg_mse_loss == mse_loss(unit_variance_mlpg(...), target) also for T > 1024, for window extents
    > 1, when the target wants a gradient and for CPU tensors (loss on means.device) -- against the dense float64
    definition from the oracle's window matrices."""
    import torch
    from nnmnkwii_amd import autograd as AF
    from oracle import mlpg as O

    def dense(windows, m, t):
        """paramgen/_mlpg.py:297-373 without the float32 cast: R = P^-1 [mask W_w]^T, y = R mu."""
        T = m.shape[-2]
        nw = len(windows)
        sd = m.shape[-1] // nw
        mw = int(max(max(l, u) for l, u, _ in windows))
        mask = O._edge_mask(T, mw)
        Ws = [O.window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for (l, u, c) in windows]
        Wt = [W if w == 0 else mask[:, None] * W for w, W in enumerate(Ws)]
        P = sum(Wt[w].T @ Ws[w] for w in range(nw))
        R = np.linalg.solve(P, np.concatenate([Wt[w].T for w in range(nw)], axis=1))       # (T, nw*T)
        mm = m.reshape(-1, T, nw, sd).transpose(0, 2, 1, 3).reshape(-1, nw * T, sd)
        y = np.einsum("tk,bkd->btd", R, mm)
        r = y - t.reshape(-1, T, sd)
        loss = (r ** 2).mean()
        g = np.einsum("tk,btd->bkd", R, 2.0 * r / r.size).reshape(-1, nw, T, sd).transpose(0, 2, 1, 3).reshape(m.shape)
        return loss, g, (-2.0 * r / r.size).reshape(t.shape)

    cases = [("std3", (2, 1100, 3), False, "cuda"),      # T > 1024
             ("wide3", (2, 90, 4), False, "cuda"),        # extents 2
             ("std3", (3, 50, 5), True, "cuda"),          # target.requires_grad
             ("std3", (40, 6), False, "cpu")]             # CPU tensors, 2-D
    for wname, shape, tgrad, dev in cases:
        windows = WINDOW_SETS[wname]
        nw = len(windows)
        torch.manual_seed(len(shape) + shape[-2])
        means = torch.rand(*shape[:-1], nw * shape[-1], dtype=torch.float64, device=dev, requires_grad=True)
        target = torch.rand(*shape, dtype=torch.float64, device=dev, requires_grad=tgrad)
        loss = AF.unit_variance_mlpg_mse_loss(windows, means, target)
        assert loss.device == means.device
        loss.backward()
        lo, go, gt = dense(windows, means.detach().cpu().numpy(), target.detach().cpu().numpy())
        assert abs(float(loss) - lo) <= 1e-9 * lo, wname
        assert np.abs(means.grad.cpu().numpy() - go).max() <= 1e-9 * np.abs(go).max(), wname
        if tgrad:
            assert np.abs(target.grad.cpu().numpy() - gt).max() <= 1e-12 * np.abs(gt).max()


def _dense_unit_mse(windows, m, t):
    """loss and d loss / d means of mse_loss(R mu, target) from the dense float64 definition (paramgen/_mlpg.py:297-373
    without the float32 cast), built from the oracle's window matrices."""
    from oracle import mlpg as O
    T = m.shape[-2]
    nw = len(windows)
    sd = m.shape[-1] // nw
    mw = int(max(max(l, u) for l, u, _ in windows))
    mask = O._edge_mask(T, mw)
    Ws = [O.window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for (l, u, c) in windows]
    Wt = [W if w == 0 else mask[:, None] * W for w, W in enumerate(Ws)]
    P = sum(Wt[w].T @ Ws[w] for w in range(nw))
    R = np.linalg.solve(P, np.concatenate([Wt[w].T for w in range(nw)], axis=1))
    mm = m.reshape(-1, T, nw, sd).transpose(0, 2, 1, 3).reshape(-1, nw * T, sd)
    r = np.einsum("tk,bkd->btd", R, mm) - t.reshape(-1, T, sd)
    g = np.einsum("tk,btd->bkd", R, 2.0 * r / r.size).reshape(-1, nw, T, sd).transpose(0, 2, 1, 3).reshape(m.shape)
    return (r ** 2).mean(), g


_SLOW = {
    # legal window sets whose P^-1 does NOT decay to 2^-26 within 24 frames: the FIR form's table test refuses them
    "dynamic-x4": [(0, 0, np.array([1.0])), (1, 1, 4.0 * np.array([-0.5, 0.0, 0.5])), (1, 1, 4.0 * np.array([1.0, -2.0, 1.0]))],
    "static-0.3": [(0, 0, np.array([0.3])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))],
    "wide3-x3": [(0, 0, np.array([1.0])), (2, 2, 3.0 * np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0),
                 (2, 2, 3.0 * np.array([-1.0, 16.0, -30.0, 16.0, -1.0]) / 12.0)],
}


@pytest.mark.parametrize("wname,T", [("std3", 1100), ("wide3", 200), ("wide3", 1100), ("dynamic-x4", 1100), ("dynamic-x4", 300),
                                     ("static-0.3", 1100), ("static-0.3", 128), ("wide3-x3", 1100)])
def test_fused_mse_loss_float32_routes_ask_the_library(wname, T):
    """Round-5 ADVICE: for float32 batches with T > 1024 or window extents of 2 the fused node used to be chosen by shape
    alone, and mlpg_hip_unit_mse_step then returned EINVAL for window sets that fail the FIR form's decay test.  Now
    mlpg_hip_unit_mse_form -- the function the step itself decides with -- is asked: whatever it says, the loss and the
    gradient equal the dense float64 definition (float32 tolerance), and the call never raises."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import autograd as AF
    windows = _SLOW.get(wname) or WINDOW_SETS[wname]
    nw = len(windows)
    B, sd = 3, 5
    torch.manual_seed(T)
    means = torch.rand(B, T, nw * sd, dtype=torch.float32, device="cuda", requires_grad=True)
    target = torch.rand(B, T, sd, dtype=torch.float32, device="cuda")
    form = _hip.unit_mse_form(means.device, torch.float32, False, B, T, nw * sd, windows)
    ext = max(max(l, u) for l, u, _ in windows)
    if wname in _SLOW:
        assert form != 2, "the FIR form must refuse a window set whose inverse decays this slowly"
    if form == 0:
        assert T > 1024 or ext > 1                  # (the one-launch kernel takes everything else)
    n0 = int(_hip.lib().mlpg_hip_launch_count(7)), int(_hip.lib().mlpg_hip_launch_count(5))
    loss = AF.unit_variance_mlpg_mse_loss(windows, means, target)
    loss.backward()
    n1 = int(_hip.lib().mlpg_hip_launch_count(7)), int(_hip.lib().mlpg_hip_launch_count(5))
    if form == 2:
        assert n1[0] - n0[0] == 2 and n1[1] == n0[1]      # the FIR form: two launches inside one call
    elif form == 1:
        assert n1[1] - n0[1] == 1                          # the one-launch fused kernel
    else:
        assert n1[1] == n0[1]                              # two-node form: no fused launch
    lo, go = _dense_unit_mse(windows, means.detach().cpu().numpy().astype(np.float64), target.cpu().numpy().astype(np.float64))
    assert abs(float(loss) - lo) <= 2e-5 * lo, (wname, T, form)
    assert np.abs(means.grad.cpu().numpy() - go).max() <= 2e-5 * np.abs(go).max(), (wname, T, form)
    # the R-matrix form of the call reaches the same route
    if T <= 300:
        from nnmnkwii_amd import paramgen as G
        R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T)).cuda()
        m2 = means.detach().clone().requires_grad_()
        l2 = AF.unit_variance_mlpg_mse_loss(R, m2, target)
        l2.backward()
        assert abs(float(l2) - lo) <= 2e-5 * lo
        assert np.abs(m2.grad.cpu().numpy() - go).max() <= 2e-5 * np.abs(go).max()


def test_fused_step_inside_a_stream_capture_needs_a_warm_up_and_says_so():
    """The step's workspace is never created or grown while its stream is being captured (round-5 ADVICE): the call raises
    a HipExtensionError that says what to do; after one eager step on the stream the capture works and replays."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import autograd as AF
    windows = WINDOW_SETS["std3"]
    B, T, sd = 4, 160, 6
    means = torch.rand(B, T, 3 * sd, device="cuda", requires_grad=True)
    target = torch.rand(B, T, sd, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        try:
            with pytest.raises((_hip.HipExtensionError, RuntimeError)):
                with torch.cuda.graph(g, stream=side):
                    AF.unit_variance_mlpg_mse_loss(windows, means, target)
        finally:
            torch.cuda.synchronize()
        for _ in range(2):                               # the warm-up the message asks for
            means.grad = None
            AF.unit_variance_mlpg_mse_loss(windows, means, target).backward()
        torch.cuda.synchronize()
        want_loss = float(AF.unit_variance_mlpg_mse_loss(windows, means, target))
        want_grad = means.grad.clone()
        means.grad = None
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            loss = AF.unit_variance_mlpg_mse_loss(windows, means, target)
            loss.backward()
        g2.replay()
        torch.cuda.synchronize()
        assert float(loss) == want_loss
        assert torch.equal(means.grad, want_grad)
