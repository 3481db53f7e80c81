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


def test_foreign_R_rejected_loudly():
    from nnmnkwii_amd import HipExtensionError
    from nnmnkwii_amd import autograd as AF
    R = torch.rand(5, 15, device="cuda")
    with pytest.raises(HipExtensionError):
        AF.unit_variance_mlpg(R, torch.rand(5, 6, device="cuda"))
