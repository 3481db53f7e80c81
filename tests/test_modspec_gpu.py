"""modulation spectrum kernels (LDS-resident FFT) against goldens produced by the
reference's preprocessing/modspec.py and autograd/_impl/modspec.py, plus the reference's own test
assertions (tests/test_preprocessing.py:504-546) and size-independent properties at full size."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "mlpg_golden.npz"))


def _close(a, b, rel):
    scale = max(np.abs(b).max(), 1e-300)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= rel * scale, np.abs(a - b).max() / scale


@pytest.mark.parametrize("T,n", [(10, 16), (64, 64), (50, 128), (300, 1024), (1000, 4096)])
def test_modspec_inverse_and_smoothing_match_reference(golden, T, n):
    from nnmnkwii_amd.preprocessing import inv_modspec, modspec, modspec_smoothing
    x = golden["modspec/T%d-n%d/x" % (T, n)]
    for norm in (None, "ortho"):
        key = "modspec/T%d-n%d/%s" % (T, n, norm or "none")
        ms, ph = modspec(x, n=n, norm=norm, return_phase=True)
        assert ms.dtype == np.float64 and ph.dtype == np.complex128
        _close(ms, golden[key + "/ms"], 1e-12)
        # the phase of a bin is ill-conditioned where the bin is ~0: compare amplitude-weighted
        amp = np.sqrt(golden[key + "/ms"])
        _close(ph * amp, golden[key + "/phase"] * amp, 1e-11)
        _close(inv_modspec(golden[key + "/ms"], golden[key + "/phase"], norm=norm), golden[key + "/inv"], 1e-12)
        for log_domain in (True, False):
            for cutoff in (100, 25, 60):
                y = modspec_smoothing(x, 200, n=n, norm=norm, cutoff=cutoff, log_domain=log_domain)
                assert y.flags["C_CONTIGUOUS"] and y.shape == x.shape
                _close(y, golden[key + "/smooth-log%d-c%s" % (log_domain, cutoff)], 1e-11)
    assert modspec(x, n=n).shape == (n // 2 + 1, 3)


def test_float32_inputs_and_batches(golden):
    from nnmnkwii_amd.preprocessing import modphase, modspec, modspec_smoothing
    x32 = golden["modspec/f32/x"]
    ms = modspec(x32, n=64)
    assert ms.dtype == np.float32                       # numpy >= 2 keeps float32 through rfft
    _close(ms, golden["modspec/f32/ms"], 2e-6)
    ys = modspec_smoothing(x32, 200, n=64, cutoff=30)
    assert ys.dtype == np.float32
    _close(ys, golden["modspec/f32/smooth"], 5e-6)
    assert modphase(x32, n=64).dtype == np.complex64
    # a (B, T, D) batch, numpy or CUDA, is the stack of the per-utterance results
    xb = np.random.RandomState(0).rand(3, 70, 4)
    yb = modspec_smoothing(xb, 200, n=128, cutoff=40)
    for b in range(3):
        np.testing.assert_array_equal(yb[b], modspec_smoothing(xb[b], 200, n=128, cutoff=40))
    yt = modspec_smoothing(torch.from_numpy(xb).cuda(), 200, n=128, cutoff=40)
    assert yt.is_cuda and np.array_equal(yt.cpu().numpy(), yb)


def test_reference_assertions():
    """tests/test_preprocessing.py:504-546 of the reference, verbatim in spirit."""
    from nnmnkwii_amd import preprocessing as P
    np.random.seed(1234)
    generated = np.random.rand(64, 2)
    for n in [64, 128]:
        ms, phase = P.modspec(generated, n=n, return_phase=True)
        assert np.allclose(generated, P.inv_modspec(ms, phase)[:64])
    y = np.random.rand(64, 2)
    modfs = 200
    for log_domain in [True, False]:
        for norm in [None, "ortho"]:
            for n in [1024, 2048]:
                y_hat = P.modspec_smoothing(y, modfs, n=n, norm=norm, cutoff=modfs // 2, log_domain=log_domain)
                assert np.allclose(y, y_hat)
                P.modspec_smoothing(y, modfs, n=n, norm=norm, cutoff=modfs // 4, log_domain=log_domain)
    with pytest.raises(ValueError):
        P.modspec_smoothing(y, modfs, n=2048, cutoff=modfs // 2 + 1)
    with pytest.raises(RuntimeError):
        P.modspec_smoothing(y, modfs, n=32, cutoff=modfs // 2)
    assert P.modspec(y, n=100).shape == (51, 2)          # any DFT length, as numpy


@pytest.mark.parametrize("T,n", [(16, 16), (12, 32), (40, 256)])
def test_autograd_modspec_matches_reference(golden, T, n):
    from nnmnkwii_amd import autograd as AF
    for norm in (None, "ortho"):
        key = "modspec_grad/T%d-n%d-%s" % (T, n, norm or "none")
        w = torch.from_numpy(golden[key + "/w"]).cuda()
        for dev in ("cuda", "cpu"):
            y = torch.from_numpy(golden[key + "/y"]).to(dev).requires_grad_()
            ms = AF.modspec(y, n=n, norm=norm)
            assert ms.shape == (n // 2 + 1, 4) and ms.dtype == torch.float32 and ms.device == y.device
            _close(ms.detach().cpu().numpy(), golden[key + "/ms"], 2e-6)
            (ms * w.to(dev)).sum().backward()
            _close(y.grad.cpu().numpy(), golden[key + "/grad"], 5e-6)
    # float64 gradcheck (the reference keeps its own gradchecks commented out, tests/test_autograd.py:221-240)
    y = torch.rand(8, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: AF.ModSpec.apply(t, 16, None), (y,), eps=1e-6, atol=1e-6)
    assert torch.autograd.gradcheck(lambda t: AF.ModSpec.apply(t, 16, "ortho"), (y,), eps=1e-6, atol=1e-6)


@pytest.mark.parametrize("T,n", [(5, 6), (7, 7), (64, 100), (100, 100), (333, 1000), (1000, 2047), (1500, 3000),
                                 (2500, 5000), (3000, 8192), (50, 4097)])
def test_any_dft_length(T, n):
    """DFT lengths outside the in-LDS FFT (not a power of two, or > 4096): the direct transform against the numpy
    restatement of the reference (oracle/modspec.py, pinned on the reference's goldens): spectrum, phase, inverse,
    smoothing (odd n: the reference inverts at n - 1), analytic gradient; both norms."""
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import preprocessing as P
    from oracle import modspec as OM
    rng = np.random.RandomState(T + n)
    x = rng.rand(T, 5)
    for norm in (None, "ortho"):
        ms, ph = P.modspec(x, n=n, norm=norm, return_phase=True)
        mo, po = OM.modspec(x, n=n, norm=norm, return_phase=True)
        _close(ms, mo, 1e-11)
        big = mo > 1e-6 * mo.max()                        # the phase of a vanishing bin is noise
        assert np.abs(ph - po)[big].max() < 1e-8
        if n % 2 == 0:
            _close(P.inv_modspec(mo, po, norm=norm), OM.inv_modspec(mo, po, norm=norm), 1e-11)
        for log_domain in (True, False):
            for cutoff in (100, 25, 60):
                y = P.modspec_smoothing(x, 200, n=n, norm=norm, cutoff=cutoff, log_domain=log_domain)
                _close(y, OM.modspec_smoothing(x, 200, n=n, norm=norm, cutoff=cutoff, log_domain=log_domain), 1e-9)
        w = rng.rand(n // 2 + 1, 5)
        yt = torch.from_numpy(x).cuda().requires_grad_()
        (AF.modspec(yt, n=n, norm=norm) * torch.from_numpy(w).cuda()).sum().backward()
        _close(yt.grad.cpu().numpy(), OM.modspec_grad(x, w, n, norm), 1e-10)
    # batches, a column count that is not a multiple of the tile, float32 round trip
    xb = rng.rand(3, T, 21)
    msb = P.modspec(xb, n=n)
    for b in range(3):
        _close(msb[b], OM.modspec(xb[b], n=n), 1e-11)
    assert P.modspec(xb.astype(np.float32), n=n).dtype == np.float32


def test_direct_transform_equals_fft_path():
    """The same power-of-two problems through both transforms (mlpg_hip_modspec_set_direct): all four modes."""
    from nnmnkwii_amd import _hip
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(4, 700, 37, dtype=torch.float64, device="cuda", generator=gen)
    g = torch.rand(4, 513, 37, dtype=torch.float64, device="cuda", generator=gen)
    n = 1024

    def run():
        ms, ph = _hip.modspec(x, n, want_phase=True)
        return (ms, ph, _hip.inv_modspec(ms, ph), _hip.modspec_smoothing(x, n, 100, log_domain=True),
                _hip.modspec_smoothing(x, n, 100, log_domain=False, ortho=True), _hip.modspec_backward(x, g, n),
                _hip.modspec_backward(x, g, n, True))
    a = run()
    _hip.lib().mlpg_hip_modspec_set_direct(1)
    try:
        b = run()
    finally:
        _hip.lib().mlpg_hip_modspec_set_direct(0)
    for i, (u, v) in enumerate(zip(a, b)):
        if i == 1:
            continue                                      # phases of tiny bins differ; compared through the inverse
        assert float((u - v).abs().max()) <= 1e-10 * float(v.abs().max()), i


def test_full_size_properties():
    """Config-2 sized batch (256 x 1000 x 60), n = 4096: Parseval, linearity and idempotence of the
    band removal -- size-independent checks that need no CPU transform."""
    from nnmnkwii_amd import _hip
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(256, 1000, 60, dtype=torch.float64, device="cuda", generator=gen)
    n = 4096
    ms, _ = _hip.modspec(x, n)
    # Parseval for a real signal: sum x^2 = (ms[0] + 2 sum_{0<k<n/2} ms[k] + ms[n/2]) / n
    lhs = (x * x).sum(dim=1)
    rhs = (ms[:, 0] + 2.0 * ms[:, 1:n // 2].sum(dim=1) + ms[:, n // 2]) / n
    assert torch.allclose(lhs, rhs, rtol=1e-11, atol=0)
    lim = 500
    s1 = _hip.modspec_smoothing(x, n, lim, log_domain=False)
    s2 = _hip.modspec_smoothing(s1, n, lim, log_domain=False)
    # the linear band removal followed by truncation to T frames is not a projection, but removing
    # nothing is the identity and the operator is linear
    ident = _hip.modspec_smoothing(x, n, n // 2 + 1, log_domain=False)
    assert torch.allclose(ident, x, rtol=0, atol=1e-12)
    y = torch.randn(256, 1000, 60, dtype=torch.float64, device="cuda", generator=gen)
    lin = _hip.modspec_smoothing(2.0 * x - 3.0 * y, n, lim, log_domain=False)
    assert torch.allclose(lin, 2.0 * s1 - 3.0 * _hip.modspec_smoothing(y, n, lim, log_domain=False), rtol=0, atol=1e-11)
    assert s2.shape == s1.shape and torch.isfinite(s2).all()
    # smoothing lowers the high-band power of the padded trajectory
    hi_before = _hip.modspec(x, n)[0][:, lim:].sum()
    hi_after = _hip.modspec(s1, n)[0][:, lim:].sum()
    assert hi_after < 0.2 * hi_before
