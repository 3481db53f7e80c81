"""Differentiable modulation spectrum on PyTorch-ROCm tensors.

Host-side mirror of /root/reference/nnmnkwii/autograd/_impl/modspec.py:9-72.  Forward =
``mlpg_hip_modspec``; backward = ``mlpg_hip_modspec_backward`` (forward FFT, multiply by the incoming
gradient, one-sided inverse FFT -- instead of the reference's Python loop over feature dimensions
with dense ``(n/2+1, T)`` cosine / sine tables).
"""
import torch
from torch.autograd import Function

from .. import _hip
from ..preprocessing.modspec import _check_n, _norm_flag


class ModSpec(Function):
    """Modulation spectrum computation ``f : (T, D) -> (N//2+1, D)``; gradient w.r.t. ``y`` only."""

    @staticmethod
    def forward(ctx, y, n, norm):
        assert y.dim() == 2
        ctx.n = _check_n(n)
        ctx.norm = norm
        ctx.save_for_backward(y)
        dev = _hip.require_gpu(y.device if y.is_cuda else None)
        # np.fft.rfft(y, n) crops a longer signal to its first n frames (autograd/_impl/modspec.py:30-35)
        ms, _ = _hip.modspec(y.detach()[:ctx.n].to(dev)[None], ctx.n, _norm_flag(norm))
        return ms[0].to(device=y.device, dtype=y.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        (y,) = ctx.saved_tensors
        T, D = y.size()
        assert grad_output.size() == torch.Size((ctx.n // 2 + 1, D))
        dev = _hip.require_gpu(y.device if y.is_cuda else None)
        Tc = min(T, ctx.n)
        g = _hip.modspec_backward(y.detach()[:Tc].to(dev)[None], grad_output.detach().to(dev)[None], ctx.n, _norm_flag(ctx.norm))
        g = g[0].to(device=y.device, dtype=y.dtype)
        if Tc < T:  # frames beyond the DFT length do not reach the spectrum: zero gradient
            g = torch.cat([g, g.new_zeros((T - Tc, D))], dim=0)
        return g, None, None


def modspec(y, n=2048, norm=None):
    """Modulation spectrum of a ``(T, D)`` tensor (autograd/_impl/modspec.py:63-72)."""
    return ModSpec.apply(y, n, norm)
