"""What a Python torch.autograd.Function costs in the eager config-3 loop, apart from what it computes: the reference's form
(unit_variance_mlpg + MSELoss + backward) with (a) this package's node, (b) a node that does the same two library calls with
nothing else around them, (c) a node that only slices and scales (no library call at all), (d) plain torch ops (no Python
node); and the time spent INSIDE this package's forward / backward bodies (perf_counter around them)."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip, autograd as AF, paramgen as G
from nnmnkwii_amd.autograd import _mlpg as IMPL

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
PW = _hip.prepack_windows(W3)
B, T, D = 64, 500, 180
dev = torch.device("cuda", 0)
R = torch.from_numpy(G.unit_variance_mlpg_matrix(W3, T)).to(dev)
means = torch.rand(B, T, D, device=dev, requires_grad=True)
target = torch.rand(B, T, D // 3, device=dev)
loss_fn = torch.nn.MSELoss()


class Bare(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m):
        out, _ = _hip.forward(m.detach(), None, PW, want_status=False)
        return out

    @staticmethod
    def backward(ctx, g):
        grad, _ = _hip.backward(None, g, PW, D, out_dtype=g.dtype, want_status=False)
        return grad


class Trivial(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m):
        return m[..., :60] * 2.0

    @staticmethod
    def backward(ctx, g):
        out = torch.zeros(B, T, D, device=g.device, dtype=g.dtype)
        out[..., :60] = g * 2.0
        return out


def loop(make_y):
    def step():
        means.grad = None
        loss_fn(make_y(), target).backward()
    return step


def wall(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


acc = {"fwd": 0.0, "bwd": 0.0, "n": 0}
_f, _b = IMPL.UnitVarianceMLPG.forward, IMPL.UnitVarianceMLPG.backward


def timed_fwd(ctx, m, r):
    t0 = time.perf_counter()
    y = _f(ctx, m, r)
    acc["fwd"] += time.perf_counter() - t0
    acc["n"] += 1
    return y


def timed_bwd(ctx, g):
    t0 = time.perf_counter()
    y = _b(ctx, g)
    acc["bwd"] += time.perf_counter() - t0
    return y


print("(a) this package's node            %.4f ms per step" % wall(loop(lambda: AF.unit_variance_mlpg(R, means))), flush=True)
print("(b) bare node, same library calls  %.4f ms per step" % wall(loop(lambda: Bare.apply(means))), flush=True)
print("(c) node that slices and scales    %.4f ms per step" % wall(loop(lambda: Trivial.apply(means))), flush=True)
print("(d) plain torch ops, no Python node %.4f ms per step" % wall(loop(lambda: means[..., :60] * 2.0)), flush=True)
IMPL.UnitVarianceMLPG.forward = staticmethod(timed_fwd)
IMPL.UnitVarianceMLPG.backward = staticmethod(timed_bwd)
w = wall(loop(lambda: AF.unit_variance_mlpg(R, means)))
print("(a) with timers: %.4f ms per step; inside forward %.1f us, inside backward %.1f us per step" % (w, 1e6 * acc["fwd"] / acc["n"], 1e6 * acc["bwd"] / acc["n"]), flush=True)
