"""BASELINE config 3 on CPU tensors (what the reference's own benchmark runs when use_cuda is off, perf/autograd_mlpg_perf.py:56-86:
R (500, 1500) float32, means (64, 500, 180) float32 CPU tensors): autograd.unit_variance_mlpg forward / forward + backward as shipped
(CPU tensors staged through torch device tensors), against the host-memory entry points on the same arrays (forward_host with unit
variances; backward_host), and the reference's dense matmul on one thread."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import autograd as AF  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
B, T, sd = 64, 500, 60
R = torch.from_numpy(G.unit_variance_mlpg_matrix(W, T))
means = torch.rand(B, T, 3 * sd, requires_grad=True)
y = torch.rand(B, T, sd)
crit = torch.nn.MSELoss()


def wall(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    best = 1e9
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, min(ts) * 1e3


def step():
    means.grad = None
    crit(AF.unit_variance_mlpg(R, means), y).backward()


print("autograd.unit_variance_mlpg(R, means) forward, CPU tensors:   %.3f ms (min %.3f)" % wall(lambda: AF.unit_variance_mlpg(R, means.detach())))
print("the reference's loop body (forward, MSELoss, backward), CPU:  %.3f ms (min %.3f)" % wall(step))
mn = means.detach().numpy()
gn = np.random.RandomState(0).randn(B, T, sd).astype(np.float32)
print("_hip.forward_host(means, None, windows) numpy -> numpy:       %.3f ms (min %.3f)" % wall(lambda: _hip.forward_host(mn, None, W)))
print("_hip.backward_host(None, grad_out, windows) numpy -> numpy:   %.3f ms (min %.3f)" % wall(lambda: _hip.backward_host(None, gn, W, 3 * sd, out_dtype=np.float32)))
torch.set_num_threads(1)
rm = means.detach().view(B, T, 3, -1).transpose(1, 2).contiguous().view(B, -1, sd)
print("the reference: torch.matmul(R, reshaped means), 1 thread:     %.1f ms" % wall(lambda: torch.matmul(R, rm), n=3, warm=1)[0])
