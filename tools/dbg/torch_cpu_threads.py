"""How torch's CPU thread pool (128 OpenMP threads on this box by default) interacts with small CPU-tensor ops and with this library's
host-memory calls: x.sum() on 60 k floats, the literal paramgen.mlpg call before and after the first parallel torch op, at the
default thread count and after torch.set_num_threads(1)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
m = rng.randn(1000, 180)
v = rng.rand(1000, 180) + 0.1


def t(fn, n=200, warm=10):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


print("torch threads %d, OMP_NUM_THREADS=%s" % (torch.get_num_threads(), os.environ.get("OMP_NUM_THREADS")))
print("paramgen.mlpg before any torch CPU op:            %.1f us" % t(lambda: G.mlpg(m, v, W)))
x = torch.rand(1000, 60)
big = torch.rand(1000, 180)
print("x.sum() on 60 k floats:                           %.1f us" % t(lambda: x.sum()))
print("paramgen.mlpg after those:                        %.1f us" % t(lambda: G.mlpg(m, v, W)))
print("(big * 2).sum() on 180 k floats:                  %.1f us" % t(lambda: (big * 2).sum()))
print("paramgen.mlpg after those:                        %.1f us" % t(lambda: G.mlpg(m, v, W)))
y = torch.rand(1000, 180, requires_grad=True)


def fb():
    y.grad = None
    (y * 2.0)[:, :60].sum().backward()


print("a CPU autograd step on (1000, 180):               %.1f us" % t(fb))
print("paramgen.mlpg after those:                        %.1f us" % t(lambda: G.mlpg(m, v, W)))
torch.set_num_threads(1)
print("-- torch.set_num_threads(1)")
print("x.sum():                                          %.1f us" % t(lambda: x.sum()))
print("a CPU autograd step:                              %.1f us" % t(fb))
print("paramgen.mlpg:                                    %.1f us" % t(lambda: G.mlpg(m, v, W)))
