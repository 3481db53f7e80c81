"""Where the time of autograd.mlpg(...).sum().backward() on CPU tensors goes at one config-2 utterance (T = 1000, 60 static dims, float32):
cProfile of 200 steps; the two library calls alone."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import autograd as AF  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
torch.manual_seed(1234)
T, sd = 1000, 60
mt = torch.rand(T, 3 * sd, requires_grad=True)
vt = torch.rand(T, 3 * sd) + 0.1


def fb():
    mt.grad = None
    AF.mlpg(mt, vt, W).sum().backward()


for _ in range(20):
    fb()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    fb()
print("forward + backward: %.1f us per step" % ((time.perf_counter() - t0) / n * 1e6))
mn, vn = mt.detach().numpy(), vt.numpy()
go = np.ones((T, sd), dtype=np.float32)
t0 = time.perf_counter()
for _ in range(n):
    G.mlpg(mn, vn, W)
t1 = time.perf_counter()
for _ in range(n):
    G.mlpg_grad(mn, vn, W, go)
t2 = time.perf_counter()
print("paramgen.mlpg %.1f us, paramgen.mlpg_grad %.1f us" % ((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    fb()
pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(18)
