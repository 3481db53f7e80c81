"""Where the host time of BASELINE config 3's eager training step goes (64 x 500 x 180 float32, unit variances; the
reference's loop: perf/autograd_mlpg_perf.py:56-86): wall time per step of (a) the literal two-node loop
`criterion(AF.unit_variance_mlpg(R, means), y).backward()`, (b) the fused node `AF.unit_variance_mlpg_mse_loss`, (c) the
same loop with plain torch ops in place of the MLPG node (y_hat = means[..., :60] * 1.0), (d) the library call alone
(_hip.unit_mse_step), each over n steps with one synchronize at the end; then cProfile of (a) and (b).
usage: python tools/dbg/c3_eager_profile.py [n]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import autograd as AF  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
B, T, sd = 64, 500, 60
dev = torch.device("cuda:0")
R = torch.from_numpy(G.unit_variance_mlpg_matrix(W, T)).to(dev)
means = torch.rand(B, T, 3 * sd, device=dev, requires_grad=True)
y = torch.rand(B, T, sd, device=dev)
criterion = torch.nn.MSELoss()


def two_node():
    means.grad = None
    criterion(AF.unit_variance_mlpg(R, means), y).backward()


def fused():
    means.grad = None
    AF.unit_variance_mlpg_mse_loss(R, means, y).backward()


def plain():
    means.grad = None
    criterion(means[..., :sd] * 1.0, y).backward()


md = means.detach()


def call_only():
    _hip.unit_mse_step(md, y, W)


def wall(fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


for name, fn in (("two-node literal loop", two_node), ("fused node", fused), ("plain torch ops, no MLPG node", plain), ("_hip.unit_mse_step alone", call_only)):
    print("%-34s %.1f us per step" % (name, wall(fn)))
for name, fn in (("two-node literal loop", two_node), ("fused node", fused)):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    print("---- cProfile, %s, %d steps" % (name, n))
    pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(22)
