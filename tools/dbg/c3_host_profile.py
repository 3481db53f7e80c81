"""Where the host time of BASELINE config 3 through autograd goes (64 x 500 x 180 float32, unit variances):
wall clock per eager step of (a) the reference's form unit_variance_mlpg + MSELoss + backward, (b) the fused node,
(c) the same loop with the MLPG node replaced by a trivial torch op (the framework's own floor), and a cProfile of (a).
usage: python tools/dbg/c3_host_profile.py [reps]"""
import cProfile
import pstats
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip, autograd as AF, paramgen as G

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, T, D = 64, 500, 180
dev = torch.device("cuda", 0)
R = torch.from_numpy(G.unit_variance_mlpg_matrix(W3, T)).to(dev)
means = torch.rand(B, T, D, device=dev, requires_grad=True)
target = torch.rand(B, T, D // 3, device=dev)
loss_fn = torch.nn.MSELoss()


def step_ref():
    means.grad = None
    loss_fn(AF.unit_variance_mlpg(R, means), target).backward()


def step_fused():
    means.grad = None
    AF.unit_variance_mlpg_mse_loss(R, means, target).backward()


def step_floor():
    means.grad = None
    loss_fn(means[..., :60] * 2.0, target).backward()


def step_kernels_only():
    md = means.detach()
    _hip.forward(md, None, W3, want_status=False)
    _hip.backward(None, target, W3, D, out_dtype=torch.float32, want_status=False)


def wall(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (("reference form (two nodes + MSELoss)", step_ref), ("fused node", step_fused),
                 ("framework floor (slice * 2 + MSELoss)", step_floor), ("the two library calls alone", step_kernels_only)):
    print("%-42s %.4f ms per step (wall, %d steps back to back)" % (name, wall(fn, reps), reps), flush=True)

for name, fn in (("reference form", step_ref), ("the two library calls alone", step_kernels_only)):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    print("---- cProfile, %s, %d steps (tottime)" % (name, reps))
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
