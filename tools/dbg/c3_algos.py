"""Config-3 shape (64 x 500 x 180 float32, unit variances) forward / backward on every kernel that takes it."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


B, T, sd = 64, 500, 60
m = torch.rand(B, T, 3 * sd, dtype=torch.float32, device="cuda")
g = torch.randn(B, T, sd, dtype=torch.float32, device="cuda")
vg = torch.rand(3 * sd, dtype=torch.float32, device="cuda") + 0.1
for mode, var in (("unit", None), ("global", vg)):
    for name, algo in (("wave", 2), ("strip", 3), ("const", 5), ("chunk", 6)):
        f = timeit(lambda: _hip.forward(m, var, W3, None, algo=algo, want_status=False))
        b = timeit(lambda: _hip.backward(var, g, W3, 3 * sd, out_dtype=torch.float32, algo=algo, want_status=False))
        print("%-6s %-6s forward %.4f ms  backward %.4f ms  sum %.4f" % (mode, name, f, b, f + b))
