"""Config-2 shape (256 x 1000 x 180, per-frame variances, the 3-tap windows) forward / backward, float64 and float32, on the strip,
wave and chunked kernels."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


B, T, sd = 256, 1000, 60
for dt in (torch.float64, torch.float32):
    m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
    v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
    g = torch.randn(B, T, sd, dtype=dt, device="cuda")
    for name, algo in (("strip", 3), ("wave", 2), ("chunk", 6)):
        f = timeit(lambda: _hip.forward(m, v, W3, None, algo=algo, want_status=False))
        b = timeit(lambda: _hip.backward(v, g, W3, 3 * sd, out_dtype=dt, algo=algo, want_status=False))
        print("%s %-6s forward %.4f ms  backward %.4f ms" % (str(dt)[6:], name, f, b))
