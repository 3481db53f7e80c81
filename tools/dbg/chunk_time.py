"""Chunked kernel (window extents of 2) against the natural-order kernel at the config-2 shape with the reference's 5-tap
windows; MLPG_HIP_CHUNK_SLAB_MB selects the slab size."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

WIDE3 = [(0, 0, np.array([1.0])), (2, 2, np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0), (2, 2, np.array([-1.0, 16.0, -30.0, 16.0, -1.0]) / 12.0)]


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


B, T, sd = 256, 1000, 60
for dt in (torch.float64, torch.float32):
    m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
    v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
    by = m.element_size() * 7.0 * sd * B * T
    for name, algo in (("generic", 1), ("chunk", 6)):
        med, mn = timeit(lambda: _hip.forward(m, v, WIDE3, None, algo=algo, want_status=False))
        print("%s %-8s %.4f ms (min %.4f)  %.0f GB/s  frac %.3f" % (str(dt)[6:], name, med, mn, by / med / 1e6, by / med / 1e6 / 8000))
