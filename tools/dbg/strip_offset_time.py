"""Does the relative placement of the means and the variances in HBM matter to the strip kernel?  (channel / bank interleaving: both arrays
are read at the same offsets at the same time.)  Config-2 shape; the variances live in one large buffer at several byte offsets."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
B, T, sd = 256, 1000, 60
n = B * T * 3 * sd
gen = torch.Generator(device="cuda").manual_seed(1234)
m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen)
big = torch.rand(n + (64 << 20) // 8, dtype=torch.float64, device="cuda", generator=gen) + 0.1


def timeit(fn, reps=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


print("means at 0x%x" % m.data_ptr())
for rnd in range(2):
    for off_bytes in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, (1 << 20) + 4096 + 256, 3 << 20, 33 << 20):
        v = big[off_bytes // 8: off_bytes // 8 + n].view(B, T, 3 * sd)
        t, tmin = timeit(lambda: _hip.forward(m, v, W3, None, algo=3, want_status=False))
        print("variances at +%9d bytes (0x%x; distance to the means mod 1 MiB: %7d): %.4f ms (min %.4f)"
              % (off_bytes, v.data_ptr(), (v.data_ptr() - m.data_ptr()) % (1 << 20), t, tmin), flush=True)
