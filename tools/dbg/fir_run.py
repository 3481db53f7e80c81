"""Runs the FIR kernels at the config-3 shape a few times (for rocprofv3 --kernel-trace)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
B, T, sd = (int(a) for a in (sys.argv[1:4] or [64, 500, 60]))
m = torch.rand(B, T, 3 * sd, dtype=torch.float32, device="cuda")
g = torch.randn(B, T, sd, dtype=torch.float32, device="cuda")
ALGOS = [int(x) for x in (sys.argv[4:] or [7])]
for _ in range(20):
    _hip.forward(m, None, W3, None, algo=ALGOS[0], want_status=False)
    _hip.backward(None, g, W3, 3 * sd, out_dtype=torch.float32, algo=ALGOS[0], want_status=False)
torch.cuda.synchronize()
