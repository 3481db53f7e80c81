import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nnmnkwii_amd import _hip
from tools.bench_paths import gpu_time, WINDOWS
dev = torch.device("cuda", 0)
B, T = 512, 2000
for ld in (198, 200, 192 + 16):
    m = torch.randn(B, T, ld, dtype=torch.float64, device=dev)
    v = torch.rand(B, T, ld, dtype=torch.float64, device=dev) + 0.1
    for name, st in (("mgc", [(0, 60, WINDOWS)]), ("lf0", [(180, 1, WINDOWS)]), ("bap", [(183, 5, WINDOWS)]), ("bap184", [(184, 4, WINDOWS)]),
                     ("all", [(0, 60, WINDOWS), (180, 1, WINDOWS), (183, 5, WINDOWS)])):
        ms = gpu_time(lambda: _hip.forward_streams(m, v, st, want_status=False), steps=5)
        print(ld, name, round(ms, 4), flush=True)
    del m, v
