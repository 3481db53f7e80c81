import sys, torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import WINDOWS, gpu_time
B, T, sd = 256, 1000, 60
for dt in (torch.float64, torch.float32):
    m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
    v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
    for name, algo in (("strip", _hip.ALGO_STRIP), ("pipe", _hip.ALGO_PIPE)):
        ms = gpu_time(lambda: _hip.forward(m, v, WINDOWS, algo=algo, want_status=False), steps=30, warmup=5)
        by = (56.0 if dt == torch.float64 else 28.0) * sd * B * T
        print("config 2 forward %s %s: %.4f ms  frac %.3f" % (str(dt)[6:], name, ms, by / ms / 1e6 / 8000), flush=True)
