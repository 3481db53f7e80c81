"""Phase cycle counters of the strip kernel's BACKWARD (build the bwd instantiations with -DMLPG_STRIP_TIMING)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import WINDOWS
dt = torch.float32 if "f32" in sys.argv else torch.float64
B, T, sd = 256, 1000, 60
v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
go = torch.randn(B, T, sd, dtype=dt, device="cuda")
for _ in range(3):
    g, st = _hip.backward(v, go, WINDOWS, 3 * sd, out_dtype=dt, algo=_hip.ALGO_STRIP, want_status=True)
torch.cuda.synchronize()
s = st.cpu().numpy()[:8 * 16 * 16].reshape(8, 16, 16)
names = "claim assemble eliminate barrier1 level2 publish poll barrier2 l3-stage l3-sweep l2-back barrier3 backsub store".split()
mean = s.reshape(-1, 16).mean(0)
print(str(dt), {n: int(x) for n, x in zip(names, mean)}, "sum", int(mean[:14].sum()))
from tools.bench_paths import gpu_time
print("kernel ms", gpu_time(lambda: _hip.backward(v, go, WINDOWS, 3 * sd, out_dtype=dt, algo=_hip.ALGO_STRIP, want_status=False), steps=20))
