"""Phase timers of the pipelined kernel (build with -DMLPG_PIPE_TIMING): mean cycles per item per wavefront role."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import WINDOWS, gpu_time
dt = torch.float32 if "f32" in sys.argv else torch.float64
B, T, sd = 256, 1000, 60
m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
for _ in range(3):
    out, st = _hip.forward(m, v, WINDOWS, algo=_hip.ALGO_PIPE)
torch.cuda.synchronize()
s = st.cpu().numpy()[:64 * 4 * 8].reshape(64, 4, 8)
print("chunk wavefronts (rotate, level-1 body, wait u, backsub+stores, next ticket+prologue) cycles/item; items:")
for w in range(3):
    print("  wave", w, s[:, w, :5].mean(0).astype(int), "sum", int(s[:, w, :5].mean(0).sum()), "items", s[:, w, 7].mean())
print("chain wavefront (ticket, polls, level3+l2back, level2+publish, idle) cycles/item:", s[:, 3, :5].mean(0).astype(int), "sum", int(s[:, 3, :5].mean(0).sum()), "items", s[:, 3, 7].mean())
x = st.cpu().numpy()[64 * 4 * 8: 64 * 4 * 8 + 64 * 4].reshape(64, 4)
print("  finish detail (issue staging loads, loads landed + LDS writes, sweep) cycles/item:", x[:, :3].mean(0).astype(int), "(the rest of level3+l2back = decode + level-2 back-substitution)")
ms = gpu_time(lambda: _hip.forward(m, v, WINDOWS, algo=_hip.ALGO_PIPE, want_status=False), steps=30, warmup=5)
print("kernel %.4f ms -> %.0f cycles per item at 2.4 GHz (21 items per workgroup)" % (ms, ms * 1e-3 * 2.4e9 / 21))
