import sys, torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import WINDOWS, gpu_time
B, T, D = 64, 500, 180
md = torch.rand(B, T, D, device="cuda")
tg = torch.rand(B, T, D // 3, device="cuda")
for _ in range(30):
    _hip.unit_mse_step(md, tg, WINDOWS)
    _hip.forward(md, None, WINDOWS, algo=_hip.ALGO_WAVE, want_status=False)
torch.cuda.synchronize()
print("fused call %.4f ms" % gpu_time(lambda: _hip.unit_mse_step(md, tg, WINDOWS), steps=30))
