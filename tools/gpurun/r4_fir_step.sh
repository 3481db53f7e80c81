#!/usr/bin/env bash
# the training step in the FIR form: parity (FIR + autograd tests) and the config-3 step time against the one-launch kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_fir_gpu.py tests/test_autograd_gpu.py -m gpu -x -q 2>&1 | tail -12
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
def timeit(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))
B, T, sd = 64, 500, 60
m = torch.rand(B, T, 3 * sd, device="cuda"); tg = torch.rand(B, T, sd, device="cuda")
print("config-3 step, float32 (FIR form, two launches): %.4f ms" % timeit(lambda: _hip.unit_mse_step(m, tg, W3)))
L = torch.full((B,), T, dtype=torch.int32, device="cuda")
print("config-3 step, float32 with a lengths vector (one-launch kernel): %.4f ms" % timeit(lambda: _hip.unit_mse_step(m, tg, W3, lengths=L)))
PY
