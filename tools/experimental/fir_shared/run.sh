#!/usr/bin/env bash
# parity + timings of the LDS-shared FIR tiles (after applying mlpg_fir_shared.patch and rebuilding)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MLPG_FIR_SHARED=1 timeout 600 python -m pytest tests/test_fir_gpu.py -m gpu -x -q 2>&1 | tail -6
for sw in 0 1; do
echo "== MLPG_FIR_SHARED=$sw"
MLPG_FIR_SHARED=$sw python - <<'PY'
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
for (B, T, sd) in ((64, 500, 60), (256, 1000, 60)):
    m = torch.rand(B, T, 3 * sd, dtype=torch.float32, device="cuda")
    g = torch.randn(B, T, sd, dtype=torch.float32, device="cuda")
    tg = torch.rand(B, T, sd, dtype=torch.float32, device="cuda")
    f = timeit(lambda: _hip.forward(m, None, W3, None, algo=7, want_status=False))
    b = timeit(lambda: _hip.backward(None, g, W3, 3 * sd, out_dtype=torch.float32, algo=7, want_status=False))
    s = timeit(lambda: _hip.unit_mse_step(m, tg, W3))
    print("%d x %d x %d  forward %.4f ms  backward %.4f ms  step %.4f ms" % (B, T, sd, f, b, s))
PY
done
