#!/usr/bin/env bash
# FIR form of unit-variance MLPG on float32 tensors: parity and config-3 / config-2-shape timings per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_fir_gpu.py -m gpu -x -q 2>&1 | tail -12
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
for (B, T, sd) in ((64, 500, 60), (256, 1000, 60)):
    m = torch.rand(B, T, 3 * sd, dtype=torch.float32, device="cuda")
    g = torch.randn(B, T, sd, dtype=torch.float32, device="cuda")
    for name, algo in (("wave", 2), ("const", 5), ("fir", 7)):
        try:
            f = timeit(lambda: _hip.forward(m, None, W3, None, algo=algo, want_status=False))
            b = timeit(lambda: _hip.backward(None, g, W3, 3 * sd, out_dtype=torch.float32, algo=algo, want_status=False))
            print("%d x %d x %d  %-6s forward %.4f ms  backward %.4f ms  sum %.4f" % (B, T, sd, name, f, b, f + b))
        except Exception as e:
            print(name, "ERR", str(e)[:100])
PY
