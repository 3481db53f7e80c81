#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_chunk_gpu.py -m gpu -x -q 2>&1 | tail -12
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
WIDE3 = [(0, 0, np.array([1.0])), (2, 2, np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0), (2, 2, np.array([-1.0, 16.0, -30.0, 16.0, -1.0]) / 12.0)]
B, T, sd = 256, 1000, 60
v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda") + 0.1
g = torch.randn(B, T, sd, dtype=torch.float64, device="cuda")
for name, algo in (("generic", 1), ("chunk", 6)):
    for _ in range(3): _hip.backward(v, g, WIDE3, 3 * sd, out_dtype=torch.float64, algo=algo, want_status=False)
    torch.cuda.synchronize()
    evs = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _hip.backward(v, g, WIDE3, 3 * sd, out_dtype=torch.float64, algo=algo, want_status=False); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    print("backward f64 %-8s %.4f ms  frac %.3f" % (name, ms, 56.0 * sd * B * T / ms / 1e6 / 8000))
PY
