#!/usr/bin/env bash
# Round 6: a shorter first staging piece of the short host path (MLPG_HIP_HOST_FIRST_KB), interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_firstpiece
: > ${O}.txt
for r in 1 2 3; do
for kb in 0 192 0 64 0 384; do
MLPG_HIP_HOST_FIRST_KB=$kb timeout 300 python - <<'PY' | tee -a ${O}.txt
import os, sys, time, numpy as np
sys.path.insert(0, ".")
from nnmnkwii_amd import paramgen as G
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
out = []
for T, sd, dt in ((1000, 60, np.float64), (1000, 60, np.float32), (2000, 60, np.float64)):
    m, v = rng.randn(T, 3 * sd).astype(dt), (rng.rand(T, 3 * sd) + 0.1).astype(dt)
    for _ in range(30): G.mlpg(m, v, W)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); G.mlpg(m, v, W); ts.append(time.perf_counter() - t0)
    out.append("T=%d %s: %.1f us (min %.1f)" % (T, np.dtype(dt).name, np.median(ts) * 1e6, np.min(ts) * 1e6))
print("first piece %4s KB: " % os.environ.get("MLPG_HIP_HOST_FIRST_KB") + "  |  ".join(out))
PY
done
done
