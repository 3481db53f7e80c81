#!/usr/bin/env bash
# Round 6: the short path's staging helper threads (MLPG_HIP_HOST_HELPERS = 0 / 3), interleaved processes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_helpers_ab
: > ${O}.txt
for r in 1 2 3 4; do
for h in 0 3; do
MLPG_HIP_HOST_HELPERS=$h timeout 300 python - <<'PY' | tee -a ${O}.txt
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
        t0 = time.perf_counter(); y = G.mlpg(m, v, W); ts.append(time.perf_counter() - t0)
    out.append("T=%d %s %.1f" % (T, np.dtype(dt).name[-2:], np.median(ts) * 1e6))
utts = [(rng.randn(1000, 180), rng.rand(1000, 180) + 0.1) for _ in range(256)]
for m, v in utts[:8]: G.mlpg(m, v, W)
t0 = time.perf_counter(); ys = [G.mlpg(m, v, W) for m, v in utts]; loop = time.perf_counter() - t0
t0 = time.perf_counter()
for m, v in utts: y = G.mlpg(m, v, W)
loop2 = time.perf_counter() - t0
print("helpers %s: " % os.environ.get("MLPG_HIP_HOST_HELPERS") + " | ".join(out) + " | loop keeping results %.1f us/call, dropping them %.1f us/call" % (loop / 256 * 1e6, loop2 / 256 * 1e6))
PY
done
done
