#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_lit2
timeout 900 python -m pytest tests/test_literal_calls_gpu.py tests/test_autograd_gpu.py tests/test_parity_r2_gpu.py tests/test_host_multi_gpu.py -m gpu -q > ${O}_tests.log 2>&1
echo "pytest rc=$?" >> ${O}_tests.log
tail -25 ${O}_tests.log
echo "== single-utterance kernel times"
timeout 300 python tools/dbg/single_utt_kernels.py 2>&1 | tee ${O}_single_utt_kernels.txt
echo "== kernel trace of the literal calls"
rocprofv3 --kernel-trace --stats -d ${O}_trace -o run -- python - <<'PY' > ${O}_trace.log 2>&1
import sys, numpy as np
sys.path.insert(0, ".")
from nnmnkwii_amd import paramgen as G
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
for T, sd in ((100, 2), (1000, 60)):
    m, v = rng.randn(T, 3 * sd), rng.rand(T, 3 * sd) + 0.1
    for _ in range(100):
        G.mlpg(m, v, W)
PY
f=$(find ${O}_trace -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" > ${O}_trace.txt 2>&1; rm -rf ${O}_trace
head -30 ${O}_trace.txt
