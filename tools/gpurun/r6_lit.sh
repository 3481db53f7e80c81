#!/usr/bin/env bash
# Round 6: the short path of mlpg_hip_forward_host (one small call) -- parity / route tests, the literal calls timed beside the
# reference (bench_paths --only lit), where the call's time goes (MLPG_HIP_HOST_TRACE), flag polling against hipStreamSynchronize.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_lit
timeout 900 python -m pytest tests/test_literal_calls_gpu.py tests/test_autograd_gpu.py tests/test_parity_r2_gpu.py tests/test_host_multi_gpu.py -m gpu -q -x > ${O}_tests.log 2>&1
echo "pytest rc=$?" >> ${O}_tests.log
tail -15 ${O}_tests.log
echo "== literal calls"
timeout 600 python tools/bench_paths.py --only lit > ${O}.jsonl 2>${O}.err; echo "rc=$?"
cat ${O}.jsonl; tail -5 ${O}.err
echo "== literal calls, hipStreamSynchronize instead of the flag word"
MLPG_HIP_HOST_SMALL_WAIT=sync timeout 600 python tools/bench_paths.py --only lit --quick > ${O}_sync.jsonl 2>${O}_sync.err; echo "rc=$?"
cut -c1-330 ${O}_sync.jsonl
echo "== python profile"
timeout 300 python tools/dbg/lit_profile.py > ${O}_profile.txt 2>&1; echo "rc=$?"
head -64 ${O}_profile.txt
echo "== host trace"
MLPG_HIP_HOST_TRACE=1 timeout 120 python - <<'PY' 2>&1 | tail -24
import sys, numpy as np
sys.path.insert(0, ".")
from nnmnkwii_amd import paramgen as G
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
for T, sd in ((100, 2), (1000, 60), (2000, 60)):
    m, v = rng.randn(T, 3 * sd), rng.rand(T, 3 * sd) + 0.1
    for _ in range(6):
        G.mlpg(m, v, W)
PY
