#!/bin/bash
# round 6: mlpg_hip_backward_host (the literal paramgen.mlpg_grad call / autograd.MLPG on CPU tensors): tests, soak, timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_literal_calls_gpu.py tests/test_mlpg_gpu.py tests/test_autograd_gpu.py -x -q > gpurun_out/bwd_tests.log 2>&1
tail -n 15 gpurun_out/bwd_tests.log
timeout 300 python tools/dbg/lit_soak.py 60 7 > gpurun_out/bwd_soak.log 2>&1
tail -n 4 gpurun_out/bwd_soak.log
python tools/bench_paths.py --only lit > gpurun_out/bwd_lit.jsonl 2>gpurun_out/bwd_lit.err
grep -o '"path": "lit-c[12][a-z]*-autograd[^}]*' gpurun_out/bwd_lit.jsonl | cut -c1-420
MLPG_HIP_HOST_TRACE=1 python - > gpurun_out/bwd_trace.txt 2>&1 <<'PY'
import numpy as np
from nnmnkwii_amd import paramgen as G
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
for T, sd in ((100, 2), (1000, 60)):
    m = np.zeros((T, 3 * sd)); v = rng.rand(T, 3 * sd) + 0.1; go = rng.randn(T, sd)
    for _ in range(6):
        G.mlpg_grad(m, v, W, go)
PY
tail -n 4 gpurun_out/bwd_trace.txt
