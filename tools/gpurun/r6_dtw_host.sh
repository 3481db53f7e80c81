#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dtw_gpu.py tests/test_align_gpu.py tests/test_host_multi_gpu.py tests/test_soak_gpu.py -x -q > gpurun_out/dtw_tests.log 2>&1
tail -n 4 gpurun_out/dtw_tests.log
python tools/bench_paths.py --only lit > gpurun_out/dtw_lit.jsonl 2>gpurun_out/dtw_lit.err
grep -o '"path": "lit-dtw[^}]*' gpurun_out/dtw_lit.jsonl | cut -c1-300
MLPG_HIP_HOST_TRACE=1 python - 2>&1 <<'PY' | tail -n 3
import numpy as np
from nnmnkwii_amd.preprocessing.alignment import DTWAligner
rng = np.random.RandomState(0)
X = np.zeros((1, 900, 25)); Y = np.zeros((1, 900, 25))
X[0, :812] = np.cumsum(rng.randn(812, 25), 0) * 0.1
Y[0, :777] = np.cumsum(rng.randn(777, 25), 0) * 0.1
al = DTWAligner()
for _ in range(5):
    al.transform((X, Y))
PY
