#!/usr/bin/env bash
# chunked kernel for window extents of 2: one build per argument (-D flags), parity + timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in "${@:-}"; do
MLPG_HIP_EXTRA_FLAGS="$f" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_chunk > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
echo "== [$f] $(timeout 900 python -m pytest tests/test_chunk_gpu.py -m gpu -x -q 2>&1 | tail -1)"
timeout 300 python tools/dbg/chunk_time.py 2>&1 | grep -v amdgpu.ids | grep chunk
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_chunk > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
