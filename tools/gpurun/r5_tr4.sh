#!/usr/bin/env bash
# Round 5: soaks that reach the transposed form (batches without lengths), the short soak tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 200 python tools/dbg/soak_strip.py 100 601 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_soakf_strip_tr.log; echo "rc=$?" >> gpurun_out/r05_soakf_strip_tr.log; tail -n 3 gpurun_out/r05_soakf_strip_tr.log
timeout 200 python tools/dbg/mlpg_soak.py 80 602 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_soakf_streams_tr.log; echo "rc=$?" >> gpurun_out/r05_soakf_streams_tr.log; tail -n 2 gpurun_out/r05_soakf_streams_tr.log
timeout 300 python -m pytest tests/test_soak_gpu.py tests/test_strip_tr_gpu.py -m gpu -q 2>&1 | tail -n 3
