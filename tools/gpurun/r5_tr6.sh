#!/usr/bin/env bash
# Round 5: the transposed form with a lengths vector (per-lane frame counts): parity, soaks with ragged lengths, timing unchanged without lengths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_strip_tr_gpu.py tests/test_streams_gpu.py tests/test_strip_gpu.py tests/test_util_gpu.py tests/test_soak_gpu.py -m gpu -q -x 2>&1 | tail -n 12
timeout 200 python tools/dbg/soak_strip.py 90 701 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_soakf_strip_trl.log; echo "rc=$?" >> gpurun_out/r05_soakf_strip_trl.log; tail -n 3 gpurun_out/r05_soakf_strip_trl.log
timeout 200 python tools/dbg/mlpg_soak.py 60 702 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_soakf_streams_trl.log; echo "rc=$?" >> gpurun_out/r05_soakf_streams_trl.log; tail -n 2 gpurun_out/r05_soakf_streams_trl.log
timeout 200 python tools/dbg/narrow_time.py frame 2>&1 | grep -v amdgpu.ids | grep float64 | head -8
