#!/usr/bin/env bash
# Round 5: the transposed form with global / unit variances: parity, timing, config 5 with global variances
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_strip_tr_gpu.py tests/test_streams_gpu.py tests/test_const_gpu.py -m gpu -q -x 2>&1 | tail -n 12
timeout 300 python tools/dbg/narrow_time.py global 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_tr_narrow_global.txt
timeout 300 python tools/dbg/narrow_time.py unit 2>&1 | grep -v amdgpu.ids | grep "float64" | tee gpurun_out/r05_tr_narrow_unit.txt
timeout 300 python tools/bench_paths.py --only c5 2>&1 | grep '"path"' | tee gpurun_out/r5_tr5_c5.jsonl | cut -c1-150
