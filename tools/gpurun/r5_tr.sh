#!/usr/bin/env bash
# Round 5: the strip kernel's transposed form for narrow streams: parity, timing against the wave-per-system kernel, config 5.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_strip_tr_gpu.py tests/test_streams_gpu.py tests/test_strip_gpu.py -m gpu -q -x 2>&1 | tail -n 15
timeout 300 python tools/dbg/narrow_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_tr_narrow.txt
timeout 300 python tools/bench_paths.py --only c5 2>&1 | grep '"path"' | tee gpurun_out/r5_tr_c5.jsonl | cut -c1-200
