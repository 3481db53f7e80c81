#!/usr/bin/env bash
# chunked kernel: kernel trace of tools/dbg/chunk_time.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chunk_gpu.py -m gpu -x -q 2>&1 | tail -3
rm -rf gpurun_out/chunkprof; rocprofv3 --kernel-trace --stats -d gpurun_out/chunkprof -o run -- python tools/dbg/chunk_time.py > gpurun_out/chunkprof.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/chunkprof -name "*.db" | head -1) 2>&1 | cut -c1-90,112-200 | head -14; rm -rf gpurun_out/chunkprof
grep "chunk\|generic" gpurun_out/chunkprof.log
