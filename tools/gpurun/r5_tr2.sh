#!/usr/bin/env bash
# Round 5: kernel trace of the config-5 paths with the transposed form in place
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/r5_tr_c5_trace -o run -- python tools/bench_paths.py --only c5 > gpurun_out/r5_tr_c5_trace.log 2>&1
f=$(find gpurun_out/r5_tr_c5_trace -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" > gpurun_out/r5_tr_c5_trace.txt 2>&1; rm -rf gpurun_out/r5_tr_c5_trace
cut -c1-250 gpurun_out/r5_tr_c5_trace.txt | head -24
grep '"path"' gpurun_out/r5_tr_c5_trace.log | cut -c1-120
