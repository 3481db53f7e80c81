#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/dbg/c3_cpu_tensors.py > gpurun_out/c3_cpu_tensors.txt 2>&1
cat gpurun_out/c3_cpu_tensors.txt
