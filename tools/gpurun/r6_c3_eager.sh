#!/bin/bash
# round 6: host time of config 3's eager step, piece by piece; the GPU suite behind a host_api change; bench_paths c3 (graph legs)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/dbg/c3_eager_profile.py 2000 > gpurun_out/c3_eager_profile.txt 2>&1
tail -n 80 gpurun_out/c3_eager_profile.txt | head -n 12
python tools/bench_paths.py --only c3 > gpurun_out/c3_paths.jsonl 2>gpurun_out/c3_paths.err
grep -o '"ms_hip_graph_replay": [^,]*' gpurun_out/c3_paths.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1
tail -n 3 gpurun_out/tests.log
