#!/usr/bin/env bash
# Round 5: host-side costs -- where the eager config-3 step spends its time (cProfile), the numpy -> numpy entry of bench_paths
# after its timing fix, DTWAligner.transform; plus the -m gpu suite on the ADVICE fixes (XCC probe, const table cache after a capture).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_host
timeout 900 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1; echo "pytest rc=$?" >> ${O}_tests.log; tail -n 3 ${O}_tests.log
timeout 300 python tools/dbg/c3_host_profile.py 300 > ${O}_c3_profile.txt 2>&1; grep -v amdgpu.ids ${O}_c3_profile.txt | head -80
timeout 300 python tools/bench_paths.py --only c2h 2>&1 | grep '"path"' | tee ${O}_c2h.jsonl | cut -c1-420
timeout 300 python tools/dbg/dtw_transform_time.py 2>&1 | grep -v amdgpu.ids | tee ${O}_dtw_transform.txt | tail -n 20
