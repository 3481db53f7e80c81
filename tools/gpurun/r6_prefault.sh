#!/bin/bash
# round 6: first touch of the result array's pages while the device works (short host path), A/B in alternating processes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/prefault_ab.txt
: > $O
for i in 1 2; do
  MLPG_HIP_HOST_PREFAULT=0 python tools/dbg/host_sizes.py "no prefault ($i)" >> $O 2>&1
  python tools/dbg/host_sizes.py "prefault ($i)" >> $O 2>&1
done
grep -v amdgpu.ids $O | cut -c1-150
for i in 1 2; do
  for pf in 0 1; do
    MLPG_HIP_HOST_PREFAULT=$pf python tools/bench_paths.py --only litq 2>/dev/null | grep -o '"path": "lit-c2-loop[^}]*' | cut -c1-160 | sed "s/^/prefault=$pf  /"
  done
done
