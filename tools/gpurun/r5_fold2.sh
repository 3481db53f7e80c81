#!/usr/bin/env bash
# Round 5: where the folded launch end's time goes -- (a) the agent-scope release fence (L2 write-back) at every workgroup's exit,
# (b) write-through (sc1) row stores instead, which need no write-back.  Interleaved, forward float64 / float32 at the config-2 shape.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_fold2
: > ${O}.txt
B=$PWD/tools/dbg/bin
for round in 1 2 3; do
  for v in default nofold foldnf foldsc1nt foldsc1 nofoldsc1nt; do
    if [ $v = default ]; then unset NNMNKWII_AMD_SO; else export NNMNKWII_AMD_SO=$B/libmlpg_hip_$v.so; fi
    timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  done
done
