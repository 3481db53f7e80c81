#!/usr/bin/env bash
# float32 strip forward with different ring depths (each argument: a depth)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="-DMLPG_STRIP_RING_F32=$r" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f32 > /dev/null 2>&1
  for rep in 1 2; do
  python - <<PY 2>&1 | grep -v amdgpu
import sys, torch
sys.path.insert(0, '.')
from tools.bench_paths import WINDOWS, gpu_time
from nnmnkwii_amd import _hip
B,T,sd=256,1000,60
m=torch.randn(B,T,3*sd,dtype=torch.float32,device='cuda'); v=torch.rand(B,T,3*sd,dtype=torch.float32,device='cuda')+0.1
pw=_hip.prepack_windows(WINDOWS)
print("ring $r  f32 strip %.4f ms" % gpu_time(lambda: _hip.forward(m,v,pw,algo=3,want_status=False),steps=30,warmup=5))
PY
  done
done
