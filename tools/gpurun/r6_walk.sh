#!/usr/bin/env bash
# Round 6: the walk form of the strip kernel: parity, then interleaved timing against the strip kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_walk
: > ${O}.txt
timeout 900 python -m pytest tests/test_walk_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee -a ${O}.txt
for round in 1 2 3; do
for w in 0 1; do
  echo "== MLPG_STRIP_WALK=$w" | tee -a ${O}.txt
  MLPG_STRIP_WALK=$w timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
done
for w in 0 1; do
MLPG_STRIP_WALK=$w timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-traffic --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('walk $w: ms_per_step %.4f  kernel_ms %.4f  cold %.4f  regions median %.4f  parity %.2e / tight %.2e' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['cold_protocol']['ms_per_step'], r['repeat_regions']['ms_per_step_median'], r['parity_rel_err_vs_oracle'], r['parity_rel_err_vs_oracle_tight_dynamic_variances']))" | tee -a ${O}.txt
done
