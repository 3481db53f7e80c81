#!/usr/bin/env bash
# Round 6: the start ramp under bench.py's own protocol (settled clocks, 9 repeat regions), interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_stagger_bench
: > ${O}.txt
for round in 1 2 3; do
for us in ${SWEEP:-0 20 0 24 0 16}; do
  MLPG_STRIP_STAGGER_US=$us timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-traffic --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('stagger $us us: ms_per_step %.4f  kernel_ms %.4f  cold %.4f  regions median %.4f min %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['cold_protocol']['ms_per_step'], r['repeat_regions']['ms_per_step_median'], r['repeat_regions']['ms_per_step_min']))" | tee -a ${O}.txt
done
done
