#!/usr/bin/env bash
# Round 5: bench.py's official region with and without a leg of the same steps in front of the warm-up (clock settling)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r05_bench_precondition.txt
for r in 1 2 3; do
  for pc in 0 200 50; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --precondition $pc --no-cpu-baseline --no-secondary --no-traffic --regions 3 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('precondition %3d: ms_per_step %.4f  value %.4e  roofline.frac %.3f  kernel_ms %.4f  kernel_ms_steady %.4f  repeat_regions %s' % ($pc, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_steady'], d['repeat_regions']['ms_per_step_all']))" | tee -a gpurun_out/r05_bench_precondition.txt
  done
done
