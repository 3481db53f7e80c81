#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_walk4
: > ${O}.txt
timeout 900 python -m pytest tests/test_walk_gpu.py -m gpu -q 2>&1 | tail -15 | tee -a ${O}.txt
for round in 1 2 3; do
for w in 0 1; do
MLPG_STRIP_WALK=$w timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-traffic --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('walk $w: ms_per_step %.4f  kernel_ms %.4f  frac %.3f  cold %.4f  regions median %.4f  parity %.2e / tight %.2e' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['cold_protocol']['ms_per_step'], r['repeat_regions']['ms_per_step_median'], r['parity_rel_err_vs_oracle'], r['parity_rel_err_vs_oracle_tight_dynamic_variances']))" | tee -a ${O}.txt
done
done
for w in 0 1 0 1; do
echo "== paths, walk $w" | tee -a ${O}.txt
MLPG_STRIP_WALK=$w timeout 600 python tools/bench_paths.py --only c2t,c2b,c5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)
        if 'ms' in r: print('  %-62s %.4f ms' % (r['path'], r['ms']))" | tee -a ${O}.txt
done
