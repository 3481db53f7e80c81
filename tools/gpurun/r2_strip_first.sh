#!/usr/bin/env bash
# round 2, first GPU contact of the strip kernel: its parity tests, then bench (strip vs wave)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_strip_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -25
for a in 3 2; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --algo $a 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('algo $a kernel_ms %.4f steady %.4f frames/s %.3e  GB/s %.1f  frac %.3f  err %.2e' % (rf['kernel_ms'], rf['kernel_ms_steady'], r['value'], rf['achieved'], rf['frac'], r['parity_rel_err_vs_oracle']))"
done
