#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
for T in 1000 500 256; do
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --frames $T 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('T=$T kernel_ms %.4f  frames/s %.3e  GB/s %.1f  frac %.3f' % (rf['kernel_ms'], r['value'], rf['achieved'], rf['frac']))"
done
