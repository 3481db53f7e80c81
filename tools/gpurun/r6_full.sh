#!/usr/bin/env bash
# Round 6: the whole -m gpu suite, smoke(), the bench line in the driver's command form, the literal calls.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_full
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1
echo "pytest rc=$?" >> ${O}_tests.log
tail -6 ${O}_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (driver form)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench.json 2>${O}_bench.err; echo "rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r6_full_bench.json") if l.startswith("{")][-1])
sec = r.pop("secondary", {})
print(json.dumps({k: v for k, v in r.items() if k not in ("config", "repeat_regions", "cpu_baseline")}, indent=None)[:1800])
print("cold:", r.get("cold_protocol"))
for k, v in sec.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "frac", "us_per_call", "cpu_us_per_call", "speedup_vs_cpu", "us_forward", "us_forward_backward", "cpu_us_forward", "error", "rel_err_vs_oracle_last_utterance")})
PY
tail -3 ${O}_bench.err
