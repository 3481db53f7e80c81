"""MLPG_HIP_ALGO_AUTO's routing as a tested table: profiles/r05_auto_routing.json holds, for 29
representative launches, the kernel family AUTO picked on the MI355X (read off the library's launch counters by
tools/auto_routing.py) and the time of every kernel that accepts the launch.  The GPU test re-derives the routes with the
library as built; the CPU test checks that the recorded choice is the fastest recorded kernel, or within 6 % of it."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "profiles", "r05_auto_routing.json")


def _rows():
    with open(TABLE) as f:
        return json.load(f)


def test_recorded_auto_choice_is_the_fastest_recorded_kernel_or_close():
    rows = _rows()
    assert len(rows) >= 20
    for r in rows:
        ms = r["ms"]
        kinds = r["auto"].split("+")
        # (the transposed strip form has no algorithm number of its own: MLPG_HIP_ALGO_STRIP takes it for narrow streams)
        key = "strip" if kinds[0] == "strip_tr" else kinds[0]
        assert len(kinds) == 1 and key in ms, r
        best = min(ms.values())
        assert ms[key] <= 1.06 * best + 0.002, (r["case"], r["direction"], r["auto"], ms)


@pytest.mark.gpu
def test_auto_routes_as_recorded():
    import torch
    sys.path.insert(0, ROOT)
    from nnmnkwii_amd import _hip
    from tools import auto_routing as AR
    rows = {(r["case"], r["direction"]): r for r in _rows()}
    for case in AR.CASES:
        fn = AR.runner(case, torch, _hip)
        fn(_hip.ALGO_AUTO)                      # first call: tables, scratch
        torch.cuda.synchronize()
        got = AR.route_of(fn, _hip)
        torch.cuda.synchronize()
        assert got == rows[(case[0], case[7])]["auto"], (case, got, rows[(case[0], case[7])]["auto"])
        del fn
        torch.cuda.empty_cache()
