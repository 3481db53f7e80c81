"""config 5 in place from one (B, T, 198) batch: every stream alone, combinations of them and the one-call form
(mlpg_hip_forward_streams merges streams that share their windows into one strip launch: DESIGN.md K1m)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def child():
    import torch
    from nnmnkwii_amd import _hip
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench_paths import WINDOWS, gpu_time
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    B, T = 512, 2000
    m = torch.randn(B, T, 198, dtype=torch.float64, device=dev, generator=gen)
    v = torch.rand(B, T, 198, dtype=torch.float64, device=dev, generator=gen) + 0.1
    streams = [(0, 60, WINDOWS), (180, 1, WINDOWS), (183, 5, WINDOWS)]
    out = {}
    bap4 = (183, 4, WINDOWS)      # not a real stream (its windows are 5 apart): only the shape of the launch matters here
    for nm, ss in (("mgc", streams[:1]), ("lf0", streams[1:2]), ("bap", streams[2:]), ("narrow", streams[1:]), ("mgc+lf0", streams[:2]),
                   ("mgc+4", [streams[0], bap4]), ("mgc+bap", [streams[0], streams[2]]), ("all", streams)):
        out[nm] = round(gpu_time(lambda: _hip.forward_streams(m, v, ss, want_status=False), steps=20), 4)
    print(out, flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["child"]:
        child()
    else:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], check=False)
