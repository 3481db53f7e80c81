"""Raw PCIe rates of this box (pinned host memory): what the host-pointer entry points can reach at best."""
import time, torch
n = 368640000 // 4
h = torch.empty(n, dtype=torch.float32).pin_memory()
h2 = torch.empty(n, dtype=torch.float32).pin_memory()
d = torch.empty(n, dtype=torch.float32, device="cuda")
d2 = torch.empty(n, dtype=torch.float32, device="cuda")
ho = torch.empty(n // 6, dtype=torch.float32).pin_memory()
do = torch.empty(n // 6, dtype=torch.float32, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = t(lambda: d.copy_(h, non_blocking=True))
print("H2D 368.6 MB: %.2f ms  %.1f GB/s" % (a * 1e3, 0.36864 / a))
def both():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
a = t(both)
print("2 x H2D on two streams: %.2f ms  %.1f GB/s" % (a * 1e3, 2 * 0.36864 / a))
a = t(lambda: ho.copy_(do, non_blocking=True))
print("D2H 61.4 MB: %.2f ms  %.1f GB/s" % (a * 1e3, 0.06144 / a))
def duplex():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): ho.copy_(do, non_blocking=True)
a = t(duplex)
print("H2D 368.6 MB + D2H 61.4 MB concurrently: %.2f ms" % (a * 1e3))
