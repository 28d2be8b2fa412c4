"""PCIe copy rates of the box: pinned host <-> device, one direction at a time and both at once (tools/exp)."""
import time, torch
dev = torch.device("cuda:0")
n = 256 << 20
h_a = torch.empty(n, dtype=torch.uint8).pin_memory(); h_b = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device=dev); d_b = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def h2d():
    with torch.cuda.stream(s1): d_a.copy_(h_a, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_b.copy_(d_b, non_blocking=True)
def both():
    h2d(); d2h()
print("H2D %.1f GB/s" % (n / run(h2d) / 1e9))
print("D2H %.1f GB/s" % (n / run(d2h) / 1e9))
t = run(both)
print("both at once: %.1f GB/s each way (%.2f ms for 256 MiB each)" % (n / t / 1e9, t * 1e3))
for piece in (8 << 20, 1 << 20):
    def h2d_p():
        with torch.cuda.stream(s1):
            for o in range(0, n, piece): d_a[o:o + piece].copy_(h_a[o:o + piece], non_blocking=True)
    print("H2D in %d MiB pieces %.1f GB/s" % (piece >> 20, n / run(h2d_p) / 1e9))
