"""what does fresh device memory cost on this box?  hipMalloc through torch (caching allocator miss) for several block sizes,
first write to it, and re-use of a cached block"""
import os, sys, time
import torch
dev = torch.device("cuda")
torch.zeros(1, device=dev); torch.cuda.synchronize()
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t0
keep = []
for gb in (0.25, 1.0, 2.35, 2.35, 2.35, 8.0, 32.0):
    n = int(gb * 1e9 / 8)
    x, ta = t(lambda: torch.empty(n, dtype=torch.float64, device=dev))
    _, tw = t(lambda: x.fill_(1.0))
    _, tw2 = t(lambda: x.fill_(2.0))
    keep.append(x)
    print("fresh %.2f GB: alloc %.2f ms (%.1f ms/GB), first fill %.2f ms, second fill %.2f ms" % (gb, 1e3 * ta, 1e3 * ta / gb, 1e3 * tw, 1e3 * tw2))
del keep, x
_, tf = t(lambda: torch.cuda.empty_cache())
print("empty_cache (hipFree of ~48 GB): %.1f ms" % (1e3 * tf))
x, ta = t(lambda: torch.empty(int(2.35e9 / 8), dtype=torch.float64, device=dev))
print("2.35 GB after the free: alloc %.2f ms" % (1e3 * ta))
del x
x, ta = t(lambda: torch.empty(int(2.35e9 / 8), dtype=torch.float64, device=dev))
print("2.35 GB from the cache: alloc %.3f ms" % (1e3 * ta))
# allocation while the GPU is busy
y = torch.empty(int(4e9 / 8), dtype=torch.float64, device=dev)
for _ in range(50):
    y.mul_(1.0001)
t0 = time.perf_counter()
z = torch.empty(int(2.0e9 / 8), dtype=torch.float64, device=dev)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("2 GB fresh alloc with 50 kernels queued: host blocked %.2f ms, queue drained after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
