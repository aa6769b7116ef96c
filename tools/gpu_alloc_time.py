"""what does a fresh device allocation cost?  (setup of a 20-atom molecule allocates 2.0 GB of ERI tiles + 2.4 GB of AO values)"""
import time, torch
torch.cuda.init()
x = torch.empty(16, device="cuda"); torch.cuda.synchronize()
for gb in (0.25, 1, 2, 4, 8, 16):
    n = int(gb * 2**30 / 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    a.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a.zero_(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("%.2f GB: alloc %.2f ms, first touch (zero_) %.2f ms, second zero_ %.2f ms" % (gb, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    del a
    torch.cuda.empty_cache()
