"""does fresh-VRAM cost come from memory another process just released, and do parallel allocations clear faster?
usage: python tools/gpu_alloc_cost3.py dirty | single | threads N"""
import ctypes, time, threading, sys
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
GB = 1 << 30
def malloc(n):
    p = ctypes.c_void_p(); rc = hip.hipMalloc(ctypes.byref(p), n); assert rc == 0, rc; return p
mode = sys.argv[1]
hip.hipDeviceSynchronize()
if mode == "dirty":
    ps = [malloc(4 * GB) for _ in range(60)]
    for p in ps:
        hip.hipMemset(p, 1, 4 * GB)
    hip.hipDeviceSynchronize()
    print("dirtied 240 GiB")
elif mode == "single":
    t0 = time.perf_counter(); ts = []
    for i in range(36):
        t1 = time.perf_counter(); malloc(4 * GB); ts.append(1e3 * (time.perf_counter() - t1))
    print("single thread: 36 x 4 GiB in %.0f ms; per call ms: %s" % (1e3 * (time.perf_counter() - t0), " ".join("%.0f" % t for t in ts)))
else:
    n = int(sys.argv[2])
    per = 36 // n
    def worker():
        for _ in range(per):
            malloc(4 * GB)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker) for _ in range(n)]
    [x.start() for x in th]; [x.join() for x in th]
    print("%d threads: %d x 4 GiB in %.0f ms" % (n, per * n, 1e3 * (time.perf_counter() - t0)))
