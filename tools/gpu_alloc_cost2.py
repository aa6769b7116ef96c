"""fresh-VRAM cost by allocation route (ctypes on libamdhip64): hipMalloc, hipMallocAsync, hipExtMallocWithFlags, two threads"""
import ctypes, time, threading, sys
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMallocAsync.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
GB = 1 << 30
def tm(fn):
    t0 = time.perf_counter(); r = fn(); hip.hipDeviceSynchronize(); return r, 1e3 * (time.perf_counter() - t0)
def malloc(n):
    p = ctypes.c_void_p(); rc = hip.hipMalloc(ctypes.byref(p), n); assert rc == 0, rc; return p
hip.hipDeviceSynchronize()
ptrs = []
# burn the "fast" region first
for i in range(6):
    p, t = tm(lambda: malloc(2 * GB)); ptrs.append(p); print("hipMalloc 2 GiB #%d: %.1f ms" % (i, t))
p, t = tm(lambda: malloc(16 * GB)); ptrs.append(p); print("hipMalloc 16 GiB: %.1f ms (%.1f ms/GiB)" % (t, t / 16))
def masync(n):
    p = ctypes.c_void_p(); rc = hip.hipMallocAsync(ctypes.byref(p), n, None); assert rc == 0, rc; return p
p, t = tm(lambda: masync(8 * GB)); ptrs.append(p); print("hipMallocAsync 8 GiB: %.1f ms (%.1f ms/GiB)" % (t, t / 8))
for flag, name in ((0x3, "uncached"), (0x1, "finegrained")):
    def ext(n):
        p = ctypes.c_void_p(); rc = hip.hipExtMallocWithFlags(ctypes.byref(p), n, flag); return p if rc == 0 else None
    p, t = tm(lambda: ext(8 * GB)); print("hipExtMallocWithFlags(%s) 8 GiB: %.1f ms (%.1f ms/GiB) %s" % (name, t, t / 8, "ok" if p else "FAILED"))
    if p: ptrs.append(p)
res = {}
def worker(k):
    t0 = time.perf_counter(); res[k] = (malloc(8 * GB), 0); res[k] = (res[k][0], 1e3 * (time.perf_counter() - t0))
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
[x.start() for x in th]; [x.join() for x in th]
print("4 threads x hipMalloc 8 GiB at once: wall %.1f ms (%.1f ms/GiB aggregate), per thread %s" % (
    1e3 * (time.perf_counter() - t0), 1e3 * (time.perf_counter() - t0) / 32, ["%.0f" % res[k][1] for k in range(4)]))
# free and re-allocate within the process
big = ptrs.pop(6)
_, t = tm(lambda: hip.hipFree(big)); print("hipFree 16 GiB: %.1f ms" % t)
p, t = tm(lambda: malloc(16 * GB)); print("hipMalloc 16 GiB again: %.1f ms" % t)
p2, t = tm(lambda: malloc(100 * GB)); print("hipMalloc 100 GiB: %.1f ms (%.1f ms/GiB)" % (t, t / 100))
