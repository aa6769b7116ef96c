"""per-chunk timeline of the Vxc kernel (one block): DQC_AMD_LIB=dqc_amd/libdqc_amd_vwutrace.so python tools/gpu_vxc_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqc_amd import lib
ngrid, nao, ld = 353400, 208, 208
g = torch.Generator(device="cuda").manual_seed(1)
ao = torch.randn((4, ngrid, ld), dtype=torch.float64, device="cuda", generator=g) * 0.1
if os.environ.get("SPARSE"):  # AO-like magnitudes: mostly tiny
    ao *= torch.exp(-30 * torch.rand((4, ngrid, ld), dtype=torch.float64, device="cuda", generator=g) ** 2)
w = torch.rand(ngrid, dtype=torch.float64, device="cuda", generator=g)
vrho = torch.randn(ngrid, dtype=torch.float64, device="cuda", generator=g)
vgrad = torch.randn((3, ngrid), dtype=torch.float64, device="cuda", generator=g)
f = lambda: lib.grid_vxc(ao, nao, w, vrho, vgrad)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): f()
e1.record(); torch.cuda.synchronize()
print("kernel + memset + symmetrize: %.4f ms" % (e0.elapsed_time(e1) / 30))
buf = np.zeros(2 * 4 * 128, dtype=np.int64)
lib.load().dqc_debug_vwu_trace(buf.ctypes.data_as(ctypes.c_void_p))
c0, w0, c1, w1 = buf[508:512]
print("main loop of one block: %.1f us, %d shader cycles -> %.3f GHz" % ((w1 - w0) / 100.0, c1 - c0, (c1 - c0) / ((w1 - w0) * 10.0)))
if os.environ.get("CHUNKS"):
    t = buf.reshape(2, 128, 4).astype(np.float64) / 100.0
    b = t[1]
    per = b[3:85, 0] - b[2:84, 0]
    wait = b[2:85, 1] - b[2:85, 0]
    print("producer wave 8 of one block: chunk period mean %.2f us (min %.2f max %.2f); window-open -> loads landed (incl. ~0.2 us stamp) mean %.2f max %.2f" % (per.mean(), per.min(), per.max(), wait.mean(), wait.max()))
