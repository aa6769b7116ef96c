"""per-phase timeline of the fused grid kernel (block 0): DQC_AMD_LIB=dqc_amd/libdqc_amd_fgtrace.so python tools/gpu_fused_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqc_amd import lib
dev = torch.device("cuda:0")
nao, ngrid, nocc = 208, 353400, 46
ld = lib.padded_nao(nao)
g = torch.Generator().manual_seed(1)
ao = lib.ao_from((torch.randn((4, ngrid, nao), dtype=torch.float64, generator=g) * torch.exp(-3 * torch.rand((4, ngrid, nao), dtype=torch.float64, generator=g))).to(dev))
w = torch.rand(ngrid, dtype=torch.float64, generator=g).to(dev)
fac = lib.pad_factor((torch.randn((nao, nocc), dtype=torch.float64, generator=g) * 0.3).to(dev), ld)
terms = [(1.0, "gga_x_pbe"), (1.0, "gga_c_pbe")]
for _ in range(3):
    lib.grid_fused(ao, nao, w, fac, terms)
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 64, dtype=np.int64)
rc = lib.load().dqc_debug_fused_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(2, 64, 8).astype(np.float64) / 100.0  # us
t0 = t[0, 0, 0]
names_c = ["half1", "wait Ba", "half2", "wait Bb", "wait Bc(window)"]
print("consumer wave 0 (us): per chunk [half1, wait Ba, half2, wait Bb, window wait]   producer wave 8: [stage, wait Ba, density, wait Bb, window, prefetch issue]")
for c in range(2, 12):
    a = t[0, c]; b = t[1, c]
    print("chunk %2d  cons: %5.2f %5.2f %5.2f %5.2f %5.2f | prod: %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f | period %.2f" % (
        c, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[6] - a[4],
        b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3], b[5] - b[4], b[6] - b[5], t[0, c + 1, 0] - a[0]))
