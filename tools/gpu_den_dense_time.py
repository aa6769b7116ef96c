"""time the full-matrix density kernel (dm not in factor form) on the C5 shape, synthetic inputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
ngrid, nao = 353400, 208
ld = lib.padded_nao(nao)
g = torch.Generator(device="cuda").manual_seed(1)
ao = lib.ao_from(torch.randn((4, ngrid, nao), dtype=torch.float64, device="cuda", generator=g) * 0.1)
d = torch.randn((nao, nao), dtype=torch.float64, device="cuda", generator=g); d = d + d.T
dp = lib.pad_matrix(d, ld)
f = lambda: lib.grid_density(ao, nao, dp, True)
rho, grho = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("dense density (GGA) lib %s: %.4f ms  checksums %.10e %.10e" % (os.path.basename(os.environ.get("DQC_AMD_LIB", "default")), e0.elapsed_time(e1) / 20, float(rho.sum()), float(grho.sum())))
