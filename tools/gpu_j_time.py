"""time the Coulomb tile stream of the C5 shape (synthetic tiles); DQC_J_NBLK / DQC_AMD_LIB select tuning variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
nao = 208
nt = int(lib.load().dqc_eri_tile_count(nao))
g = torch.Generator(device="cuda").manual_seed(1)
tiles = torch.randn(nt * 4096, dtype=torch.float64, device="cuda", generator=g)
tiles *= torch.exp(-30 * torch.rand(nt * 4096, dtype=torch.float64, device="cuda", generator=g) ** 2)
dm = torch.randn((nao, nao), dtype=torch.float64, device="cuda", generator=g); dm = dm + dm.T
work = lib.jk_workspace(nao, "cuda")
f = lambda: lib.jk(tiles, dm, work, with_k=False)[0]
j = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 40
print("J nblk/CU %s lib %s: %.4f ms  %.0f GB/s  checksum %.10e" % (os.environ.get("DQC_J_NBLK", "12"), os.path.basename(os.environ.get("DQC_AMD_LIB", "default")), ms, nt * 32768 / ms / 1e6, float(j.sum())))
