"""time the one-pass J + 2K contraction (UHF Fock build) and J + K (RHF) of the C5 shape on synthetic tiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
nao = 208
nt = int(lib.load().dqc_eri_tile_count(nao))
g = torch.Generator(device="cuda").manual_seed(1)
tiles = torch.randn(nt * 4096, dtype=torch.float64, device="cuda", generator=g)
tiles *= torch.exp(-30 * torch.rand(nt * 4096, dtype=torch.float64, device="cuda", generator=g) ** 2)
d = torch.randn((3, nao, nao), dtype=torch.float64, device="cuda", generator=g); d = d + d.transpose(1, 2)
work = lib.jk_workspace(nao, "cuda")
def timeit(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_jk = timeit(lambda: lib.jk(tiles, d[0], work, with_k=True))
t_u = timeit(lambda: lib.jk_multi(tiles, d[:1], d[1:3]))
t_1 = timeit(lambda: lib.jk_multi(tiles, d[:1], d[1:2]))
J, K = lib.jk_multi(tiles, d[:1], d[1:3])
print("RHF J+K %.3f ms | multi J+K %.3f ms | UHF J+2K one pass %.3f ms (%.2fx)  checksums %.10e %.10e" % (t_jk, t_1, t_u, t_u / t_jk, float(J.sum()), float(K.sum())))
