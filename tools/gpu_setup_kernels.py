"""one-off setup kernels of a C5 molecule, timed call by call (HIP events, warm): S / T / V (dqc_int1e), AO values + gradient
(dqc_eval_gto deriv 0 / 1 / 2), ERI tile fill"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
mol = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3")
h = mol.get_hamiltonian()
mol.setup_grid()
rg = mol.get_grid().get_rgrid().to("cuda")
tab, dev = h._tab, h.device
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
for w in ("ovlp", "kin", "nuc"):
    ms, m = t(lambda: lib.int1e(w, tab, dev))
    print("int1e %-4s %.3f ms   checksum %.12e  asym %.1e" % (w, ms, float(m.abs().sum()), float((m - m.T).abs().max())))
for d in (0, 1, 2):
    ms, ao = t(lambda: lib.eval_gto(tab, rg, d))
    nb = ao.numel() * 8
    print("eval_gto deriv %d  %.3f ms  %.2f GB written = %.2f TB/s" % (d, ms, nb / 1e9, nb / ms / 1e9))
ms, tl = t(lambda: lib.eri_tiles(tab, dev), 3)
print("eri fill %.3f ms (%.3f GB)" % (ms, tl.numel() * 8 / 1e9))
