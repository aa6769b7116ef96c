"""one-off setup of C5 molecules, stage by stage (each stage closed by a device synchronise), then a host profile"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from dqc_amd.xc import get_xc
from tests import molecules as M

def stages(i, acc):
    def lap(name, t0):
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    t = time.perf_counter()
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3"); t = lap("Mol(): basis tables, overlap, orthogonaliser", t)
    mol.setup_grid(); t = lap("Mol.setup_grid: Becke/Lebedev grid", t)
    h = mol.get_hamiltonian()
    xc = get_xc("gga_x_pbe+gga_c_pbe")
    h.setup_grid(mol.get_grid(), xc); t = lap("Hamilton.setup_grid: AO values + gradients on the grid", t)
    h.build(); t = lap("Hamilton.build: T, V, ERI tile fill", t)
    return mol

acc = {}
stages(0, {})
n = 6
for i in range(1, 1 + n):
    stages(i, acc)
for k, v in acc.items():
    print("%-60s %8.2f ms / molecule" % (k, 1e3 * v / n))
print("%-60s %8.2f ms / molecule" % ("total", 1e3 * sum(acc.values()) / n))
t0 = time.perf_counter()
qcs = [dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe") for i in range(8, 16)]
torch.cuda.synchronize()
print("8 x KS(Mol(...)) back to back without intermediate syncs: %.2f ms / molecule" % (1e3 * (time.perf_counter() - t0) / 8))
pr = cProfile.Profile(); pr.enable()
qc = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(20), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe"); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
