"""bench-like setup of N C5 molecules that STAY resident, with and without the one-request memory reserve; host profile"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.batch import molecule_bytes, reserve_device_memory
from dqc_amd.xc import get_xc
from tests import molecules as M
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reserve = (sys.argv[2] == "reserve") if len(sys.argv) > 2 else True
sync = (sys.argv[3] == "sync") if len(sys.argv) > 3 else True
dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(31), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe"); torch.cuda.synchronize()
torch.cuda.empty_cache()
brk = {}
def lap(name, t):
    if sync:
        torch.cuda.synchronize()
    now = time.perf_counter(); brk[name] = brk.get(name, 0.0) + now - t; return now
t00 = time.perf_counter()
if reserve:
    brk["reserve"] = reserve_device_memory(n * molecule_bytes(208, 353400) + (4 << 30), dev)
keep = []
pr = cProfile.Profile(); pr.enable()
for i in range(n):
    t = time.perf_counter()
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3"); t = lap("Mol", t)
    mol.setup_grid(); t = lap("grid", t)
    mol.get_hamiltonian().setup_grid(mol.get_grid(), get_xc("gga_x_pbe+gga_c_pbe")); t = lap("ao", t)
    mol.get_hamiltonian().build(); t = lap("build", t)
    keep.append(dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")); t = lap("engine", t)
torch.cuda.synchronize()
pr.disable()
tot = time.perf_counter() - t00
print("n %d reserve %s sync %s: total %.3f s = %.1f ms / molecule; stages (ms/molecule): %s" % (
    n, reserve, sync, tot, 1e3 * tot / n, {k: round(1e3 * v / n, 2) for k, v in brk.items()}))
if len(sys.argv) > 4:
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
