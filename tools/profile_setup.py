"""Where does the one-off setup go? (run on the GPU box)  One C5 molecule, warm (third of three), under cProfile; then the 32-molecule
batch the way bench.py sets it up (tables + batched orthogonalisers, then build + grid + AO), wall clock per stage and cProfile."""
import sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, ".")
import dqc_amd
from dqc_amd.batch import prepare_orthogonalisers
from tests import molecules as M


def build(i):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
    torch.cuda.synchronize()
    return qc


def top(pr, n=28):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(n)
    return "\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000]


t0 = time.perf_counter(); build(0); print("first setup (cold process) %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); build(1); print("second setup %.3f s" % (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); build(2); w = time.perf_counter() - t0; pr.disable()
print("third setup %.3f s (under cProfile)" % w)
print(top(pr))

nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
mols = []
for i in range(nmol):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    mol.get_hamiltonian()
    mols.append(mol)
prepare_orthogonalisers([m.get_hamiltonian() for m in mols])
torch.cuda.synchronize(); t1 = time.perf_counter()
qcs = [dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe") for m in mols]
t2h = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
pr.disable()
print("batch of %d: tables + orthogonalisers %.3f s, build + grid + AO %.3f s (host returned after %.3f s)" % (nmol, t1 - t0, t2 - t1, t2h - t1))
print(top(pr, 40))
