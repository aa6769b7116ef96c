import sys, time
sys.path.insert(0, "/root/repo")
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
import cProfile, pstats
def one(i):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
    torch.cuda.synchronize()
    return qc
one(0)
t0 = time.perf_counter(); one(1); print("setup one molecule %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); one(2); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
