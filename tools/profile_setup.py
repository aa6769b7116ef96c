"""Where does the one-off setup of a C5 molecule go? (run on the GPU box)"""
import sys, time, cProfile, pstats
import torch
sys.path.insert(0, ".")
import dqc_amd
from tests import molecules as M
def build(i):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
    torch.cuda.synchronize()
    return qc
build(0)
t0 = time.perf_counter(); build(1); print("setup wall %.3f s" % (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); build(2); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
