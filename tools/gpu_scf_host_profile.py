"""where does the HOST time of one SCF iteration go (single molecule, qc.run())?"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
mol = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3")
dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe").run()
qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
torch.cuda.synchronize()
t0 = time.perf_counter(); qc.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("run: %.1f ms, %d iterations, %.3f ms/iteration" % (1e3 * dt, qc.niter, 1e3 * dt / qc.niter))
qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
pr = cProfile.Profile(); pr.enable(); qc.run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
