"""one full RKS-PBE SCF of C5 molecule 0 (for rocprofv3 --kernel-trace --stats: where does an SCF iteration's GPU time go?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
mol = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3")
qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
qc.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
qc.run()
torch.cuda.synchronize()
print("SCF", time.perf_counter() - t0, "s", qc.niter, "iterations", float(qc.energy()))
