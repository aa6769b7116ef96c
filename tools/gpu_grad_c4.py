"""nuclear gradient of config C4 (naphthalene / cc-pVTZ, RKS PBE, sg3): SCF and gradient wall time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
m = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz", grid="sg3", device="cuda")
t0 = time.perf_counter()
qc = dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe").run()
torch.cuda.synchronize(); t1 = time.perf_counter()
g = qc.nuclear_gradient()
torch.cuda.synchronize(); t2 = time.perf_counter()
g = qc.nuclear_gradient()
torch.cuda.synchronize(); t3 = time.perf_counter()
print("C4 naphthalene/cc-pVTZ PBE: E %.8f niter %d SCF incl. setup %.2f s  gradient %.2f s (warm %.2f s)  max|g| %.4f  |sum g| %.1e" % (
    float(qc.energy()), qc.niter, t1 - t0, t2 - t1, t3 - t2, float(g.abs().max()), float(g.sum(0).abs().max())))
