"""time the nuclear gradient on the 20-atom C5 molecule (RHF and RKS-LDA, cc-pVDZ)"""
import sys, os, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dqc_amd
from tests import molecules as M
print = functools.partial(print, flush=True)
dev = torch.device("cuda:0")
for xc in ("gga_x_pbe+gga_c_pbe", "lda_x+lda_c_pw", None):
    m = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev)
    t0 = time.perf_counter()
    qc = (dqc_amd.KS(m, xc=xc) if xc else dqc_amd.HF(m)).run()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g = qc.nuclear_gradient()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    g = qc.nuclear_gradient()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("%s: E = %.8f  niter %d  SCF incl. setup %.2f s   gradient %.3f s (warm %.3f s)   max |g| %.4f  |sum g| %.1e" %
          (("RKS " + xc) if xc else "RHF", float(qc.energy()), qc.niter, t1 - t0, t2 - t1, t3 - t2, float(g.abs().max()), float(g.sum(0).abs().max())))
