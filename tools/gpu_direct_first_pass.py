"""first, second, third screened Coulomb pass of a fresh process (naphthalene / cc-pVTZ): what a cold start costs"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
tab = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz").get_hamiltonian()._tab
D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 34, np.eye(tab.nao), 3), device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
ctx = lib.DirectContext(tab, dev)
torch.cuda.synchronize(); print("context %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.jk(D, False, 1e-13)
    torch.cuda.synchronize(); print("pass %d: %.1f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
