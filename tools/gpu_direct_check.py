"""screened direct Coulomb / exchange matrices against the unscreened pass of the same context (quick consistency check)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
for name, geo, basis in (("H2O", M.H2O, "cc-pvdz"), ("benzene", M.benzene(), "cc-pvtz")):
    tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
    D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 34, np.eye(tab.nao), 3), device=dev)
    ctx = lib.DirectContext(tab, dev)
    J0, K0 = ctx.jk(D, True, 0.0)
    for tau in (1e-13, 1e-9):
        J1, K1 = ctx.jk(D, True, tau)
        J2, _ = ctx.jk(D, False, tau)
        torch.cuda.synchronize()
        print(name, "tau %g  dJ(jk) %.2e  dK %.2e  dJ(j) %.2e   |J| %.2e" % (tau, float((J1 - J0).abs().max()), float((K1 - K0).abs().max()), float((J2 - J0).abs().max()), float(J0.abs().max())), flush=True)
