"""per-kernel times of the C4 Fock build (naphthalene / cc-pVTZ, nao 412, RKS PBE sg3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
mol = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz", grid="sg3")
eng = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine
h = eng.hamilton
n = eng.shape[-1]
dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda")))
orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()
core = eng.knvext.fullmatrix()
acc = None
for it in range(6):
    names, ev = h.timed_fock_kernels(h.ao_orb2dm(orb, eng.orb_weight), core)
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(names))]
    if it >= 1:
        acc = t if acc is None else [a + b for a, b in zip(acc, t)]
print("C4 nao %d ngrid %d nocc %d:" % (h._nao_ao, h.rgrid.shape[0], orb.shape[1]), " ".join("%s %.3f ms" % (nm, a / 5) for nm, a in zip(names, acc)))
