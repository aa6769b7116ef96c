"""steady-state unrestricted Kohn-Sham (PBE) Fock build of one C5 molecule, as bench.py's uks_pbe row times it"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.utils.datastruct import SpinParam
from tests import molecules as M
dev = torch.device("cuda:0")
qc = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev), xc="gga_x_pbe+gga_c_pbe", restricted=False)
eng, h = qc._engine, qc._engine.hamilton
n = eng.shape[-1]
z = torch.zeros((n, n), dtype=torch.float64, device=dev)
dm = eng.scp2dm(eng.dm2scp(SpinParam(u=z, d=z)))
f = eng.dm2scp(dm)
orbs = [eng._eigvecs(f[0])[..., :eng.norb.u].contiguous(), eng._eigvecs(f[1])[..., :eng.norb.d].contiguous()]
def build():
    d = SpinParam(u=h.ao_orb2dm(orbs[0], eng.orb_weight.u), d=h.ao_orb2dm(orbs[1], eng.orb_weight.d))
    return eng.dm2scp(d)
for _ in range(3):
    f1 = build()
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f1 = build()
    e1.record(); torch.cuda.synchronize()
    print("UKS PBE build %.3f ms   checksum %.12f" % (e0.elapsed_time(e1) / 20, float(f1.abs().sum())), flush=True)
