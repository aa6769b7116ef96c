import sys, os
sys.path.insert(0, "/root/repo")
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
mol = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3")
eng = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine
h = eng.hamilton
n = eng.shape[-1]
dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda")))
orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()
d = h.ao_orb2dm(orb, eng.orb_weight)
fac = h._factor_of(d)
f = lambda: lib.grid_density_lr(h._ao, h._nao_ao, fac[0], True)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("density_lr NCT env", os.environ.get("DQC_LR_NCT"), "%.3f ms" % (e0.elapsed_time(e1) / 20))
