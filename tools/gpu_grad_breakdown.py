"""where a C5 RKS PBE nuclear gradient goes: one-electron, two-electron (dqc_eri_grad), XC (grid) parts, wall time each (warm)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib, gradient as G
from tests import molecules as M
dev = torch.device("cuda:0")
qc = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev), xc="gga_x_pbe+gga_c_pbe").run()
def wall(fn, k=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k
eng = qc._engine; h = eng.hamilton
X = h._orthozer
d = X @ qc._dm @ X.T; d = (d + d.T) * 0.5
T = lib.cart2sph_matrix(h._tab, dev)
dc = (T.T @ d @ T).contiguous()
grad = torch.zeros((20, 3), dtype=torch.float64, device=dev)
print("total            %.4f s" % wall(lambda: qc.nuclear_gradient()))
print("int1e_grad       %.4f s" % wall(lambda: lib.int1e_grad(grad, dc, dc, h._tab, h._zs)))
print("eri_grad (J)     %.4f s" % wall(lambda: lib.eri_grad(grad, dc, 0.0, h._tab)))
print("eri_grad (J+K)   %.4f s" % wall(lambda: lib.eri_grad(grad, dc, 1.0, h._tab)))
print("xc_gradient      %.4f s" % wall(lambda: G._xc_gradient(eng, [d])))
print("eval_gto deriv 3 %.4f s" % wall(lambda: lib.eval_gto(h._tab, h.rgrid, 3)))
