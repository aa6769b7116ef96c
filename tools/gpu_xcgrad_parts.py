"""wall time of the pieces of dqc_amd.gradient._xc_gradient on the C5 molecule (RKS PBE)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib, gradient as G
from dqc_amd.grid import get_predefined_grid
from dqc_amd.utils.datastruct import ValGrad
from tests import molecules as M
dev = torch.device("cuda:0")
qc = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev), xc="gga_x_pbe+gga_c_pbe").run()
eng = qc._engine; h = eng.hamilton; mol = eng.get_system()
X = h._orthozer
d = X @ qc._dm @ X.T; d = (d + d.T) * 0.5
def T(name, fn, k=3):
    r = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = fn()
    torch.cuda.synchronize(); print("%-28s %.2f ms" % (name, (time.perf_counter() - t0) / k * 1e3), flush=True)
    return r
nao, ld = h._nao_ao, h._ld
ao = T("eval_gto deriv 3", lambda: lib.eval_gto(h._tab, h.rgrid, 3))
dp = lib.pad_matrix(d, ld); lda = ao.shape[-1]; dq = dp[:lda, :lda]
rho, grho = T("grid_density", lambda: lib.grid_density(ao[:4], nao, dp, True))
b = T("b = ao0 @ D", lambda: ao[0] @ dq)
c = T("c = 3 x (ao_i @ D)", lambda: [ao[1 + i] @ dq for i in range(3)])
dens = ValGrad(value=rho, grad=grho)
edens = T("get_edensityxc", lambda: h.xc.get_edensityxc(dens))
pot = T("get_vxc", lambda: h.xc.get_vxc(dens))
w = h.dvolume
q, per_ao = T("fused terms kernel", lambda: lib.grid_xc_gradient_terms(ao, nao, b, c, w, pot.value, pot.grad, grho))
ao_atom = G._ao_owner(h, dev)
g = torch.zeros((20, 3), dtype=torch.float64, device=dev)
T("segment sums by owner", lambda: G._sum_by_owner(mol, q))
T("index_add ao", lambda: g.index_add_(0, ao_atom, -2.0 * per_ao))
def becke():
    pos = mol.atompos.to(dtype=torch.float64, device=dev).clone().requires_grad_(True)
    grid = get_predefined_grid(mol._grid_inp, mol.atomzs.tolist(), pos, dtype=torch.float64, device=dev)
    loss = (grid.get_dvolume() * edens.detach()).sum()
    return torch.autograd.grad(loss, pos)[0]
T("becke weight derivative", becke)
T("whole _xc_gradient", lambda: G._xc_gradient(eng, [d]))
