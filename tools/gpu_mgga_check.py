"""GPU check of the meta-GGA path: AO laplacian, pair kernels, SCAN functional, SCF energies."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import dqc_amd
from dqc_amd import lib
from oracle import basis as ob, natives as nat, grid as og, xc as oxc, hamilton as oh
from tests import molecules as M
dev = torch.device("cuda")
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
t = ob.make_tables(M.CH4, "cc-pvtz"); tab = lib.Tables(t.atm, t.bas, t.env)
rg, dv = og.get_predefined_grid("sg2", t.atomzs, t.atompos); rg, dv = rg[:-5], dv[:-5]
ao = lib.eval_gto(tab, torch.as_tensor(rg, device=dev), 2).cpu().numpy()
print("ao lapl", rel(ao[4][:, :tab.nao], nat.eval_gto(t, rg, 2).T), "grad", rel(ao[1:4, :, :tab.nao], nat.eval_gto(t, rg, 1).transpose(0, 2, 1)))
aod = lib.eval_gto(tab, torch.as_tensor(rg, device=dev), 2)
D = M.seeded_dm_ao(tab.nao, 10, nat.int1e("ovlp", t), 3); Dp = lib.pad_matrix(torch.as_tensor(D, device=dev), aod.shape[-1])
a, b = ao[1][:, :tab.nao], ao[4][:, :tab.nao]
print("density_pair", rel(lib.grid_density_pair(aod[1], aod[4], tab.nao, Dp).cpu().numpy(), np.einsum("gi,ij,gj->g", a, D, b)))
v = np.random.default_rng(0).standard_normal(rg.shape[0])
m = (a * (dv * v)[:, None]).T @ b; m = 0.5 * (m + m.T)
print("vxc_pair", rel(lib.grid_vxc_pair(aod[1], aod[4], tab.nao, torch.as_tensor(dv, device=dev), torch.as_tensor(v, device=dev)).cpu().numpy()[:tab.nao, :tab.nao], m))
rng = np.random.default_rng(1); n = 4000
rho = rng.uniform(0, 1.5, n) ** 2; g = rng.standard_normal((3, n)) * rho; sig = (g * g).sum(0)
tau = sig / (8 * np.maximum(rho, 1e-30)) + rng.uniform(0, 1, n) * rho ** (5 / 3) * 5
x = oxc.get_xc("mgga_x_scan+0.3*gga_c_pbe")
e, vr, vg, vt = lib.xc_eval_mgga(x.terms, *(torch.as_tensor(q, device=dev) for q in (rho, g, tau)))
er, vrr, vsr, vtr = x.compute_mgga(rho, sig, tau)
ok = rho > 1e-6
print("scan e", rel(e.cpu().numpy(), er), "vrho", rel(vr.cpu().numpy()[ok], vrr[ok]), "vgrad", rel(vg.cpu().numpy()[:, ok], (2 * vsr * g)[:, ok]), "vtau", rel(vt.cpu().numpy()[ok], vtr[ok]))
for sym, d, ref in [("Li", 5.0, -14.8687500), ("N", 2.0, -109.055074), ("C O", 2.0, -112.836255)]:
    s = sym.split(); aa, bb = (s[0], s[0]) if len(s) == 1 else s
    md = "%s %g 0 0; %s %g 0 0" % (aa, -d / 2, bb, d / 2)
    mol = dqc_amd.Mol(md, basis="6-311++G**", grid=4)
    qc = dqc_amd.KS(mol, xc="mgga_x_scan").run(fwd_options={"maxiter": 150}); eg = float(qc.energy())
    eo, _ = oh.run_scf(md, "6-311++G**", xc="mgga_x_scan", grid=4, maxiter=150)
    print("SCAN", sym, "gpu", eg, "oracle", eo, "ref", ref, "gpu-oracle %.2e" % (eg - eo), "gpu-ref %.2e" % (eg - ref), qc.niter, qc.converged)
