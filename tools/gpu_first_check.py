"""First GPU parity check of every kernel against the oracle (run through gpurun)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import basis as ob, natives as nat, grid as og, xc as oxc
from dqc_amd import lib

dev = torch.device("cuda")
H2O = ([8, 1, 1], [[0, 0, 0.2156], [0, 1.4749, -0.8625], [0, -1.4749, -0.8625]])
CH4 = ([6, 1, 1, 1, 1], [[0, 0, 0], [1.186, 1.186, 1.186], [-1.186, -1.186, 1.186], [-1.186, 1.186, -1.186], [1.186, -1.186, -1.186]])

def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

for name, mol, bas in [("h2o/sto-3g", H2O, "sto-3g"), ("h2o/cc-pvdz", H2O, "cc-pvdz"), ("ch4/cc-pvtz", CH4, "cc-pvtz")]:
    t = ob.make_tables(mol, bas)
    tab = lib.Tables(t.atm, t.bas, t.env)
    print("==", name, "nao", tab.nao, flush=True)
    for w in ("ovlp", "kin", "nuc"):
        g = lib.int1e(w, tab, dev).cpu().numpy()
        print("  int1e", w, rel(g, nat.int1e(w, t)))
    t0 = time.time(); tiles = lib.eri_tiles(tab, dev); torch.cuda.synchronize(); t1 = time.time()
    dense = lib.eri_dense(tiles, tab.nao).cpu().numpy()
    ref = nat.int2e(t)
    print("  eri   ", rel(dense, ref), "absmax", np.abs(dense - ref).max(), "time %.3fs" % (t1 - t0), flush=True)
    rng = np.random.default_rng(1)
    A = rng.standard_normal((tab.nao, tab.nao)); D = A @ A.T / tab.nao
    work = lib.jk_workspace(tab.nao, dev)
    J, K = lib.jk(tiles, torch.as_tensor(D, device=dev), work, True)
    Jr = np.einsum("ij,ijkl->kl", D, ref); Kr = np.einsum("il,ijkl->jk", D, ref)
    print("  J     ", rel(J.cpu().numpy(), Jr), " K", rel(K.cpu().numpy(), Kr))
    J2, _ = lib.jk(tiles, torch.as_tensor(D, device=dev), work, False)
    print("  J-only", rel(J2.cpu().numpy(), Jr))
    # grid
    rg, dv = og.get_predefined_grid("sg2", t.atomzs, t.atompos)
    rgd = torch.as_tensor(rg, device=dev)
    ao = lib.eval_gto(tab, rgd, 1)
    ao_ref = nat.eval_gto(t, rg, 0).T; gao_ref = nat.eval_gto(t, rg, 1).transpose(0, 2, 1)
    aoh = ao.cpu().numpy()
    print("  ao    ", rel(aoh[0][:, :tab.nao], ao_ref), " grad", rel(aoh[1:, :, :tab.nao], gao_ref), "pad", np.abs(aoh[:, :, tab.nao:]).max() if aoh.shape[-1] > tab.nao else 0.0)
    ld = ao.shape[-1]
    Dp = lib.pad_matrix(torch.as_tensor(D, device=dev), ld)
    rho, grho = lib.grid_density(ao, tab.nao, Dp, True)
    dmao = ao_ref @ D
    rho_ref = np.einsum("ri,ri->r", dmao, ao_ref); grho_ref = 2 * np.einsum("ri,dri->dr", dmao, gao_ref)
    print("  rho   ", rel(rho.cpu().numpy(), rho_ref), " grho", rel(grho.cpu().numpy(), grho_ref))
    ao0 = ao[0].contiguous()
    rho0, _ = lib.grid_density(ao0, tab.nao, Dp, False)
    print("  rho(LDA path)", rel(rho0.cpu().numpy(), rho_ref))
    for xcs in ("lda_x+lda_c_pw", "gga_x_pbe+gga_c_pbe"):
        x = oxc.get_xc(xcs)
        e, v, vg = lib.xc_eval(x.terms, rho, grho if x.family == 2 else None)
        sig = np.einsum("dr,dr->r", grho_ref, grho_ref)
        er, vr, vs = x.compute(rho_ref, sig)
        print("  xc", xcs, "e", rel(e.cpu().numpy(), er), "vrho", rel(v.cpu().numpy(), vr), end="")
        wd = torch.as_tensor(dv, device=dev)
        if x.family == 2:
            vgr = 2 * vs[None] * grho_ref
            print(" vgrad", rel(vg.cpu().numpy(), vgr), end="")
            vm = lib.grid_vxc(ao, tab.nao, wd, v, vg)
            vb = vr[:, None] * ao_ref + 2 * np.einsum("dr,dri->ri", vgr, gao_ref)
        else:
            vm = lib.grid_vxc(ao0, tab.nao, wd, v, None)
            vb = vr[:, None] * ao_ref
        m = (ao_ref * dv[:, None]).T @ vb; m = 0.5 * (m + m.T)
        print(" vxc", rel(vm.cpu().numpy()[:tab.nao, :tab.nao], m))
print("DONE")
