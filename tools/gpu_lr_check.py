"""GPU check of the rank-n_occ density kernel: parity with the dense kernel / numpy and A/B timing (C5 molecule)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import molecules as M
from oracle import basis as ob, grid as og, natives as nat
from dqc_amd import lib

dev = torch.device("cuda:0")


def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


for name, mol, basis, gridn, nocc in [("h2o", M.H2O, "cc-pvdz", "sg2", 5), ("ch4-tz", M.CH4, "cc-pvtz", "sg2", 5),
                                      ("c5", M.c5_molecule(0), "cc-pvdz", "sg3", 46),
                                      ("c5-r20", M.c5_molecule(1), "cc-pvdz", "sg3", 20),
                                      ("c5-r70", M.c5_molecule(1), "cc-pvdz", "sg3", 70),
                                      ("c5-r100", M.c5_molecule(1), "cc-pvdz", "sg3", 100),
                                      ("naph-tz", M.naphthalene(), "cc-pvtz", "sg2", 34)]:
    t = ob.make_tables(mol, basis)
    tab = lib.Tables(t.atm, t.bas, t.env)
    rg, dv = og.get_predefined_grid(gridn, t.atomzs, t.atompos)
    if name == "h2o":
        rg = rg[:-7]
    ao = lib.eval_gto(tab, torch.as_tensor(rg, device=dev), 1)
    nao, ld = tab.nao, lib.padded_nao(tab.nao)
    rng = np.random.default_rng(5)
    L = rng.standard_normal((nao, nocc)) / np.sqrt(nao)
    D = L @ L.T
    Dp = lib.pad_matrix(torch.as_tensor(D, device=dev), ld)
    fac = lib.pad_factor(torch.as_tensor(L, device=dev), ld)
    for gga in (False, True):
        r0, g0 = lib.grid_density(ao, nao, Dp, gga)
        r1, g1 = lib.grid_density_lr(ao, nao, fac, gga)
        er = float((r0 - r1).abs().max() / r0.abs().max())
        eg = float((g0 - g1).abs().max() / g0.abs().max()) if gga else 0.0
        t0 = timeit(lambda: lib.grid_density(ao, nao, Dp, gga))
        t1 = timeit(lambda: lib.grid_density_lr(ao, nao, fac, gga))
        print("%-8s nao %3d r %3d G %6d gga %d  |rho| err %.1e grad err %.1e   dense %.3f ms  lr %.3f ms" %
              (name, nao, nocc, rg.shape[0], gga, er, eg, t0, t1), flush=True)
        assert (er < 1e-12 and eg < 1e-12) or os.environ.get("DQC_AMD_LIB")
    if name in ("h2o", "ch4-tz") and not os.environ.get("DQC_AMD_LIB"):
        a = ao.cpu().numpy()[:, :, :nao]
        rr = np.einsum("gi,ij,gj->g", a[0], D, a[0])
        assert np.abs(r1.cpu().numpy() - rr).max() / np.abs(rr).max() < 1e-12
print("LR OK")
