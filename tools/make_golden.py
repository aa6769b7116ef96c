"""tools/make_golden.py -- BUILD-CONTAINER ONLY.  Generates tests/golden/*.npz, *.json.

Golden vectors are produced by the REFERENCE'S OWN Python (dqc.Mol, HamiltonCGTO, _HFEngine,
_KSEngine, SCF_QCCalc, dqc.grid) imported from /root/reference through tools/ref_harness.py, with the
oracle's C restatement of libcint/libcgto/libxc plugged in at the native seams.
The fixtures are *data*: inputs and expected outputs.  All matrices are stored in the AO basis
(libcint AO order), which is invariant to the eigenvector conventions of the orthogonaliser:
    dm_orth = X^T S dm_ao S X ,   M_ao = S X M_orth X^T S .

Usage:  python tools/make_golden.py [case ...]     (default: all small cases)
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import dqc  # noqa: E402
from dqc.qccalc.hf import _HFEngine  # noqa: E402,F401
from oracle import natives, basis as obasis  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
ANG = 1.0 / 0.52917721092


def benzene():
    rcc, rch = 1.397 * ANG, 1.084 * ANG
    zs, pos = [], []
    for k in range(6):
        a = np.pi / 3 * k
        zs.append(6)
        pos.append([rcc * np.cos(a), rcc * np.sin(a), 0.0])
    for k in range(6):
        a = np.pi / 3 * k
        zs.append(1)
        pos.append([(rcc + rch) * np.cos(a), (rcc + rch) * np.sin(a), 0.0])
    return zs, pos


H2O = ([8, 1, 1], [[0, 0, 0.2156], [0, 1.4749, -0.8625], [0, -1.4749, -0.8625]])  # test_properties.py:21-27

CASES = {
    # name: (moldesc, basis, xc (None = RHF), grid)
    "h2o_sto3g_rhf": (H2O, "sto-3g", None, None),
    "h2o_ccpvdz_rhf": (H2O, "cc-pvdz", None, None),
    "h2o_ccpvdz_lda_sg3": (H2O, "cc-pvdz", "lda_x+lda_c_pw", "sg3"),
    "h2o_ccpvdz_pbe_sg3": (H2O, "cc-pvdz", "gga_x_pbe+gga_c_pbe", "sg3"),
    "ch4_ccpvtz_pbe_sg2": (([6, 1, 1, 1, 1], [[0, 0, 0], [1.186, 1.186, 1.186], [-1.186, -1.186, 1.186],
                                               [-1.186, 1.186, -1.186], [1.186, -1.186, -1.186]]),
                           "cc-pvtz", "gga_x_pbe+gga_c_pbe", "sg2"),
    "h2o_ccpvdz_scan_sg2": (H2O, "cc-pvdz", "mgga_x_scan", "sg2"),
    "benzene_ccpvdz_rhf": (benzene(), "cc-pvdz", None, None),
    "benzene_ccpvdz_lda_sg3": (benzene(), "cc-pvdz", "lda_x+lda_c_pw", "sg3"),
}
# unrestricted cases: (moldesc, basis, xc, grid, spin)
CASES_POL = {
    "no_321g_uhf": (([7, 8], [[-1.0, 0, 0], [1.0, 0, 0]]), "3-21G", None, None, 1),
    # planar methyl radical (non-degenerate 2A2''), r_CH = 2.04 Bohr
    "ch3_ccpvdz_upbe_sg2": (([6, 1, 1, 1], [[0, 0, 0.0], [2.04, 0, 0], [-1.02, 1.766691, 0], [-1.02, -1.766691, 0]]),
                            "cc-pvdz", "gga_x_pbe+gga_c_pbe", "sg2", 1),
    "o2_ccpvdz_ulda_sg2": (([8, 8], [[-1.14, 0, 0], [1.14, 0, 0]]), "cc-pvdz", "lda_x+lda_c_pw", "sg2", 2),
}
SMALL = ["h2o_sto3g_rhf", "h2o_ccpvdz_rhf", "h2o_ccpvdz_lda_sg3", "h2o_ccpvdz_pbe_sg3", "ch4_ccpvtz_pbe_sg2"]


def seeded_dm_ao(nao, nel, S, seed):
    """symmetric PSD pseudo-density with tr(D S) = nel"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((nao, max(nel // 2, 1)))
    D = A @ A.T
    D *= nel / np.trace(D @ S)
    return D


def run_case(name):
    moldesc, basis, xc, grid = CASES[name]
    t0 = time.time()
    kw = {} if grid is None else {"grid": grid}
    mol = rh.ref_mol(moldesc, basis, **kw)
    qc = dqc.HF(mol) if xc is None else dqc.KS(mol, xc=xc)
    qc.run()
    eng = qc._engine if hasattr(qc, "_engine") else qc.get_system()  # SCF_QCCalc keeps ._engine
    e_tot = float(qc.energy())
    dm = qc.aodm()
    hamilt = mol.get_hamiltonian()
    X = hamilt._orthozer._orthozer.detach()
    tabs = obasis.make_tables(moldesc, basis)
    S = torch.as_tensor(natives.int1e("ovlp", tabs))
    SX = S @ X
    to_ao = lambda m: (SX @ m @ SX.T).numpy()  # noqa: E731
    dm_to_ao = lambda d: (X @ d @ X.T).numpy()  # noqa: E731  (unconvert_dm)
    out = {
        "atomzs": np.array(moldesc[0]), "atompos": np.array(moldesc[1], dtype=float),
        "e_tot": e_tot,
        "e_core": float(hamilt.get_e_hcore(dm)), "e_elrep": float(hamilt.get_e_elrep(dm)),
        "e_nuc": float(mol.get_nuclei_energy()),
        "dm_conv_ao": dm_to_ao(dm),
        "hcore_ao": to_ao(hamilt.get_kinnucl().fullmatrix()),
        "fock_conv_ao": to_ao(eng.dm2scp(dm)),
        "nao": X.shape[0],
    }
    if xc is None:
        out["e_exch"] = float(hamilt.get_e_exchange(dm))
    else:
        out["e_xc"] = float(hamilt.get_e_xc(dm))
        g = mol.get_grid()
        rg, dv = g.get_rgrid().numpy(), g.get_dvolume().numpy()
        out["ngrid"] = rg.shape[0]
        out["grid_checksum"] = np.array([dv.sum(), (dv * rg[:, 0]).sum(), (dv * (rg ** 2).sum(-1)).sum()])
    # probe set: seeded AO densities -> J, -K/2, Vxc, e_xc, rho samples (all AO basis)
    nel = int(sum(moldesc[0]))
    XtS = SX.T  # X^T S
    for k in range(3):
        Dao = seeded_dm_ao(tabs.nao, nel, S.numpy(), 1234 + k)
        dmo = XtS @ torch.as_tensor(Dao) @ XtS.T
        out["probe%d_dm_ao" % k] = Dao
        out["probe%d_J_ao" % k] = to_ao(hamilt.get_elrep(dmo).fullmatrix())
        if xc is None:
            out["probe%d_Khalf_ao" % k] = to_ao(hamilt.get_exchange(dmo).fullmatrix())
        else:
            out["probe%d_vxc_ao" % k] = to_ao(hamilt.get_vxc(dmo).fullmatrix())
            out["probe%d_exc" % k] = float(hamilt.get_e_xc(dmo))
            dens = hamilt._dm2densinfo(dmo)
            idx = np.linspace(0, out["ngrid"] - 1, 64).astype(int)
            out["probe_idx"] = idx
            out["probe%d_rho" % k] = dens.value.numpy()[idx]
            if dens.grad is not None:
                out["probe%d_grho" % k] = dens.grad.numpy()[:, idx]
            if dens.kin is not None:
                out["probe%d_kin" % k] = dens.kin.numpy()[idx]
                out["probe%d_lapl" % k] = dens.lapl.numpy()[idx]
    np.savez_compressed(os.path.join(GOLD, "ref_%s.npz" % name), **out)
    print("%-28s E = %.10f  (%.1f s)" % (name, e_tot, time.time() - t0), flush=True)


# density-fitted cases: (moldesc, basis, xc, grid, auxbasis) through the reference's own
# Mol.densityfit -> DFMol (dqc/system/mol.py:170-204, dqc/df/dfmol.py) with the even-tempered test auxiliary basis
CASES_DF = {
    "h2o_ccpvdz_pbe_sg2_etb": (H2O, "cc-pvdz", "gga_x_pbe+gga_c_pbe", "sg2", "etb"),
    "ch4_ccpvtz_lda_sg2_etb": (CASES["ch4_ccpvtz_pbe_sg2"][0], "cc-pvtz", "lda_x+lda_c_pw", "sg2", "etb"),
}


def run_case_df(name):
    from dqc.utils.datastruct import CGTOBasis
    moldesc, basis, xc, grid, auxname = CASES_DF[name]
    t0 = time.time()
    mol = rh.ref_mol(moldesc, basis, grid=grid)
    aux = [[CGTOBasis(angmom=l, alphas=torch.tensor(a, dtype=torch.float64), coeffs=torch.tensor(c, dtype=torch.float64))
            for (l, a, c) in obasis.even_tempered_aux(int(z))] for z in moldesc[0]]
    mol.densityfit(method="coulomb", auxbasis=aux)
    qc = dqc.KS(mol, xc=xc)
    qc.run()
    eng = qc._engine
    dm = qc.aodm()
    hamilt = mol.get_hamiltonian()
    X = hamilt._orthozer._orthozer.detach()
    tabs = obasis.make_tables(moldesc, basis)
    S = torch.as_tensor(natives.int1e("ovlp", tabs))
    SX = S @ X
    to_ao = lambda m: (SX @ m @ SX.T).numpy()  # noqa: E731
    df = hamilt.df
    j2c, j3c = df.j2c.numpy(), df.j3c.numpy()
    rng = np.random.default_rng(4)
    pi = rng.integers(0, j3c.shape[0], 64), rng.integers(0, j3c.shape[1], 64), rng.integers(0, j3c.shape[2], 64)
    out = {"atomzs": np.array(moldesc[0]), "atompos": np.array(moldesc[1], dtype=float), "e_tot": float(qc.energy()),
           "e_elrep": float(hamilt.get_e_elrep(dm)), "dm_conv_ao": (X @ dm @ X.T).numpy(),
           "fock_conv_ao": to_ao(eng.dm2scp(dm)), "naux": j2c.shape[0],
           "j2c_diag": np.diag(j2c).copy(), "j2c_row0": j2c[0].copy(), "j2c_fro": np.linalg.norm(j2c),
           "j3c_fro": np.linalg.norm(j3c), "j3c_probe_idx": np.stack(pi), "j3c_probe": j3c[pi],
           "j3c_sum_k": j3c.sum(-1)}
    nel = int(sum(moldesc[0]))
    XtS = SX.T
    for k in range(2):
        Dao = seeded_dm_ao(tabs.nao, nel, S.numpy(), 4321 + k)
        dmo = XtS @ torch.as_tensor(Dao) @ XtS.T
        out["probe%d_dm_ao" % k] = Dao
        out["probe%d_J_ao" % k] = to_ao(hamilt.get_elrep(dmo).fullmatrix())
    np.savez_compressed(os.path.join(GOLD, "refdf_%s.npz" % name), **out)
    print("%-28s E = %.10f  naux %d (%.1f s)" % (name, out["e_tot"], j2c.shape[0], time.time() - t0), flush=True)


def run_case_pol(name):
    """UHF / UKS through the reference's own polarised engine code (SpinParam plumbing of hf.py / ks.py / hcgto.py)"""
    from dqc.utils.datastruct import SpinParam
    moldesc, basis, xc, grid, spin = CASES_POL[name]
    t0 = time.time()
    kw = {"spin": spin}
    if grid is not None:
        kw["grid"] = grid
    mol = rh.ref_mol(moldesc, basis, **kw)
    qc = dqc.HF(mol) if xc is None else dqc.KS(mol, xc=xc)
    qc.run()
    eng = qc._engine
    dm = qc.aodm()
    assert isinstance(dm, SpinParam)
    hamilt = mol.get_hamiltonian()
    X = hamilt._orthozer._orthozer.detach()
    tabs = obasis.make_tables(moldesc, basis)
    S = torch.as_tensor(natives.int1e("ovlp", tabs))
    SX = S @ X
    to_ao = lambda m: (SX @ m @ SX.T).numpy()  # noqa: E731
    fock = eng.dm2scp(dm)
    out = {"atomzs": np.array(moldesc[0]), "atompos": np.array(moldesc[1], dtype=float), "spin": spin,
           "e_tot": float(qc.energy()), "dm_u_ao": (X @ dm.u @ X.T).numpy(), "dm_d_ao": (X @ dm.d @ X.T).numpy(),
           "fock_u_ao": to_ao(fock[0]), "fock_d_ao": to_ao(fock[1])}
    # probe: seeded spin densities -> polarised Vxc / exchange in the AO basis
    nel = int(sum(moldesc[0]))
    XtS = SX.T
    Du = seeded_dm_ao(tabs.nao, nel + spin, S.numpy(), 77) * 0.5
    Dd = seeded_dm_ao(tabs.nao, nel - spin, S.numpy(), 78) * 0.5
    dmo = SpinParam(u=XtS @ torch.as_tensor(Du) @ XtS.T, d=XtS @ torch.as_tensor(Dd) @ XtS.T)
    out["probe_du_ao"], out["probe_dd_ao"] = Du, Dd
    if xc is None:
        ex = hamilt.get_exchange(dmo)
        out["probe_ku_ao"], out["probe_kd_ao"] = to_ao(ex.u.fullmatrix()), to_ao(ex.d.fullmatrix())
    else:
        v = hamilt.get_vxc(dmo)
        out["probe_vu_ao"], out["probe_vd_ao"] = to_ao(v.u.fullmatrix()), to_ao(v.d.fullmatrix())
        out["probe_exc"] = float(hamilt.get_e_xc(dmo))
    np.savez_compressed(os.path.join(GOLD, "refpol_%s.npz" % name), **out)
    print("%-28s E = %.10f  (%.1f s)" % (name, out["e_tot"], time.time() - t0), flush=True)


def write_kats():
    """literal known answers held by the reference's own tests"""
    kat = {
        "_source": "literals copied from the reference test-suite (values only)",
        "rhf_321g": {"tol_rel": 1e-7, "src": "dqc/test/test_hf.py:18-32,47,51",
                     "cases": [["H", 1.0, -1.07195346], ["Li", 5.0, -14.7683688], ["N", 2.0, -108.298897],
                               ["F", 2.5, -197.636373], ["C O", 2.0, -112.078732]]},
        "rks_6311ppgss": {"tol_abs": 1.3e-3, "src": "dqc/test/test_ks.py:40-63,89-111",
                          "lda_x": [["H", 1.0, -0.979143262], ["Li", 5.0, -14.3927863482],
                                    ["N", 2.0, -107.726124018], ["F", 2.5, -197.005308558],
                                    ["C O", 2.0, -111.490687029]],
                          "gga_x_pbe": [["H", 1.0, -1.0682173104], ["Li", 5.0, -14.8282511868],
                                        ["N", 2.0, -108.980200151], ["F", 2.5, -198.772971537],
                                        ["C O", 2.0, -112.754279785]]},
        "h2_density": {"src": "dqc/test/test_hamilton.py:95-142", "atoms_z": 0.8, "basis": "3-21G",
                       "z": [0.0, 0.4, 0.8], "rho": [0.18742819, 0.23469519, 0.30250292]},
        "grid_gauss_integral": {"src": "dqc/test/test_grid.py:16-78", "value": 15.7496099457224,
                                "sg3_points": {"1": 16710, "6": 17674, "7": 18286, "8": 18946}},
        "uhf_321g": {"tol_rel": 1e-7, "src": "dqc/test/test_hf.py:141-206",
                     "atoms": [[1, 1, -4.96198609e-01], [3, 1, -7.38151326e+00], [5, 1, -2.43897617e+01], [8, 2, -7.43936572e+01]],
                     "mols": [[[7, 8], 2.0, 1, -1.28477807e+02]]},
        "uks_6311ppgss": {"tol_abs": 1.3e-3, "src": "dqc/test/test_ks.py:296-345 (atoms: grid 4, O2: grid 3)",
                          "atoms": {"lda_x": [[1, 1, -0.456918307830999], [3, 1, -7.19137615551071], [8, 2, -73.987463670134]],
                                    "gga_x_pbe": [[1, 1, -0.49413365762347017], [3, 1, -7.408839641982052], [8, 2, -74.77107826628823]]},
                          "o2": {"lda_x": -148.149998931489, "lda_x+lda_c_pw": -1.49259447e+02, "gga_x_pbe": -149.64097658035521}},
        "nuclei_energy": {"src": "dqc/test/test_system.py:61-72", "z": [1, 4], "dist": 1.5, "value": 4 / 1.5},
    }
    with open(os.path.join(GOLD, "reference_literals.json"), "w") as f:
        json.dump(kat, f, indent=1)


# fractional mode (mol.py:402-443): floating-point atomzs, spin given, last orbital partially occupied.
# (moldesc, basis, xc, grid, spin) -- the H2-like systems of test_hf.py:245-256 / test_ks.py:521-534 and a heavier one
CASES_FRAC = {
    "h2_z120_z125_321g_rhf": (([1.2, 1.25], [[-0.5, 0, 0], [0.5, 0, 0]]), "3-21G", None, None, 0),
    "h2_z120_z125_6311ppgss_lda_sg3": (([1.2, 1.25], [[-0.5, 0, 0], [0.5, 0, 0]]), "6-311++G**", "lda_x", "sg3", 0),
    # water with Z = (8.3, 1.1, 1.1): 10.5 electrons, the (non-degenerate) 4a1 orbital holds the half electron
    "h2o_z83_z11_ccpvdz_pbe_sg2": (([8.3, 1.1, 1.1], H2O[1]), "cc-pvdz", "gga_x_pbe+gga_c_pbe", "sg2", 0),
}


def run_case_frac(name):
    moldesc, basis, xc, grid, spin = CASES_FRAC[name]
    t0 = time.time()
    kw = {"spin": spin}
    if grid is not None:
        kw["grid"] = grid
    zs = torch.tensor(moldesc[0], dtype=torch.float64)
    mol = rh.ref_mol((zs, torch.tensor(moldesc[1], dtype=torch.float64)), basis, **kw)
    qc = dqc.HF(mol, restricted=True) if xc is None else dqc.KS(mol, xc=xc, restricted=True)
    qc.run()
    dm = qc.aodm()
    hamilt = mol.get_hamiltonian()
    X = hamilt._orthozer._orthozer.detach()
    out = {"atomzs": np.array(moldesc[0], dtype=float), "atompos": np.array(moldesc[1], dtype=float), "spin": float(spin),
           "e_tot": float(qc.energy()), "e_nuc": float(mol.get_nuclei_energy()),
           "orb_weight": mol.get_orbweight().numpy(), "orb_weight_u": mol.get_orbweight(polarized=True).u.numpy(),
           "orb_weight_d": mol.get_orbweight(polarized=True).d.numpy(), "dm_conv_ao": (X @ dm @ X.T).numpy()}
    np.savez_compressed(os.path.join(GOLD, "reffrac_%s.npz" % name), **out)
    print("%-36s E = %.10f  weights %s (%.1f s)" % (name, out["e_tot"], out["orb_weight"], time.time() - t0), flush=True)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    write_kats()
    for c in (sys.argv[1:] or SMALL + list(CASES_POL) + list(CASES_DF) + list(CASES_FRAC)):
        (run_case_pol if c in CASES_POL else (run_case_df if c in CASES_DF else
                                              (run_case_frac if c in CASES_FRAC else run_case)))(c)
