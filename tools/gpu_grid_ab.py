"""grid kernels (factor-form density, Vxc) of the BASELINE configs on their REAL AO data, timed call by call with HIP events;
fractions on ALGORITHMIC flops / bytes (nao, not the padded widths).  A/B runs: DQC_AO_ALIGN=2|8|16 (row stride of the AO arrays).
usage: python tools/gpu_grid_ab.py [C3 C3pbe C4 C5 ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
CFG = {"C3": (M.benzene(), "cc-pvdz", "lda_x+lda_c_pw"), "C3pbe": (M.benzene(), "cc-pvdz", "gga_x_pbe+gga_c_pbe"),
       "C4": (M.naphthalene(), "cc-pvtz", "gga_x_pbe+gga_c_pbe"), "C5": (M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe")}


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name in (sys.argv[1:] or ["C3", "C3pbe", "C4", "C5"]):
    mol_in, basis, xc = CFG[name]
    os.environ["DQC_AMD_ERI"] = "direct"  # (no tile store: only the grid pass is timed here)
    mol = dqc_amd.Mol(mol_in, basis=basis, grid="sg3")
    qc = dqc_amd.KS(mol, xc=xc)
    eng, h = qc._engine, qc._engine.hamilton
    n = eng.shape[-1]
    dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda")))
    orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()
    d = h.ao_orb2dm(orb, eng.orb_weight)
    fac = h._factor_of(d)
    gga = h.xcfamily == 2
    nao, G, c = h._nao_ao, h.rgrid.shape[0], 4 if gga else 1
    rho, grho = lib.grid_density_lr(h._ao, nao, fac[0], gga)
    _, v, vg = lib.xc_eval(h.xc.terms, rho, grho, want_e=False, want_v=True)
    t_d = timeit(lambda: lib.grid_density_lr(h._ao, nao, fac[0], gga))
    t_v = timeit(lambda: lib.grid_vxc(h._ao, nao, h.dvolume, v, vg))
    r = fac[0][0].shape[1]
    by = 8.0 * c * G * nao
    fl_v = 2.0 * G * nao * nao + 4.0 * c * G * nao
    fl_d = 2.0 * (2.0 * G * nao * r) + 2.0 * c * G * nao
    # parity of the two kernels against torch on the same device data
    a = h._ao[..., :nao] if gga else h._ao[:, :nao].unsqueeze(0)
    L = fac[0][0][:nao]
    al = a[0] @ L
    rho_ref = (al * al).sum(1)
    psi = (h.dvolume * v)[:, None] * a[0]
    if gga:
        psi = psi + 2 * (h.dvolume[None, :, None] * vg[:, :, None] * a[1:]).sum(0)
    mref = a[0].T @ psi
    mref = 0.5 * (mref + mref.T)
    vm = lib.grid_vxc(h._ao, nao, h.dvolume, v, vg)
    print(json.dumps({"config": name, "ao_align": os.environ.get("DQC_AO_ALIGN", "2"), "nao": nao, "ld": h._ld, "lda": h._lda, "ngrid": G, "norb_pad": r,
                      "density_lr_ms": round(t_d, 4), "density_frac_hbm": round(by / t_d / 1e6 / 8000, 3), "density_tf": round(fl_d / t_d / 1e9, 1),
                      "vxc_ms": round(t_v, 4), "vxc_frac_mfma": round(fl_v / t_v / 1e9 / 78.6, 3), "vxc_frac_hbm": round(by / t_v / 1e6 / 8000, 3),
                      "rho_err": float((rho - rho_ref).abs().max() / rho_ref.abs().max()),
                      "vxc_err": float((vm[:nao, :nao] - mref).abs().max() / mref.abs().max()),
                      "vxc_pad_max": float(vm[nao:].abs().max()) if h._ld > nao else 0.0}), flush=True)
    del mol, qc, eng, h
    torch.cuda.empty_cache()
