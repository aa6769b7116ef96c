"""Run one BASELINE.json config end to end on the GPU and print energy + timings.
usage: python tools/run_config.py C1|C2|C3|C4|C5 [--scf]"""
import sys, time, json
import torch
sys.path.insert(0, ".")
import dqc_amd
from tests import molecules as M

CFG = {
    "C1": (M.H2O, "sto-3g", None),
    "C2": (M.benzene(), "cc-pvdz", None),
    "C3": (M.benzene(), "cc-pvdz", "lda_x+lda_c_pw"),
    "C4": (M.naphthalene(), "cc-pvtz", "gga_x_pbe+gga_c_pbe"),
    "C5": (M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe"),
}
name = sys.argv[1]
mol_in, basis, xc = CFG[name]
torch.cuda.synchronize()
t0 = time.perf_counter()
mol = dqc_amd.Mol(mol_in, basis=basis, grid="sg3")
qc = dqc_amd.HF(mol) if xc is None else dqc_amd.KS(mol, xc=xc)
torch.cuda.synchronize()
t1 = time.perf_counter()
eng = qc._engine
h = eng.hamilton
n = eng.shape[-1]
dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda")))
orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()   # D as every SCF iteration produces it: ao_orb2dm(C_occ, n)
for _ in range(2):
    eng.dm2scp(h.ao_orb2dm(orb, eng.orb_weight))
torch.cuda.synchronize()
t2 = time.perf_counter()
K = 10
for _ in range(K):
    eng.dm2scp(h.ao_orb2dm(orb, eng.orb_weight))
torch.cuda.synchronize()
t3 = time.perf_counter()
out = {"config": name, "nao": h._nao_ao, "ld": h._ld, "ngrid": int(h.rgrid.shape[0]) if h.is_grid_set else 0,
       "tiles_GB": h._tiles.numel() * 8 / 1e9, "setup_s": t1 - t0, "dm2scp_ms": 1e3 * (t3 - t2) / K,
       "mem_GB": torch.cuda.max_memory_allocated() / 1e9}
if "--scf" in sys.argv:
    t4 = time.perf_counter()
    qc.run()
    e = float(qc.energy())
    torch.cuda.synchronize()
    out.update({"energy": e, "niter": qc.niter, "converged": qc.converged, "accepted": qc.accepted, "scf_error": qc.scf_error, "scf_s": time.perf_counter() - t4})
print(json.dumps(out), flush=True)
