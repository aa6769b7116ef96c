"""GPU check: SCF with the purification step vs the eigh step -- same energies / iteration counts, time to energy"""
import sys, os, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dqc_amd
from tests import molecules as M
print = functools.partial(print, flush=True)
dev = torch.device("cuda:0")
for name, mol, basis, xc in [("h2o", M.H2O, "cc-pvdz", "gga_x_pbe+gga_c_pbe"), ("benzene-rhf", M.benzene(), "cc-pvdz", None),
                             ("c5-pbe", M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe"), ("c5-rhf", M.c5_molecule(0), "cc-pvdz", None)]:
    m = dqc_amd.Mol(mol, basis=basis, grid="sg3", device=dev)
    res = {}
    for diag in ("eigh", "purify", "eigh", "purify"):
        qc = dqc_amd.KS(m, xc=xc) if xc else dqc_amd.HF(m)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        qc.run(fwd_options={"diag": diag})
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        res[diag] = (float(qc.energy()), qc.niter, t, qc.converged)
    (ee, ne, te, ce), (ep, npu, tp, cp) = res["eigh"], res["purify"]
    print("%-12s eigh: E %.10f niter %d %.3f s | purify: E %.10f niter %d %.3f s | dE %.1e  speed-up %.2fx" % (name, ee, ne, te, ep, npu, tp, ep - ee, te / tp))
    assert ce and cp and abs(ee - ep) < 1e-8
print("PURIFY OK")
