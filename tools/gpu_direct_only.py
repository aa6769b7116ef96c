"""C4 (naphthalene / cc-pVTZ) RKS PBE direct SCF from the core guess: wall time per iteration, host time inside the direct passes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DQC_AMD_ERI"] = "direct"
import torch, dqc_amd
from tests import molecules as M
dev = torch.device("cuda:0")
for rep in range(2):
    mol = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz", grid="sg3", device=dev)
    qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
    mol.get_hamiltonian()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    qc.run()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("direct scf %.3f s / %d it = %.1f ms per iteration  E %.9f  driver %s" % (t1 - t0, qc.niter, 1e3 * (t1 - t0) / qc.niter, float(qc.energy()), qc.driver_used), flush=True)
    del qc, mol
