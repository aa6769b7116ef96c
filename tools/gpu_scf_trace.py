"""max|[F,D]| per SCF iteration of a C5 molecule (convergence profile of the DIIS driver)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
for i in (0, 5):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
    gen = qc._run_gen()
    req = next(gen)
    errs = []
    try:
        while True:
            host = req.cpu().numpy()
            errs.append(float(host[0]))
            req = gen.send(host)
    except StopIteration:
        pass
    print("molecule %d: %d iterations, E = %.10f" % (i, qc.niter, float(qc.energy())))
    print("  " + " ".join("%.1e" % e for e in errs))
