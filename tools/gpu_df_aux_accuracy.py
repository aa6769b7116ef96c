"""accuracy and size of the auxiliary sets against the exact Coulomb operator: converged LDA energies of vitamin C / benzene / H2O
(cc-pVDZ), exact J vs density-fitted J with auxbasis = etb, autoaux:beta"""
import sys, os, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
warnings.simplefilter("ignore")
for name, geo in (("H2O", M.H2O), ("benzene", M.benzene()), ("vitamin C", M.c5_molecule(0))):
    ex = float(dqc_amd.KS(dqc_amd.Mol(geo, basis="cc-pvdz"), xc="lda_x+lda_c_pw").run().energy())
    for aux in ("etb", "autoaux:2.5", "autoaux:2.2", "autoaux", "autoaux:1.8", None):
        mol = dqc_amd.Mol(geo, basis="cc-pvdz")
        mol.densityfit(auxbasis=aux) if aux else mol.densityfit()
        qc = dqc_amd.KS(mol, xc="lda_x+lda_c_pw")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        qc.run(); e = float(qc.energy()); dt = time.perf_counter() - t0
        h = mol.get_hamiltonian()
        print("%-10s aux %-12s naux %5d  E_df - E_exact = %+.3e Ha  (%d iterations, %.3f s)" % (name, aux or "(default)", h.df.j2c.shape[0], e - ex, qc.niter, dt), flush=True)
