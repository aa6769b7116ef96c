"""Stress test of the large-shape code paths (ld > 512: unspecialised Vxc kernel, multi-panel density kernels, n_occ > 128
-> full-matrix density kernel, 1521-block purification): three vitamin-C molecules 60 Bohr apart, RKS-PBE / cc-pVDZ with
the density-fitted Coulomb operator.  Non-interacting copies: E(3 mol) == 3 E(1 mol) up to the dipole-dipole term."""
import sys, os, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dqc_amd
from tests import molecules as M
print = functools.partial(print, flush=True)
zs, pos = M.c5_molecule(0)
pos = np.array(pos)
xc = "gga_x_pbe+gga_c_pbe"
t0 = time.perf_counter()
m1 = dqc_amd.Mol((zs, pos.tolist()), basis="cc-pvdz", grid="sg2").densityfit(auxbasis="etb")
q1 = dqc_amd.KS(m1, xc=xc).run()
e1 = float(q1.energy())
print("1 molecule : E = %.8f  niter %d  %.2f s" % (e1, q1.niter, time.perf_counter() - t0))
t0 = time.perf_counter()
zs3 = list(zs) * 3
pos3 = np.concatenate([pos, pos + [60.0, 0, 0], pos + [0, 60.0, 0]])
m3 = dqc_amd.Mol((zs3, pos3.tolist()), basis="cc-pvdz", grid="sg2").densityfit(auxbasis="etb")
q3 = dqc_amd.KS(m3, xc=xc).run(fwd_options={"maxiter": 150})  # three degenerate copies: plain DIIS needs ~80-90 iterations
torch.cuda.synchronize()
e3 = float(q3.energy())
h = m3.get_hamiltonian()
print("3 molecules: E = %.8f  niter %d converged %s  %.2f s   nao %d ld %d naux %d ngrid %d  mem %.1f GB" %
      (e3, q3.niter, q3.accepted, time.perf_counter() - t0, h._nao_ao, h._ld, h.df.j2c.shape[0], h.rgrid.shape[0], torch.cuda.max_memory_allocated() / 1e9))
print("E(3) - 3 E(1) = %.2e Ha   last max|[F,D]| = %.2e (1 molecule: %.2e)" % (e3 - 3 * e1, q3.scf_error, q1.scf_error))
assert q3.accepted and abs(e3 - 3 * e1) < 2e-4
print("BIG OK")
