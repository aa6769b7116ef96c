"""tools/make_converged_golden.py -- BUILD-CONTAINER ONLY (hours of CPU): converged SCF energies of the two big BASELINE configs
by the ORACLE engine (oracle/hamilton.py, the CPU restatement of the reference's data flow hf.py / ks.py:157-187), committed as
tests/golden/oracle_converged_<case>.npz so that the GPU tests pin the SCF fixed point of C4 and C5 against the checker and not
against an earlier GPU run.

  c5m0 : molecule 0 of the C5 set (vitamin C, 20 atoms), RKS PBE / cc-pVDZ / sg3; packed-s4 ERI matrix (3.8 GB)
  c4   : naphthalene RKS PBE / cc-pVTZ / sg3 (nao 412); packed-s8 ERI triangle (29 GB; orc_int2e_s8 / orc_symv_s8)

The file holds: e_tot and its parts, the iteration count, max|F_out - F_in| at exit, checksums of the converged AO density
(sum |D|, tr(D S), and D contracted with a seeded probe), the wall times of the ERI fill / grid setup / SCF loop and the core
count.  Usage: python tools/make_converged_golden.py c5m0 c4
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import basis as ob, hamilton as oh, natives  # noqa: E402
from tests import molecules as M  # noqa: E402

XC = "gga_x_pbe+gga_c_pbe"
CASES = {
    "c5m0": (lambda: M.c5_molecule(0), "cc-pvdz", "s4"),
    "c4": (M.naphthalene, "cc-pvtz", "s8"),
    "benzene": (M.benzene, "cc-pvdz", "s8"),  # (small: checks this script and the s8 store against the s4 goldens)
}


def run(case, tol=1e-9):
    mol, basis, mode = CASES[case]
    t = ob.make_tables(mol(), basis)
    w0 = time.perf_counter()
    eng = oh.Engine(t, xc=XC, grid="sg3", eri_mode=mode)
    w_setup = time.perf_counter() - w0
    print("%s: nao %d, setup (ERI %s + grid) %.1f s" % (case, t.nao, mode, w_setup), flush=True)
    w0 = time.perf_counter()
    e = eng.run(maxiter=100, tol=tol)
    w_scf = time.perf_counter() - w0
    dm = eng.dm
    res = float((eng.dm2scp(dm) - eng.dm2scp(eng.scp2dm(eng.dm2scp(dm)))).abs().max())
    parts = eng.energy_parts(dm)
    X = eng.h.X
    dao = (X @ dm @ X.T).numpy()
    S = natives.int1e("ovlp", t)
    probe = np.random.default_rng(20260929).normal(size=dao.shape)
    probe = probe + probe.T
    parts.pop("e_tot")
    out = dict(e_tot=e, niter=eng.niter, fock_residual=res, dm_abs_sum=float(np.abs(dao).sum()), dm_trace_s=float((dao * S).sum()),
               dm_probe=float((dao * probe).sum()), probe_seed=20260929, wall_setup_s=w_setup, wall_scf_s=w_scf,
               cores=natives.num_threads(), tol=tol, xc=XC, basis=basis, grid="sg3", eri_mode=mode, **parts)
    print(case, out, flush=True)
    np.savez(os.path.join(ROOT, "tests", "golden", "oracle_converged_%s.npz" % case), **out)


if __name__ == "__main__":
    for c in sys.argv[1:] or ["c5m0"]:
        run(c)
