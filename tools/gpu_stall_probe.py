"""repeat the lockstep SCF of the 32-molecule C5 batch: converged / stalled counts, iteration histogram, and the error
trajectory of any molecule that stalls (DQC_AMD_SCF_TRACE-style), to tell a reproducible stagnation from run-to-run noise"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd.batch import run_lockstep
from tests import molecules as M
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nmol = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mols = [dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3") for i in range(nmol)]
for m in mols: m.get_hamiltonian()
os.environ["DQC_AMD_SCF_TRACE"] = "1"
for rep in range(nrep):
    qcs = [dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe") for m in mols]
    buf = io.StringIO()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        run_lockstep(qcs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    its = [q.niter for q in qcs]
    st = [i for i, q in enumerate(qcs) if q.stalled]
    print("rep %d: %.2f s, %d iterations, converged %d, stalled %s, max iters %d, eigh fallbacks %d" % (
        rep, t1 - t0, sum(its), sum(q.converged for q in qcs), st, max(its), sum(q.eigh_fallbacks for q in qcs)), flush=True)
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("lockstep it")]
    for i in st:
        # the trace prints one column per molecule of the GROUP the molecule ran in; print every line's min / the molecule column if whole batch
        print("   molecule %d: niter %d, error %.2e" % (i, qcs[i].niter, qcs[i].scf_error))
    if st:
        for l in lines[-60:]:
            cols = l.split(":")[1].split()
            print("   ", l.split(":")[0], " ".join(cols[:16]))
