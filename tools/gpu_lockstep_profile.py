"""the 32-molecule C5 lockstep SCF (bench: batch_scf) alone -- for a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/pl -- python $REPO/tools/gpu_lockstep_profile.py [nmol]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.batch import run_lockstep, prepare_orthogonalisers
from tests import molecules as M
nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kw = {}
if len(sys.argv) > 2: kw["group_size"] = int(sys.argv[2])
if len(sys.argv) > 3: kw["inflight"] = int(sys.argv[3])
if len(sys.argv) > 4: kw["nstreams"] = int(sys.argv[4])
mols = [dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3") for i in range(nmol)]
prepare_orthogonalisers([m.get_hamiltonian() for m in mols])
qcs = [dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe") for m in mols]
torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_lockstep(qcs, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    its = sum(q.niter for q in qcs)
    print("lockstep SCF of %d molecules %s: %.3f s, %d molecule-iterations -> %.1f /s" % (nmol, kw, dt, its, its / dt), flush=True)
