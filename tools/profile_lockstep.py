"""lockstep SCF of N C5 molecules (for rocprofv3 --kernel-trace --stats: where does a batch iteration's GPU time go?)
usage: python tools/profile_lockstep.py [nmol] [group] [inflight]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.batch import run_lockstep
from tests import molecules as M
nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 16
group = int(sys.argv[2]) if len(sys.argv) > 2 else 16
infl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mols = [dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3") for i in range(nmol)]
qcs = [dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe") for m in mols]
run_lockstep(qcs[:2])
torch.cuda.synchronize()
t0 = time.perf_counter()
run_lockstep(qcs, group_size=group, inflight=infl)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
it = sum(q.niter for q in qcs)
print("lockstep %d molecules (group %d, inflight %d): %.3f s, %d iterations, %.3f ms per molecule-iteration" % (nmol, group, infl, dt, it, 1e3 * dt / it))
