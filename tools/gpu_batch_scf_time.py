"""time-to-energy of a batch of C5 molecules with N molecules in flight (batch.run_concurrent); env GPU_MAX_HW_QUEUES etc. apply.
usage: python tools/gpu_batch_scf_time.py [nmol] [inflight ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.batch import run_concurrent
from tests import molecules as M
nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 16
infl = [int(a) for a in sys.argv[2:]] or [8]
mols = [dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3") for i in range(nmol)]
def fresh():
    return [dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe") for m in mols]
run_concurrent(fresh()[:4], max_inflight=4)
torch.cuda.synchronize()
for n in infl:
    qcs = fresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_concurrent(qcs, max_inflight=n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = sum(q.niter for q in qcs)
    print("GPU_MAX_HW_QUEUES=%s  %d molecules, %d in flight: %.3f s, %d iterations, %.3f ms/iteration, accepted %d" % (
        os.environ.get("GPU_MAX_HW_QUEUES"), nmol, n, dt, it, 1e3 * dt / it, sum(q.accepted for q in qcs)))
