"""several one-molecule SCF drivers in flight (batch.run_concurrent): does the one-launch projector (persistent workers that must all
be resident on one XCD) hold up when several of them are queued at once?  Counts eigh fallbacks and times the batch, with the
persistent kernel (default) and with the multi-launch purification (DQC_AMD_PURIFY=launch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.batch import run_concurrent
from tests import molecules as M
nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, mk in (("c5", lambda i: dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe")),
                 ("benzene", lambda i: dqc_amd.KS(dqc_amd.Mol(M.benzene(), basis="cc-pvdz", grid="sg3"), xc="lda_x+lda_c_pw"))):
    qcs = [mk(i) for i in range(nmol)]
    torch.cuda.synchronize()
    for inflight in (4, 8, 16):
        for q in qcs:
            q.eigh_fallbacks = 0
        t0 = time.perf_counter()
        run_concurrent(qcs, max_inflight=inflight)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        its = sum(q.niter for q in qcs)
        print("%s x %d, %2d in flight (DQC_AMD_PURIFY=%s): %.3f s, %d iterations -> %.0f /s, eigh fallbacks %d, energy[0] %.10f" % (
            name, nmol, inflight, os.environ.get("DQC_AMD_PURIFY", "persistent"), dt, its, its / dt, sum(getattr(q, "eigh_fallbacks", 0) for q in qcs), float(qcs[0].energy())), flush=True)
    del qcs
    torch.cuda.empty_cache()
