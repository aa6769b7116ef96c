"""molecules x SCF iterations per second for a batch of small molecules on ONE GPU, serial loop
(`qc.run()` one after the other) vs `dqc_amd.batch.run_concurrent` (one stream per molecule in flight).
usage: python tools/concurrent_small.py [nmol] [out.txt]      (default 256 benzene RKS LDA/cc-pVDZ sg3, BASELINE config C3)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dqc_amd  # noqa: E402
from dqc_amd.batch import run_concurrent  # noqa: E402
from tests import molecules as M  # noqa: E402

nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = []


def build(kind, n):
    qcs = []
    for i in range(n):
        if kind == "benzene":
            zs, pos = M.benzene()
            xc, basis, grid = "lda_x+lda_c_pw", "cc-pvdz", "sg3"
        else:
            zs, pos = M.H2O
            xc, basis, grid = "gga_x_pbe+gga_c_pbe", "cc-pvdz", "sg3"
        pos = np.array(pos) + np.random.default_rng(100 + i).normal(0, 0.02, (len(zs), 3))
        qcs.append(dqc_amd.KS(dqc_amd.Mol((zs, pos.tolist()), basis=basis, grid=grid), xc=xc))
    return qcs


for kind, n in (("h2o", nmol), ("benzene", nmol)):
    t0 = time.perf_counter()
    qa = build(kind, n)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    for q in qa:
        q.run()
    ea = [float(q.energy()) for q in qa]
    torch.cuda.synchronize()
    t_serial = time.perf_counter() - t0
    its = sum(q.niter for q in qa)
    del qa
    qb = build(kind, n)
    torch.cuda.synchronize()
    for inflight in (8, 16, 32):
        for q in qb:
            q._has_run = False
        t0 = time.perf_counter()
        run_concurrent(qb, max_inflight=inflight)
        eb = [float(q.energy()) for q in qb]
        torch.cuda.synchronize()
        t_conc = time.perf_counter() - t0
        itb = sum(q.niter for q in qb)
        line = ("%-8s x%-4d setup %.2fs | serial %.3fs (%d its, %.0f mol-it/s) | %2d streams %.3fs (%d its, %.0f mol-it/s) x%.2f | max|dE| %.1e all accepted %s"
                % (kind, n, t_setup, t_serial, its, its / t_serial, inflight, t_conc, itb, itb / t_conc, (itb / t_conc) / (its / t_serial),
                   max(abs(a - b) for a, b in zip(ea, eb)), all(q.accepted for q in qb)))
        print(line, flush=True)
        out.append(line)
    del qb
    torch.cuda.empty_cache()
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
