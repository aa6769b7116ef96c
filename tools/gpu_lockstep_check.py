"""lockstep batched SCF vs the one-molecule driver: energies, iteration counts, wall time.
usage: python tools/gpu_lockstep_check.py [small|c5] [nmol]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from dqc_amd.batch import run_concurrent, run_lockstep
from tests import molecules as M

what = sys.argv[1] if len(sys.argv) > 1 else "small"
nmol = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda")

# unit checks of the batched kernels
rng = np.random.default_rng(1)
for m in (1, 2, 5, 12):
    H = 12
    E = rng.standard_normal((3, H, 500)) * 10.0 ** rng.uniform(-2, 0, (3, H, 1))
    G = np.einsum("mik,mjk->mij", E, E)
    c = lib.diis_solve(torch.as_tensor(G, device=dev), m).cpu().numpy()
    for b in range(3):
        g = G[b, :m, :m]
        B = np.zeros((m + 1, m + 1)); B[:m, :m] = g / g.diagonal().max(); B[m, :m] = B[:m, m] = -1
        rhs = np.zeros(m + 1); rhs[m] = -1
        ref = np.linalg.lstsq(B, rhs, rcond=None)[0][:m]
        assert np.allclose(c[b, :m], ref, rtol=1e-6, atol=1e-9), (m, c[b, :m], ref)
        assert np.all(c[b, m:] == 0)
print("diis_solve == numpy lstsq")

if what == "small":
    def mk(i):
        zs, pos = M.H2O
        pos = np.array(pos) + np.random.default_rng(100 + i).normal(0, 0.05, (3, 3))
        return dqc_amd.Mol((zs, pos.tolist()), basis="cc-pvdz", grid="sg2")
    xc = "gga_x_pbe+gga_c_pbe"
elif what == "benzene":
    def mk(i):
        zs, pos = M.benzene()
        pos = np.array(pos) + np.random.default_rng(100 + i).normal(0, 0.03, (12, 3))
        return dqc_amd.Mol((zs, pos.tolist()), basis="cc-pvdz", grid="sg3")
    xc = "lda_x+lda_c_pw"
else:
    def mk(i):
        return dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    xc = "gga_x_pbe+gga_c_pbe"
mols = [mk(i) for i in range(nmol)]
fresh = lambda: [dqc_amd.KS(m, xc=xc) for m in mols]
qa = fresh()
torch.cuda.synchronize(); t0 = time.perf_counter()
run_concurrent(qa, max_inflight=8)
ea = [float(q.energy()) for q in qa]
torch.cuda.synchronize(); ta = time.perf_counter() - t0
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(16, 2), (8, 2), (nmol, 1), (max(2, nmol // 4), 4)]
for gs, infl in cfgs:
    qb = fresh()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_lockstep(qb, group_size=gs, inflight=infl, nstreams=int(os.environ.get('NSTREAMS', '3')))
    torch.cuda.synchronize(); tl = time.perf_counter() - t0
    eb = [float(q.energy()) for q in qb]
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
    de = max(abs(a - b) for a, b in zip(ea, eb))
    print("%s x%d group %d inflight %d: concurrent %.3f s (%d it), lockstep %.3f s scf (+%.3f s energies) (%d it, fallbacks %d); max |dE| = %.2e; conv %d/%d"
          % (what, nmol, gs, infl, ta, sum(q.niter for q in qa), tl, tb - tl, sum(q.niter for q in qb),
             sum(getattr(q, "eigh_fallbacks", 0) for q in qb), de, sum(q.converged for q in qb), nmol))
    print("   molecule-iterations/s: concurrent %.0f  lockstep %.0f" % (sum(q.niter for q in qa) / ta, sum(q.niter for q in qb) / tl))
