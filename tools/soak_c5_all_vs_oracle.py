"""every molecule of the 32-molecule C5 batch against the CPU restatement (the graded test covers molecules 0 and 17): Fock matrix of
the core-guess density (AO representation), rho / grad rho on the whole sg3 grid, total energy.  ~20 s of oracle per molecule.
usage: python tools/soak_c5_all_vs_oracle.py [first last]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from oracle import basis as ob, hamilton as oh
from tests import molecules as M
dev = torch.device("cuda")
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 32)
worst = [0.0, 0.0, 0.0, 0.0]
for imol in range(lo, hi):
    t0 = time.perf_counter()
    geo = M.c5_molecule(imol)
    eng = dqc_amd.KS(dqc_amd.Mol(geo, basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe")._engine
    n = eng.shape[-1]
    dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device=dev)))
    hg = eng.hamilton
    o = oh.Engine(ob.make_tables(geo, "cc-pvdz"), xc="gga_x_pbe+gga_c_pbe", grid="sg3", eri_mode="s4")
    S = hg._ovlp_ao.cpu()
    Dao = (hg._orthozer @ dm @ hg._orthozer.T).cpu()
    Xinv = o.h.X.T @ S
    dmo = Xinv @ Dao @ Xinv.T
    dmo = (dmo + dmo.T) * 0.5
    SXo, SXg = S @ o.h.X, hg._ovlp_ao @ hg._orthozer
    dF = float(((SXg @ eng.dm2scp(dm) @ SXg.T).cpu() - SXo @ o.dm2scp(dmo) @ SXo.T).abs().max())
    dE = abs(float(eng.dm2energy(dm)) - float(o.dm2energy(dmo)))
    di = hg._dm2densinfo(dm)
    rho_o, grho_o = o.h.dm2densinfo(dmo)
    dr = float((di.value.cpu() - rho_o).abs().max()) / float(rho_o.abs().max())
    dg = float((di.grad.cpu() - grho_o).abs().max()) / float(grho_o.abs().max())
    worst = [max(a, b) for a, b in zip(worst, (dF, dE, dr, dg))]
    print("molecule %2d: max|dF| %.1e  |dE| %.1e Ha  rel rho %.1e  rel grad rho %.1e   (%.0f s)" % (imol, dF, dE, dr, dg, time.perf_counter() - t0), flush=True)
    del eng, hg, o
    torch.cuda.empty_cache()
print("worst over molecules %d..%d: max|dF| %.1e  |dE| %.1e Ha  rel rho %.1e  rel grad rho %.1e" % (lo, hi - 1, *worst))
