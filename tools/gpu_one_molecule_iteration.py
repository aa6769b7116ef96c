"""where one SCF iteration of ONE 20-atom molecule goes (the reference's use: one Mol, one KS(...).run()): the whole run per
iteration, the hipGraph step (purification + Fock build) alone, its parts alone"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.graph import GraphedSCFStep, GraphedFock
from dqc_amd.purify import projector_from_fock
from tests import molecules as M
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
geo, basis, xc = {"c5": (M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe"), "c3": (M.benzene(), "cc-pvdz", "lda_x+lda_c_pw"),
                  "c2": (M.benzene(), "cc-pvdz", None)}[name]
def mk():
    mol = dqc_amd.Mol(geo, basis=basis, grid="sg3")
    return dqc_amd.KS(mol, xc=xc) if xc else dqc_amd.HF(mol)
mk().run()
qc = mk()
torch.cuda.synchronize(); t0 = time.perf_counter(); qc.run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("%s: run() %.1f ms, %d iterations -> %.3f ms per iteration (%.0f /s), eigh fallbacks %s" % (name, 1e3 * t, qc.niter, 1e3 * t / qc.niter, qc.niter / t, getattr(qc, "eigh_fallbacks", 0)))
e1 = float(qc.energy())
torch.cuda.synchronize(); t0 = time.perf_counter(); qc.run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("%s: run() AGAIN on the same object (graphs captured) %.1f ms, %d iterations -> %.3f ms per iteration (%.0f /s); energy %.10f (first run %.10f) driver %s" % (
    name, 1e3 * t, qc.niter, 1e3 * t / qc.niter, qc.niter / t, float(qc.energy()), e1, qc.driver_used))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable(); qc.run(); torch.cuda.synchronize(); pr.disable()
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(14); print("\n".join(l for l in sio.getvalue().splitlines() if l.strip())[:2600])
os.environ["DQC_AMD_SCF_DRIVER"] = "host"
qh = mk()
torch.cuda.synchronize(); t0 = time.perf_counter(); qh.run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("%s: host-driven loop (DQC_AMD_SCF_DRIVER=host) %.1f ms, %d iterations -> %.3f ms per iteration; energy %.10f" % (name, 1e3 * t, qh.niter, 1e3 * t / qh.niter, float(qh.energy())))
os.environ["DQC_AMD_SCF_DRIVER"] = "device"
eng = qc._engine
def ev(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
F = qc._fock.clone()
st = GraphedSCFStep(eng)
print("  graph step (purification + orth factor + Fock build)  %.3f ms" % ev(lambda: st(F)))
gf = GraphedFock(eng)
orb = eng.scp2orb(F).contiguous()
print("  graph Fock build alone                                 %.3f ms" % ev(lambda: gf(orb)))
print("  eager Fock build                                       %.3f ms" % ev(lambda: eng.dm2scp(eng.hamilton.ao_orb2dm(orb, eng.orb_weight))))
print("  projector_from_fock (64 TC2 launches + 2 McWeeny), eager %.3f ms" % ev(lambda: projector_from_fock(F, eng.norb)))
print("  torch.linalg.eigh                                      %.3f ms" % ev(lambda: torch.linalg.eigh(F), 5))
loop = getattr(qc, "_devloop", None)
if loop is not None and loop.graph is not None:
    print("  device-loop iteration graph (DIIS + purification + Fock build)  %.3f ms" % ev(lambda: loop.graph.replay()))
    from dqc_amd import lib
    print("  dqc_diis_solve_dev alone                               %.3f ms" % ev(lambda: lib.diis_solve_dev(loop.gram, loop.count, loop.coef)))
    def diis_ops():
        a = torch.bmm(loop.fock, loop.dm); err = a - a.transpose(-2, -1); evv = err.reshape(1, -1)
        row = (loop.eh * evv).sum(-1)
        return (loop.coef[0].unsqueeze(-1) * loop.fh).sum(0), row, err.abs().max()
    print("  DIIS tensor algebra (eager, incl. launch gaps)         %.3f ms" % ev(diis_ops))
