"""deterministic mode (dqc_set_deterministic): bit-reproducibility of the Fock build and of whole SCF runs, distance to the fp64-atomic
result, and what it costs"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M

def builds(eng, dm, k=5):
    return [eng.dm2scp(dm.clone()) for _ in range(k)]

def timeit(eng, orb, k=30):
    h = eng.hamilton
    for _ in range(3):
        eng.dm2scp(h.ao_orb2dm(orb, eng.orb_weight))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        eng.dm2scp(h.ao_orb2dm(orb, eng.orb_weight))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k

for label, geo, basis, xc in (("C5 RKS PBE", M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe"), ("benzene RHF", M.benzene(), "cc-pvdz", None),
                              ("H2O RKS LDA", M.H2O, "cc-pvdz", "lda_x+lda_c_pw")):
    mol = dqc_amd.Mol(geo, basis=basis, grid="sg3")
    qc = dqc_amd.KS(mol, xc=xc) if xc else dqc_amd.HF(mol)
    eng = qc._engine
    n = eng.shape[-1]
    dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda")))
    orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()
    lib.set_deterministic(False)
    fa = builds(eng, dm)
    ta = timeit(eng, orb)
    lib.set_deterministic(True)
    fd = builds(eng, dm)
    td = timeit(eng, orb)
    print("%-12s atomics: identical builds %s (max diff %.1e), %.3f ms | deterministic: identical builds %s, %.3f ms (%+.1f %%) | max |F_det - F_atomic| %.1e" % (
        label, all(torch.equal(f, fa[0]) for f in fa[1:]), max(float((f - fa[0]).abs().max()) for f in fa[1:]), ta,
        all(torch.equal(f, fd[0]) for f in fd[1:]), td, 100 * (td / ta - 1), float((fd[0] - fa[0]).abs().max())))
    es, dms = [], []
    for _ in range(2):
        q = (dqc_amd.KS(mol, xc=xc) if xc else dqc_amd.HF(mol)).run()
        es.append(float(q.energy())); dms.append(q.aodm().clone())
    print("             two SCF runs in deterministic mode: energies %.12f %.12f, identical bits: energy %s, density %s, iterations %d" % (
        es[0], es[1], es[0] == es[1], torch.equal(dms[0], dms[1]), q.niter))
    lib.set_deterministic(False)
    q = (dqc_amd.KS(mol, xc=xc) if xc else dqc_amd.HF(mol)).run()
    print("             fp64-atomic SCF energy %.12f (difference %.1e)" % (float(q.energy()), float(q.energy()) - es[0]))
