"""where does the SCF stall come from?  (a) noise of the Fock build: two builds of the same D; (b) the convergence trace with the
Pulay system solved unscaled (round 2) and with the Gram block normalised"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, dqc_amd
    from tests import molecules as M
    for i in (0, 5):
        mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
        qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
        eng = qc._engine
        gen = qc._run_gen({} if False else "1e", {"maxiter": 60})
        req = next(gen)
        errs = []
        try:
            while True:
                host = req.cpu().numpy()
                errs.append(float(host[0]))
                req = gen.send(host)
        except StopIteration:
            pass
        print("scale=%s molecule %d: %d iterations, conv %s stalled %s, E = %.10f" % (
            os.environ.get("DQC_AMD_DIIS_SCALE"), i, qc.niter, qc.converged, qc.stalled, float(qc.energy())))
        print("  " + " ".join("%.1e" % e for e in errs))
        if i == 0:
            dm = qc.aodm()
            fs = [eng.dm2scp(dm.clone()) for _ in range(6)]
            dF = max(float((f - fs[0]).abs().max()) for f in fs[1:])
            dC = max(float(((f - fs[0]) @ dm - dm @ (f - fs[0])).abs().max()) for f in fs[1:])
            h = eng.hamilton
            sv = torch.linalg.eigvalsh(h._ovlp_ao)
            print("  noise of the build: max|F1-F2| = %.2e  max|[F1-F2, D]| = %.2e   smin(S) = %.2e  |X|max = %.1f" % (
                dF, dC, float(sv[0]), float(h._orthozer.abs().max())))
else:
    for sc in ("0", "1"):
        env = dict(os.environ, DQC_AMD_DIIS_SCALE=sc)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
