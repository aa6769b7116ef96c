"""End-to-end GPU check: product SCF energies vs golden (reference-generated) fixtures + timing at C5 scale."""
import sys, time, glob, os
import numpy as np, torch
sys.path.insert(0, ".")
import dqc_amd
from dqc_amd import lib

dev = torch.device("cuda")
for f in sorted(glob.glob("tests/golden/ref_*.npz")):
    g = np.load(f)
    name = os.path.basename(f)[4:-4]
    parts = name.split("_")
    basis = {"sto3g": "sto-3g", "ccpvdz": "cc-pvdz", "ccpvtz": "cc-pvtz"}[parts[1]]
    xc = None if parts[2] == "rhf" else {"lda": "lda_x+lda_c_pw", "pbe": "gga_x_pbe+gga_c_pbe"}[parts[2]]
    grid = parts[3] if len(parts) > 3 else "sg3"
    t0 = time.time()
    mol = dqc_amd.Mol((g["atomzs"].tolist(), g["atompos"]), basis=basis, grid=grid)
    qc = dqc_amd.HF(mol) if xc is None else dqc_amd.KS(mol, xc=xc)
    t1 = time.time()
    qc.run()
    e = float(qc.energy())
    torch.cuda.synchronize(); t2 = time.time()
    print("%-28s E=%.10f ref=%.10f diff=%.2e iters=%d conv=%s setup %.2fs scf %.2fs" % (name, e, float(g["e_tot"]), e - float(g["e_tot"]), qc.niter, qc.converged, t1 - t0, t2 - t1), flush=True)
print("DONE")
