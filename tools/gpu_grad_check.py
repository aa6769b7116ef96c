"""GPU check of the analytic nuclear gradient: vs central finite differences of the GPU SCF energy, translational
invariance, and (small cases) the oracle's finite-difference gradient"""
import sys, os, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dqc_amd
print = functools.partial(print, flush=True)
dev = torch.device("cuda:0")


def energy(mol, basis, xc, grid):
    spin = None
    if isinstance(basis, tuple):
        basis, spin = basis
    m = dqc_amd.Mol(mol, basis=basis, grid=grid, device=dev, spin=spin)
    if DF:
        m.densityfit(auxbasis="etb")
    qc = (dqc_amd.KS(m, xc=xc) if xc else dqc_amd.HF(m)).run(fwd_options={"f_tol": 1e-11, "maxiter": 200})
    return qc


cases = [("h2-321g-rhf", ([1, 1], [[0, 0, -0.7], [0, 0, 0.7]]), "3-21G", None, "sg2"),
         ("h2o-sto3g-rhf", ([8, 1, 1], [[0, 0, 0.2217], [0, 1.4309, -0.8867], [0.1, -1.4309, -0.8867]]), "sto-3g", None, "sg2"),
         ("h2o-ccpvdz-rhf", ([8, 1, 1], [[0, 0, 0.2217], [0, 1.4309, -0.8867], [0.1, -1.4309, -0.8867]]), "cc-pvdz", None, "sg2"),
         ("h2-321g-lda", ([1, 1], [[0, 0, -0.7], [0, 0, 0.7]]), "3-21G", "lda_x", 4),
         ("lih-321g-pbe", ([3, 1], [[0, 0.1, -1.5], [0, 0, 1.5]]), "3-21G", "gga_x_pbe+gga_c_pbe", 4),
         ("h2o-ccpvdz-pbe", ([8, 1, 1], [[0, 0, 0.2217], [0, 1.4309, -0.8867], [0.1, -1.4309, -0.8867]]), "cc-pvdz", "gga_x_pbe+gga_c_pbe", "sg2"),
         ("lih-321g-scan", ([3, 1], [[0, 0.1, -1.5], [0, 0, 1.5]]), "3-21G", "mgga_x_scan", 4),
         ("h2o-ccpvdz-scan", ([8, 1, 1], [[0, 0, 0.2217], [0, 1.4309, -0.8867], [0.1, -1.4309, -0.8867]]), "cc-pvdz", "mgga_x_scan", "sg2"),
         ("h2o-ccpvdz-lda", ([8, 1, 1], [[0, 0, 0.2217], [0, 1.4309, -0.8867], [0.1, -1.4309, -0.8867]]), "cc-pvdz", "lda_x+lda_c_pw", "sg2")]
if "--pol" in sys.argv:  # unrestricted: (basis, spin)
    CH3 = ([6, 1, 1, 1], [[0, 0, 0.05], [2.039, 0, 0], [-1.0195, 1.7658, 0], [-1.0195, -1.7658, 0.1]])
    cases = [("ch3-321g-uhf", CH3, ("3-21G", 1), None, 4),
             ("ch3-321g-ulda", CH3, ("3-21G", 1), "lda_x+lda_c_pw", 4),
             ("ch3-ccpvdz-upbe", CH3, ("cc-pvdz", 1), "gga_x_pbe+gga_c_pbe", "sg2"),
             ("o2-321g-upbe", ([8, 8], [[0, 0, -1.14], [0.05, 0, 1.14]]), ("3-21G", 2), "gga_x_pbe+gga_c_pbe", 4)]
h = 1e-3
DF = "--df" in sys.argv
if DF:
    cases = [c for c in cases if c[3]]
import dqc_amd.grid as _g
for name, mol, basis, xc, grid in cases + [(c[0] + "-nocut",) + c[1:] for c in cases if c[3]]:
    _g._BECKE_CUT = 2.0 if name.endswith("-nocut") else 0.74
    qc = energy(mol, basis, xc, grid)
    g = qc.nuclear_gradient().cpu().numpy()
    pos = np.array(mol[1], dtype=float)
    gfd = np.zeros_like(pos)
    for a in range(len(mol[0])):
        for d in range(3):
            e = []
            for sgn in (1, -1):
                p = pos.copy(); p[a, d] += sgn * h
                e.append(float(energy((mol[0], p.tolist()), basis, xc, grid).energy()))
            gfd[a, d] = (e[0] - e[1]) / (2 * h)
    print(name, "max |analytic - FD| = %.2e   |sum_A g_A| = %.2e   |g| max %.4f" % (np.abs(g - gfd).max(), np.abs(g.sum(0)).max(), np.abs(g).max()))
    if np.abs(g - gfd).max() > 1e-6:
        print(" analytic\n", g, "\n FD\n", gfd)
print("GRAD CHECK DONE")
