"""Nuclear-gradient fixtures: central finite differences (h = 1e-3 Bohr) of the ORACLE's SCF energies -- by construction
what the reference's autograd returns (its own gradient tests are gradcheck, test_hf.py:82-111, test_ks.py:117-137).
KS cases are generated with the Becke sparsification cut (mu < 0.74) switched off so that E(R) is smooth (with the cut
the finite difference carries 1e-4 jumps that an analytic / autograd derivative does not have); the GPU tests switch it
off on their side as well.  Writes tests/golden/oracle_fd_gradients.json.     usage: python tools/make_grad_golden.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import hamilton as oh, grid as og

H2O = ([8, 1, 1], [[0, 0, 0.2217], [0, 1.4309, -0.8867], [0.1, -1.4309, -0.8867]])
LIH = ([3, 1], [[0, 0.1, -1.5], [0, 0, 1.5]])
CH3 = ([6, 1, 1, 1], [[0, 0, 0.05], [2.039, 0, 0], [-1.0195, 1.7658, 0], [-1.0195, -1.7658, 0.1]])
CASES = {
    # name: (mol, basis, xc, grid, spin, auxbasis)
    "h2-321g-rhf": (([1, 1], [[0, 0, -0.7], [0, 0, 0.7]]), "3-21G", None, 4, None, None),
    "h2o-sto3g-rhf": (H2O, "sto-3g", None, 4, None, None),
    "lih-321g-lda": (LIH, "3-21G", "lda_x", 4, None, None),
    "lih-321g-pbe": (LIH, "3-21G", "gga_x_pbe+gga_c_pbe", 4, None, None),
    "lih-321g-pbe-df": (LIH, "3-21G", "gga_x_pbe+gga_c_pbe", 4, None, "etb"),
    "ch3-321g-uhf": (CH3, "3-21G", None, 4, 1, None),
    "ch3-321g-upbe": (CH3, "3-21G", "gga_x_pbe+gga_c_pbe", 4, 1, None),
}

if __name__ == "__main__":
    og.BECKE_CUT = 2.0
    out = {"_how": __doc__, "h": 1e-3}
    for name, (mol, basis, xc, grid, spin, aux) in CASES.items():
        t0 = time.time()
        kw = {"maxiter": 300}
        if aux:
            kw["auxbasis"] = aux
        g = oh.nuclear_gradient_fd(mol, basis, xc=xc, grid=grid, h=1e-3, spin=spin, **kw)
        out[name] = {"atomzs": mol[0], "atompos": mol[1], "basis": basis, "xc": xc, "grid": grid, "spin": spin,
                     "auxbasis": aux, "becke_cut": "off" if xc else "n/a", "gradient": g.tolist()}
        print("%-18s %.1f s  max|g| %.5f  sum %.1e" % (name, time.time() - t0, np.abs(g).max(), np.abs(g.sum(0)).max()), flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_fd_gradients.json"), "w"), indent=1)
