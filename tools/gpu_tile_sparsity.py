"""how sparse is the stored ERI tile set?  fraction of 8^4 tiles whose largest |integral| is below a threshold
(the reference contracts every quartet, molintor.py:676 prescreen NULL; a skipped tile changes J by less than the threshold)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
for name, mol, basis in (("C5 vitamin C / cc-pVDZ", M.c5_molecule(0), "cc-pvdz"), ("naphthalene / cc-pVTZ", M.naphthalene(), "cc-pvtz")):
    h = dqc_amd.Mol(mol, basis=basis).get_hamiltonian().build()
    t = h._tiles.reshape(-1, 4096).abs().amax(dim=1)
    print(name, "tiles", t.numel(), " ".join("<%g: %.3f" % (th, float((t < th).double().mean())) for th in (1e-8, 1e-10, 1e-12, 1e-14, 1e-16)))
    del h, t
    torch.cuda.empty_cache()
