import sys
sys.path.insert(0, "/root/repo")
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
for name, mol, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("benzene", M.benzene(), "cc-pvdz"), ("CH4 tz", M.CH4, "cc-pvtz"), ("naphthalene tz", M.naphthalene(), "cc-pvtz")):
    h = dqc_amd.Mol(mol, basis=basis).get_hamiltonian()
    lib.eri_tiles(h._tab, dev); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): t = lib.eri_tiles(h._tab, dev)
    e1.record(); torch.cuda.synchronize()
    print(name, "eri fill %.2f ms" % (e0.elapsed_time(e1) / 3))
