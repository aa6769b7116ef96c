"""stored-tile J + K (jk_stream_kernel) and J (j_stream_kernel) per call: benzene / cc-pVDZ, a C5 molecule, naphthalene / cc-pVTZ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
print("DQC_JK_STREAM_OCC =", os.environ.get("DQC_JK_STREAM_OCC", "(default 3)"))
for name, geo, bas in (("benzene", M.benzene(), "cc-pvdz"), ("C5", M.c5_molecule(0), "cc-pvdz"), ("naphthalene", M.naphthalene(), "cc-pvtz")):
    tab = dqc_amd.Mol(geo, basis=bas).get_hamiltonian()._tab
    D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 20, np.eye(tab.nao), 3), device=dev)
    tiles = lib.eri_tiles(tab, dev); work = lib.jk_workspace(tab.nao, dev)
    for wk in (False, True):
        for _ in range(5): lib.jk(tiles, D, work, wk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): lib.jk(tiles, D, work, wk)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print("%s nao %d with_k=%s: %.4f ms  (%.2f TB/s on nao^4 bytes)" % (name, tab.nao, wk, ms, tab.nao ** 4 / ms / 1e9), flush=True)
