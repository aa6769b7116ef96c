"""RHF J + K of one density: jk_tiles_kernel<true> (dqc_jk_from_tiles) against jk_multi_kernel<1> (dqc_jk_from_tiles_multi, nj = nk = 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
def ev(fn, k=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz"), ("benzene", M.benzene(), "cc-pvdz"), ("H2O", M.H2O, "cc-pvdz")):
    tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
    tiles = lib.eri_tiles(tab, dev)
    D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 60, np.eye(tab.nao), 3), device=dev)
    work = lib.jk_workspace(tab.nao, dev)
    D1 = D.unsqueeze(0).contiguous()
    t1 = min(ev(lambda: lib.jk(tiles, D, work, True)) for _ in range(3))
    t2 = min(ev(lambda: lib.jk_multi(tiles, D1, D1)) for _ in range(3))
    J1, K1 = lib.jk(tiles, D, work, True)
    J2, K2 = lib.jk_multi(tiles, D1, D1)
    print("%-8s J+K: jk_tiles<true> %.3f ms | jk_multi<1> %.3f ms | max rel diff J %.1e K %.1e" % (
        name, t1, t2, float((J1 - J2[0]).abs().max() / J1.abs().max()), float((K1 - K2[0]).abs().max() / K1.abs().max())))
    del tiles
