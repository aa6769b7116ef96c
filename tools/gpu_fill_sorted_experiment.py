"""experiment: ERI fill of the same molecule with its shells SORTED by (l, contraction depth, atom) -- AO blocks of the tile store
then hold functions of one class, so a 64-byte line is written by one class launch -- against the natural shell order.
Run under rocprofv3 (tools/eri_kernel_sum.py) or timed here with events (whole fill, memset included)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz")):
    tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
    bas = tab.bas
    order = sorted(range(len(bas)), key=lambda i: (int(bas[i][1]), -int(bas[i][2]), int(bas[i][0]), i))
    tab2 = lib.Tables(tab.atm, bas[order], tab.env)
    for label, t in (("natural", tab), ("sorted", tab2)):
        for _ in range(2): lib.eri_tiles(t, dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): x = lib.eri_tiles(t, dev)
        e1.record(); torch.cuda.synchronize()
        print("%s %s shell order: fill %.2f ms (events, memset and host included)" % (name, label, e0.elapsed_time(e1) / 3), flush=True)
        del x
