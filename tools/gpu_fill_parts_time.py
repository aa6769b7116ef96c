"""sharded ERI fill: time of every slice of a store cut into N equal-byte slices (naphthalene / cc-pVTZ and a C5 molecule): the
slowest slice is what a rank of an N-GPU one-molecule run waits for"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz")):
    tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
    for n in (1, 2, 4, 8):
        ts = []
        for r in range(n):
            t0, t1 = lib.tile_slice(tab.nao, r, n)[:2]
            buf = lib.eri_tiles_part(tab, dev, t0, t1); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); buf = lib.eri_tiles_part(tab, dev, t0, t1); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)); del buf
        print("%s N=%d: slices %s ms -> max %.1f, sum %.1f" % (name, n, " ".join("%.1f" % t for t in ts), max(ts), sum(ts)), flush=True)
