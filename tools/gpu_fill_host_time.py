"""host time of one dqc_eri_fill_tiles call (it only enqueues) against its device time: C5 molecule"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
tab = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz").get_hamiltonian()._tab
tiles = torch.empty(lib.eri_store_doubles(tab.nao), dtype=torch.float64, device=dev)
def fill():
    with lib._on(dev) as st_:
        lib._check(lib.load().dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
for env in ({}, {"DQC_ERI_WMAP": "0"}):
    fill(); torch.cuda.synchronize()
    hs, ds = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fill(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(t1 - t0); ds.append(t2 - t0)
    print("host %.2f ms, host + device %.2f ms" % (1e3 * min(hs), 1e3 * min(ds)), flush=True)
    break
