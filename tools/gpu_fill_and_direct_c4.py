"""one tile-store fill and one unscreened direct Coulomb pass of naphthalene / cc-pVTZ (C4): per-class kernel times of the two modes
side by side under rocprofv3 (tools/direct_vs_fill_classes.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
tab = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz").get_hamiltonian()._tab
D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 34, np.eye(tab.nao), 3), device=dev)
ctx = lib.DirectContext(tab, dev)
for _ in range(2):
    ctx.jk(D, False, float(os.environ.get("DQC_TAU", "0")))
torch.cuda.synchronize()
L = lib.load()
tiles = torch.empty(lib.eri_store_doubles(tab.nao), dtype=torch.float64, device="cuda")
for _ in range(2):
    with lib._on(tiles.device) as st_:
        lib._check(L.dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
torch.cuda.synchronize()
