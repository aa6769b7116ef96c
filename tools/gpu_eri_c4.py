"""three ERI fills of naphthalene / cc-pVTZ into ONE tile buffer (for rocprofv3 kernel traces: kernel time only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
h = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz").get_hamiltonian()
tab = h._tab
L = lib.load()
tiles = torch.empty(lib.eri_store_doubles(tab.nao), dtype=torch.float64, device="cuda")
for _ in range(3):
    with lib._on(tiles.device) as st_:
        lib._check(L.dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
torch.cuda.synchronize()
