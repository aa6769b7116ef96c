import sys
sys.path.insert(0, "/root/repo")
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
h = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz").get_hamiltonian()
for _ in range(3): lib.eri_tiles(h._tab, torch.device("cuda"))
torch.cuda.synchronize()
