"""ERI fill of the C4 molecule (naphthalene / cc-pVTZ, f shells on C): run under tools/eri_kernel_sum.sh-style rocprof"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
h = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz").get_hamiltonian()
t = lib.eri_tiles(h._tab, torch.device("cuda"))
del t
torch.cuda.empty_cache()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); t = lib.eri_tiles(h._tab, torch.device("cuda")); e1.record(); torch.cuda.synchronize()
print("C4 ERI fill: %.1f ms (events, incl. host table building)" % e0.elapsed_time(e1))
