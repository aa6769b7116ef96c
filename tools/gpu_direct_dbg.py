"""direct J (and J + K) pass of naphthalene / cc-pVTZ without screening, under the DQC_ERI_DBG switches (1: no primitive loops,
2: no output phase): where the direct path's time goes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
tab = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz").get_hamiltonian()._tab
D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 34, np.eye(tab.nao), 3), device=dev)
ctx = lib.DirectContext(tab, dev)
TAU = float(os.environ.get("DQC_TAU", "0"))
for wk in (False, True):
    ctx.jk(D, wk, TAU); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.jk(D, wk, TAU); ctx.jk(D, wk, TAU); e1.record(); torch.cuda.synchronize()
    print("DBG=%s  with_k=%s  %.1f ms per pass" % (os.environ.get("DQC_ERI_DBG", "0"), wk, e0.elapsed_time(e1) / 2), flush=True)
