"""steady-state Fock builds of one BASELINE config (for rocprofv3 passes): setup, 2 warm-up builds, K timed builds.
usage: python tools/config_step.py C2|C3|C3pbe|C4|C5 [K]      prints one JSON line (shape + ms per build by HIP events)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
CFG = {"C2": (M.benzene(), "cc-pvdz", None), "C3": (M.benzene(), "cc-pvdz", "lda_x+lda_c_pw"),
       "C3pbe": (M.benzene(), "cc-pvdz", "gga_x_pbe+gga_c_pbe"), "C4": (M.naphthalene(), "cc-pvtz", "gga_x_pbe+gga_c_pbe"),
       "C5": (M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe"), "n264": (None, None, None)}
name = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mol_in, basis, xc = CFG[name]
mol = dqc_amd.Mol(mol_in, basis=basis, grid="sg3")
qc = dqc_amd.HF(mol) if xc is None else dqc_amd.KS(mol, xc=xc)
eng, h = qc._engine, qc._engine.hamilton
n = eng.shape[-1]
dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda")))
orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()
for _ in range(2):
    eng.dm2scp(h.ao_orb2dm(orb, eng.orb_weight))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    eng.dm2scp(h.ao_orb2dm(orb, eng.orb_weight))
e1.record()
torch.cuda.synchronize()
# the same build as the SCF loop issues it: one hipGraph replay (dqc_amd/graph.py) -- what an iteration pays; the eager figure above
# carries ~15 small launches of Python / torch host time per build, which dominates for benzene-size molecules
from dqc_amd.graph import GraphedFock
gf = GraphedFock(eng)
for _ in range(3):
    gf(orb)
torch.cuda.synchronize()
g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g0.record()
for _ in range(K):
    gf(orb)
g1.record()
torch.cuda.synchronize()
graph_ms = g0.elapsed_time(g1) / K
print(json.dumps({"config": name, "fock_build_graph_ms": graph_ms, "nao": h._nao_ao, "ld": h._ld, "nocc": int(eng.norb), "ngrid": int(h.rgrid.shape[0]) if h.is_grid_set else 0,
                  "xc": xc, "tile_bytes": h._tiles.numel() * 8, "fock_build_ms": e0.elapsed_time(e1) / K}))
