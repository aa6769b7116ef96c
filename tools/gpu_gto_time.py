"""eval_gto per derivative level on the C5 molecule and naphthalene / cc-pVTZ: kernel time by HIP events around repeated calls of one
prepared launch (the basis upload of every call is outside the bracket only approximately: see rocprofv3 for kernel durations)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
print("library:", os.environ.get("DQC_AMD_LIB", "product"))
for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz")):
    mol = dqc_amd.Mol(geo, basis=basis, grid="sg3")
    h = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine.hamilton
    for deriv in (0, 1, 3):
        x = lib.eval_gto(h._tab, h.rgrid, deriv); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = lib.eval_gto(h._tab, h.rgrid, deriv)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%s deriv %d: %.3f ms, %.2f TB/s of writes" % (name, deriv, ms, x.numel() * 8 / ms / 1e9), flush=True)
        del x, y
