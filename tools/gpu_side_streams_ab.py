"""A/B of the side streams of the ERI class launches (DQC_SIDE_STREAMS, read once per process): fill C5 / C4, direct Coulomb pass C4,
C5 gradient.  usage: DQC_SIDE_STREAMS=n python tools/gpu_side_streams_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
out = ["streams %s" % os.environ.get("DQC_SIDE_STREAMS", "default")]
for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz")):
    tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
    tiles = torch.empty(lib.eri_store_doubles(tab.nao), dtype=torch.float64, device=dev)
    def fill():
        with lib._on(dev) as st_:
            lib._check(lib.load().dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
    fill(); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fill(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out.append("fill %s %.2f ms" % (name, 1e3 * min(ts)))
    if name == "C4":
        D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 34, np.eye(tab.nao), 3), device=dev)
        ctx = lib.DirectContext(tab, dev)
        ctx.jk(D, False, 1e-13); torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.jk(D, False, 1e-13); ctx.jk(D, False, 1e-13); torch.cuda.synchronize()
        out.append("direct J C4 %.1f ms" % (1e3 * (time.perf_counter() - t0) / 2))
        del ctx
    del tiles
qg = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev), xc="gga_x_pbe+gga_c_pbe").run()
g0 = qg.nuclear_gradient(); torch.cuda.synchronize()
t0 = time.perf_counter(); g = qg.nuclear_gradient(); torch.cuda.synchronize()
out.append("gradient C5 %.4f s  (sum %.1e)" % (time.perf_counter() - t0, float(g.sum(0).abs().max())))
print(" | ".join(out), flush=True)
