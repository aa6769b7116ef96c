"""is the tile store the same with and without the side streams?  four worker processes (DQC_SIDE_STREAMS = 0, 0, 4, 4), one small
and one mid-size molecule: max |difference| of the stores"""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
for name, geo in (("h2o", M.CH4), ("benzene", M.benzene())):
    tab = dqc_amd.Mol(geo, basis="cc-pvtz").get_hamiltonian()._tab
    tiles = torch.empty(lib.eri_store_doubles(tab.nao), dtype=torch.float64, device=dev)
    for rep in range(2):
        tiles.fill_(float("nan"))
        with lib._on(dev) as st_:
            lib._check(lib.load().dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
        torch.cuda.synchronize()
    t = tiles.cpu().numpy()
    print(name, "nan", int(np.isnan(t).sum()), "sum %.17g abs %.17g" % (t.sum(), np.abs(t).sum()), flush=True)
    if name == "h2o":
        np.save(sys.argv[2], t)
"""
outs = []
with tempfile.TemporaryDirectory() as d:
    for i, n in enumerate(("0", "0", "4", "4")):
        f = os.path.join(d, "t%d.npy" % i)
        r = subprocess.run([sys.executable, "-c", W, ROOT, f], env=dict(os.environ, DQC_SIDE_STREAMS=n), capture_output=True, text=True, timeout=600)
        print("streams", n, "|", " | ".join(ln for ln in r.stdout.splitlines() if ln[:3] in ("h2o", "ben")), r.stderr[-300:] if r.returncode else "", flush=True)
        outs.append(np.load(f))
    for i in range(1, 4):
        dlt = np.abs(outs[i] - outs[0])
        print("h2o store %d vs 0: max diff %.3e, differing elements %d" % (i, dlt.max(), int((dlt > 0).sum())))
