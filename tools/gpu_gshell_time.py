"""tools/gpu_gshell_time.py -- cost of the runtime-class kernel: CH4 in a cc-pVQZ-shaped basis (cc-pVTZ plus one diffuse
shell per angular momentum and a g shell on C / an f shell on H: C 5s4p3d2f1g, H 4s3p2d1f, nao 175).  Fill time of the tile
store with (a) the g classes through the runtime kernel (the product path), (b) every class through it."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import basis as ob  # noqa: E402
from dqc_amd import lib  # noqa: E402
from tests import molecules as M  # noqa: E402



def shells(z):
    base = [(l, list(a), None, c) for (l, a, c) in ob.loadbasis(z, "cc-pvtz")]
    extra = {6: [(0, 0.06), (1, 0.05), (2, 0.2), (3, 0.5), (4, 1.0)], 1: [(0, 0.03), (1, 0.12), (2, 0.35), (3, 1.0)]}[z]
    return base, extra


def tables(with_g=True):
    zs, pos = M.CH4
    per_atom = []
    for z in zs:
        base, extra = shells(int(z))
        sh = [(l, np.asarray(a, float), np.asarray(c, float)) for (l, a, _, c) in base]
        for (l, a) in extra:
            if l == 4 and not with_g:
                continue
            sh.append((l, np.array([a]), ob.wfnormalize(l, [a], [1.0])))
        per_atom.append(sh)
    t = ob.Tables(zs, pos, per_atom)
    return lib.Tables(t.atm, t.bas, t.env)


def fill_ms(tab, n=5):
    lib.eri_tiles(tab, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tl = lib.eri_tiles(tab, dev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, tl


tg, tn = tables(True), tables(False)
if not torch.cuda.is_available():
    print('nao', tg.nao, tn.nao)
    sys.exit(0)
dev = torch.device("cuda")
lib.load()
print("nao with g", tg.nao, "without", tn.nao)
ms_n, _ = fill_ms(tn)
ms_g, tiles_g = fill_ms(tg)
print("fill without the g shell %.2f ms; with it %.2f ms" % (ms_n, ms_g))
lib.set_generic_eri(True)
ms_n2, _ = fill_ms(tn)
ms_g2, tiles_g2 = fill_ms(tg)
lib.set_generic_eri(False)
print("every class through the runtime kernel: %.2f ms / %.2f ms" % (ms_n2, ms_g2))
d = (lib.eri_dense(tiles_g, tg.nao) - lib.eri_dense(tiles_g2, tg.nao)).abs().max().item()
print("max |difference| of the two fills", d)
