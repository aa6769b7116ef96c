"""tools/gpu_gshell_check.py -- g shells through the runtime-class integral kernel (eri_generic.hpp) against the oracle:
full ERI tensor, direct J / K, int3c2e / int2c2e with g auxiliary shells, and the nuclear gradient against finite
differences.  Run on the GPU box:  python tools/gpu_gshell_check.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import basis as ob, natives as nat  # noqa: E402
from dqc_amd import lib  # noqa: E402

dev = torch.device("cuda")
lib.load()

BAS_A = [(0, [1.3, 0.4], [0.6, 0.5]), (1, [0.9], [1.0]), (2, [1.1], [1.0]), (3, [0.8], [1.0]), (4, [1.2, 0.5], [0.7, 0.4])]
BAS_B = [(0, [0.7], [1.0]), (1, [1.4, 0.5], [0.3, 0.8]), (2, [0.6], [1.0]), (4, [0.9], [1.0])]
MOL = ([8, 1], [[0.0, 0.1, -0.2], [0.3, -0.2, 1.6]])


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


t = ob.make_tables(MOL, [BAS_A, BAS_B])
tab = lib.Tables(t.atm, t.bas, t.env)
print("nao", tab.nao)
t0 = time.time()
ref = nat.int2e(t)
print("oracle int2e %.1f s" % (time.time() - t0))
tiles = lib.eri_tiles(tab, dev)
torch.cuda.synchronize()
t0 = time.time()
tiles = lib.eri_tiles(tab, dev)
torch.cuda.synchronize()
print("gpu fill %.3f s" % (time.time() - t0))
dense = lib.eri_dense(tiles, tab.nao).cpu().numpy()
print("eri max abs err", np.abs(dense - ref).max(), "max", np.abs(ref).max())
# per shell-quartet class error
loc = t.ao_loc
ls = t.bas[:, 1]
worst = {}
for i in range(t.nbas):
    for j in range(t.nbas):
        for k in range(t.nbas):
            for l in range(t.nbas):
                e = np.abs(dense[loc[i]:loc[i + 1], loc[j]:loc[j + 1], loc[k]:loc[k + 1], loc[l]:loc[l + 1]] -
                           ref[loc[i]:loc[i + 1], loc[j]:loc[j + 1], loc[k]:loc[k + 1], loc[l]:loc[l + 1]]).max()
                key = tuple(sorted([tuple(sorted([ls[i], ls[j]], reverse=True)), tuple(sorted([ls[k], ls[l]], reverse=True))], reverse=True))
                worst[key] = max(worst.get(key, 0.0), e)
bad = {k: v for k, v in worst.items() if v > 1e-11}
print("classes", len(worst), "bad", bad)

# direct J / K
D = np.random.default_rng(1).standard_normal((tab.nao, tab.nao))
Ds = 0.5 * (D + D.T)
J, K = lib.jk_direct(tab, torch.as_tensor(D, device=dev), True)
Jr = np.einsum("ij,ijkl->kl", Ds, ref)
Kr = np.einsum("il,ijkl->jk", Ds, ref)
print("direct J", rel(J.cpu().numpy(), Jr), "K", rel(K.cpu().numpy(), Kr))

# density-fitting integrals, g auxiliary shells
aux = [[(0, [0.5], [1.0]), (1, [0.7], [1.0]), (2, [0.8], [1.0]), (3, [0.9], [1.0]), (4, [1.0], [1.0]), (4, [0.4], [1.0])],
       [(0, [0.6], [1.0]), (4, [0.7], [1.0])]]
tc, orb, ax = ob.make_tables_df(MOL, [BAS_A, BAS_B], aux)
tabc = lib.Tables(tc.atm, tc.bas, tc.env)
j2 = lib.int2c2e(tabc, ax, dev).cpu().numpy()
j3 = lib.int3c2e(tabc, orb, ax, dev).cpu().numpy()
r2, r3 = nat.int2c2e(tc, ax), nat.int3c2e(tc, orb, ax)
print("int2c2e", rel(j2, r2), "int3c2e", rel(j3, r3))
