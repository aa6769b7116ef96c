"""how much (16-point chunk, 16-AO tile) sparsity would an AO REORDERING buy?  The AO matrix of a C5 molecule, per-AO per-chunk
max |value, gradient|; tile sparsity for the native order (atom by atom) and for AOs sorted by spatial range (tight functions
of neighbouring atoms grouped into tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
mol = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3")
eng = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine
h = eng.hamilton
ao = h._ao  # (4, G, ld)
n = h._nao_ao
G = ao.shape[1]
nch = G // 16
m = ao[:, :nch * 16, :n].abs().amax(0).reshape(nch, 16, n).amax(1)  # (chunks, nao)
for th in (1e-10, 1e-12, 1e-14):
    live = m > th  # (chunks, nao)
    frac_ao = 1.0 - live.double().mean().item()
    def tile_sparsity(order):
        lv = live[:, order]
        pad = (-n) % 16
        if pad:
            lv = torch.cat([lv, torch.zeros((nch, pad), dtype=torch.bool, device=lv.device)], 1)
        t = lv.reshape(nch, -1, 16).any(2)
        return 1.0 - t.double().mean().item(), t
    nat = torch.arange(n, device=ao.device)
    s_nat, _ = tile_sparsity(nat)
    rng = live.double().sum(0)  # chunks where the AO is live
    s_rng, t_rng = tile_sparsity(torch.argsort(rng))
    # range class first, then atom order inside the class (keeps neighbours together)
    cls = torch.bucketize(rng / nch, torch.tensor([0.35, 0.6, 0.85], device=ao.device))
    order2 = torch.argsort(cls * 1000 + nat, stable=True)
    s_cls, t_cls = tile_sparsity(order2)
    # Vxc tile PAIRS (i, j): a pair is skippable when either tile is dead in the chunk
    def pair_sparsity(t):
        a = t.double()
        livep = torch.einsum("ci,cj->ij", a, a) / nch
        return 1.0 - livep.mean().item()
    print("threshold %.0e: dead (chunk, AO) pairs %.3f | dead (chunk, 16-AO tile): native order %.3f, sorted by range %.3f, range class + atom order %.3f | dead Vxc tile pairs: range-sorted %.3f" % (
        th, frac_ao, s_nat, s_rng, s_cls, pair_sparsity(t_rng)))
