"""one C4 molecule (naphthalene / cc-pVTZ, RKS PBE, sg3) spread over N GPUs with the tile store sliced (shard_over(eri='tiles')):
what ONE rank does per Fock build, timed on this one GPU for N = 1, 2, 4, 8 -- its slice of the J stream and its slab of the
grid passes (the slowest rank's slice is reported).  The all_reduce of the partial (J + Vxc) matrix -- 1.4 MB -- and of E_xc is
NOT in these numbers (one RCCL all_reduce of that size: tens of microseconds on xGMI); a projection, not a multi-GPU measurement"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
def ev(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
mol = dqc_amd.Mol(M.naphthalene(), basis="cc-pvtz", grid="sg3")
qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
eng, h = qc._engine, mol.get_hamiltonian()
n, ld = h._nao_ao, h._ld
z = torch.zeros((eng.shape[-1],) * 2, dtype=torch.float64, device=dev)
dm = eng.scp2dm(eng.dm2scp(z))
t_full = ev(lambda: eng.dm2scp(dm))
dao = h._unconvert_dm(dm).contiguous()
fac = h._factor_of(dm)
tiles, work = h._tiles, h._jkwork
G = h.rgrid.shape[0]
print("C4 nao %d, %d grid points, tile store %.1f GB: whole Fock build on one GPU %.3f ms" % (n, G, tiles.numel() * 8 / 1e9, t_full))
for N in (1, 2, 4, 8):
    tj = max(ev(lambda: lib.jk_part(tiles[lib.load().dqc_eri_tile_offset(n, t0):], dao, work, False, t0, t1)) for t0, t1, _ in
             [lib.tile_slice(n, r, N) for r in range(N)])
    g0, g1 = 0, G // N
    ao = h._ao[:, g0:g1].contiguous()
    w = h.dvolume[g0:g1].contiguous()
    def grid():
        rho, grho = lib.grid_density_lr(ao, n, fac[0], True)
        exc, v, vg = lib.xc_eval_quad(h.xc.terms, rho, grho, w)
        return lib.grid_vxc(ao, n, w, v, vg)
    tg = ev(grid)
    print("N = %d: slowest J slice %.3f ms + grid slab (%d points) %.3f ms = %.3f ms per Fock build per rank (+ one all_reduce of %.1f MB); store per rank %.1f GB" % (
        N, tj, g1 - g0, tg, tj + tg, ld * ld * 8 / 1e6, max(s[2] for s in [lib.tile_slice(n, r, N) for r in range(N)]) * 8 / 1e9), flush=True)
