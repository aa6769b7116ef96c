"""direct SCF (dqc_jk_direct): J and J + K per call against the stored-tile kernels and the one-off fill, C5 and C4 shapes; then a
system whose tile store does not fit one GPU (naphthalene dimer, cc-pVTZ: nao 824, 0.46 TB of tiles) through a few SCF steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
def ev(fn, k=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz")):
    h = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()
    tab = h._tab
    D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 60, np.eye(tab.nao), 3), device=dev)
    tiles = torch.empty(lib.eri_store_doubles(tab.nao), dtype=torch.float64, device=dev)
    def fill():
        with lib._on(dev) as st_:
            lib._check(lib.load().dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
    tf = ev(fill)
    work = lib.jk_workspace(tab.nao, dev)
    tj, tjk = ev(lambda: lib.jk(tiles, D, work, False), 10), ev(lambda: lib.jk(tiles, D, work, True), 10)
    dj, djk = ev(lambda: lib.jk_direct(tab, D, False)), ev(lambda: lib.jk_direct(tab, D, True))
    Jt, Kt = lib.jk(tiles, D, work, True)
    Jd, Kd = lib.jk_direct(tab, D, True)
    print("%s nao %d: fill %.2f ms | stored J %.3f, J+K %.3f ms | direct J %.2f, J+K %.2f ms | max rel diff J %.1e K %.1e" % (
        name, tab.nao, tf, tj, tjk, dj, djk, float((Jd - Jt).abs().max() / Jt.abs().max()), float((Kd - Kt).abs().max() / Kt.abs().max())))
    del tiles
if len(sys.argv) > 1 and sys.argv[1] == "big":
    zs, pos = M.naphthalene()
    pos = np.array(pos)
    zs2, pos2 = list(zs) + list(zs), np.concatenate([pos, pos + np.array([0.0, 0.0, 6.6])]).tolist()
    t0 = time.perf_counter()
    mol = dqc_amd.Mol((zs2, pos2), basis="cc-pvtz", grid="sg2")
    qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
    h = mol.get_hamiltonian()
    torch.cuda.synchronize()
    print("naphthalene dimer / cc-pVTZ: nao %d, tile store would be %.0f GB -> direct %s; setup %.1f s" % (
        h._nao_ao, lib.eri_store_doubles(h._nao_ao) * 8 / 1e9, h._direct, time.perf_counter() - t0))
    t0 = time.perf_counter()
    qc.run(fwd_options={"maxiter": int(sys.argv[2]) if len(sys.argv) > 2 else 6})
    torch.cuda.synchronize()
    print("  %d SCF iterations in %.1f s (%.2f s each), max|[F,D]| %.2e, E = %.8f" % (qc.niter, time.perf_counter() - t0, (time.perf_counter() - t0) / qc.niter, qc.scf_error, float(qc.energy())))
