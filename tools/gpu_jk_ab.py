"""A/B of the stored J and J + K passes on the same box: the current library against a pre-packing build (DQC_AMD_LIB)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, dqc_amd
    from dqc_amd import lib
    from tests import molecules as M
    dev = torch.device("cuda")
    def ev(fn, k=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k
    L = lib.load()
    for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz"), ("benzene", M.benzene(), "cc-pvdz")):
        tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
        n = L.dqc_eri_store_doubles(tab.nao) if hasattr(L, "dqc_eri_store_doubles") else L.dqc_eri_tile_count(tab.nao) * 4096
        tiles = torch.empty(n, dtype=torch.float64, device=dev)
        with lib._on(dev) as st_:
            lib._check(L.dqc_eri_fill_tiles(lib._ptr(tiles), *tab.args(), st_), "fill")
        D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 60, np.eye(tab.nao), 3), device=dev)
        work = lib.jk_workspace(tab.nao, dev)
        res = []
        for _ in range(3):
            res.append((ev(lambda: lib.jk(tiles, D, work, False)), ev(lambda: lib.jk(tiles, D, work, True)),
                        ev(lambda: lib.jk_multi(tiles, D.unsqueeze(0), torch.stack([D * 0.6, D * 0.4])), 10)))
        print("%-22s %-8s store %.3f GB | J %.3f  J+K %.3f  UHF J+2K %.3f ms (best of 3)" % (
            os.path.basename(lib.libpath()), name, n * 8 / 1e9, min(r[0] for r in res), min(r[1] for r in res), min(r[2] for r in res)), flush=True)
        del tiles
else:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for so in ("libdqc_amd_prepack.so", "libdqc_amd.so", "libdqc_amd_prepack.so", "libdqc_amd.so"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DQC_AMD_LIB=os.path.join(here, "dqc_amd", so)))
