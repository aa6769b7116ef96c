"""UHF trio J[D_u + D_d], K[D_u], K[D_d] in one tile pass: stream form against the grid-stride kernel (DQC_JK_MULTI_IMPL=grid)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, dqc_amd
    from dqc_amd import lib
    from tests import molecules as M
    dev = torch.device("cuda")
    def ev(fn, k=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k
    for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz"), ("benzene", M.benzene(), "cc-pvdz")):
        tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
        tiles = lib.eri_tiles(tab, dev)
        D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 60, np.eye(tab.nao), 3), device=dev)
        dj, dk = D.unsqueeze(0).contiguous(), torch.stack([D * 0.6, D * 0.4])
        t3 = min(ev(lambda: lib.jk_multi(tiles, dj, dk)) for _ in range(3))
        t2 = min(ev(lambda: lib.jk_multi(tiles, None, dk)) for _ in range(3))
        print("%-7s impl %-6s J+2K %.3f ms | 2K %.3f ms" % (name, os.environ.get("DQC_JK_MULTI_IMPL", "stream"), t3, t2), flush=True)
        del tiles
else:
    for impl in ("stream", "grid", "stream", "grid"):
        env = dict(os.environ)
        env["DQC_JK_MULTI_IMPL"] = impl
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
