import os, sys, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1:
    import numpy as np, torch, dqc_amd
    from dqc_amd import lib
    from tests import molecules as M
    dev = torch.device("cuda")
    out = []
    for name, geo, bas in (("benzene", M.benzene(), "cc-pvdz"), ("C5", M.c5_molecule(0), "cc-pvdz"), ("C4", M.naphthalene(), "cc-pvtz")):
        tab = dqc_amd.Mol(geo, basis=bas).get_hamiltonian()._tab
        D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 20, np.eye(tab.nao), 3), device=dev)
        tiles = lib.eri_tiles(tab, dev); work = lib.jk_workspace(tab.nao, dev)
        best = 1e9
        for rep in range(3):
            for _ in range(5): lib.jk(tiles, D, work, False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): lib.jk(tiles, D, work, False)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        out.append("%s %.4f ms" % (name, best))
        del tiles
    print("j nblk %-8s %s" % (os.environ.get("DQC_J_NBLK", "shipped"), " | ".join(out)), flush=True)
else:
    for v in ("3072", "16384", "32768", "65536", "3072", "16384", "32768"):
        env = dict(os.environ)
        if v: env["DQC_J_NBLK"] = v
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, stderr=subprocess.DEVNULL)
