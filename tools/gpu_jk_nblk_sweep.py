"""block-count sweep of the two stored-tile stream kernels on the small shapes (benzene nao 114, a C5 molecule nao 208):
DQC_J_NBLK (j_stream_kernel, shipped 3072) and DQC_JK_NBLK (jk_stream_kernel, shipped max(1024, min(6144, tiles / 8)))"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, dqc_amd
    from dqc_amd import lib
    from tests import molecules as M
    dev = torch.device("cuda")
    wk = sys.argv[2] == "jk"
    out = []
    for name, geo in (("benzene", M.benzene()), ("C5", M.c5_molecule(0))):
        tab = dqc_amd.Mol(geo, basis="cc-pvdz").get_hamiltonian()._tab
        D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 20, np.eye(tab.nao), 3), device=dev)
        tiles = lib.eri_tiles(tab, dev); work = lib.jk_workspace(tab.nao, dev)
        best = 1e9
        for rep in range(3):
            for _ in range(5): lib.jk(tiles, D, work, wk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): lib.jk(tiles, D, work, wk)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 100)
        out.append("%s %.4f ms" % (name, best))
    print("%-3s nblk %-8s %s" % (sys.argv[2], os.environ.get("DQC_JK_NBLK" if wk else "DQC_J_NBLK", "shipped"), " | ".join(out)), flush=True)
else:
    for kind, var, vals in (("j", "DQC_J_NBLK", ("", "512", "1024", "1536", "2048", "4096", "6144")), ("jk", "DQC_JK_NBLK", ("", "512", "768", "1024", "2048", "3072", "4096"))):
        for v in vals:
            env = dict(os.environ)
            if v: env[var] = v
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", kind], env=env, stderr=subprocess.DEVNULL)
