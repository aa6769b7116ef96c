"""where does the RHF J + K pass go?  jk_tiles_kernel<true> with pieces compiled out (variant libraries libdqc_amd_abl_*.so built from
edited copies of jk.hip; wrong results, timing only)"""
import os, sys, subprocess, glob
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
    out = []
    for name, geo, basis in (("C5", M.c5_molecule(0), "cc-pvdz"), ("benzene", M.benzene(), "cc-pvdz")):
        tab = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()._tab
        tiles = lib.eri_tiles(tab, dev)
        D = torch.as_tensor(M.seeded_dm_ao(tab.nao, 60, np.eye(tab.nao), 3), device=dev)
        work = lib.jk_workspace(tab.nao, dev)
        out.append("%s J+K %.3f ms (J only %.3f)" % (name, min(ev(lambda: lib.jk(tiles, D, work, True)) for _ in range(3)),
                                                  min(ev(lambda: lib.jk(tiles, D, work, False)) for _ in range(3))))
        del tiles
    print("%-36s %s" % (os.path.basename(lib.libpath()), " | ".join(out)), flush=True)
else:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for so in sorted(glob.glob(os.path.join(here, "dqc_amd", "libdqc_amd_abl_*.so"))):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DQC_AMD_LIB=so))
