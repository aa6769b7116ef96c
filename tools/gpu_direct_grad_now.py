"""quick numbers: C4 direct SCF per iteration (bench.py direct_scf leg) and the C5 RKS PBE gradient (warm)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
import bench
from tests import molecules as M
dev = torch.device("cuda:0")
XC = "gga_x_pbe+gga_c_pbe"
if "grad" in sys.argv or len(sys.argv) == 1:
    qg = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev), xc=XC).run()
    qg.nuclear_gradient(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g = qg.nuclear_gradient(); torch.cuda.synchronize()
    print("C5 gradient warm %.4f s  |sum g| %.1e" % (time.perf_counter() - t0, float(g.sum(0).abs().max())), flush=True)
    del qg
if "direct" in sys.argv or len(sys.argv) == 1:
    d = bench.direct_scf_leg(dev)
    for m in ("tiles", "direct"):
        print(m, "scf %.3f s / %d it = %.1f ms per iteration, setup %.2f s, E %.9f" % (d[m]["scf_s"], d[m]["iterations"], 1e3 * d[m]["scf_s"] / d[m]["iterations"], d[m]["setup_s"], d[m]["energy_ha"]))
    print("dE", d["energy_diff_ha"], {k: d["direct"][k] for k in d["direct"] if k not in ("scf_s", "iterations", "energy_ha")})
