import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, dqc_amd
from tests import molecules as M
qcs = [dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe") for i in range(4)]
for q in qcs: q.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for q in qcs:
    q.eigh_fallbacks = 0
    q.run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("TC2 iters", os.environ.get("DQC_AMD_TC2_ITERS"), "4 molecules %.3f s" % dt, "iterations", [q.niter for q in qcs], "eigh fallbacks", [q.eigh_fallbacks for q in qcs])
