"""Round 6: throughput of a batch's Fock builds against (a) the cap on the CUs the one-block-per-CU Vxc kernel occupies
(dqc_set_vxc_cus: the rest of the chip is left to the Coulomb / density kernels other streams have queued), (b) the number of ordinary
streams the builds are dealt to, (c) the CU-partition form (dqc_amd.batch.CuPartition).  NMOL C5 molecules (default 8), the bench's own
step.  Writes gpurun_out/vxc_cus_sweep.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dqc_amd
from dqc_amd import lib
from dqc_amd.batch import CuPartition
from tests import molecules as M

dev = torch.device("cuda")
NMOL = int(os.environ.get("NMOL", "8"))
out = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.append(s)


engines, orbs = [], []
for i in range(NMOL):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    eng = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine
    n = eng.shape[-1]
    dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device=dev)))
    orbs.append(eng.scp2orb(eng.dm2scp(dm)).contiguous())
    engines.append(eng)
torch.cuda.synchronize()
say("%d C5 molecules, ngrid (live) %d of %d" % (NMOL, engines[0].hamilton.rgrid.shape[0], engines[0].hamilton.ngrid_full))


def rate(streams, passes=6):
    def step():
        for k, (eng, orb) in enumerate(zip(engines, orbs)):
            with torch.cuda.stream(streams[k % len(streams)]):
                eng.dm2scp(eng.hamilton.ao_orb2dm(orb, eng.orb_weight))
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        step()
    torch.cuda.synchronize()
    return NMOL * passes / (time.perf_counter() - t0)


say("ordinary streams; rows: Vxc CU cap, columns: streams")
ns_list = [1, 2, 3, 4, 8]
say("%6s " % "cap" + " ".join("%8d" % n for n in ns_list))
for cap in [0, 248, 240, 232, 224, 216, 208, 192, 176, 160]:
    lib.set_vxc_cus(cap)
    row = []
    for ns in ns_list:
        streams = [torch.cuda.Stream() for _ in range(ns)]
        row.append(rate(streams))
    say("%6d " % (cap or 256) + " ".join("%8.1f" % r for r in row))
lib.set_vxc_cus(0)
say("CU partition (grid pass on 32 - k CUs per XCD, Coulomb stream on k); rows: k, columns: grid streams")
gs_list = [1, 2, 3]
say("%6s " % "k" + " ".join("%8d" % n for n in gs_list))
for k in (4, 8, 12):
    row = []
    for gs in gs_list:
        p = CuPartition(dev, k, gs)
        row.append(rate(p.grid_streams))
        p.close()
    say("%6d " % k + " ".join("%8.1f" % r for r in row))

os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/vxc_cus_sweep.txt", "w") as f:
    f.write("\n".join(out) + "\n")
