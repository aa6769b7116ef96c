"""setup of the 32-molecule C5 batch with the orthogonalisers from one batched eigh (batch.prepare_orthogonalisers): how often
should the host wait for the device between molecules?  (never: the host runs ahead and the pinned staging blocks of the
stream-ordered uploads are not recycled; every molecule: the next molecule's host work does not overlap this one's ERI fill)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.batch import molecule_bytes, reserve_device_memory, prepare_orthogonalisers
from tests import molecules as M
dev = torch.device("cuda")
n = 32
dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(31), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe"); torch.cuda.synchronize()
r = reserve_device_memory(n * molecule_bytes(208, 353400) + (4 << 30), dev)
print("reserve %.2f s" % r)
for every, batched in ((1, True), (1, False), (1, True), (0, True), (0, False), (1, False)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mols = [dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3") for i in range(n)]
    for m in mols: m.get_hamiltonian()
    if batched: prepare_orthogonalisers([m.get_hamiltonian() for m in mols])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    keep = []
    for k, m in enumerate(mols):
        keep.append(dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe"))
        if every and (k + 1) % every == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("batched eigh %s, device sync every %d molecules: tables + orthogonalisers %.3f s, engines %.3f s, total %.3f s" % (batched, every, t1 - t0, t2 - t1, t2 - t0), flush=True)
    del keep, mols
