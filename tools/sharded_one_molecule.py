"""one molecule over N GPUs (HamiltonMI355.shard_over): SCF of the naphthalene dimer / cc-pVTZ (nao 824: no tile store fits one GPU)
launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_one_molecule.py [maxiter]
(on a 1-GPU box: add `--all-on-gpu0` to run the N ranks on device 0 over gloo -- correctness only; `--tiles`: one C4 molecule with
its stored tile store spread over the ranks instead of the direct path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import dqc_amd
from tests import molecules as M
one = "--all-on-gpu0" in sys.argv
rank, world, lrank = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
dev = torch.device("cuda", 0 if one else lrank)
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group("gloo" if one else "nccl")
zs, pos = M.naphthalene()
pos = np.array(pos)
geo = (list(zs) + list(zs), np.concatenate([pos, pos + np.array([0.0, 0.0, 6.6])]).tolist())
small = "--small" in sys.argv
tiles = "--tiles" in sys.argv  # the stored path with the tile store spread over the ranks (C4: naphthalene / cc-pVTZ, 30 GB of tiles)
if tiles:
    mol = dqc_amd.Mol(M.naphthalene(), basis="cc-pvdz" if small else "cc-pvtz", grid="sg3", device=dev)
else:
    mol = dqc_amd.Mol(M.naphthalene() if small else geo, basis="cc-pvdz" if small else "cc-pvtz", grid="sg2", device=dev)
h = mol.get_hamiltonian()
if world > 1:
    h.shard_over(eri="tiles" if tiles else "direct")
elif not tiles:
    h.use_direct_eri(True)
t0 = time.perf_counter()
qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
torch.cuda.synchronize(dev); t1 = time.perf_counter()
it = [a for a in sys.argv[1:] if a.isdigit()]
qc.run(fwd_options={"maxiter": int(it[0]) if it else 50})
torch.cuda.synchronize(dev); t2 = time.perf_counter()
if rank == 0:
    print("ranks %d: nao %d, grid points on this rank %d, setup %.2f s, %d SCF iterations in %.2f s (%.3f s each), E = %.10f, converged %s" % (
        world, h._nao_ao, h.rgrid.shape[0], t1 - t0, qc.niter, t2 - t1, (t2 - t1) / qc.niter, float(qc.energy()), qc.converged), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
