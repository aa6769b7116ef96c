"""GPU check / timing of the density-fitted Coulomb path on the C5 molecule (etb auxiliary basis)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import functools
import torch
import dqc_amd
print = functools.partial(print, flush=True)
from tests import molecules as M

dev = torch.device("cuda:0")
xc = "gga_x_pbe+gga_c_pbe"
mol_x = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev)
eng_x = dqc_amd.KS(mol_x, xc=xc)._engine
torch.cuda.synchronize(); t0 = time.perf_counter()
mol_d = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3", device=dev).densityfit(auxbasis="etb")
eng_d = dqc_amd.KS(mol_d, xc=xc)._engine
torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
h = mol_d.get_hamiltonian()
print("naux %d  j3c %.1f MB  setup incl. grid/AO %.3f s" % (h.df.j2c.shape[0], h.df.j3c.numel() * 8 / 1e6, t_setup))
n = eng_x.shape[-1]
dm = eng_x.scp2dm(eng_x.dm2scp(torch.zeros((n, n), dtype=torch.float64, device=dev)))
Jx = eng_x.hamilton.get_elrep(dm).fullmatrix()
Jd = eng_d.hamilton.get_elrep(dm).fullmatrix()
print("max |J_df - J_exact| = %.2e  (|J| max %.2f)" % (float((Jx - Jd).abs().max()), float(Jx.abs().max())))


def timeit(f, nrep=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nrep):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nrep * 1e3


orb = eng_x.scp2orb(eng_x.dm2scp(dm)).contiguous()
print("get_elrep exact %.3f ms   DF %.3f ms" % (timeit(lambda: eng_x.hamilton.get_elrep(dm.clone())), timeit(lambda: eng_d.hamilton.get_elrep(dm.clone()))))
tx = timeit(lambda: eng_x.dm2scp(eng_x.hamilton.ao_orb2dm(orb, eng_x.orb_weight)))
td = timeit(lambda: eng_d.dm2scp(eng_d.hamilton.ao_orb2dm(orb, eng_d.orb_weight)))
print("Fock build (dm2scp) exact-J %.3f ms = %.1f it/s   DF-J %.3f ms = %.1f it/s" % (tx, 1e3 / tx, td, 1e3 / td))
ex = float(dqc_amd.KS(mol_x, xc=xc).run().energy())
ed = float(dqc_amd.KS(mol_d, xc=xc).run().energy())
print("E exact-J %.8f   E DF-J(etb) %.8f   diff %.2e Ha" % (ex, ed, ed - ex))
