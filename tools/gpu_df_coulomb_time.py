import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
warnings.simplefilter("ignore")
m = dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz").densityfit()
h = m.get_hamiltonian(); h.build()
df = h.df
n = h._nao_ao
d = torch.randn(n, n, dtype=torch.float64, device="cuda"); d = d + d.T
for _ in range(5): J = lib.df_coulomb(df.j3c, df._inv_j2c, d, df._work)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): J = lib.df_coulomb(df.j3c, df._inv_j2c, d, df._work)
e1.record(); torch.cuda.synchronize()
naux = df.j2c.shape[0]
ms = e0.elapsed_time(e1) / 100
by = 2 * 4.0 * n * (n + 1) * naux + 8.0 * naux * naux
print("dqc_df_coulomb nao %d naux %d: %.4f ms -> %.2f TB/s of algorithmic bytes (%.2f of 8 TB/s)" % (n, naux, ms, by / ms / 1e9, by / ms / 8e9))
