"""cold-process costs that land in the first molecule's setup (run on the GPU box, fresh process): library load, first launches of
the vendor libraries, first molecule"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import torch
t1 = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2 = time.perf_counter()
import dqc_amd
from dqc_amd import lib
t3 = time.perf_counter()
lib.load()
t4 = time.perf_counter()
lib.padded_nao(10)
a = torch.randn(208, 208, dtype=torch.float64, device="cuda"); a = a + a.T
torch.cuda.synchronize(); t5 = time.perf_counter()
b = a @ a
torch.cuda.synchronize(); t6 = time.perf_counter()
torch.linalg.eigh(a)
torch.cuda.synchronize(); t7 = time.perf_counter()
torch.linalg.eigh(torch.stack([a] * 4))
torch.cuda.synchronize(); t8 = time.perf_counter()
torch.linalg.cholesky(b + 1e3 * torch.eye(208, dtype=torch.float64, device="cuda"))
torch.cuda.synchronize(); t9 = time.perf_counter()
from tests import molecules as M
qc = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(0), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe")
torch.cuda.synchronize(); t10 = time.perf_counter()
qc2 = dqc_amd.KS(dqc_amd.Mol(M.c5_molecule(1), basis="cc-pvdz", grid="sg3"), xc="gga_x_pbe+gga_c_pbe")
torch.cuda.synchronize(); t11 = time.perf_counter()
print("import torch %.2f s | first CUDA tensor %.2f | import dqc_amd %.2f | lib.load (dlopen of libdqc_amd.so) %.2f | randn %.3f | first GEMM %.3f | "
      "first eigh %.3f | first batched eigh %.3f | first cholesky %.3f | first C5 molecule %.3f | second C5 molecule %.3f" % (
          t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8, t10 - t9, t11 - t10))
