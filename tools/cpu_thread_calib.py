import sys, time, os, torch
sys.path.insert(0, ".")
from oracle import basis as ob, hamilton as oh, natives as nat
from tests import molecules as M
t = ob.make_tables(M.benzene(), "cc-pvdz")
os.environ.setdefault("OMP_NUM_THREADS", "64")
t0 = time.time(); eng = oh.Engine(t, xc="gga_x_pbe+gga_c_pbe", grid="sg3", eri_mode="s4"); print("setup", time.time() - t0, "omp", nat.num_threads())
n = eng.h.nao
dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64)))
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    eng.dm2scp(dm)
    t0 = time.time()
    for _ in range(3): eng.dm2scp(dm)
    print("threads", nt, "dm2scp s", (time.time() - t0) / 3, flush=True)
