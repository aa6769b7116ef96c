import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from tests import molecules as M
name = sys.argv[1]
geo, basis, xc = {"c5": (M.c5_molecule(0), "cc-pvdz", "gga_x_pbe+gga_c_pbe"), "c3": (M.benzene(), "cc-pvdz", "lda_x+lda_c_pw")}[name]
qc = dqc_amd.KS(dqc_amd.Mol(geo, basis=basis, grid="sg3"), xc=xc)
qc.run()
loop = qc._devloop
eng = qc._engine
import dqc_amd.devscf as D
def T():
    torch.cuda.synchronize(); return time.perf_counter()
n = loop.n
t0 = T()
z = torch.zeros((n, n), dtype=torch.float64, device="cuda")
f0 = eng.dm2scp(z)
t1 = T()
loop._build_from(f0.reshape(1, n, n))
t2 = T()
for t in (loop.fh, loop.eh, loop.gram, loop.count): t.zero_()
pe = float(loop.perr)
t3 = T()
K = 15
for k in range(K):
    loop.graph.replay()
t4h = time.perf_counter()
t4 = T()
print("%s: f0 build %.2f ms, first step (eager) %.2f ms, zero+sync %.2f ms, %d replays: host returned after %.2f ms, done after %.2f ms (%.3f per replay)" % (
    name, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), K, 1e3*(t4h-t3), 1e3*(t4-t3), 1e3*(t4-t3)/K))
# replay + event + host read pattern
evs = [torch.cuda.Event(), torch.cuda.Event()]
st = torch.cuda.current_stream()
t5 = T()
for k in range(K):
    loop.graph.replay(); evs[k % 2].record(st)
    if k >= 1:
        evs[(k - 1) % 2].synchronize(); x = float(loop.host[(k - 1) % 2, 0])
t6 = T()
print("   with the lagged host look: %.3f ms per iteration" % (1e3*(t6-t5)/K))
