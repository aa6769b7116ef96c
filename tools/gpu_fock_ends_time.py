"""(eager event timings of single library calls are HOST-bound below ~20 us per call: use rocprofv3 for the kernels' own durations)
Round 6: the fused small-matrix ends of a Fock build (csrc/fock.hip) timed alone by HIP events against the torch GEMM form they
replace, for benzene / cc-pVDZ (nao 114), a C5 molecule (208) and naphthalene / cc-pVTZ-size matrices (412: random X, no tiles).
Writes gpurun_out/fock_ends_time.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib

dev = torch.device("cuda")
out = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.append(s)


def ev(fn, k=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / k  # us


g = torch.Generator().manual_seed(1)
for n, rp in ((114, 32), (208, 48), (412, 48)):
    north = n
    x = torch.randn((n, north), generator=g, dtype=torch.float64).to(dev).contiguous()
    dm = torch.randn((north, north), generator=g, dtype=torch.float64).to(dev)
    orb = torch.zeros((lib.padded_nao(n), rp), dtype=torch.float64, device=dev)
    orb[:n] = torch.randn((n, rp), generator=g, dtype=torch.float64).to(dev)
    work = lib.jk_workspace(n, dev)
    work.normal_()
    vm = torch.randn((lib.padded_nao(n),) * 2, generator=g, dtype=torch.float64).to(dev)
    core = torch.randn((north, north), generator=g, dtype=torch.float64).to(dev)
    t_prep_dm = ev(lambda: lib.fock_prep(work, x, n, True, dm=dm))
    t_prep_fac = ev(lambda: lib.fock_prep(work, x, n, False, orb=orb))
    t_fin_k = ev(lambda: lib.fock_finish(work, x, n, True, core=core))
    t_fin_v = ev(lambda: lib.fock_finish(work, x, n, False, vxc_ao=vm, core=core))
    # the torch form of the same steps (eager launches back to back on one stream: the GPU time of the chain, as a graph would replay it)
    J = torch.randn((n, n), generator=g, dtype=torch.float64).to(dev)

    def torch_prep():
        return x @ (0.5 * (dm + dm.T)) @ x.T

    def torch_fin():
        m = x.T @ (J - 0.5 * J + vm[:n, :n]) @ x
        e = 0.5 * (J * J).sum()
        return core + 0.5 * (m + m.T), e

    gp = torch.cuda.CUDAGraph()
    torch_prep(); torch_fin(); torch.cuda.synchronize()
    with torch.cuda.graph(gp):
        a_ = torch_prep()
    gf = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf):
        b_ = torch_fin()
    t_tp, t_tf = ev(gp.replay), ev(gf.replay)
    say("nao %3d: fock_prep (X D X^T) %6.1f us | from the factor %6.1f us | torch graph %6.1f us    fock_finish J+K %6.1f us | J+V %6.1f us | torch graph %6.1f us"
        % (n, t_prep_dm, t_prep_fac, t_tp, t_fin_k, t_fin_v, t_tf))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/fock_ends_time.txt", "w").write("\n".join(out) + "\n")
