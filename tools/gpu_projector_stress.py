"""stress of the one-launch projector (dqc_projector_tc2: persistent kernel, workers on one XCD, bounded spin barriers): N calls on
Fock-like matrices of several sizes -- how many gave up (err = 1e300), the distribution of the call durations (HIP events), and the
same from inside hipGraph replays"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for n, nocc in ((114, 21), (208, 46), (24, 5), (250, 60)):
    g = torch.Generator().manual_seed(n)
    q, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, generator=g))
    ev = torch.cat([torch.linspace(-20.0, -0.3, nocc, dtype=torch.float64), torch.linspace(0.1, 4.0, n - nocc, dtype=torch.float64)])
    f = ((q * ev) @ q.T).to(dev)
    pref = (q[:, :nocc] @ q[:, :nocc].T).to(dev)
    lib.projector_tc2(f, nocc, 64, 1e-13)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    errs = []
    for i in range(N):
        evs[i][0].record()
        p, e = lib.projector_tc2(f, nocc, 64, 1e-13)
        evs[i][1].record()
        errs.append(e)
    torch.cuda.synchronize()
    ts = torch.tensor([a.elapsed_time(b) for a, b in evs])
    es = torch.stack(errs).cpu()
    bad = int((es > 1e-9).sum())
    print("n %3d: %d calls, gave up / not converged %d, max err of the rest %.1e, |P - P_eigh| %.1e; ms per call: median %.3f  p99 %.3f  max %.3f" % (
        n, N, bad, float(es[es <= 1e-9].max()) if bad < N else float("nan"), float((p - pref).abs().max()), float(ts.median()), float(ts.kthvalue(int(0.99 * N))[0]), float(ts.max())), flush=True)
    # inside a graph
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        lib.projector_tc2(f, nocc, 64, 1e-13)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    nbad = torch.zeros((), dtype=torch.int64, device=dev)
    with torch.cuda.graph(gr):
        pg, eg = lib.projector_tc2(f, nocc, 64, 1e-13)
        nbad += (eg > 1e-9)       # counted ON THE DEVICE in every replay: the replays below are back to back, no host sync between
    nbad.zero_()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(N):
        gr.replay()
    t1.record(); torch.cuda.synchronize()
    print("       graph replays back to back: %.3f ms each, failures %d of %d, |P - P_eigh| %.1e" % (t0.elapsed_time(t1) / N, int(nbad), N, float((pg - pref).abs().max())), flush=True)
