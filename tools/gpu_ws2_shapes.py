"""vxc_ws2_kernel rectangle shapes (DQC_WS2_NR / DQC_WS2_NC): GGA Vxc time on synthetic data for several basis sizes, every
configuration checked against a torch reference of the same contraction"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
dev = torch.device("cuda")
def ev(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for nao, G, cfgs in ((232, 100000, [None, (2, 2)]), (264, 206304, [None, (3, 2), (2, 2), (2, 3)]), (390, 150000, [None, (4, 3), (3, 3)]),
                     (412, 310420, [None, (4, 3), (3, 3), (3, 4)]), (520, 150000, [None, (5, 4), (4, 4)]), (624, 200000, [None, (5, 4)])):
    ld = lib.padded_nao(nao)
    ao = lib.ao_from(torch.randn((4, G, nao), dtype=torch.float64, device=dev))
    w = torch.rand(G, dtype=torch.float64, device=dev); v = torch.randn(G, dtype=torch.float64, device=dev); vg = torch.randn((3, G), dtype=torch.float64, device=dev)
    a = ao[:, :, :nao]
    psi = w[:, None] * (v[:, None] * a[0] + 2.0 * (vg[:, :, None] * a[1:4]).sum(0))
    m = a[0].T @ psi
    ref = 0.5 * (m + m.T)
    del psi, m, a
    for c in cfgs:
        for k in ("DQC_WS2_NR", "DQC_WS2_NC"): os.environ.pop(k, None)
        if c: os.environ["DQC_WS2_NR"], os.environ["DQC_WS2_NC"] = str(c[0]), str(c[1])
        try:
            t = ev(lambda: lib.grid_vxc(ao, nao, w, v, vg))
            out = lib.grid_vxc(ao, nao, w, v, vg)
            err = float((out[:nao, :nao] - ref).abs().max() / ref.abs().max())
            fl = 2.0 * G * nao * nao + 8.0 * G * nao
            print("nao %d (T %d) NR x NC %s: %.3f ms = %.1f TF = %.2f of peak, rel err vs torch %.1e" % (nao, ld // 16, c or "default", t, fl / t / 1e9, fl / t / 1e9 / 78.6, err), flush=True)
        except Exception as e:
            print("nao %d %s: %r" % (nao, c, str(e)[:120]))
    del ao, ref
