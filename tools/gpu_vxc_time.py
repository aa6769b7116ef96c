"""time the Vxc kernel alone on the C5 shape (used with DQC_AMD_LIB ablation builds)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
dev = torch.device("cuda:0")
nao, ngrid = 208, 353400
ld = lib.padded_nao(nao)
ao = torch.randn((4, ngrid, ld), dtype=torch.float64, device=dev)
w = torch.rand(ngrid, dtype=torch.float64, device=dev)
v = torch.randn(ngrid, dtype=torch.float64, device=dev)
vg = torch.randn((3, ngrid), dtype=torch.float64, device=dev)
for gga in (True, False):
    f = lambda: lib.grid_vxc(ao if gga else ao[0], nao, w, v, vg if gga else None)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print("vxc gga=%d: %.3f ms" % (gga, e0.elapsed_time(e1) / 20))
