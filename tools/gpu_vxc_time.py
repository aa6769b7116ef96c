"""time the Vxc kernel alone on the C5 shape (used with DQC_AMD_LIB ablation builds); argv[1] = data kind:
randn (default) | zeros | small (randn * 1e-3 with 90 % exact zeros) | ones"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "randn"
nao, ngrid = 208, 353400
ld = lib.padded_nao(nao)
torch.manual_seed(0)
if kind == "zeros":
    ao = lib.ao_from(torch.zeros((4, ngrid, nao), dtype=torch.float64, device=dev))
elif kind == "ones":
    ao = lib.ao_from(torch.ones((4, ngrid, nao), dtype=torch.float64, device=dev))
else:
    ao = lib.ao_from(torch.randn((4, ngrid, nao), dtype=torch.float64, device=dev))
    if kind == "small":
        ao = ao * 1e-3 * (torch.rand_like(ao) > 0.9)
w = torch.rand(ngrid, dtype=torch.float64, device=dev)
v = torch.randn(ngrid, dtype=torch.float64, device=dev)
vg = torch.randn((3, ngrid), dtype=torch.float64, device=dev)
for gga in (True, False):
    f = lambda: lib.grid_vxc(ao if gga else ao[0], nao, w, v, vg if gga else None)
    vm = f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print("vxc %s gga=%d lib %s: %.4f ms  checksum %.12e" % (kind, gga, os.path.basename(os.environ.get("DQC_AMD_LIB", "default")),
                                                          e0.elapsed_time(e1) / 20, float(vm.sum())))
