import sys
sys.path.insert(0, "/root/repo")
import torch
from dqc_amd import lib
dev = torch.device("cuda")
terms = [(1.0, "gga_x_pbe"), (1.0, "gga_c_pbe")]
for n in (64, 64 * 256, 64 * 1024, 64 * 4096, 353400):
    rho = torch.rand(n, dtype=torch.float64, device=dev) + 0.01
    g = torch.randn((3, n), dtype=torch.float64, device=dev)
    f = lambda: lib.xc_eval(terms, rho, g, want_e=False, want_v=True)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("n=%d waves=%d: %.2f us per call" % (n, n // 64, e0.elapsed_time(e1) / 50 * 1e3))
