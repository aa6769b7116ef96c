"""fused grid kernel vs the three-kernel path on synthetic data (run under `timeout`: a barrier bug would hang)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqc_amd import lib
dev = torch.device("cuda:0")
terms = [(1.0, "gga_x_pbe"), (1.0, "gga_c_pbe")]
for nao, ngrid, nocc in ((208, 353400, 46), (170, 30011, 20), (200, 5000, 64), (208, 17, 5), (208, 4099, 33)):
    ld = lib.padded_nao(nao)
    g = torch.Generator().manual_seed(nao + ngrid)
    ao = torch.randn((4, ngrid, nao), dtype=torch.float64, generator=g) * torch.exp(-3 * torch.rand((4, ngrid, nao), dtype=torch.float64, generator=g))
    ao = lib.ao_from(ao.to(dev))
    w = torch.rand(ngrid, dtype=torch.float64, generator=g).to(dev)
    c = (torch.randn((nao, nocc), dtype=torch.float64, generator=g) * 0.3).to(dev)
    fac = lib.pad_factor(c, ld)
    assert lib.grid_fused_supported(nao, fac[0].shape[1]), (nao, fac[0].shape)
    rho, grho = lib.grid_density_lr(ao, nao, fac, True)
    e, v, vg = lib.xc_eval(terms, rho, grho, want_e=True, want_v=True)
    vref = lib.grid_vxc(ao, nao, w, v, vg)
    exc_ref = float((w * e).sum())
    vm, r2, g2, exc = lib.grid_fused(ao, nao, w, fac, terms, want_dens=True, want_exc=True)
    torch.cuda.synchronize()
    sc = float(vref.abs().max())
    print("nao %d ngrid %d nocc %d: rho %.2e grho %.2e vxc %.2e exc %.2e" % (
        nao, ngrid, nocc, float((r2 - rho).abs().max() / rho.abs().max()), float((g2 - grho).abs().max() / grho.abs().max()),
        float((vm - vref).abs().max()) / sc, abs(float(exc) - exc_ref) / abs(exc_ref)), flush=True)
    if ngrid > 100000:
        for name, f in (("fused", lambda: lib.grid_fused(ao, nao, w, fac, terms)),
                        ("3 kernels", lambda: lib.grid_vxc(ao, nao, w, *lib.xc_eval(terms, *lib.grid_density_lr(ao, nao, fac, True), want_e=False, want_v=True)[1:]))):
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            print("   %-10s %.3f ms" % (name, e0.elapsed_time(e1) / 10), flush=True)
    del ao
