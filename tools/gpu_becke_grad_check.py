"""analytic Becke-weight derivative (dqc_becke_weights_grad through _BeckeWeightsFn) against torch autograd through the element-wise
expression (the CPU path of dqc_amd.grid._becke_weights), same loss sum_g w_g c_g"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dqc_amd
from dqc_amd.grid import get_predefined_grid
from tests import molecules as M
for geo, grid in ((M.H2O, "sg2"), (M.benzene(), "sg3"), (M.H2O, 3), (M.c5_molecule(0), "sg3")):
    zs, pos0 = geo
    pos0 = torch.as_tensor(pos0, dtype=torch.float64)
    res = []
    for dev in ("cpu", "cuda"):
        pos = pos0.to(dev).clone().requires_grad_(True)
        g = get_predefined_grid(grid, list(zs), pos, dtype=torch.float64, device=dev)
        c = torch.sin(torch.arange(g.get_dvolume().shape[0], dtype=torch.float64, device=dev) * 0.37) + 0.2
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = (g.get_dvolume() * c).sum()
        gr = torch.autograd.grad(loss, pos)[0]
        torch.cuda.synchronize()
        res.append((gr.cpu(), time.perf_counter() - t0))
    print("natm %d grid %s: max |analytic - autograd| %.2e (max |g| %.2e), cpu %.3f s, gpu %.4f s" % (
        len(zs), grid, float((res[0][0] - res[1][0]).abs().max()), float(res[0][0].abs().max()), res[0][1], res[1][1]), flush=True)
