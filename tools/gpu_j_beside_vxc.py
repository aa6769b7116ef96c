"""Round 6: the Vxc kernel allocates 136 VGPRs per wave, three waves per SIMD = 408 of 512; a Coulomb-stream block of <= 104 VGPRs per
wave fits beside it on every CU.  Does it run there?  J alone, Vxc alone, density alone, then J on one stream beside Vxc / density on
another (ordinary streams, whole chip).  Writes gpurun_out/j_beside_vxc.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dqc_amd
from dqc_amd import lib
from tests import molecules as M

dev = torch.device("cuda")
out = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.append(s)


def make(i):
    mol = dqc_amd.Mol(M.c5_molecule(i), basis="cc-pvdz", grid="sg3")
    eng = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")._engine
    h = eng.hamilton
    n = eng.shape[-1]
    dm = eng.scp2dm(eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device=dev)))
    orb = eng.scp2orb(eng.dm2scp(dm)).contiguous()
    d = h.ao_orb2dm(orb, eng.orb_weight)
    return eng, h, d


eng, h, d = make(0)
eng2, h2, d2 = make(1)
fac = h._factor_of(d)[0]
nao = h._nao_ao
dao = (fac[0] @ fac[1])[:nao, :nao].contiguous()
terms = h.xc.terms
w = h.dvolume
rho, grho = lib.grid_density_lr(h._ao, nao, fac, True)
_, v, vg = lib.xc_eval(terms, rho, grho, want_e=False, want_v=True)


def coulomb():
    return lib.jk(h2._tiles, dao, h2._jkwork, False)[0]


def vxc():
    return lib.grid_vxc(h._ao, nao, w, v, vg)


def density():
    return lib.grid_density_lr(h._ao, nao, fac, True)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def both(fa, fb, n=20):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        fa()
    with torch.cuda.stream(sb):
        fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(sa):
            fa()
        with torch.cuda.stream(sb):
            fb()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa)
    eb.record(sb)
    ea.synchronize()
    ta = time.perf_counter() - t0
    eb.synchronize()
    tb = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * max(ta, tb) / n, 1e3 * ta / n, 1e3 * tb / n


say("J kernel build:", os.environ.get("DQC_BUILD_NOTE", "?"))
tj, tv, td = timed(coulomb), timed(vxc), timed(density)
say("alone: J %.4f ms   Vxc %.4f ms   density %.4f ms" % (tj, tv, td))
for label, fb, tb in (("Vxc", vxc, tv), ("density", density, td)):
    for _ in range(2):
        r = both(coulomb, fb)
        say("J | %-8s on two streams: %.4f ms per pair (J done %.4f, other done %.4f); serial sum %.4f -> hidden %.0f %% of J"
            % (label, r[0], r[1], r[2], tj + tb, 100 * (tj + tb - r[0]) / tj))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/j_beside_vxc.txt", "a").write("\n".join(out) + "\n")
