"""Round 6, VERDICT r5 item 1(b): does the Coulomb stream hide beside the grid pass when each gets its own compute units?
The Vxc kernel holds most of the VGPRs and LDS of a CU, so two launches on ordinary streams never share the chip (8 streams: 724 it/s against 741 for
the serial sum of the three hot kernels).  Here: J on CUs [32 - k, 32) of every XCD, the grid pass (density, functional, Vxc) of
ANOTHER molecule on CUs [0, 32 - k); each alone on its partition, then both at once.  One C5 molecule's arrays serve both roles
(different buffers are touched: tiles vs AO matrix), a second molecule's tile store is used when --two is given.
Writes gpurun_out/cu_partition_curves.txt."""
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
G = h.rgrid.shape[0]
say("C5 molecule: nao", nao, "ngrid", G, "device CUs", lib.device_cu_count())


def coulomb(hh=h2):
    return lib.jk(hh._tiles, dao, hh._jkwork, False)[0]


def grid_pass():
    rho, grho = lib.grid_density_lr(h._ao, nao, fac, True)
    _, v, vg = lib.xc_eval(terms, rho, grho, want_e=False, want_v=True)
    return lib.grid_vxc(h._ao, nao, w, v, vg)


def parts():
    rho, grho = lib.grid_density_lr(h._ao, nao, fac, True)
    _, v, vg = lib.xc_eval(terms, rho, grho, want_e=False, want_v=True)
    return rho, grho, v, vg


def timed(fn, stream, n=20):
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / n


main = torch.cuda.current_stream()
tj = timed(coulomb, main)
tg = timed(grid_pass, main)
rho, grho, v, vg = parts()
td = timed(lambda: lib.grid_density_lr(h._ao, nao, fac, True), main)
tx = timed(lambda: lib.xc_eval(terms, rho, grho, want_e=False, want_v=True), main)
tv = timed(lambda: lib.grid_vxc(h._ao, nao, w, v, vg), main)
say("whole chip, one stream:  J %.4f ms   grid pass %.4f ms  (density %.4f, xc %.4f, vxc %.4f)   serial sum %.4f ms -> %.1f builds/s"
    % (tj, tg, td, tx, tv, tj + tg, 1e3 / (tj + tg)))

# two ordinary streams, J beside the grid pass (what rounds 2-5 did with 8 streams)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(sa, sb, n=20, fa=coulomb, fb=grid_pass):
    """n iterations of fa on sa and fb on sb, started together; wall time per iteration (host clock around device syncs)"""
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


r = both(s1, s2)
say("two ordinary streams (J | grid pass): %.4f ms per pair  (J stream done at %.4f, grid stream at %.4f) -> %.1f builds/s" % (r[0], r[1], r[2], 1e3 / r[0]))

say("")
say("partitioned: J on k CUs of every XCD (8k in all), the grid pass on the other 32 - k")
say("%4s %6s | %9s %9s %9s %9s %9s | %9s %9s %9s | %8s" % ("k", "J CUs", "J alone", "grid", "density", "xc", "vxc", "together", "J done", "grid done", "builds/s"))
best = None
for k in [int(x) for x in os.environ.get("KS", "2,3,4,5,6,8,12,16").split(",")]:
    pj = lib.partition_stream(dev, 32 - k, 32)
    pg = lib.partition_stream(dev, 0, 32 - k)
    tj_ = timed(coulomb, pj.stream)
    tg_ = timed(grid_pass, pg.stream)
    td_ = timed(lambda: lib.grid_density_lr(h._ao, nao, fac, True), pg.stream)
    tx_ = timed(lambda: lib.xc_eval(terms, rho, grho, want_e=False, want_v=True), pg.stream)
    tv_ = timed(lambda: lib.grid_vxc(h._ao, nao, w, v, vg), pg.stream)
    r = both(pj.stream, pg.stream)
    say("%4d %6d | %9.4f %9.4f %9.4f %9.4f %9.4f | %9.4f %9.4f %9.4f | %8.1f" % (k, pj.cus, tj_, tg_, td_, tx_, tv_, r[0], r[1], r[2], 1e3 / r[0]))
    if best is None or r[0] < best[1]:
        best = (k, r[0])
    pj.close()
    pg.close()
say("best split: k = %d -> %.4f ms per build = %.1f builds/s (serial whole-chip: %.1f)" % (best[0], best[1], 1e3 / best[1], 1e3 / (tj + tg)))

# the J kernel against the number of CUs it may use (is it bound by HBM or by the CUs?)
say("")
say("J stream alone on 8 k CUs:")
for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    p = lib.partition_stream(dev, 32 - k, 32)
    t = timed(coulomb, p.stream)
    say("  %3d CUs  %.4f ms  %.2f TB/s  (%.1f GB/s per CU)" % (p.cus, t, 1.872e9 / t / 1e9, 1.872e9 / t / 1e6 / p.cus))
    p.close()
say("density pass alone on 8 k CUs:")
for k in (16, 24, 28, 30, 32):
    p = lib.partition_stream(dev, 0, k)
    t = timed(lambda: lib.grid_density_lr(h._ao, nao, fac, True), p.stream)
    say("  %3d CUs  %.4f ms  %.2f TB/s" % (p.cus, t, 2.366e9 / t / 1e9))
    p.close()

os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/cu_partition_curves.txt", "w") as f:
    f.write("\n".join(out) + "\n")
