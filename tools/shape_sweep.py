"""Per-kernel time and achieved fp64 TF / GB/s of the grid kernels over basis sizes and occupied-space widths
(VERDICT r1 item 5): nao in {114, 208, 264, 412, 624} x n_occ in {21, 46, 70, 128}, GGA, synthetic AO arrays of the
real sg3 grid sizes.  Run on the GPU box, optionally under `rocprofv3 --kernel-trace --stats` (tools/rocpd_summary.py
then gives the same averages from the profiler).     usage: python tools/shape_sweep.py [out.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dqc_amd import lib  # noqa: E402

dev = torch.device("cuda:0")
PEAK_TF, PEAK_GBS = 78.6, 8000.0
SHAPES = [(114, 206304), (208, 353400), (264, 206304), (412, 310000), (624, 500000)]
NOCC = [21, 46, 70, 128]


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    lines = ["%-5s %-7s %-5s %-22s %9s %9s %8s %8s" % ("nao", "ngrid", "nocc", "kernel", "ms", "TFLOP/s", "GB/s", "frac")]
    for nao, ngrid in SHAPES:
        ld = lib.padded_nao(nao)
        g = torch.Generator(device="cpu").manual_seed(nao)
        ao = lib.ao_empty(4, ngrid, nao, dev, zero=True)
        # AO-like magnitudes (many small values): exp(-|N(0, 4)|^2) * N(0, 1)
        blk = 65536
        for c in range(4):
            for s in range(0, ngrid, blk):
                e = min(s + blk, ngrid)
                x = torch.randn((e - s, nao), dtype=torch.float64, generator=g)
                ao[c, s:e, :nao] = (torch.exp(-(2 * torch.randn((e - s, nao), dtype=torch.float64, generator=g)) ** 2) * x).to(dev)
        w = torch.rand(ngrid, dtype=torch.float64, generator=g).to(dev)
        v = torch.randn(ngrid, dtype=torch.float64, generator=g).to(dev)
        vg = torch.randn((3, ngrid), dtype=torch.float64, generator=g).to(dev)
        by = 8.0 * 4 * ngrid * nao
        t = timeit(lambda: lib.grid_vxc(ao, nao, w, v, vg))
        fl = 2.0 * ngrid * ld * ld
        lines.append("%-5d %-7d %-5s %-22s %9.3f %9.1f %8.0f %8.2f" % (nao, ngrid, "-", "vxc (GGA)", t, fl / t / 1e9, by / t / 1e6,
                                                                      max(fl / t / 1e9 / PEAK_TF, by / t / 1e6 / PEAK_GBS)))
        dm = torch.randn((nao, nao), dtype=torch.float64, generator=g).to(dev)
        dmp = lib.pad_matrix(dm + dm.T, ld)
        t = timeit(lambda: lib.grid_density(ao, nao, dmp, True))
        lines.append("%-5d %-7d %-5s %-22s %9.3f %9.1f %8.0f %8.2f" % (nao, ngrid, "-", "density (full D)", t, fl / t / 1e9, by / t / 1e6,
                                                                      max(fl / t / 1e9 / PEAK_TF, by / t / 1e6 / PEAK_GBS)))
        for r in NOCC:
            if r >= nao:
                continue
            c = torch.randn((nao, r), dtype=torch.float64, generator=g).to(dev)
            # wider than the widest instantiation: column panels, as HamiltonMI355._factor_of does
            pan = [c] if lib.padded_norb(r) else [c[:, i:i + (r + 1) // 2].contiguous() for i in range(0, r, (r + 1) // 2)]
            facs = [lib.pad_factor(p, ld) for p in pan]
            t = timeit(lambda: [lib.grid_density_lr(ao, nao, f, True) for f in facs])
            rp = sum(f[0].shape[1] for f in facs)
            fl2 = 4.0 * ngrid * ld * rp
            lines.append("%-5d %-7d %-5d %-22s %9.3f %9.1f %8.0f %8.2f" % (nao, ngrid, r, "density (factor, %d)" % rp, t, fl2 / t / 1e9,
                                                                          by * len(facs) / t / 1e6,
                                                                          max(fl2 / t / 1e9 / PEAK_TF, by * len(facs) / t / 1e6 / PEAK_GBS)))
        del ao
        torch.cuda.empty_cache()
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
