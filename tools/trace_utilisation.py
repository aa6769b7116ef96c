"""utilisation of a rocprofv3 --kernel-trace csv: over the LAST `frac` of the trace (the steady part), the share of wall time with
(a) any kernel running, (b) one of the hot kernels (vxc / density / j_stream / jk_stream) running, and the sum of hot-kernel
durations over wall time (> 1: several at once).  usage: python tools/trace_utilisation.py <dir> [frac=0.3]"""
import csv, glob, sys
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t1 = max(r[1] for r in rows)
t0 = t1 - int((t1 - rows[0][0]) * frac)
sel = [r for r in rows if r[0] >= t0]
HOT = ("vxc_ws", "density_lr_kernel", "j_stream_kernel", "jk_stream_kernel", "density_kernel")
def union(iv):
    iv = sorted(iv)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return u + ce - cs
wall = t1 - t0
allk = [(s, e) for s, e, n in sel]
hot = [(s, e) for s, e, n in sel if any(h in n for h in HOT)]
print("%s: window %.1f ms, %d kernels (%d hot)" % (f, wall / 1e6, len(sel), len(hot)))
print("some kernel running %.3f of the time; a hot kernel running %.3f; sum of hot durations / wall %.2f; sum of all durations / wall %.2f" % (
    union(allk) / wall, union(hot) / wall, sum(e - s for s, e in hot) / wall, sum(e - s for s, e in allk) / wall))
# gaps between hot kernels (no hot kernel running): histogram
iv = sorted(hot)
gaps, ce = [], iv[0][1]
for s, e in iv[1:]:
    if s > ce: gaps.append((s - ce) / 1e3)
    ce = max(ce, e)
gaps.sort(reverse=True)
print("time without a hot kernel: %.1f ms in %d gaps; the 10 longest (us): %s" % (sum(gaps) / 1e3, len(gaps), ", ".join("%.0f" % g for g in gaps[:10])))
