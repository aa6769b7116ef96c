"""timeline of ONE device-loop SCF iteration from a rocprofv3 kernel trace of tools/gpu_devloop_probe.py <name>:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/<tag> -- python tools/gpu_devloop_probe.py c5
    python tools/devloop_timeline.py gpurun_out/<tag> > profiles/<tag>_devloop_timeline.txt
takes the last 10 occurrences of the iteration's first kernel as iteration boundaries, prints per kernel: launches per iteration,
busy microseconds per iteration, and the iteration's span / union-busy / idle time (streams overlap: union over all queues)"""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the iteration starts with the commutator bmm of devscf._iteration -- find the most frequent periodic pattern: use the DIIS kernel as anchor
anchor = [i for i, r in enumerate(rows) if "diis_solve" in r[2]]
if len(anchor) < 12:
    sys.exit("no device-loop iterations in the trace")
last = anchor[-11:]
spans, busy_u, per = [], [], collections.OrderedDict()
for a, b in zip(last[:-1], last[1:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    spans.append((t1 - t0) / 1e3)
    iv = sorted((s, e) for s, e, _ in seg)
    u, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            u += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    u += cur_e - cur_s
    busy_u.append(u / 1e3)
    for s, e, n in seg:
        k = n.split("(")[0][:100]
        c = per.setdefault(k, [0, 0.0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e3
        c[2] = (s - t0) / 1e3
n = len(spans)
print("%s: %d iterations (anchored at dqc diis_solve_kernel)" % (f, n))
print("iteration span %.1f us (min %.1f max %.1f), some kernel running %.1f us, idle %.1f us" % (sum(spans) / n, min(spans), max(spans), sum(busy_u) / n, (sum(spans) - sum(busy_u)) / n))
print("%8s %10s %10s  kernel" % ("launches", "busy us", "starts at"))
for k, (c, t, at) in sorted(per.items(), key=lambda kv: kv[1][2]):
    print("%8.1f %10.1f %10.1f  %s" % (c / n, t / n, at, k))
print("sum of kernel durations per iteration %.1f us" % (sum(v[1] for v in per.values()) / n))
