"""per-class kernel time of the fill (mode 0) and of the direct Coulomb pass (mode 6) from one rocpd database"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start) from kernels group by %s" % (name_col, name_col)).fetchall()
t = {}
for n, c, s in rows:
    m = re.search(r"eri_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", n)
    if not m:
        continue
    cls, mode = tuple(int(x) for x in m.groups()[:4]), int(m.group(5))
    t.setdefault(cls, {}).setdefault(mode, [0.0, 0])
    t[cls][mode][0] += s / 1e3 / 2  # (two passes of each mode)
    t[cls][mode][1] += 1
tot = {0: 0.0, 6: 0.0}
out = []
for cls, d in t.items():
    f, j = d.get(0, [0.0])[0], d.get(6, [0.0])[0]
    tot[0] += f; tot[6] += j
    out.append((j - f, cls, f, j))
out.sort(reverse=True)
print("class (la lb|lc ld)   fill us   direct J us   J - fill")
for dlt, cls, f, j in out[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print("(%d %d|%d %d)   %9.1f   %9.1f   %+9.1f" % (*cls, f, j, dlt))
print("sum: fill %.1f ms, direct J %.1f ms" % (tot[0] / 1e3, tot[6] / 1e3))
