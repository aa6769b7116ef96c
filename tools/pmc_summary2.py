"""per-kernel table of several PMC counters from a rocprofv3 rocpd database: python tools/pmc_summary2.py <db> [name-filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
tab = {}
for k, c, n, v in rows:
    if flt in k:
        tab.setdefault(k, {})[c] = (n, v)
cs = sorted({c for k in tab for c in tab[k]})
print("%-60s %6s " % ("kernel", "n") + " ".join("%16s" % c[-16:] for c in cs))
for k in sorted(tab, key=lambda k: -max(v[1] for v in tab[k].values())):
    n = max(v[0] for v in tab[k].values())
    print("%-60s %6d " % (k[:60], n) + " ".join("%16.4g" % tab[k].get(c, (0, 0))[1] for c in cs))
