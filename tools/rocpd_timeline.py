"""Print the last N kernel intervals (start, end relative to the first of them, us) of a rocprofv3 rocpd database whose names match a pattern.
usage: python tools/rocpd_timeline.py <results.db> <substring,substring,...> [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
pats = sys.argv[2].split(",")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
rows = [r for r in rows if any(p in r[0] for p in pats)][-n:]
t0 = rows[0][1]
for nm, s, e in rows:
    print("%-40s start %10.1f  end %10.1f  dur %8.1f us" % (nm[:40], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
