"""Per-kernel PMC totals from a rocprofv3 rocpd sqlite db (run on the GPU box; prints a small table).
usage: python tools/pmc_summary.py <db> <counter-name>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'")]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("# columns:", cols)
name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
q = ("select %s, counter_name, count(*), avg(value), sum(value) from counters_collection "
     "group by %s, counter_name order by 5 desc" % (name_col, name_col))
for r in cur.execute(q).fetchall()[:14]:
    print("%-70s %-12s n=%-5d avg=%.6g sum=%.6g" % (r[0][:70], r[1], r[2], r[3], r[4]))
