"""sum of the ERI class kernels' GPU time in a rocprofv3 rocpd database: python tools/eri_kernel_sum.py <db> <nfills>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
tot, n = cur.execute("select sum(end-start), count(*) from kernels where %s like '%%eri_kernel%%'" % name_col).fetchone()
print("eri kernels: %.2f ms per fill (%d launches, %s fills)" % (tot / 1e6 / int(sys.argv[2]), n, sys.argv[2]))
