"""kernel-time sums of a traced direct SCF by category (rocpd database): direct ERI kernels, grid kernels, rocSOLVER / rocBLAS, the rest"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start) from kernels group by %s" % (name_col, name_col)).fetchall()
cat = {}
for n, c, s in rows:
    m = re.search(r"eri_kernel<\d+, \d+, \d+, \d+, (\d+)", n)
    if m:
        k = "eri mode " + m.group(1)
    elif "rocsolver" in n:
        k = "rocsolver"
    elif "rocblas" in n or "Cijk" in n:
        k = "rocblas / gemm"
    elif "dqc::" in n:
        k = "dqc " + n.split("dqc::")[1].split("<")[0].split("(")[0]
    else:
        k = "other (torch element-wise, copies)"
    cat.setdefault(k, [0, 0.0])
    cat[k][0] += c; cat[k][1] += s / 1e6
for k, (c, s) in sorted(cat.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-48s %7d launches %9.2f ms" % (k, c, s))
