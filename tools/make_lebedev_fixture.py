"""Build dqc_amd/data/lebedev.npz from the Lebedev-Laikov quadrature tables.

The tables are published numerical data (V.I. Lebedev & D.N. Laikov, Doklady
Mathematics 59 (1999) 477); the reference keeps them as text files
(dqc/datasets/lebedevquad/lebedev_XXX.txt, rows = phi[deg] theta[deg] weight,
see dqc/grid/lebedev_grid.py:10-24).  This script (run once, in the build
container, where /root/reference exists) converts them to one compressed npz
so that both the product grid builder and the oracle read the same *data*.
Only data travels; no reference source is copied.
"""
import glob
import os
import re

import numpy as np

src = "/root/reference/dqc/datasets/lebedevquad"
out = {}
for f in sorted(glob.glob(os.path.join(src, "lebedev_*.txt"))):
    prec = int(re.search(r"lebedev_(\d+)\.txt", f).group(1))
    out["prec%03d" % prec] = np.loadtxt(f)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dqc_amd", "data", "lebedev.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, len(out), "orders", sum(v.shape[0] for v in out.values()), "points")
