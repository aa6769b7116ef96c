"""Compile every HIP source for gfx950 with -save-temps and list the kernels whose metadata reports VGPR spills or scratch.
usage: python tools/check_spills.py [out.txt]        (CPU only: hipcc cross-compiles; takes a few minutes)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dqc_amd import build as b  # noqa: E402


def one(src):
    d = tempfile.mkdtemp(prefix="spill_")
    subprocess.run([b._hipcc()] + b.FLAGS + ["-I" + os.path.join(ROOT, "include"), "-save-temps", "-c", os.path.join(b.CSRC, src), "-o",
                    os.path.join(d, "x.o")], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(d, asm)).read()
    md = s[s.find("amdhsa.kernels"):]
    rows = []
    for e in md.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", e).group(1)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1))  # noqa: E731
        rows.append((src, name, g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
    return rows


def main():
    with ThreadPoolExecutor(max_workers=4) as ex:
        rows = [r for rs in ex.map(one, b.SOURCES) for r in rs]
    bad = [r for r in rows if r[3] > 0 or r[4] > 0]
    lines = ["%d kernels in %d sources; %d with VGPR spills or scratch" % (len(rows), len(b.SOURCES), len(bad))]
    for src, name, v, sp, sc in bad:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        lines.append("%-10s %-100s vgpr %3d spill %3d scratch %4d" % (src, dem[:100], v, sp, sc))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
