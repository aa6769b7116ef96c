import sys, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
out = bench.f_row_legs(dev, K=10)
def show(d, ind=0):
    for k, v in d.items():
        if k == "entries":
            for r in v:
                print(" " * ind + "  %-26s x%.1f  %.4f ms  %s" % (r["entry"], r["calls_per_build"], r["ms_per_call"], ("%s %.2f" % (r["bound"], r["frac"])) if "frac" in r else ""))
        elif isinstance(v, dict):
            print(" " * ind + k + ":"); show(v, ind + 2)
        else:
            print(" " * ind + "%s: %s" % (k, v))
show(out)
