"""Build libdqc_amd.so (gfx950) in-tree with hipcc.  `python -m dqc_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdqc_amd.so")
SOURCES = ["host.hip", "int1e.hip", "eri.hip", "df.hip", "grad.hip", "jk.hip", "gto.hip", "becke.hip", "grid.hip", "xc.hip", "purify.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "dqc_amd.h"))
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_variant(name, defines):
    """perf-bisection builds: grid.hip recompiled with -D<define>, linked as libdqc_amd_<name>.so
    (select with the DQC_AMD_LIB environment variable); never used by tests or bench defaults"""
    build()
    hipcc = _hipcc()
    obj = os.path.join(OBJ, "grid_%s.o" % name)
    subprocess.check_call([hipcc] + FLAGS + ["-D" + d for d in defines] +
                          ["-c", os.path.join(CSRC, "grid.hip"), "-o", obj])
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES if s != "grid.hip"] + [obj]
    out = os.path.join(HERE, "libdqc_amd_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
