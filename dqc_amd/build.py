"""Build libdqc_amd.so (gfx950) in-tree with hipcc.  `python -m dqc_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdqc_amd.so")
SOURCES = ["host.hip", "int1e.hip", "eri.hip", "df.hip", "grad.hip", "jk.hip", "fock.hip", "gto.hip", "becke.hip", "grid_density.hip", "grid_vxc.hip", "grid_xcgrad.hip", "xc.hip", "purify.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result", "-Wno-int-to-pointer-cast"]


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


def _deps(path, seen=None):
    """the file and every header it includes with quotes, transitively (an edit of eri_core.hpp does not rebuild the grid kernels)"""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), flags=re.M):
            _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, sorted(_deps(src))):
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


GRID_SOURCES = ["grid_density.hip", "grid_vxc.hip"]


def build_variant(name, defines, force=False):
    """perf-bisection builds: the grid sources recompiled with -D<define>, linked as libdqc_amd_<name>.so and selected with the
    DQC_AMD_LIB environment variable; never built by build_all(), never used by tests or bench defaults"""
    build()
    hipcc = _hipcc()
    out = os.path.join(HERE, "libdqc_amd_%s.so" % name)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES if s not in GRID_SOURCES]
    for src in GRID_SOURCES:
        obj = os.path.join(OBJ, "%s_%s.o" % (src.replace(".hip", ""), name))
        if force or _stale(obj, sorted(_deps(os.path.join(CSRC, src)))):
            subprocess.check_call([hipcc] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    if force or _stale(out, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_all(force=False, verbose=False):
    """the product library (one .so)"""
    return build(force=force, verbose=verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
