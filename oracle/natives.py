"""oracle/natives.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle_cint.so (the C restatement of the libcint /
libcgto arithmetic the reference reaches through dqclibs:
dqc/hamilton/intor/molintor.py:590-708, dqc/hamilton/intor/gtoeval.py:196-239).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_cint.so")
        src = os.path.join(_HERE, "cint_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _tab(t):
    return (_p(t.atm, ctypes.c_int), ctypes.c_int(t.natm), _p(t.bas, ctypes.c_int),
            ctypes.c_int(t.nbas), _p(t.env))


def int1e(which, t, zs=None):
    """which: 'ovlp' | 'kin' | 'nuc' -> (nao, nao); 'r0' -> (3, nao, nao), 'r0r0' -> (9, nao, nao): multipole moments about
    the origin (intor.int1e("r0" * n), hcgto.py:117-125)"""
    if which in ("r0", "r0r0"):
        codes = range(3, 6) if which == "r0" else range(6, 15)
        return np.stack([int1e(c, t) for c in codes])
    code = which if isinstance(which, int) else {"ovlp": 0, "kin": 1, "nuc": 2}[which]
    out = np.zeros((t.nao, t.nao))
    zp = None
    if zs is not None:
        zs = np.ascontiguousarray(zs, dtype=np.float64)
        zp = _p(zs)
    lib().orc_int1e(ctypes.c_int(code), _p(out), *_tab(t), zp)
    return out


def int2e_s4(t):
    """packed (npair, npair), pair index i(i+1)/2+j"""
    npair = t.nao * (t.nao + 1) // 2
    out = np.zeros((npair, npair))
    lib().orc_int2e_s4(_p(out), *_tab(t))
    return out


def int2e_s8(t):
    """packed lower triangle of the s4 matrix: out[P (P + 1) / 2 + Q], P >= Q AO-pair indices (29 GB for nao 412)"""
    npair = t.nao * (t.nao + 1) // 2
    out = np.zeros(npair * (npair + 1) // 2)
    lib().orc_int2e_s8(_p(out), *_tab(t))
    return out


def symv_s8(packed, x):
    """y = M x with M the symmetric matrix held as the packed lower triangle of int2e_s8"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().orc_symv_s8(_p(y), _p(packed), _p(x), ctypes.c_longlong(x.size))
    return y


def fills4(packed, nao):
    out = np.empty((nao, nao, nao, nao))
    lib().orc_fills4(_p(out), _p(np.ascontiguousarray(packed)), ctypes.c_int(nao))
    return out


def int2e(t):
    """dense (nao,)*4 like molintor.elrep after S4Symmetry expansion"""
    return fills4(int2e_s4(t), t.nao)


def int2e_quartets(t, quartets):
    """spherical blocks (sa, sb, sc, sd) of the listed shell quartets (nq, 4) -- for bases whose packed matrix
    does not fit the host (the numbers are those int2e_s4 would scatter)"""
    q = np.ascontiguousarray(quartets, dtype=np.int32).reshape(-1, 4)
    dims = 2 * t.bas[q, 1] + 1                      # (nq, 4)
    sizes = np.prod(dims, axis=1).astype(np.int64)
    offs = np.zeros(len(q) + 1, dtype=np.int64)
    np.cumsum(sizes, out=offs[1:])
    out = np.zeros(int(offs[-1]))
    lib().orc_int2e_quartets(_p(out), offs.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), _p(q, ctypes.c_int),
                             ctypes.c_int(len(q)), *_tab(t))
    return [out[offs[i]:offs[i + 1]].reshape(tuple(int(d) for d in dims[i])) for i in range(len(q))]


def int3c2e(tc, orb_range, aux_range):
    """(ij|k) over the concatenated tables `tc`: orbital shells [s0, s1), auxiliary shells [k0, k1) -> (nao, nao, naux)"""
    (s0, s1), (k0, k1) = orb_range, aux_range
    nao = int(tc.ao_loc[s1] - tc.ao_loc[s0])
    naux = int(tc.ao_loc[k1] - tc.ao_loc[k0])
    out = np.zeros((nao, nao, naux))
    lib().orc_int3c2e(_p(out), *_tab(tc), *(ctypes.c_int(int(v)) for v in (s0, s1, k0, k1)))
    return out


def int2c2e(tc, aux_range):
    """(k|l) over auxiliary shells [k0, k1) of the concatenated tables -> (naux, naux)"""
    k0, k1 = aux_range
    naux = int(tc.ao_loc[k1] - tc.ao_loc[k0])
    out = np.zeros((naux, naux))
    lib().orc_int2c2e(_p(out), *_tab(tc), ctypes.c_int(int(k0)), ctypes.c_int(int(k1)))
    return out


def eval_gto(t, rgrid, deriv=0):
    """deriv 0: (nao, ngrid); 1: (3, nao, ngrid); 2: laplacian (nao, ngrid)"""
    rgrid = np.ascontiguousarray(rgrid, dtype=np.float64)
    ng = rgrid.shape[0]
    shape = (3, t.nao, ng) if deriv == 1 else (t.nao, ng)
    out = np.zeros(shape)
    lib().orc_eval_gto(ctypes.c_int(deriv), _p(out), _p(rgrid), ctypes.c_int(ng), *_tab(t))
    return out


def cart2sph(l):
    nc = (l + 1) * (l + 2) // 2
    out = np.zeros((2 * l + 1, nc))
    lib().orc_cart2sph(ctypes.c_int(l), _p(out))
    return out


def boys(mmax, T):
    out = np.zeros(mmax + 1)
    lib().orc_boys(ctypes.c_int(mmax), ctypes.c_double(T), _p(out))
    return out


def num_threads():
    return lib().orc_num_threads()
