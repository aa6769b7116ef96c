"""oracle/basis.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's basis handling:

* Gaussian94 parsing, one CGTOBasis per contraction column
  (dqc/api/loadbasis.py:54-83, `_expand_angmoms` :131-152, name normalisation :115-122)
* CGTOBasis.wfnormalize_  (dqc/utils/datastruct.py:34-61, gaussian_int dqc/utils/misc.py:53-56)
* LibcintWrapper atm/bas/env tables (dqc/hamilton/intor/lcintwrap.py:37-123)
* parse_moldesc (dqc/api/parser.py:8-62), periodic-table symbols

The basis *data* files live in dqc_amd/data/basis (shared data, no code shared).
"""
import os
from math import gamma

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dqc_amd", "data", "basis")

SYMBOLS = ["X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne",
           "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar"]
SYM2Z = {s: i for i, s in enumerate(SYMBOLS)}
_SPDF = {"s": 0, "p": 1, "d": 2, "f": 3, "g": 4, "h": 5, "i": 6}


def normalize_basisname(name):
    # dqc/api/loadbasis.py:115-122
    b = name.lower()
    for a, r in (("+", "p"), ("*", "s"), ("(", "_"), (")", "_"), (",", "_")):
        b = b.replace(a, r)
    return b


def gaussian_int(n, alpha):
    # int_0^inf x^n exp(-alpha x^2) dx   (dqc/utils/misc.py:53-56)
    n1 = (n + 1) * 0.5
    return gamma(n1) / (2 * alpha ** n1)


def wfnormalize(l, alphas, coeffs):
    """dqc/utils/datastruct.py:34-61"""
    alphas = np.asarray(alphas, dtype=np.float64)
    coeffs = np.asarray(coeffs, dtype=np.float64)
    coeffs = coeffs / np.sqrt(gaussian_int(2 * l + 2, 2 * alphas))
    ee = alphas[:, None] + alphas[None, :]
    ee = gaussian_int(2 * l + 2, ee)
    s1 = 1.0 / np.sqrt(np.einsum("a,ab,b", coeffs, ee, coeffs))
    return coeffs * s1


def loadbasis(atomz, name):
    """Returns list of (angmom, alphas, normalized coeffs); dqc/api/loadbasis.py:11-83."""
    fpath = os.path.join(_DATA, normalize_basisname(name), "%02d.gaussian94" % atomz)
    if not os.path.exists(fpath):
        raise RuntimeError("basis %s for Z=%d is not in the fixture set (%s)" % (name, atomz, fpath))
    with open(fpath) as f:
        lines = f.read().split("\n")
    while True:  # skip header
        line = lines.pop(0)
        if line == "" or line.startswith("!"):
            continue
        break
    res = []
    while lines:
        line = lines.pop(0)
        if line.startswith("**"):
            break
        desc = line.split()
        nlines = int(desc[1])
        alphas, coeffsT = [], []
        for _ in range(nlines):
            ac = [float(x.replace("D", "E")) for x in lines.pop(0).split()]
            alphas.append(ac[0])
            coeffsT.append(ac[1:])
        coeffs = list(zip(*coeffsT))
        s = desc[0]
        if len(s) != len(coeffs):
            assert len(coeffs) % len(s) == 0
            s = s * (len(coeffs) // len(s))
        for c, ch in zip(coeffs, s.lower()):
            l = _SPDF[ch]
            res.append((l, np.array(alphas), wfnormalize(l, alphas, c)))
    return res


def parse_moldesc(moldesc):
    """'H 1 0 0; H -1 0 0' (Bohr) or (atomzs, atompos) -> (int array, (natm,3) array)."""
    if isinstance(moldesc, str):
        zs, pos = [], []
        for line in moldesc.split(";"):
            t = line.split()
            if not t:
                continue
            zs.append(SYM2Z[t[0]] if not t[0].lstrip("-").isdigit() else int(t[0]))
            pos.append([float(x) for x in t[1:4]])
        return np.array(zs), np.array(pos, dtype=np.float64)
    zs, pos = moldesc
    zs = [SYM2Z[z] if isinstance(z, str) else z for z in zs]
    return np.array(zs), np.array(pos, dtype=np.float64)


class Tables:
    """libcint-style atm/bas/env (dqc/hamilton/intor/lcintwrap.py:37-123)."""

    def __init__(self, atomzs, atompos, shells_per_atom):
        ptr = 20
        atm, bas, env = [], [], [0.0] * ptr
        for ia, (z, pos, shells) in enumerate(zip(atomzs, atompos, shells_per_atom)):
            atm.append([int(z), ptr, 1, ptr + 3, 0, 0])
            env.extend([float(x) for x in pos])
            env.append(0.0)
            ptr += 4
            for (l, alphas, coeffs) in shells:
                ng = len(alphas)
                bas.append([ia, l, ng, 1, 0, ptr, ptr + ng, 0])
                env.extend([float(x) for x in alphas])
                env.extend([float(x) for x in coeffs])
                ptr += 2 * ng
        self.atm = np.array(atm, dtype=np.int32)
        self.bas = np.array(bas, dtype=np.int32)
        self.env = np.array(env, dtype=np.float64)
        self.atomzs = np.array(atomzs)
        self.atompos = np.array(atompos, dtype=np.float64)
        self.nbas = len(bas)
        self.natm = len(atm)
        loc = [0]
        for b in bas:
            loc.append(loc[-1] + 2 * b[1] + 1)
        self.ao_loc = np.array(loc, dtype=np.int32)
        self.nao = loc[-1]


def make_tables(moldesc, basis):
    zs, pos = parse_moldesc(moldesc)
    if isinstance(basis, str):
        shells = [loadbasis(int(z), basis) for z in zs]
    else:  # list (per atom) of lists of (l, alphas, raw coeffs)
        shells = [[(l, np.asarray(a, float), wfnormalize(l, a, c)) for (l, a, c) in ab] for ab in basis]
    return Tables(zs, pos, shells)


# ------------------------------------------------------------------------------------------------
# density fitting: auxiliary basis + concatenated tables
# ------------------------------------------------------------------------------------------------
def even_tempered_aux(atomz, beta=2.5):
    """A reproducible even-tempered auxiliary basis used by the DF parity tests and available in the product as
    auxbasis="etb" (the reference's named JK-fit sets are external data that does not ship with it; DFMol accepts any
    list of CGTOBasis, dqc/system/mol.py:193-198).  Uncontracted shells, exponents a0 * beta^k:
        Z <= 2 :  s x 6 (a0 0.15), p x 3 (0.4), d x 1 (0.9)
        Z  > 2 :  s x 9 (a0 0.15), p x 6 (0.25), d x 4 (0.35), f x 2 (0.6)
    Returns a list of (angmom, alphas, raw coeffs)."""
    if atomz <= 2:
        spec = [(0, 6, 0.15), (1, 3, 0.4), (2, 1, 0.9)]
    else:
        spec = [(0, 9, 0.15), (1, 6, 0.25), (2, 4, 0.35), (3, 2, 0.6)]
    out = []
    for l, n, a0 in spec:
        for k in range(n):
            out.append((l, [a0 * beta ** k], [1.0]))
    return out


def make_tables_df(moldesc, basis, auxbasis="etb"):
    """Concatenated tables in the layout LibcintWrapper.concatenate produces (lcintwrap.py:299-370): the atoms appear
    twice (orbital parent, then auxiliary parent), shells = orbital shells followed by auxiliary shells.
    Returns (tables, (s0, s1), (k0, k1))."""
    zs, pos = parse_moldesc(moldesc)
    if isinstance(basis, str):
        orb = [loadbasis(int(z), basis) for z in zs]
    else:
        orb = [[(l, np.asarray(a, float), wfnormalize(l, a, c)) for (l, a, c) in ab] for ab in basis]
    if isinstance(auxbasis, str):
        if not auxbasis.startswith("etb"):
            raise RuntimeError("auxiliary basis %s is not in the fixture set" % auxbasis)
        beta = float(auxbasis.split(":")[1]) if ":" in auxbasis else 2.5
        auxbasis = [even_tempered_aux(int(z), beta) for z in zs]
    aux = [[(l, np.asarray(a, float), wfnormalize(l, a, c)) for (l, a, c) in ab] for ab in auxbasis]
    n = len(zs)
    t = Tables(list(zs) + list(zs), list(pos) + list(pos), orb + aux)
    nsh_orb = sum(len(x) for x in orb)
    return t, (0, nsh_orb), (nsh_orb, t.nbas)
