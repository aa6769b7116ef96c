"""tools/ref_harness.py -- BUILD-CONTAINER ONLY (needs /root/reference; never travels to the GPU box,
never imported by tests/, bench.py or the product).

Makes the reference's *own* Python layers importable by pre-seeding sys.modules with stubs for the
un-vendored third-party packages, and plugs the oracle's C restatement of the
native arithmetic into the four seams:

  dqc.hamilton.intor.molintor.Intor.calc      <- oracle.natives.int1e / int2e
  dqc.hamilton.intor.gtoeval.gto_evaluator    <- oracle.natives.eval_gto
  pylibxc.LibXCFunctional                     <- oracle.xc (closed forms)
  xitorch.{LinearOperator, linalg, optimize}  <- dense eigh + DIIS fixed point

With that, dqc.Mol / HamiltonCGTO / _HFEngine / _KSEngine / SCF_QCCalc / dqc.grid run unmodified, and
tools/make_golden.py records their outputs as fixtures under tests/golden/.
"""
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
from oracle import natives as _nat  # noqa: E402
from oracle import xc as _oxc  # noqa: E402
from oracle import basis as _obasis  # noqa: E402


# ------------------------------------------------------------------ xitorch stub
class EditableModule:
    def getparamnames(self, methodname, prefix=""):
        return []


class LinearOperator(EditableModule):
    def __init__(self, mat, is_hermitian=False):
        self._mat = mat
        self.is_hermitian = is_hermitian
        self.shape = mat.shape
        self.dtype = mat.dtype
        self.device = mat.device

    @staticmethod
    def m(mat, is_hermitian=None):
        return LinearOperator(mat, bool(is_hermitian))

    def fullmatrix(self):
        return self._mat

    def mm(self, x):
        return self._mat @ x

    def __add__(self, other):
        return LinearOperator(self._mat + other._mat, self.is_hermitian and other.is_hermitian)

    def __sub__(self, other):
        return LinearOperator(self._mat - other._mat, self.is_hermitian and other.is_hermitian)

    def __mul__(self, f):
        return LinearOperator(self._mat * f, self.is_hermitian)

    __rmul__ = __mul__

    @property
    def H(self):
        return LinearOperator(self._mat.transpose(-2, -1).conj(), self.is_hermitian)

    def _getparamnames(self, prefix=""):
        return [prefix + "_mat"]


def _tomat(A):
    return A.fullmatrix() if isinstance(A, LinearOperator) else A


def symeig(A, neig=None, mode="lowest", M=None, **kw):
    a = _tomat(A)
    if M is not None:
        m = _tomat(M)
        s, U = torch.linalg.eigh(m)
        X = U * s ** -0.5
        e, C = torch.linalg.eigh(X.transpose(-2, -1) @ a @ X)
        C = X @ C
    else:
        e, C = torch.linalg.eigh(a)
    if neig is not None:
        if mode == "lowest":
            e, C = e[..., :neig], C[..., :neig]
        else:
            e, C = e[..., -neig:], C[..., -neig:]
    return e, C


def lsymeig(A, neig=None, M=None, **kw):
    return symeig(A, neig, "lowest", M, **kw)


def equilibrium(fcn, y0, params=(), bck_options=None, method=None, **fwd):
    """fixed point y = fcn(y) by Pulay DIIS on the residual (the reference uses Broyden-1,
    dqc/qccalc/scf_qccalc.py:48-53; only the converged point is compared)."""
    y = y0
    ys, rs = [], []
    for it in range(200):
        fy = fcn(y, *params)
        r = fy - y
        if r.abs().max() < 1e-11:
            return fy
        ys.append(fy.reshape(-1))
        rs.append(r.reshape(-1))
        if len(ys) > 10:
            ys.pop(0)
            rs.pop(0)
        n = len(ys)
        B = -torch.ones((n + 1, n + 1), dtype=y.dtype)
        B[n, n] = 0
        for a in range(n):
            for b in range(n):
                B[a, b] = rs[a] @ rs[b]
        rhs = torch.zeros(n + 1, dtype=y.dtype)
        rhs[n] = -1
        try:
            c = torch.linalg.solve(B, rhs)[:n]
        except Exception:
            c = torch.zeros(n, dtype=y.dtype)
            c[-1] = 1
        y = sum(ci * yi for ci, yi in zip(c, ys)).reshape(y0.shape)
    return fy


def _install_stubs():
    xt = types.ModuleType("xitorch")
    xt.EditableModule = EditableModule
    xt.LinearOperator = LinearOperator
    la = types.ModuleType("xitorch.linalg")
    la.symeig, la.lsymeig = symeig, lsymeig
    la.solve = lambda A, B, **kw: torch.linalg.solve(_tomat(A), B)
    op = types.ModuleType("xitorch.optimize")
    op.equilibrium = equilibrium
    op.minimize = None
    gr = types.ModuleType("xitorch.grad")
    gr.hess = gr.jac = None
    xt.linalg, xt.optimize, xt.grad = la, op, gr
    sys.modules.update({"xitorch": xt, "xitorch.linalg": la, "xitorch.optimize": op, "xitorch.grad": gr})

    class _C:
        def CINTcgto_spheric(self, sh, bas):
            import ctypes
            arr = ctypes.cast(bas, ctypes.POINTER(ctypes.c_int))
            return 2 * arr[int(sh.value if hasattr(sh, "value") else sh) * 8 + 1] + 1

        def __getattr__(self, name):
            raise AttributeError("native symbol %s is not available in the harness" % name)

    dl = types.ModuleType("dqclibs")
    dl.CINT = dl.CGTO = dl.CPBC = dl.CSYMM = lambda: _C()
    sys.modules["dqclibs"] = dl
    sys.modules["h5py"] = types.ModuleType("h5py")
    ver = types.ModuleType("dqc._version")
    ver.get_version = lambda: "harness"
    sys.modules["dqc._version"] = ver

    # pylibxc stub on top of the oracle closed forms
    px = types.ModuleType("pylibxc")
    pxf = types.ModuleType("pylibxc.functional")

    class LibXCFunctional:
        def __init__(self, name, spin):
            self.spin = spin  # the reference builds both; only the unpolarised one may be evaluated
            self.name = name
            self._fam = 4 if name in _oxc._FUNCS_MGGA else _oxc._FUNCS[name][0]

        def get_family(self):
            return self._fam

        def compute(self, inp, do_exc=True, do_vxc=False, **kw):
            rho = np.asarray(inp["rho"])
            res = {}
            if self.spin in ("polarized", 2):  # rho (n,2), sigma (n,3) -- libxc's polarised layout
                rho = rho.reshape(-1, 2)
                n = rho.shape[0]
                sg = np.asarray(inp["sigma"]).reshape(-1, 3) if "sigma" in inp else np.zeros((n, 3))
                e, vr, vs = _oxc._FUNCS_POL[self.name](rho[:, 0], rho[:, 1], sg[:, 0], sg[:, 1], sg[:, 2])
                tot = rho.sum(-1)
                if do_exc:
                    res["zk"] = np.where(tot > _oxc.DENS_THRESHOLD, e / np.where(tot > 0, tot, 1), 0.0)[:, None]
                if do_vxc:
                    res["vrho"] = np.stack(vr, axis=-1)
                    if self._fam == 2:
                        res["vsigma"] = np.stack(vs, axis=-1)
                return res
            rho = rho.reshape(-1)
            sigma = np.asarray(inp["sigma"]).reshape(-1) if "sigma" in inp else None
            if self._fam == 4:  # meta-GGA: inputs rho, sigma, lapl, tau -> zk, vrho, vsigma, vlapl, vtau
                e, vr, vs, vt = _oxc._FUNCS_MGGA[self.name](rho, sigma, np.asarray(inp["tau"]).reshape(-1))
                if do_exc:
                    res["zk"] = np.where(rho > _oxc.DENS_THRESHOLD, e / np.where(rho > 0, rho, 1), 0.0)[:, None]
                if do_vxc:
                    res.update({"vrho": vr[:, None], "vsigma": vs[:, None], "vlapl": np.zeros_like(vr)[:, None],
                                "vtau": vt[:, None]})
                return res
            e, vr, vs = _oxc._FUNCS[self.name][1](rho, sigma)
            if do_exc:
                res["zk"] = np.where(rho > _oxc.DENS_THRESHOLD, e / np.where(rho > 0, rho, 1), 0.0)[:, None]
            if do_vxc:
                res["vrho"] = vr[:, None]
                if self._fam == 2:
                    res["vsigma"] = vs[:, None]
            return res

    px.LibXCFunctional = pxf.LibXCFunctional = LibXCFunctional
    px.functional = pxf
    sys.modules["pylibxc"] = px
    sys.modules["pylibxc.functional"] = pxf


_install_stubs()
sys.path.insert(0, "/root/reference")
import dqc  # noqa: E402
import dqc.hamilton.intor.molintor as _molintor  # noqa: E402
import dqc.hamilton.intor.gtoeval as _gtoeval  # noqa: E402
import dqc.hamilton.intor as _intor  # noqa: E402


class _T:  # adapter: LibcintWrapper -> oracle table struct
    def __init__(self, w):
        self.atm, self.bas, self.env = w.atm_bas_env
        self.natm, self.nbas = self.atm.shape[0], self.bas.shape[0]
        self.nao = int(w.nao()) if callable(getattr(w, "nao", None)) else int(w.nao)


def _intor_init(self, int_nmgr, wrappers):
    self.int_nmgr = int_nmgr
    self.wrappers = wrappers
    self.int_type = int_nmgr.int_type


class _TC:  # adapter for concatenated wrappers (LibcintWrapper.concatenate): tables + ao_loc
    def __init__(self, w):
        self.atm, self.bas, self.env = w.atm_bas_env
        self.natm, self.nbas = self.atm.shape[0], self.bas.shape[0]
        loc = [0]
        for b in self.bas:
            loc.append(loc[-1] + 2 * int(b[1]) + 1)
        self.ao_loc = loc


def _calc(self):
    w = self.wrappers[0]
    name = self.int_nmgr.get_intgl_name(w.spherical)
    if name.startswith("int2c2e"):  # coul2c(auxbw): dfmol.py:35
        assert self.wrappers[0].shell_idxs == self.wrappers[1].shell_idxs
        return torch.as_tensor(_nat.int2c2e(_TC(w), w.shell_idxs), dtype=w.dtype)
    if name.startswith("int3c2e"):  # coul3c(basisw, basisw, auxbw): dfmol.py:37-38
        w0, w1, w2 = self.wrappers
        assert w0.shell_idxs == w1.shell_idxs
        return torch.as_tensor(_nat.int3c2e(_TC(w0), w0.shell_idxs, w2.shell_idxs), dtype=w.dtype)
    assert all(ww is w for ww in self.wrappers), "harness: full-range integrals only"
    assert w.shell_idxs == (0, w.atm_bas_env[1].shape[0])
    t = _T(w)
    if name.startswith("int1e_ovlp"):
        out = _nat.int1e("ovlp", t)
    elif name.startswith("int1e_kin"):
        out = _nat.int1e("kin", t)
    elif name.startswith("int1e_nuc"):
        out = _nat.int1e("nuc", t)
    elif name.startswith("int1e_rinv"):
        # fractional-Z nuclear attraction (molintor.py:105-112): 1/|r - R| about env[4:7] (libcint PTR_RINV_ORIG, set by
        # LibcintWrapper.centre_on_r).  The origin is always one of the atoms: = -(nuclear attraction of a unit charge there)
        env = np.asarray(t.env)
        orig = env[4:7]
        hit = [i for i in range(t.natm) if np.allclose(env[int(t.atm[i][1]):int(t.atm[i][1]) + 3], orig, atol=1e-14)]
        assert len(hit) == 1, "harness: rinv origin is not a (unique) atom position"
        zs = np.zeros(t.natm)
        zs[hit[0]] = 1.0
        out = -_nat.int1e("nuc", t, zs)
    elif name.startswith("int2e"):
        out = _nat.int2e(t)
    else:
        raise RuntimeError("harness: integral %s is not provided" % name)
    return torch.as_tensor(out, dtype=w.dtype)


_molintor.Intor.__init__ = _intor_init
_molintor.Intor.calc = _calc


def _gto_evaluator(wrapper, shortname, rgrid, to_transpose):
    t = _T(wrapper)
    deriv = {"": 0, "ip": 1, "lapl": 2}[shortname]
    out = _nat.eval_gto(t, rgrid.detach().numpy(), deriv)
    out = torch.as_tensor(out, dtype=wrapper.dtype)
    if to_transpose:
        out = out.transpose(-2, -1).contiguous()
    return out


_gtoeval.gto_evaluator = _gto_evaluator


def ref_basis(atomz, name):
    """list of reference CGTOBasis objects from the repo's basis fixtures"""
    from dqc.utils.datastruct import CGTOBasis
    return [CGTOBasis(angmom=l, alphas=torch.tensor(a), coeffs=torch.tensor(c), normalized=True)
            for (l, a, c) in _obasis.loadbasis(atomz, name)]


def ref_mol(moldesc, basis, **kw):
    zs, pos = _obasis.parse_moldesc(moldesc)
    bas = [ref_basis(int(z), basis) for z in zs]
    return dqc.Mol((torch.tensor(zs), torch.tensor(pos)), basis=bas, **kw)
