"""Exchange-correlation objects with the reference's BaseXC surface (dqc/xc/base_xc.py:8-195):
`.family`, `get_edensityxc(densinfo)`, `get_vxc(densinfo)`, '+' and scalar '*' combinators, and
`get_xc("lda_x + gga_c_pbe")` (dqc/api/getxc.py:38-59).  Functional evaluation runs in the HIP kernel
dqc_xc_eval (csrc/xc.hip) for trees made of the supported libxc names; any other object exposing the
BaseXC methods (e.g. a user CustomXC, dqc/xc/custom_xc.py:7-25) is called as-is on device tensors by the
Hamiltonian."""
import re

import torch

from . import lib
from .utils.datastruct import ValGrad, SpinParam

_FAMILY = {"lda_x": 1, "lda_c_pw": 1, "lda_c_pw_mod": 1, "lda_c_vwn": 1, "gga_x_pbe": 2, "gga_c_pbe": 2, "gga_x_b88": 2, "gga_c_lyp": 2,
           "gga_x_pbe_r": 2, "gga_x_pbe_sol": 2, "gga_x_rpbe": 2, "gga_c_pbe_sol": 2, "mgga_x_scan": 4, "mgga_c_scan": 4, "mgga_x_tpss": 4, "mgga_c_tpss": 4,
           # round 4: exchange GGAs given by an enhancement factor (one table entry each, csrc/xc_funcs.hpp), PZ81, P86
           "gga_x_pw91": 2, "gga_x_b86": 2, "gga_x_g96": 2, "gga_x_pw86": 2, "gga_x_optx": 2, "gga_x_wc": 2, "lda_c_pz": 1, "gga_c_p86": 2}


class BaseXC:
    @property
    def family(self):
        raise NotImplementedError

    def get_edensityxc(self, densinfo):
        raise NotImplementedError

    def get_vxc(self, densinfo):
        raise NotImplementedError

    def __add__(self, other):
        return LibXC(self.terms + other.terms) if isinstance(self, LibXC) and isinstance(other, LibXC) \
            else _SumXC(self, other)

    def __mul__(self, f):
        if isinstance(self, LibXC):
            return LibXC([(c * float(f), n) for c, n in self.terms])
        return _MulXC(self, f)

    __rmul__ = __mul__

    def getparamnames(self, methodname, prefix=""):
        return []


class LibXC(BaseXC):
    """weighted sum of libxc functionals evaluated by the fused HIP kernel (unpolarised)"""

    def __init__(self, terms):
        self.terms = list(terms)
        for _, n in self.terms:
            if n not in _FAMILY:
                raise ValueError("libxc functional %s is not available in dqc_amd (supported: %s)"
                                 % (n, sorted(_FAMILY)))

    @property
    def family(self):
        return max([1] + [_FAMILY[n] for _, n in self.terms])

    def _flat(self, densinfo):
        rho = densinfo.value
        grad = densinfo.grad
        assert rho.dim() == 1, "batched densities are looped by the caller"
        return rho.contiguous(), (None if (grad is None or self.family == 1) else grad.contiguous())

    def _pol_mgga(self, densinfo, want_e, want_v):
        """spin-polarised meta-GGA (libxc_wrapper.py polarised family-4 branch).  Exchange functionals obey the spin-scaling
        relation E_x[rho_u, rho_d] = 1/2 E_x[2 rho_u] + 1/2 E_x[2 rho_d], so mgga_x_* terms run through the unpolarised
        kernel on (2 rho_s, 2 grad rho_s, 2 tau_s); d e / d rho_s, d e / d grad rho_s, d e / d tau_s are then exactly the
        kernel's outputs at the scaled arguments.  mgga_c_* terms run through the polarised correlation kernel
        (dqc_xc_eval_mgga_pol2), LDA / GGA terms through the polarised LDA / GGA kernel.  Returns
        (edens, [ValGrad_u, ValGrad_d])"""
        mx = [(c, n) for c, n in self.terms if n.startswith("mgga_x_")]
        mc = [(c, n) for c, n in self.terms if n.startswith("mgga_c_")]
        rest = [(c, n) for c, n in self.terms if _FAMILY[n] != 4]
        e, pots = 0.0, []
        for d in (densinfo.u, densinfo.d):
            if mx:
                es, v, vg, vt = lib.xc_eval_mgga(mx, (2.0 * d.value).contiguous(), (2.0 * d.grad).contiguous(),
                                                 (2.0 * d.kin).contiguous(), want_e=want_e, want_v=want_v)
                if want_e:
                    e = e + 0.5 * es
            else:
                v = vg = vt = None
            z = torch.zeros_like
            pots.append(ValGrad(value=v if v is not None else z(d.value), grad=vg if vg is not None else z(d.grad), lapl=None,
                                kin=vt if vt is not None else z(d.value)) if want_v else None)
        if mc:  # correlation: the general polarised form (depends on rho_u, rho_d, |grad rho|^2, tau_u + tau_d)
            u, d = densinfo.u, densinfo.d
            ec, (vu, vd), vgc, vtc = lib.xc_eval_mgga_pol2(mc, u.value.contiguous(), d.value.contiguous(), u.grad.contiguous(),
                                                           d.grad.contiguous(), u.kin.contiguous(), d.kin.contiguous(),
                                                           want_e=want_e, want_v=want_v)
            if want_e:
                e = e + ec
            if want_v:  # (one gradient potential per spin: TPSS correlation sees sigma_uu, sigma_ud, sigma_dd separately)
                for p_, v_, g_ in zip(pots, (vu, vd), vgc):
                    p_.value = p_.value + v_
                    p_.grad = p_.grad + g_
                    p_.kin = p_.kin + vtc
        if rest:
            gga = max(_FAMILY[n] for _, n in rest) == 2
            gu = densinfo.u.grad.contiguous() if gga else None
            gd = densinfo.d.grad.contiguous() if gga else None
            er, (vu, vd), (ggu, ggd) = lib.xc_eval_pol(rest, densinfo.u.value.contiguous(), densinfo.d.value.contiguous(),
                                                       gu, gd, want_e=want_e, want_v=want_v)
            if want_e:
                e = e + er
            if want_v:
                for p_, v_, g_ in zip(pots, (vu, vd), (ggu, ggd)):
                    p_.value = p_.value + v_
                    if g_ is not None:
                        p_.grad = p_.grad + g_
        return e, pots

    def get_edensityxc(self, densinfo):
        if isinstance(densinfo, SpinParam) and self.family == 4:
            return self._pol_mgga(densinfo, True, False)[0]
        if isinstance(densinfo, SpinParam):  # polarised (libxc.py:66-85 polarised branch)
            (ru, gu), (rd, gd) = self._flat(densinfo.u), self._flat(densinfo.d)
            if not self.terms:
                return torch.zeros_like(ru)
            return lib.xc_eval_pol(self.terms, ru, rd, gu, gd, want_e=True, want_v=False)[0]
        if self.family == 4:  # meta-GGA: rho, grad, tau (ValGrad.kin); libxc.py:124-186 MGGA branch
            return lib.xc_eval_mgga(self.terms, densinfo.value.contiguous(), densinfo.grad.contiguous(),
                                    densinfo.kin.contiguous(), want_e=True, want_v=False)[0]
        rho, grad = self._flat(densinfo)
        if not self.terms:
            return torch.zeros_like(rho)
        e, _, _ = lib.xc_eval(self.terms, rho, grad, want_e=True, want_v=False)
        return e

    def get_vxc(self, densinfo):
        if isinstance(densinfo, SpinParam) and self.family == 4:
            pu, pd = self._pol_mgga(densinfo, False, True)[1]
            return SpinParam(u=pu, d=pd)
        if isinstance(densinfo, SpinParam):  # polarised (libxc.py:40-63 polarised branch)
            (ru, gu), (rd, gd) = self._flat(densinfo.u), self._flat(densinfo.d)
            if not self.terms:
                z = lambda r, g: ValGrad(value=torch.zeros_like(r), grad=None if g is None else torch.zeros_like(g))  # noqa: E731
                return SpinParam(u=z(ru, gu), d=z(rd, gd))
            _, (vu, vd), (ggu, ggd) = lib.xc_eval_pol(self.terms, ru, rd, gu, gd, want_e=False, want_v=True)
            return SpinParam(u=ValGrad(value=vu, grad=ggu), d=ValGrad(value=vd, grad=ggd))
        if self.family == 4:
            _, v, vg, vt = lib.xc_eval_mgga(self.terms, densinfo.value.contiguous(), densinfo.grad.contiguous(),
                                            densinfo.kin.contiguous(), want_e=False, want_v=True)
            # no functional of the kernel set depends on the laplacian (libxc returns vlapl = 0 for SCAN): lapl=None
            return ValGrad(value=v, grad=vg, lapl=None, kin=vt)
        rho, grad = self._flat(densinfo)
        if not self.terms:
            return ValGrad(value=torch.zeros_like(rho), grad=None if grad is None else torch.zeros_like(grad))
        _, v, vg = lib.xc_eval(self.terms, rho, grad, want_e=False, want_v=True)
        return ValGrad(value=v, grad=vg)


    def get_vxc_and_exc(self, densinfo, w):
        """(potentials, E_xc = sum_g w_g e_g as a (1,) device tensor) from ONE pass over the grid, for the unpolarised LDA / GGA
        kernel set; (potentials, None) otherwise.  What get_vxc + get_edensityxc (libxc.py:40-85) give in two passes."""
        if isinstance(densinfo, SpinParam) or self.family == 4 or not self.terms:
            return self.get_vxc(densinfo), None
        rho, grad = self._flat(densinfo)
        exc, v, vg = lib.xc_eval_quad(self.terms, rho, grad, w)
        return ValGrad(value=v, grad=vg), exc


class _SumXC(BaseXC):
    def __init__(self, a, b):
        self.a, self.b = a, b

    @property
    def family(self):
        return max(self.a.family, self.b.family)

    def get_edensityxc(self, densinfo):
        return self.a.get_edensityxc(densinfo) + self.b.get_edensityxc(densinfo)

    def get_vxc(self, densinfo):
        # every ValGrad field (value, grad, lapl, kin), restricted or SpinParam (base_xc.py:147-160 AddBaseXC)
        va, vb = self.a.get_vxc(densinfo), self.b.get_vxc(densinfo)
        return SpinParam.apply_fcn(lambda x, y: x + y, va, vb)


class _MulXC(BaseXC):
    def __init__(self, a, f):
        self.a, self.f = a, f

    @property
    def family(self):
        return self.a.family

    def get_edensityxc(self, densinfo):
        return self.a.get_edensityxc(densinfo) * self.f

    def get_vxc(self, densinfo):
        v = self.a.get_vxc(densinfo)  # base_xc.py:175-186 MulBaseXC
        return SpinParam.apply_fcn(lambda x: x * self.f, v)


def get_libxc(name):
    return LibXC([(1.0, name)])


def get_xc(xcstr):
    """"lda_x + gga_c_pbe", "0.7*lda_x" ... ; None -> zero functional (reference ks.py:59-60 accepts None)"""
    if xcstr is None:
        return LibXC([])
    if isinstance(xcstr, BaseXC) or hasattr(xcstr, "get_vxc"):
        return xcstr
    terms = []
    for tok in xcstr.replace(" ", "").split("+"):
        m = re.fullmatch(r"(?:([0-9.eE+-]+)\*)?([a-zA-Z0-9_]+)", tok)
        if m is None:
            raise ValueError("cannot parse xc term: %s" % tok)
        terms.append((float(m.group(1)) if m.group(1) else 1.0, m.group(2).lower()))
    return LibXC(terms)
