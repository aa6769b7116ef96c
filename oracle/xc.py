"""oracle/xc.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the exchange-correlation functionals the reference obtains
from the un-vendored dependency `pylibxc2>=6.0.0` (libxc; /root/reference/setup.py:58)
through dqc/xc/libxc.py:19-115 and dqc/xc/libxc_wrapper.py:380-413:

  * unpolarised inputs rho (n,), sigma = |grad rho|^2 (n,)   (libxc.py:124-186)
  * deriv=0 returns zk*rho  = energy per unit volume          (libxc_wrapper.py:400-411)
  * deriv=1 returns vrho, vsigma; the potential handed to the Hamiltonian is
    ValGrad(value=vrho, grad=2*vsigma*grad rho)               (libxc.py:188-242)
  * "a + b" / "c * a" combinators                             (dqc/xc/base_xc.py:197-268,
                                                               dqc/api/getxc.py:38-59)

Functional definitions:
  lda_x     : closed form in dqc/test/test_xc.py:390-391, 416-417
  lda_c_pw  : PW92, constants of dqc/test/test_xc.py:393-414 (unpolarised branch, xi = 0)
  gga_x_pbe : closed form of dqc/test/test_xc.py:419-425 with libxc's kappa = 0.8040,
              mu = 0.2195149727645171 (the test's mu = 0.21951 agrees to allclose)
  gga_c_pbe : no formula in the reference; Perdew-Burke-Ernzerhof PRL 77, 3865 (1996)
              with libxc's beta = 0.06672455060314922, gamma = (1-ln2)/pi^2 and the
              "modified" PW92 constants (a = gamma) for its LDA part  -- PARITY UNPINNED
              against an executed libxc.

Densities at or below DENS_THRESHOLD contribute zero (libxc dens_threshold behaviour).
"""
import re

import numpy as np

DENS_THRESHOLD = 1e-15

_PW_ALPHA1 = 0.21370
_PW_BETA = (7.5957, 3.5876, 1.6382, 0.49294)
_PW_A = 0.0310907
_PW_A_MOD = 0.0310906908696548950  # (1 - ln 2)/pi^2, used by lda_c_pw_mod inside gga_c_pbe
_PBE_KAPPA = 0.8040
_PBE_MU = 0.2195149727645171
_PBE_BETA = 0.06672455060314922
_PBE_GAMMA = (1.0 - np.log(2.0)) / np.pi ** 2


def _safe(rho):
    mask = rho > DENS_THRESHOLD
    return mask, np.where(mask, rho, 1.0)


# ---- each functional returns (e, vrho, vsigma) ; e = energy per unit volume ----
def lda_x(rho, sigma=None):
    mask, r = _safe(rho)
    c = (3.0 / np.pi) ** (1.0 / 3)
    e = -0.75 * c * r ** (4.0 / 3)
    v = -c * r ** (1.0 / 3)
    z = np.zeros_like(rho)
    return np.where(mask, e, 0.0), np.where(mask, v, 0.0), z


def _pw92_eps(rs, a):
    """eps_c(rs) (unpolarised) and d eps / d rs"""
    b1, b2, b3, b4 = _PW_BETA
    sq = np.sqrt(rs)
    q0 = -2.0 * a * (1.0 + _PW_ALPHA1 * rs)
    q1 = 2.0 * a * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs)
    q1p = a * (b1 / sq + 2.0 * b2 + 3.0 * b3 * sq + 4.0 * b4 * rs)
    lg = np.log1p(1.0 / q1)
    eps = q0 * lg
    deps = -2.0 * a * _PW_ALPHA1 * lg - q0 * q1p / (q1 * q1 + q1)
    return eps, deps


def lda_c_pw(rho, sigma=None, a=_PW_A):
    mask, r = _safe(rho)
    rs = (3.0 / (4.0 * np.pi * r)) ** (1.0 / 3)
    eps, deps = _pw92_eps(rs, a)
    e = r * eps
    v = eps - rs / 3.0 * deps
    z = np.zeros_like(rho)
    return np.where(mask, e, 0.0), np.where(mask, v, 0.0), z


def gga_x_pbe(rho, sigma, kappa=_PBE_KAPPA, mu=_PBE_MU, rpbe=False):
    """the PBE exchange family (libxc gga_x_pbe.c parameter sets): PBE (kappa 0.804, mu 0.21951), revPBE `gga_x_pbe_r`
    (kappa = 1.245; Zhang, Yang, PRL 80, 890), PBEsol `gga_x_pbe_sol` (mu = 10/81; Perdew et al., PRL 100, 136406) and RPBE
    `gga_x_rpbe` (F = 1 + kappa (1 - exp(-mu s^2 / kappa)); Hammer, Hansen, Norskov, PRB 59, 7413)"""
    mask, r = _safe(rho)
    A = -0.75 * (3.0 / np.pi) ** (1.0 / 3)
    c2 = 4.0 * (3.0 * np.pi ** 2) ** (2.0 / 3)
    r13 = r ** (1.0 / 3)
    r43 = r * r13
    s2 = sigma / (c2 * r43 * r43)
    if rpbe:
        ex_ = np.exp(-mu * s2 / kappa)
        F = 1.0 + kappa * (1.0 - ex_)
        Fp = mu * ex_
    else:
        den = 1.0 + mu * s2 / kappa
        F = 1.0 + kappa - kappa / den
        Fp = mu / (den * den)  # dF/d(s2)
    e = A * r43 * F
    vrho = (4.0 / 3.0) * A * r13 * F + A * r43 * Fp * (-8.0 / 3.0) * s2 / r
    vsigma = A * r43 * Fp / (c2 * r43 * r43)
    return np.where(mask, e, 0.0), np.where(mask, vrho, 0.0), np.where(mask, vsigma, 0.0)


def gga_c_pbe(rho, sigma, beta=_PBE_BETA):
    """PBE correlation; beta = 0.046: PBEsol `gga_c_pbe_sol` (Perdew et al., PRL 100, 136406)"""
    mask, r = _safe(rho)
    g, b = _PBE_GAMMA, beta
    rs = (3.0 / (4.0 * np.pi * r)) ** (1.0 / 3)
    eps, deps = _pw92_eps(rs, _PW_A_MOD)
    deps_drho = deps * (-rs / (3.0 * r))
    kf = (3.0 * np.pi ** 2 * r) ** (1.0 / 3)
    ks2 = 4.0 * kf / np.pi
    # t^2 = sigma / (4 ks^2 rho^2) ;  d t2/d rho = -(7/3) t2 / rho
    t2 = sigma / (4.0 * ks2 * r * r)
    dt2_drho = -(7.0 / 3.0) * t2 / r
    dt2_dsig = 1.0 / (4.0 * ks2 * r * r)
    ex = np.expm1(-eps / g)  # exp(-eps/gamma) - 1
    Ac = (b / g) / ex
    dA_deps = (b / g) * (ex + 1.0) / (g * ex * ex)
    At2 = Ac * t2
    num = 1.0 + At2
    den = 1.0 + At2 + At2 * At2
    X = (b / g) * t2 * num / den
    H = g * np.log1p(X)
    # partials of X wrt t2 and A
    dX_dt2 = (b / g) * (num / den + t2 * (Ac * den - num * (Ac + 2.0 * Ac * At2)) / (den * den))
    dX_dA = (b / g) * t2 * (t2 * den - num * (t2 + 2.0 * At2 * t2)) / (den * den)
    dH_dX = g / (1.0 + X)
    dH_drho = dH_dX * (dX_dt2 * dt2_drho + dX_dA * dA_deps * deps_drho)
    dH_dsig = dH_dX * dX_dt2 * dt2_dsig
    e = r * (eps + H)
    vrho = (eps + H) + r * (deps_drho + dH_drho)
    vsigma = r * dH_dsig
    return np.where(mask, e, 0.0), np.where(mask, vrho, 0.0), np.where(mask, vsigma, 0.0)


_FUNCS = {"lda_x": (1, lda_x), "lda_c_pw": (1, lda_c_pw),
          "gga_x_pbe": (2, gga_x_pbe), "gga_c_pbe": (2, gga_c_pbe)}


class XC:
    """Linear combination of libxc-named functionals; family 1 (LDA) / 2 (GGA)
    (dqc/xc/base_xc.py:13-18)."""

    def __init__(self, terms):
        self.terms = terms  # list of (coef, name)
        self.family = max([1] + [_FUNCS[n][0] for _, n in terms])

    def compute(self, rho, sigma):
        e = np.zeros_like(rho)
        vr = np.zeros_like(rho)
        vs = np.zeros_like(rho)
        for c, n in self.terms:
            ee, vv, ss = _FUNCS[n][1](rho, sigma)
            e += c * ee
            vr += c * vv
            vs += c * ss
        return e, vr, vs

    def get_edensityxc(self, rho, grad=None):
        sigma = None if grad is None else np.einsum("dr,dr->r", grad, grad)
        return self.compute(rho, sigma)[0]

    def get_vxc(self, rho, grad=None):
        """returns (vrho, vgrad) with vgrad = 2*vsigma*grad (libxc.py:239)"""
        sigma = None if grad is None else np.einsum("dr,dr->r", grad, grad)
        _, vr, vs = self.compute(rho, sigma)
        if self.family == 1 or grad is None:
            return vr, None
        return vr, 2.0 * vs[None, :] * grad


def get_xc(xcstr):
    """'lda_x + gga_c_pbe', '0.5*lda_x' ... (dqc/api/getxc.py:38-59); None/'' -> zero functional"""
    terms = []
    if xcstr:
        for tok in xcstr.replace(" ", "").split("+"):
            m = re.fullmatch(r"(?:([0-9.eE+-]+)\*)?([a-z0-9_]+)", tok)
            if m is None or m.group(2) not in _FUNCS:
                raise ValueError("unsupported xc term: %s" % tok)
            terms.append((float(m.group(1)) if m.group(1) else 1.0, m.group(2)))
    return XC(terms)


# =====================================================================================================
# Spin-polarised functionals (reference: polarised branches of dqc/xc/libxc.py:124-242 and
# CalcLDALibXCPol / CalcGGALibXCPol in dqc/xc/libxc_wrapper.py; inputs rho_u, rho_d, sigma_uu, sigma_ud,
# sigma_dd; outputs e = zk*(rho_u+rho_d), vrho (2), vsigma (3)).  Derivatives by forward-mode dual arrays;
# tests check them against finite differences and against the unpolarised functions at rho_u = rho_d.
#   lda_x, gga_x_pbe : exact spin scaling  E[ru, rd] = (E[2 ru] + E[2 rd]) / 2
#   lda_c_pw         : PW92 with the zeta interpolation of dqc/test/test_xc.py:393-414
#   gga_c_pbe        : PBE correlation with phi(zeta) and libxc's modified-PW92 constants -- PARITY UNPINNED
# =====================================================================================================
class Dual:
    """value + gradient w.r.t. the 5 polarised inputs (numpy arrays)"""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v, self.d = v, d

    @staticmethod
    def var(v, i, n=5):
        d = [np.zeros_like(v) for _ in range(n)]
        d[i] = np.ones_like(v)
        return Dual(v, d)

    @staticmethod
    def const(c, like):
        return Dual(np.full_like(like.v, c), [np.zeros_like(like.v) for _ in like.d])

    def _wrap(self, o):
        return o if isinstance(o, Dual) else Dual.const(o, self)

    def __add__(self, o):
        o = self._wrap(o)
        return Dual(self.v + o.v, [a + b for a, b in zip(self.d, o.d)])

    __radd__ = __add__

    def __sub__(self, o):
        o = self._wrap(o)
        return Dual(self.v - o.v, [a - b for a, b in zip(self.d, o.d)])

    def __rsub__(self, o):
        return self._wrap(o) - self

    def __neg__(self):
        return Dual(-self.v, [-a for a in self.d])

    def __mul__(self, o):
        o = self._wrap(o)
        return Dual(self.v * o.v, [a * o.v + self.v * b for a, b in zip(self.d, o.d)])

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._wrap(o)
        q = self.v / o.v
        return Dual(q, [(a - q * b) / o.v for a, b in zip(self.d, o.d)])

    def __rtruediv__(self, o):
        return self._wrap(o) / self

    def fn(self, f, df):
        g = df(self.v)
        return Dual(f(self.v), [a * g for a in self.d])

    def pow(self, p):
        return self.fn(lambda x: x ** p, lambda x: p * x ** (p - 1))

    def log1p(self):
        return self.fn(np.log1p, lambda x: 1.0 / (1.0 + x))

    def expm1(self):
        return self.fn(np.expm1, np.exp)

    def sqrt(self):
        return self.pow(0.5)


_PW_POL = {  # dqc/test/test_xc.py:399-405 : index 0 paramagnetic, 1 ferromagnetic, 2 -alpha_c
    "a": (0.0310907, 0.01554535, 0.0168869), "alpha1": (0.21370, 0.20548, 0.11125),
    "beta1": (7.5957, 14.1189, 10.357), "beta2": (3.5876, 6.1977, 3.6231),
    "beta3": (1.6382, 3.3662, 0.88026), "beta4": (0.49294, 0.62517, 0.49671)}
_PW_A_MOD3 = (0.0310906908696548950, 0.01554534543482745, 0.0168868639403896)
_FZ20 = 1.709920934161365617563962776245


def _pw92_pol_eps(rho, zeta, a3):
    rs = ((3.0 / (4.0 * np.pi)) / rho).pow(1.0 / 3)
    sq = rs.sqrt()
    g = []
    for i in range(3):
        a, p = a3[i], _PW_POL
        q1 = 2.0 * a * (p["beta1"][i] * sq + p["beta2"][i] * rs + p["beta3"][i] * (rs * sq) + p["beta4"][i] * (rs * rs))
        g.append((-2.0 * a) * (1.0 + p["alpha1"][i] * rs) * (1.0 / q1).log1p())
    fz = ((1.0 + zeta).pow(4.0 / 3) + (1.0 - zeta).pow(4.0 / 3) - 2.0) / (2.0 ** (4.0 / 3) - 2.0)
    z4 = (zeta * zeta) * (zeta * zeta)
    return g[0] + z4 * fz * (g[1] - g[0] + g[2] / _FZ20) - fz * g[2] / _FZ20


def _pol_inputs(ru, rd, suu, sud, sdd):
    return [Dual.var(np.asarray(x, dtype=np.float64), i) for i, x in enumerate((ru, rd, suu, sud, sdd))]


def _safe_zeta(ru, rd):
    rho = ru + rd
    z = (ru - rd) / rho
    # keep (1 +- zeta)^(1/3) differentiable for fully polarised points (libxc zeta_threshold behaviour)
    z.v = np.clip(z.v, -1.0 + 1e-10, 1.0 - 1e-10)
    return rho, z


def _finish(e, mask):
    z = lambda a: np.where(mask, a, 0.0)  # noqa: E731
    return z(e.v), (z(e.d[0]), z(e.d[1])), (z(e.d[2]), z(e.d[3]), z(e.d[4]))


def _masked(ru, rd):
    ru, rd = np.asarray(ru, float), np.asarray(rd, float)
    mask = (ru + rd) > DENS_THRESHOLD
    fl = 0.5 * DENS_THRESHOLD
    return mask, np.where(mask, np.maximum(ru, fl), 1.0), np.where(mask, np.maximum(rd, fl), 1.0)


def lda_x_pol(ru, rd, suu=None, sud=None, sdd=None):
    mask, ru_, rd_ = _masked(ru, rd)
    z0 = np.zeros_like(ru_)
    u, d, _, _, _ = _pol_inputs(ru_, rd_, z0, z0, z0)
    c = -0.75 * (3.0 / np.pi) ** (1.0 / 3) * 2.0 ** (1.0 / 3)
    return _finish(c * (u.pow(4.0 / 3) + d.pow(4.0 / 3)), mask)


def lda_c_pw_pol(ru, rd, suu=None, sud=None, sdd=None):
    mask, ru_, rd_ = _masked(ru, rd)
    z0 = np.zeros_like(ru_)
    u, d, _, _, _ = _pol_inputs(ru_, rd_, z0, z0, z0)
    rho, zeta = _safe_zeta(u, d)
    return _finish(rho * _pw92_pol_eps(rho, zeta, _PW_POL["a"]), mask)


def gga_x_pbe_pol(ru, rd, suu, sud, sdd, kappa=_PBE_KAPPA, mu=_PBE_MU, rpbe=False):
    mask, ru_, rd_ = _masked(ru, rd)
    u, d, suu_, sud_, sdd_ = _pol_inputs(ru_, rd_, suu, sud, sdd)
    A = -0.75 * (3.0 / np.pi) ** (1.0 / 3)
    c2 = 4.0 * (3.0 * np.pi ** 2) ** (2.0 / 3)

    def ex(r, s):  # unpolarised E_x[r, s]
        r43 = r.pow(4.0 / 3)
        s2 = s / (c2 * r43 * r43)
        if rpbe:
            F = (1.0 + kappa) - kappa * (((-mu / kappa) * s2).expm1() + 1.0)
        else:
            F = (1.0 + kappa) - kappa / (1.0 + (mu / kappa) * s2)
        return A * r43 * F

    e = 0.5 * (ex(2.0 * u, 4.0 * suu_) + ex(2.0 * d, 4.0 * sdd_))
    return _finish(e, mask)


def gga_c_pbe_pol(ru, rd, suu, sud, sdd, beta=_PBE_BETA):
    mask, ru_, rd_ = _masked(ru, rd)
    u, d, suu_, sud_, sdd_ = _pol_inputs(ru_, rd_, suu, sud, sdd)
    rho, zeta = _safe_zeta(u, d)
    g_, b_ = _PBE_GAMMA, beta
    eps = _pw92_pol_eps(rho, zeta, _PW_A_MOD3)
    phi = 0.5 * ((1.0 + zeta).pow(2.0 / 3) + (1.0 - zeta).pow(2.0 / 3))
    phi3 = phi * phi * phi
    sig = suu_ + 2.0 * sud_ + sdd_
    kf = ((3.0 * np.pi ** 2) * rho).pow(1.0 / 3)
    ks2 = (4.0 / np.pi) * kf
    t2 = sig / (4.0 * phi * phi * ks2 * rho * rho)
    Ac = (b_ / g_) / (-(eps / (g_ * phi3))).expm1()
    At2 = Ac * t2
    X = (b_ / g_) * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2)
    H = g_ * phi3 * X.log1p()
    return _finish(rho * (eps + H), mask)


_FUNCS_POL = {"lda_x": lda_x_pol, "lda_c_pw": lda_c_pw_pol, "gga_x_pbe": gga_x_pbe_pol, "gga_c_pbe": gga_c_pbe_pol}


def compute_pol(xc, ru, rd, gu=None, gd=None):
    """XC (linear combination) on spin densities -> e, (vrho_u, vrho_d), (vgrad_u, vgrad_d) with
    vgrad_u = 2 vsigma_uu grad_u + vsigma_ud grad_d (dqc/xc/libxc.py:205-215); gradients (3, n) or None"""
    n = ru.shape[0]
    if xc.family == 2:
        suu, sud, sdd = (gu * gu).sum(0), (gu * gd).sum(0), (gd * gd).sum(0)
    else:
        suu = sud = sdd = np.zeros(n)
    e = np.zeros(n)
    vr = [np.zeros(n), np.zeros(n)]
    vs = [np.zeros(n), np.zeros(n), np.zeros(n)]
    for c, nm in xc.terms:
        ee, vv, ss = _FUNCS_POL[nm](ru, rd, suu, sud, sdd)
        e += c * ee
        for i in range(2):
            vr[i] += c * vv[i]
        for i in range(3):
            vs[i] += c * ss[i]
    if xc.family == 2:
        vgu = 2.0 * vs[0][None] * gu + vs[1][None] * gd
        vgd = 2.0 * vs[2][None] * gd + vs[1][None] * gu
    else:
        vgu = vgd = None
    return e, (vr[0], vr[1]), (vgu, vgd), vs


# =====================================================================================================
# meta-GGA: SCAN exchange (mgga_x_scan), unpolarised.  Closed form = scan_e_true of
# dqc/test/test_xc.py:427-455 (Sun, Ruzsinszky, Perdew PRL 115, 036402); inputs rho, sigma, tau (the
# functional does not depend on the laplacian: vlapl = 0).  Derivatives by dual arrays over (rho, sigma, tau).
# Reference conventions (dqc/xc/libxc.py:124-242): outputs e = zk*rho, vrho, vsigma, vtau; the potential
# handed to the Hamiltonian is ValGrad(value=vrho, grad=2 vsigma grad rho, lapl=vlapl, kin=vtau).
# =====================================================================================================
def _dexp(x):
    """exp with the argument clipped at 50 (only ever active in the branch np.where discards)"""
    c = Dual(np.minimum(x.v, 50.0), [np.where(x.v < 50.0, a, 0.0) for a in x.d])
    return c.fn(np.exp, np.exp)


def mgga_x_scan(rho, sigma, tau):
    rho, sigma, tau = (np.asarray(a, dtype=np.float64) for a in (rho, sigma, tau))
    mask = rho > DENS_THRESHOLD
    r_ = np.where(mask, rho, 1.0)
    s_ = np.where(mask, np.maximum(sigma, 1e-40), 1.0)
    t_ = np.where(mask, np.maximum(tau, 1e-20), 1.0)
    r, sg, ta = Dual.var(r_, 0, 3), Dual.var(s_, 1, 3), Dual.var(t_, 2, 3)
    a1, c1x, c2x, dx = 4.9479, 0.667, 0.8, 1.24
    mu_ak = 10.0 / 81
    b2 = (5913 / 405000.0) ** 0.5
    b1 = 511 / 13500.0 / (2 * b2)
    b3, k1, h0 = 0.5, 0.065, 1.174
    b4 = mu_ak ** 2 / k1 - 1606 / 18225.0 - b1 ** 2
    kf2 = ((3 * np.pi ** 2) * r).pow(2.0 / 3)              # kf^2
    s2 = sg / (4.0 * r * r * kf2)
    tau_w = sg / (8.0 * r)
    tau_unif = 0.3 * kf2 * r
    alpha = (ta - tau_w) / tau_unif
    oma = 1.0 - alpha
    x = mu_ak * s2 * (1.0 + (b4 / mu_ak) * s2 * _dexp((-abs(b4) / mu_ak) * s2)) + \
        (b1 * s2 + b2 * oma * _dexp((-b3) * oma * oma)).pow(2.0)
    h1 = 1.0 + k1 * (1.0 - k1 / (k1 + x))
    gs = 1.0 - _dexp((-a1) / s2.pow(0.25))                   # sqrt(s) = s2^(1/4)
    # switching function: exp(-c1x a/(1-a)) for a < 1, -dx exp(c2x/(1-a)) for a > 1, 0 at a = 1
    av = alpha.v
    lo = av < 1.0
    safe = np.where(np.abs(1.0 - av) < 1e-12, 0.5, av)       # dummy value where the branch is exactly zero
    al = Dual(safe, alpha.d)
    om = 1.0 - al
    f_lo = _dexp((-c1x) * al / om)
    f_hi = (-dx) * _dexp(c2x / om)
    sel = lambda a, b: np.where(np.abs(1.0 - av) < 1e-12, 0.0, np.where(lo, a, b))  # noqa: E731
    fa = Dual(sel(f_lo.v, f_hi.v), [sel(p, q) for p, q in zip(f_lo.d, f_hi.d)])
    Fx = (h1 + fa * (h0 - h1)) * gs
    ex = (-0.75 * (3.0 / np.pi) ** (1.0 / 3)) * r.pow(4.0 / 3)
    e = ex * Fx
    z = lambda a: np.where(mask, a, 0.0)  # noqa: E731
    return z(e.v), z(e.d[0]), z(e.d[1]), z(e.d[2])


# =====================================================================================================
# SCAN correlation (mgga_c_scan): Sun, Ruzsinszky, Perdew, PRL 115, 036402 (2015), eqs. (9)-(17) and the supplementary
# material, with libxc's constants (b1c, b2c, b3c, chi_infinity, beta(rs), the modified-PW92 LSDA part).  General
# spin-polarised form in the four quantities it depends on -- rho_u, rho_d, sigma = |grad(rho_u + rho_d)|^2,
# tau = tau_u + tau_d -- differentiated with dual arrays.  The reference reaches it through pylibxc
# (dqc/xc/libxc.py:105-115, libxc_wrapper.py:221-378); no closed form or literal exists in the reference:
# PARITY UNPINNED against libxc (as gga_c_pbe); pinned by its exact constraints (uniform gas, one-electron
# limit, spin symmetry) and finite differences in tests/.
# =====================================================================================================
_SCAN_C = dict(b1c=0.0285764, b2c=0.0889, b3c=0.125541, c1c=0.64, c2c=1.5, dc=0.7, chi_inf=0.12802585262625815,
               gcnst=2.3631, gamma=0.031090690869654895)


def _scan_c_energy(u, d, sg, ta):
    """rho * eps_c as a Dual; u, d, sg, ta: Duals of rho_u, rho_d, sigma_total, tau_total"""
    c = _SCAN_C
    rho, zeta = _safe_zeta(u, d)
    rs = ((3.0 / (4.0 * np.pi)) / rho).pow(1.0 / 3)
    kf = ((3.0 * np.pi ** 2) * rho).pow(1.0 / 3)
    s2 = sg / (4.0 * kf * kf * rho * rho)                     # reduced gradient squared (total density)
    phi = 0.5 * ((1.0 + zeta).pow(2.0 / 3) + (1.0 - zeta).pow(2.0 / 3))
    phi3 = phi * phi * phi
    dx = 0.5 * ((1.0 + zeta).pow(4.0 / 3) + (1.0 - zeta).pow(4.0 / 3))
    ds = 0.5 * ((1.0 + zeta).pow(5.0 / 3) + (1.0 - zeta).pow(5.0 / 3))
    # ---- alpha and the switching function
    tau_w = sg / (8.0 * rho)
    tau_unif = 0.3 * (kf * kf) * rho * ds
    alpha = (ta - tau_w) / tau_unif
    av = alpha.v
    near1 = np.abs(1.0 - av) < 1e-12
    al = Dual(np.where(near1, 0.5, av), alpha.d)
    om = 1.0 - al
    f_lo = _dexp((-c["c1c"]) * al / om)
    f_hi = (-c["dc"]) * _dexp(c["c2c"] / om)
    sel = lambda a, b: np.where(near1, 0.0, np.where(av < 1.0, a, b))  # noqa: E731
    fc = Dual(sel(f_lo.v, f_hi.v), [sel(p_, q_) for p_, q_ in zip(f_lo.d, f_hi.d)])
    # ---- eps_c^1: PBE-like with beta(rs) and g(A t^2) = (1 + 4 A t^2)^(-1/4)
    eps_lsda = _pw92_pol_eps(rho, zeta, _PW_A_MOD3)
    beta = 0.066725 * (1.0 + 0.1 * rs) / (1.0 + 0.1778 * rs)
    t2 = sg / (4.0 * phi * phi * ((4.0 / np.pi) * kf) * rho * rho)
    w1 = (-eps_lsda / (c["gamma"] * phi3)).expm1()
    A = beta / (c["gamma"] * w1)
    g = 1.0 / (1.0 + 4.0 * A * t2).pow(0.25)
    eps1 = eps_lsda + c["gamma"] * phi3 * (w1 * (1.0 - g)).log1p()
    # ---- eps_c^0: the alpha = 0 (single-orbital) limit
    eps_lda0 = (-c["b1c"]) / (1.0 + c["b2c"] * rs.sqrt() + c["b3c"] * rs)
    w0 = (-eps_lda0 / c["b1c"]).expm1()
    ginf = 1.0 / (1.0 + 4.0 * c["chi_inf"] * s2).pow(0.25)
    h0 = c["b1c"] * (w0 * (1.0 - ginf)).log1p()
    z2 = zeta * zeta
    z12 = (z2 * z2 * z2) * (z2 * z2 * z2)
    gc = (1.0 - c["gcnst"] * (dx - 1.0)) * (1.0 - z12)
    eps0 = (eps_lda0 + h0) * gc
    return rho * (eps1 + fc * (eps0 - eps1))


def mgga_c_scan_pol(ru, rd, sigma, tau):
    """-> e (= rho eps_c), (d/drho_u, d/drho_d), d/dsigma_total, d/dtau_total.  In libxc's polarised outputs
    vsigma = (d, 2 d, d)/dsigma_total for (uu, ud, dd) and vtau_u = vtau_d = d/dtau_total."""
    mask, ru_, rd_ = _masked(ru, rd)
    sg_ = np.where(mask, np.maximum(np.asarray(sigma, float), 1e-40), 1.0)
    ta_ = np.where(mask, np.maximum(np.asarray(tau, float), 1e-20), 1.0)
    u, d, sg, ta = (Dual.var(x, i, 4) for i, x in enumerate((ru_, rd_, sg_, ta_)))
    e = _scan_c_energy(u, d, sg, ta)
    z = lambda a: np.where(mask, a, 0.0)  # noqa: E731
    return z(e.v), (z(e.d[0]), z(e.d[1])), z(e.d[2]), z(e.d[3])


def mgga_c_scan(rho, sigma, tau):
    """unpolarised: E[rho/2, rho/2]; d/drho = (d_u + d_d)/2 -> e, vrho, vsigma, vtau"""
    rho = np.asarray(rho, float)
    e, (vu, vd), vs, vt = mgga_c_scan_pol(0.5 * rho, 0.5 * rho, sigma, tau)
    return e, 0.5 * (vu + vd), vs, vt


# =====================================================================================================
# TPSS exchange (mgga_x_tpss, libxc id 202): Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003), eqs. (5)-(10).
#   F_x = 1 + kappa - kappa / (1 + x / kappa),  p = s^2,  z = tau_W / tau (<= 1),  alpha = (tau - tau_W) / tau_unif,
#   qb = (9/20)(alpha - 1) / sqrt(1 + b alpha (alpha - 1)) + 2 p / 3,
#   x = { [10/81 + c z^2 / (1 + z^2)^2] p + 146/2025 qb^2 - 73/405 qb sqrt((3z/5)^2 / 2 + p^2 / 2) + (10/81)^2 p^2 / kappa
#         + 2 sqrt(e) (10/81) (3z/5)^2 + e mu p^3 } / (1 + sqrt(e) p)^2,   kappa 0.804, b 0.40, c 1.59096, e 1.537, mu 0.21951
# The reference reaches it through pylibxc and holds no formula or literal: PARITY against libxc UNPINNED.  Pinned by what the
# paper states: F_x = 1 for the uniform gas, and c, e were fixed so that the exchange energy of the exact hydrogen atom is the
# exact -0.3125 Ha (tests/test_oracle_cpu.py).
# =====================================================================================================
_TPSS_X = dict(kappa=0.804, b=0.40, c=1.59096, e=1.537, mu=0.21951)


def mgga_x_tpss(rho, sigma, tau):
    rho, sigma, tau = (np.asarray(a, dtype=np.float64) for a in (rho, sigma, tau))
    mask = rho > DENS_THRESHOLD
    r_ = np.where(mask, rho, 1.0)
    s_ = np.where(mask, np.maximum(sigma, 1e-40), 1.0)
    t_ = np.where(mask, np.maximum(tau, 1e-20), 1.0)
    r, sg, ta = Dual.var(r_, 0, 3), Dual.var(s_, 1, 3), Dual.var(t_, 2, 3)
    k, b, c, e, mu = (_TPSS_X[n] for n in ("kappa", "b", "c", "e", "mu"))
    kf2 = ((3 * np.pi ** 2) * r).pow(2.0 / 3)
    p = sg / (4.0 * r * r * kf2)
    tau_w = sg / (8.0 * r)
    z = tau_w / ta
    over = z.v > 1.0   # tau < tau_W cannot happen for a real density; on a grid it can by round-off: z = 1 there (a constant)
    z = Dual(np.where(over, 1.0, z.v), [np.where(over, 0.0, a) for a in z.d])
    alpha = (ta - tau_w) / (0.3 * kf2 * r)
    alpha = Dual(np.where(over, 0.0, alpha.v), [np.where(over, 0.0, a) for a in alpha.d])
    am1 = alpha - 1.0
    qb = 0.45 * am1 / (1.0 + b * alpha * am1).pow(0.5) + (2.0 / 3.0) * p
    z2 = z * z
    opz = 1.0 + z2
    t35 = 0.36 * z2  # (3 z / 5)^2
    se = np.sqrt(e)
    num = (10.0 / 81 + c * z2 / (opz * opz)) * p + (146.0 / 2025) * qb * qb \
        - (73.0 / 405) * qb * (0.5 * t35 + 0.5 * p * p).pow(0.5) + ((10.0 / 81) ** 2 / k) * p * p \
        + (2.0 * se * 10.0 / 81) * t35 + (e * mu) * p * p * p
    den = 1.0 + se * p
    x = num / (den * den)
    Fx = 1.0 + k - k / (1.0 + x / k)
    ex = (-0.75 * (3.0 / np.pi) ** (1.0 / 3)) * r.pow(4.0 / 3)
    en = ex * Fx
    zz = lambda a: np.where(mask, a, 0.0)  # noqa: E731
    return zz(en.v), zz(en.d[0]), zz(en.d[1]), zz(en.d[2])


# =====================================================================================================
# TPSS correlation (mgga_c_tpss, libxc id 231): the same letter, eqs. (11)-(14):
#   eps_revPKZB = eps_PBE(n_u, n_d, grad n_u, grad n_d) [1 + C(zeta, xi) z^2] - [1 + C(zeta, xi)] z^2 sum_s (n_s / n) epst_s,   z = tau_W / tau
#   epst_s = max[eps_PBE(n_s, 0, grad n_s, 0), eps_PBE(n_u, n_d, ...)],      eps_TPSS = eps_revPKZB [1 + d eps_revPKZB z^3],  d = 2.8
#   C(zeta, xi) = C(zeta, 0) / {1 + xi^2 [(1 + zeta)^(-4/3) + (1 - zeta)^(-4/3)] / 2}^4,  C(zeta, 0) = 0.53 + 0.87 z^2 + 0.50 z^4 + 2.26 z^6,
#   xi = |grad zeta| / (2 (3 pi^2 n)^(1/3)),   grad zeta = [(1 - zeta) grad n_u - (1 + zeta) grad n_d] / n
# Unlike SCAN it depends on sigma_uu, sigma_ud, sigma_dd separately (through |grad zeta| and the one-spin PBE terms): six variables.
# PBE = this file's gga_c_pbe (modified-PW92 LSDA part, as in libxc).  The reference holds no formula or literal: PARITY against
# libxc UNPINNED; pinned by the constraints the construction states -- no correlation for any one-electron density (zeta = 1, z = 1),
# = PBE correlation where z = 0, the uniform-gas limit -- and finite differences (tests/test_oracle_cpu.py).
# =====================================================================================================
def _pbe_c_eps(rho, zeta, sig, ferro=False):
    """PBE correlation energy per particle as a Dual; ferro: the fully polarised limit zeta = 1 in closed form (phi = 2^(-1/3),
    LSDA = PW92's ferromagnetic fit) -- (1 - zeta)^(2/3) is not differentiable there"""
    g_, b_ = _PBE_GAMMA, _PBE_BETA
    if ferro:
        a, p = _PW_A_MOD3[1], _PW_POL
        rs = ((3.0 / (4.0 * np.pi)) / rho).pow(1.0 / 3)
        sq = rs.sqrt()
        q1 = 2.0 * a * (p["beta1"][1] * sq + p["beta2"][1] * rs + p["beta3"][1] * (rs * sq) + p["beta4"][1] * (rs * rs))
        eps = (-2.0 * a) * (1.0 + p["alpha1"][1] * rs) * (1.0 / q1).log1p()
        phi2, phi3 = 2.0 ** (-2.0 / 3), 0.5
    else:
        eps = _pw92_pol_eps(rho, zeta, _PW_A_MOD3)
        phi = 0.5 * ((1.0 + zeta).pow(2.0 / 3) + (1.0 - zeta).pow(2.0 / 3))
        phi2, phi3 = phi * phi, phi * phi * phi
    kf = ((3.0 * np.pi ** 2) * rho).pow(1.0 / 3)
    t2 = sig / (4.0 * phi2 * ((4.0 / np.pi) * kf) * rho * rho)
    Ac = (b_ / g_) / (-(eps / (g_ * phi3))).expm1()
    At2 = Ac * t2
    X = (b_ / g_) * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2)
    return eps + g_ * phi3 * X.log1p()


def _dmax(a, b):
    pick = a.v >= b.v
    return Dual(np.where(pick, a.v, b.v), [np.where(pick, x, y) for x, y in zip(a.d, b.d)])


def mgga_c_tpss_pol(ru, rd, suu, sud, sdd, tau):
    """-> e (= n eps_c), (d/drho_u, d/drho_d), (d/dsigma_uu, d/dsigma_ud, d/dsigma_dd), d/dtau_total"""
    mask, ru_, rd_ = _masked(ru, rd)
    fl = lambda a, lo: np.where(mask, np.maximum(np.asarray(a, float), lo), 1.0)  # noqa: E731
    suu_, sdd_, ta_ = fl(suu, 1e-40), fl(sdd, 1e-40), fl(tau, 1e-20)
    sud_ = np.where(mask, np.asarray(sud, float), 1.0)
    u, d, guu, gud, gdd, ta = (Dual.var(x, i, 6) for i, x in enumerate((ru_, rd_, suu_, sud_, sdd_, ta_)))
    rho, zeta = _safe_zeta(u, d)
    sig = guu + 2.0 * gud + gdd
    sig.v = np.maximum(sig.v, 1e-40)
    eps = _pbe_c_eps(rho, zeta, sig)
    et_u = _dmax(_pbe_c_eps(u, None, guu, ferro=True), eps)
    et_d = _dmax(_pbe_c_eps(d, None, gdd, ferro=True), eps)
    z2_ = zeta * zeta
    c0 = 0.53 + 0.87 * z2_ + 0.50 * z2_ * z2_ + 2.26 * z2_ * z2_ * z2_
    omz, opz = 1.0 - zeta, 1.0 + zeta
    gz2 = (omz * omz * guu - 2.0 * omz * opz * gud + opz * opz * gdd) / (rho * rho)
    xi2 = gz2 / (4.0 * ((3.0 * np.pi ** 2) * rho).pow(2.0 / 3))
    cd = 1.0 + 0.5 * xi2 * (opz.pow(-4.0 / 3) + omz.pow(-4.0 / 3))
    cd2 = cd * cd
    C = c0 / (cd2 * cd2)
    z = sig / (8.0 * rho * ta)
    over = z.v > 1.0
    z = Dual(np.where(over, 1.0, z.v), [np.where(over, 0.0, a) for a in z.d])
    zz = z * z
    erev = eps * (1.0 + C * zz) - (1.0 + C) * zz * (u * et_u + d * et_d) / rho
    e = rho * erev * (1.0 + 2.8 * erev * zz * z)
    w = lambda a: np.where(mask, a, 0.0)  # noqa: E731
    return w(e.v), (w(e.d[0]), w(e.d[1])), (w(e.d[2]), w(e.d[3]), w(e.d[4])), w(e.d[5])


def mgga_c_tpss(rho, sigma, tau):
    """unpolarised: E[rho/2, rho/2; sigma/4 each]; -> e, vrho, vsigma, vtau"""
    rho, sigma = np.asarray(rho, float), np.asarray(sigma, float)
    e, (vu, vd), (a, b, c), vt = mgga_c_tpss_pol(0.5 * rho, 0.5 * rho, 0.25 * sigma, 0.25 * sigma, 0.25 * sigma, tau)
    return e, 0.5 * (vu + vd), 0.25 * (a + b + c), vt


_FUNCS_MGGA = {"mgga_x_scan": mgga_x_scan, "mgga_c_scan": mgga_c_scan, "mgga_x_tpss": mgga_x_tpss, "mgga_c_tpss": mgga_c_tpss}


class XCM(XC):
    """linear combination that may contain meta-GGA terms (family 4)"""

    def __init__(self, terms):
        self.terms = terms
        fam = [4 if n in _FUNCS_MGGA else _FUNCS[n][0] for _, n in terms]
        self.family = max([1] + fam)

    def compute_mgga(self, rho, sigma, tau):
        e, vr, vs, vt = (np.zeros_like(rho) for _ in range(4))
        for c, n in self.terms:
            if n in _FUNCS_MGGA:
                ee, a, b, t = _FUNCS_MGGA[n](rho, sigma, tau)
                vt += c * t
            else:
                ee, a, b = _FUNCS[n][1](rho, sigma)
            e += c * ee
            vr += c * a
            vs += c * b
        return e, vr, vs, vt


_get_xc_plain = get_xc


def get_xc(xcstr):  # noqa: F811  (extends the parser above with the meta-GGA names)
    if xcstr and any(n in xcstr for n in _FUNCS_MGGA):
        terms = []
        for tok in xcstr.replace(" ", "").split("+"):
            m = re.fullmatch(r"(?:([0-9.eE+-]+)\*)?([a-z0-9_]+)", tok)
            if m is None or (m.group(2) not in _FUNCS and m.group(2) not in _FUNCS_MGGA):
                raise ValueError("unsupported xc term: %s" % tok)
            terms.append((float(m.group(1)) if m.group(1) else 1.0, m.group(2)))
        return XCM(terms)
    return _get_xc_plain(xcstr)


# =====================================================================================================
# Round 3: lda_c_vwn (VWN5, libxc id 7), gga_x_b88 (106), gga_c_lyp (131).  The reference reaches them through pylibxc
# (dqc/api/getxc.py:12-36 accepts any libxc name); it holds no formula for them -- PARITY against an executed libxc is
# UNPINNED, like gga_c_pbe.  Pins used instead (tests/test_oracle_cpu.py): VWN5 against PW92 (two fits of the same quantum
# Monte-Carlo data: |difference| < 6e-4 Ha for 0.5 <= rs <= 20, zeta = 0 and 1); B88 on the exact hydrogen-atom density
# (Becke, PRA 38, 3098, table I: 0.3098 Ha); LYP vanishing on every one-spin density; central differences of every derivative.
#   VWN : Vosko, Wilk, Nusair, Can. J. Phys. 58, 1200 (1980), eq. 4.4, parameters of the "VWN5" fits (paramagnetic,
#         ferromagnetic, spin stiffness), interpolation eps_P + alpha_c f (1 - z^4) / f''(0) + (eps_F - eps_P) f z^4
#   B88 : hand-derived derivatives per spin channel, e_s = -rho_s^(4/3) [Cx + beta y / (1 + 6 beta x asinh x)], y = x^2
#   LYP : general spin form of Miehlich, Savin, Stoll, Preuss, CPL 157, 200 (1989) eq. 2 on dual arrays; the unpolarised
#         entry point evaluates it at rho_a = rho_b (the product's closed-shell kernel uses a separately simplified form)
# =====================================================================================================
_VWN = {"A": (0.0310907, 0.01554535, -1.0 / (6.0 * np.pi ** 2)), "b": (3.72744, 7.06042, 1.13107),
        "c": (12.9352, 18.0578, 13.0045), "x0": (-0.10498, -0.32500, -0.0047584)}


def _vwn_fit(x, k):
    """eps(x = sqrt(rs)) and d eps / d x of fit k (0 paramagnetic, 1 ferromagnetic, 2 spin stiffness); plain arrays"""
    A, b, c, x0 = (_VWN[n][k] for n in ("A", "b", "c", "x0"))
    Q = np.sqrt(4.0 * c - b * b)
    X, X0 = x * x + b * x + c, x0 * x0 + b * x0 + c
    at = np.arctan(Q / (2.0 * x + b))
    eps = A * (np.log(x * x / X) + 2.0 * b / Q * at
               - b * x0 / X0 * (np.log((x - x0) ** 2 / X) + 2.0 * (b + 2.0 * x0) / Q * at))
    dat = -2.0 * Q / ((2.0 * x + b) ** 2 + Q * Q)  # d atan(Q / (2x + b)) / dx
    deps = A * (2.0 / x - (2.0 * x + b) / X + 2.0 * b / Q * dat
                - b * x0 / X0 * (2.0 / (x - x0) - (2.0 * x + b) / X + 2.0 * (b + 2.0 * x0) / Q * dat))
    return eps, deps


def lda_c_vwn(rho, sigma=None):
    mask, r = _safe(rho)
    rs = (3.0 / (4.0 * np.pi * r)) ** (1.0 / 3)
    x = np.sqrt(rs)
    eps, deps_dx = _vwn_fit(x, 0)
    # v = eps + rho d eps / d rho,  d rs / d rho = -rs / (3 rho),  d x / d rs = 1 / (2 x)
    v = eps - rs / 3.0 * deps_dx / (2.0 * x)
    z = np.zeros_like(rho)
    return np.where(mask, r * eps, 0.0), np.where(mask, v, 0.0), z


_B88_BETA, _B88_CX = 0.0042, 1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3)


def _b88_spin(rs_, sss):
    """one spin channel: e_s, d e_s / d rho_s, d e_s / d sigma_ss (hand-derived)"""
    r43 = rs_ ** (4.0 / 3)
    y = sss / (r43 * r43)
    x = np.sqrt(y)
    small = x < 1e-4
    xs = np.where(small, 1.0, x)
    ax = np.where(small, 1.0 - y / 6.0, np.arcsinh(xs) / xs)  # asinh(x) / x
    g = y * ax                                                 # x asinh x
    dg = 0.5 * (ax + 1.0 / np.sqrt(1.0 + y))                   # d g / d y
    den = 1.0 + 6.0 * _B88_BETA * g
    h = _B88_BETA * y / den
    dh = _B88_BETA * (den - 6.0 * _B88_BETA * y * dg) / (den * den)
    e = -r43 * (_B88_CX + h)
    de_dr = -(4.0 / 3) * rs_ ** (1.0 / 3) * (_B88_CX + h) - r43 * dh * (-(8.0 / 3) * y / rs_)
    de_ds = -dh / r43
    return e, de_dr, de_ds


def gga_x_b88(rho, sigma):
    mask, r = _safe(rho)
    e, dr, ds = _b88_spin(0.5 * r, 0.25 * sigma)
    return np.where(mask, 2.0 * e, 0.0), np.where(mask, dr, 0.0), np.where(mask, 0.5 * ds, 0.0)


_LYP = (0.04918, 0.132, 0.2533, 0.349)


def _lyp_dual(ra, rb, saa, sab, sbb):
    a, b, c, d = _LYP
    CF = 0.3 * (3.0 * np.pi ** 2) ** (2.0 / 3)
    rho = ra + rb
    ir13 = rho.pow(-1.0 / 3)
    den = 1.0 + d * ir13
    delta = c * ir13 + d * ir13 / den
    omega = (-(c * ir13)).fn(np.exp, np.exp) / den * rho.pow(-11.0 / 3)
    sig = saa + 2.0 * sab + sbb
    br = ra * rb * (2.0 ** (11.0 / 3) * CF * (ra.pow(8.0 / 3) + rb.pow(8.0 / 3))
                    + (47.0 / 18 - 7.0 / 18 * delta) * sig
                    - (2.5 - delta / 18.0) * (saa + sbb)
                    - (delta - 11.0) / 9.0 * (ra / rho * saa + rb / rho * sbb))
    br = br - (2.0 / 3) * rho * rho * sig + ((2.0 / 3) * rho * rho - ra * ra) * sbb + ((2.0 / 3) * rho * rho - rb * rb) * saa
    return -4.0 * a / den * ra * rb / rho - a * b * omega * br


def gga_c_lyp_pol(ru, rd, suu, sud, sdd):
    mask, ru_, rd_ = _masked(ru, rd)
    u, d, suu_, sud_, sdd_ = _pol_inputs(ru_, rd_, suu, sud, sdd)
    return _finish(_lyp_dual(u, d, suu_, sud_, sdd_), mask)


def gga_c_lyp(rho, sigma):
    """closed shell = the spin form at rho_a = rho_b = rho / 2, sigma_aa = sigma_ab = sigma_bb = sigma / 4 (chain rule)"""
    h, q = 0.5 * np.asarray(rho, float), 0.25 * np.asarray(sigma, float)
    e, (vu, vd), (saa, sab, sbb) = gga_c_lyp_pol(h, h, q, q, q)
    return e, 0.5 * (vu + vd), 0.25 * (saa + sab + sbb)


def lda_c_vwn_pol(ru, rd, suu=None, sud=None, sdd=None):
    mask, ru_, rd_ = _masked(ru, rd)
    z0 = np.zeros_like(ru_)
    u, d, _, _, _ = _pol_inputs(ru_, rd_, z0, z0, z0)
    rho, zeta = _safe_zeta(u, d)
    x = ((3.0 / (4.0 * np.pi)) / rho).pow(1.0 / 6)
    fit = [x.fn(lambda t, k=k: _vwn_fit(t, k)[0], lambda t, k=k: _vwn_fit(t, k)[1]) for k in range(3)]
    fz = ((1.0 + zeta).pow(4.0 / 3) + (1.0 - zeta).pow(4.0 / 3) - 2.0) / (2.0 ** (4.0 / 3) - 2.0)
    z4 = (zeta * zeta) * (zeta * zeta)
    eps = fit[0] + fit[2] * fz * (1.0 - z4) / _FZ20 + (fit[1] - fit[0]) * fz * z4
    return _finish(rho * eps, mask)


def gga_x_b88_pol(ru, rd, suu, sud, sdd):
    mask, ru_, rd_ = _masked(ru, rd)
    eu, du, su = _b88_spin(ru_, np.asarray(suu, float))
    ed, dd, sd = _b88_spin(rd_, np.asarray(sdd, float))
    z = lambda a: np.where(mask, a, 0.0)  # noqa: E731
    return z(eu + ed), (z(du), z(dd)), (z(su), np.zeros_like(ru_), z(sd))


_FUNCS.update({"lda_c_vwn": (1, lda_c_vwn), "gga_x_b88": (2, gga_x_b88), "gga_c_lyp": (2, gga_c_lyp)})
_FUNCS_POL.update({"lda_c_vwn": lda_c_vwn_pol, "gga_x_b88": gga_x_b88_pol, "gga_c_lyp": gga_c_lyp_pol})


# ---- parameter variants of the PBE family and PW92 with full-precision constants (libxc ids 102, 116, 117, 133, 13)
def _variant(f, **kw):
    import functools
    return functools.partial(f, **kw)


def lda_c_pw_mod_pol(ru, rd, suu=None, sud=None, sdd=None):
    mask, ru_, rd_ = _masked(ru, rd)
    z0 = np.zeros_like(ru_)
    u, d, _, _, _ = _pol_inputs(ru_, rd_, z0, z0, z0)
    rho, zeta = _safe_zeta(u, d)
    return _finish(rho * _pw92_pol_eps(rho, zeta, _PW_A_MOD3), mask)


_FUNCS.update({"lda_c_pw_mod": (1, _variant(lda_c_pw, a=_PW_A_MOD)),
               "gga_x_pbe_r": (2, _variant(gga_x_pbe, kappa=1.245)), "gga_x_pbe_sol": (2, _variant(gga_x_pbe, mu=10.0 / 81.0)),
               "gga_x_rpbe": (2, _variant(gga_x_pbe, rpbe=True)), "gga_c_pbe_sol": (2, _variant(gga_c_pbe, beta=0.046))})
_FUNCS_POL.update({"lda_c_pw_mod": lda_c_pw_mod_pol,
                   "gga_x_pbe_r": _variant(gga_x_pbe_pol, kappa=1.245), "gga_x_pbe_sol": _variant(gga_x_pbe_pol, mu=10.0 / 81.0),
                   "gga_x_rpbe": _variant(gga_x_pbe_pol, rpbe=True), "gga_c_pbe_sol": _variant(gga_c_pbe_pol, beta=0.046)})


# =====================================================================================================
# Round 4: exchange GGAs given by an enhancement factor (libxc ids 103 gga_x_b86, 107 gga_x_g96, 108 gga_x_pw86, 109 gga_x_pw91,
# 110 gga_x_optx, 118 gga_x_wc), lda_c_pz (9) and gga_c_p86 (132).  The reference reaches them through pylibxc (getxc.py:12-36) and
# holds no formula: PARITY against an executed libxc is UNPINNED.  Written here per SPIN CHANNEL in the reduced gradient
# x_s = |grad rho_s| / rho_s^(4/3) of the original papers, y = x_s^2, with hand-derived F'(y) (the product evaluates one template in
# s^2 of the total density with dual numbers and spin-scales it):
#     e_s = -C_x rho_s^(4/3) F(y),   C_x = (3/2)(3/(4 pi))^(1/3);   s^2 = X2S^2 y,  X2S = 1 / (2 (6 pi^2)^(1/3))
#   B86   Becke, JCP 84, 4524 (1986):               F = 1 + (0.0036 / C_x) y / (1 + 0.004 y)
#   G96   Gill, Mol. Phys. 89, 433 (1996):          F = 1 + y^(3/4) / (137 C_x)
#   PW86  Perdew, Wang, PRB 33, 8800 (1986):        F = (1 + 1.296 s^2 + 14 s^4 + 0.2 s^6)^(1/15)
#   PW91  Perdew et al., PRB 46, 6671 (1992):       F = [1 + a s asinh(b s) + (c + d exp(-100 s^2)) s^2] / [1 + a s asinh(b s) + f s^4] with the
#         constants as libxc derives them from bt = 0.0042, alpha = 100, expo = 4 (published: 0.19645, 7.7956, 0.2743, -0.1508, 0.004)
#   OPTX  Handy, Cohen, Mol. Phys. 99, 403 (2001):  F = 1.05151 + (1.43169 / C_x) u^2,  u = 0.006 y / (1 + 0.006 y)
#   WC    Wu, Cohen, PRB 73, 235116 (2006):         F = 1 + kappa - kappa^2 / (kappa + x),  x = 10/81 s^2 + (mu - 10/81) s^2 e^(-s^2) + ln(1 + c s^4)
#   PZ81  Perdew, Zunger, PRB 23, 5048 (1981) app. C;  P86  Perdew, PRB 33, 8822 (1986) -- on dual arrays
# =====================================================================================================
_X2S = 1.0 / (2.0 * (6.0 * np.pi ** 2) ** (1.0 / 3))
_CX = 1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3)
_S2_FLOOR = 1e-40


def _enh_pw91(y):
    bt, alpha, beta = 0.0042, 100.0, 5.0 * (36.0 * np.pi) ** (-5.0 / 3)
    a, b = 6.0 * bt / _X2S, 1.0 / _X2S
    c, d, f = bt / (_CX * _X2S ** 2), -(bt - beta) / (_CX * _X2S ** 2), 1e-6 / (_CX * _X2S ** 4)
    s2 = _X2S ** 2 * y
    s = np.sqrt(s2)
    small = b * s < 1e-4
    ss = np.where(small, 1.0, s)
    ash_s = np.where(small, b * (1.0 - (b * b * s2) / 6.0), np.arcsinh(b * ss) / ss)   # asinh(b s) / s
    sas = a * s2 * ash_s
    dsas = 0.5 * a * (ash_s + b / np.sqrt(1.0 + b * b * s2))                              # d (a s asinh(b s)) / d s^2
    ex = np.exp(-alpha * s2)
    N = (c + d * ex) * s2 - f * s2 * s2
    dN = c + d * ex * (1.0 - alpha * s2) - 2.0 * f * s2
    Dn = 1.0 + sas + f * s2 * s2
    dDn = dsas + 2.0 * f * s2
    return 1.0 + N / Dn, _X2S ** 2 * (dN * Dn - N * dDn) / (Dn * Dn)


def _enh_b86(y):
    be, ga = 0.0036 / _CX, 0.004
    return 1.0 + be * y / (1.0 + ga * y), be / (1.0 + ga * y) ** 2


def _enh_g96(y):
    lo = _S2_FLOOR / _X2S ** 2          # the product floors s^2 at 1e-40 (a constant there: zero derivative)
    fl = y < lo
    ye = np.where(fl, lo, y)
    k = 1.0 / (137.0 * _CX)
    return 1.0 + k * ye ** 0.75, np.where(fl, 0.0, 0.75 * k * ye ** (-0.25))


def _enh_pw86(y):
    s2 = _X2S ** 2 * y
    P = 1.0 + 1.296 * s2 + 14.0 * s2 ** 2 + 0.2 * s2 ** 3
    return P ** (1.0 / 15), _X2S ** 2 * (1.0 / 15) * P ** (-14.0 / 15) * (1.296 + 28.0 * s2 + 0.6 * s2 ** 2)


def _enh_optx(y):
    g = 0.006
    u = g * y / (1.0 + g * y)
    return 1.05151 + (1.43169 / _CX) * u * u, 2.0 * (1.43169 / _CX) * u * g / (1.0 + g * y) ** 2


def _enh_wc(y):
    ka, mu = _PBE_KAPPA, _PBE_MU
    c = (146.0 / 2025.0) * (4.0 / 9.0) - (73.0 / 405.0) * (2.0 / 3.0) + (mu - 10.0 / 81.0)
    s2 = _X2S ** 2 * y
    ex = np.exp(-s2)
    x = (10.0 / 81.0) * s2 + (mu - 10.0 / 81.0) * s2 * ex + np.log1p(c * s2 * s2)
    dx = 10.0 / 81.0 + (mu - 10.0 / 81.0) * ex * (1.0 - s2) + 2.0 * c * s2 / (1.0 + c * s2 * s2)
    return 1.0 + ka - ka * ka / (ka + x), _X2S ** 2 * ka * ka * dx / (ka + x) ** 2


_ENH = {"gga_x_pw91": _enh_pw91, "gga_x_b86": _enh_b86, "gga_x_g96": _enh_g96, "gga_x_pw86": _enh_pw86, "gga_x_optx": _enh_optx,
        "gga_x_wc": _enh_wc}


def _x_spin(enh, rs_, sss):
    """one spin channel of an enhancement-factor exchange functional: e_s, d e_s / d rho_s, d e_s / d sigma_ss"""
    r43 = rs_ ** (4.0 / 3)
    y = sss / (r43 * r43)
    F, dF = enh(y)
    e = -_CX * r43 * F
    de_dr = -(4.0 / 3) * _CX * rs_ ** (1.0 / 3) * F + _CX * r43 * dF * (8.0 / 3) * y / rs_
    de_ds = -_CX * dF / r43
    return e, de_dr, de_ds


def _make_x(name):
    enh = _ENH[name]

    def unpol(rho, sigma):
        mask, r = _safe(rho)
        e, dr, ds = _x_spin(enh, 0.5 * r, 0.25 * np.asarray(sigma, float))
        return np.where(mask, 2.0 * e, 0.0), np.where(mask, dr, 0.0), np.where(mask, 0.5 * ds, 0.0)

    def pol(ru, rd, suu, sud, sdd):
        mask, ru_, rd_ = _masked(ru, rd)
        eu, du, su = _x_spin(enh, ru_, np.asarray(suu, float))
        ed, dd, sd = _x_spin(enh, rd_, np.asarray(sdd, float))
        z = lambda a: np.where(mask, a, 0.0)  # noqa: E731
        return z(eu + ed), (z(du), z(dd)), (z(su), np.zeros_like(ru_), z(sd))

    return unpol, pol


for _n in _ENH:
    _u, _p = _make_x(_n)
    _FUNCS[_n] = (2, _u)
    _FUNCS_POL[_n] = _p

_PZ = {"gam": (-0.1423, -0.0843), "b1": (1.0529, 1.3981), "b2": (0.3334, 0.2611), "A": (0.0311, 0.01555), "B": (-0.048, -0.0269),
       "C": (0.0020, 0.0007), "D": (-0.0116, -0.0048)}


def _sel(cond, a, b):
    """where() on dual arrays"""
    return Dual(np.where(cond, a.v, b.v), [np.where(cond, x, y) for x, y in zip(a.d, b.d)])


def _pz_channel(rs, i):
    hi = _PZ["gam"][i] / (1.0 + _PZ["b1"][i] * rs.sqrt() + _PZ["b2"][i] * rs)
    lr = rs.fn(np.log, lambda x: 1.0 / x)
    lo = _PZ["A"][i] * lr + _PZ["B"][i] + _PZ["C"][i] * (rs * lr) + _PZ["D"][i] * rs
    return _sel(rs.v >= 1.0, hi, lo)


def _pz_p86_dual(u, d, suu, sud, sdd, p86):
    rho, zeta = _safe_zeta(u, d)
    rs = ((3.0 / (4.0 * np.pi)) / rho).pow(1.0 / 3)
    eP, eF = _pz_channel(rs, 0), _pz_channel(rs, 1)
    fz = ((1.0 + zeta).pow(4.0 / 3) + (1.0 - zeta).pow(4.0 / 3) - 2.0) / (2.0 ** (4.0 / 3) - 2.0)
    e = rho * (eP + (eF - eP) * fz)
    if p86:
        a, b, g, dd_, cinf, ft = 0.023266, 7.389e-6, 8.723, 0.472, 0.001667 + 0.002568, 0.11
        rs2 = rs * rs
        Cn = 0.001667 + (0.002568 + a * rs + b * rs2) / (1.0 + g * rs + dd_ * rs2 + 1e4 * b * rs2 * rs)
        sig = suu + 2.0 * sud + sdd
        fl = sig.v < _S2_FLOOR
        sig = _sel(fl, Dual.const(_S2_FLOOR, sig), sig)
        phi = 1.745 * ft * cinf * sig.sqrt() / (Cn * rho.pow(7.0 / 6))
        dz = 2.0 ** (1.0 / 3) * ((0.5 * (1.0 + zeta)).pow(5.0 / 3) + (0.5 * (1.0 - zeta)).pow(5.0 / 3)).sqrt()
        e = e + (-phi).fn(np.exp, np.exp) * Cn * sig / (dz * rho.pow(4.0 / 3))
    return e


def lda_c_pz_pol(ru, rd, suu=None, sud=None, sdd=None):
    mask, ru_, rd_ = _masked(ru, rd)
    z0 = np.zeros_like(ru_)
    u, d, a, b, c = _pol_inputs(ru_, rd_, z0, z0, z0)
    return _finish(_pz_p86_dual(u, d, a, b, c, False), mask)


def gga_c_p86_pol(ru, rd, suu, sud, sdd):
    mask, ru_, rd_ = _masked(ru, rd)
    u, d, a, b, c = _pol_inputs(ru_, rd_, suu, sud, sdd)
    return _finish(_pz_p86_dual(u, d, a, b, c, True), mask)


def lda_c_pz(rho, sigma=None):
    h = 0.5 * np.asarray(rho, float)
    e, (vu, vd), _ = lda_c_pz_pol(h, h)
    return e, 0.5 * (vu + vd), np.zeros_like(h)


def gga_c_p86(rho, sigma):
    """closed shell = the spin form at rho_a = rho_b (chain rule), like gga_c_lyp above"""
    h, q = 0.5 * np.asarray(rho, float), 0.25 * np.asarray(sigma, float)
    e, (vu, vd), (saa, sab, sbb) = gga_c_p86_pol(h, h, q, q, q)
    return e, 0.5 * (vu + vd), 0.25 * (saa + sab + sbb)


_FUNCS.update({"lda_c_pz": (1, lda_c_pz), "gga_c_p86": (2, gga_c_p86)})
_FUNCS_POL.update({"lda_c_pz": lda_c_pz_pol, "gga_c_p86": gga_c_p86_pol})
