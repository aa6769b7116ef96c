// xc_funcs.hpp -- unpolarised LDA / GGA functionals with forward-mode first derivatives (dual numbers in (rho, sigma)),
// shared by the stand-alone XC kernel (xc.hip) and the fused grid kernel (grid.hip).  Functional forms and the references
// they are pinned to: see the header of xc.hip.
#pragma once
#include "common.hpp"

namespace dqc {

struct Dual {  // value, d/drho, d/dsigma
    double v, r, s;
};
DQC_DEV Dual mk(double v, double r = 0.0, double s = 0.0) { return Dual{v, r, s}; }
DQC_DEV Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.r + b.r, a.s + b.s}; }
DQC_DEV Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.r - b.r, a.s - b.s}; }
DQC_DEV Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.r * b.v + a.v * b.r, a.s * b.v + a.v * b.s}; }
DQC_DEV Dual operator/(Dual a, Dual b) {
    double q = a.v / b.v, ib = 1.0 / b.v;
    return {q, (a.r - q * b.r) * ib, (a.s - q * b.s) * ib};
}
DQC_DEV Dual operator+(double a, Dual b) { return {a + b.v, b.r, b.s}; }
DQC_DEV Dual operator+(Dual a, double b) { return {a.v + b, a.r, a.s}; }
DQC_DEV Dual operator-(double a, Dual b) { return {a - b.v, -b.r, -b.s}; }
DQC_DEV Dual operator-(Dual a, double b) { return {a.v - b, a.r, a.s}; }
DQC_DEV Dual operator*(double a, Dual b) { return {a * b.v, a * b.r, a * b.s}; }
DQC_DEV Dual operator*(Dual a, double b) { return {a.v * b, a.r * b, a.s * b}; }
DQC_DEV Dual operator/(Dual a, double b) { double ib = 1.0 / b; return {a.v * ib, a.r * ib, a.s * ib}; }
DQC_DEV Dual operator/(double a, Dual b) { return mk(a) / b; }
DQC_DEV Dual dlog1p(Dual a) { double d = 1.0 / (1.0 + a.v); return {log1p(a.v), a.r * d, a.s * d}; }
DQC_DEV Dual dexpm1(Dual a) { double e = exp(a.v); return {expm1(a.v), a.r * e, a.s * e}; }
DQC_DEV Dual dsqrt(Dual a) { double q = sqrt(a.v), d = 0.5 / q; return {q, a.r * d, a.s * d}; }
DQC_DEV Dual dcbrt(Dual a) { double q = cbrt(a.v), d = q / (3.0 * a.v); return {q, a.r * d, a.s * d}; }

DQC_DEV Dual dlog(Dual a) { double d = 1.0 / a.v; return {log(a.v), a.r * d, a.s * d}; }
DQC_DEV Dual datan(Dual a) { double d = 1.0 / (1.0 + a.v * a.v); return {atan(a.v), a.r * d, a.s * d}; }
DQC_DEV Dual dexp(Dual a) { double e = exp(a.v); return {e, a.r * e, a.s * e}; }
// g(y) = x asinh(x), x = sqrt(y): smooth in y = x^2 (the reduced gradient enters B88 only through it), g'(y) = (asinh(x)/x + 1/sqrt(1+y))/2
DQC_DEV double xasinhx_val(double y, double &dg) {
    const double x = sqrt(y);
    const double ax = x > 1e-4 ? asinh(x) / x : 1.0 - y / 6.0 + 3.0 * y * y / 40.0;
    dg = 0.5 * (ax + 1.0 / sqrt(1.0 + y));
    return y * ax;
}
DQC_DEV Dual dxasinhx(Dual y) { double dg; const double v = xasinhx_val(y.v, dg); return {v, y.r * dg, y.s * dg}; }

constexpr double kPi = 3.14159265358979323846;
constexpr double kPbeKappa_ = 0.8040, kPbeMu_ = 0.2195149727645171;

// number-type-generic spellings (Dual here, the five-variable D5 of the spin-polarised kernel in xc.hip) for the functionals that
// are written once as templates: the exchange functionals given by an enhancement factor, PZ81 and P86
DQC_DEV Dual dpow(Dual a, double e) { const double f = pow(a.v, e), d = e * f / a.v; return {f, a.r * d, a.s * d}; }
DQC_DEV Dual n_exp(Dual a) { return dexp(a); }
DQC_DEV Dual n_log(Dual a) { return dlog(a); }
DQC_DEV Dual n_log1p(Dual a) { return dlog1p(a); }
DQC_DEV Dual n_sqrt(Dual a) { return dsqrt(a); }
DQC_DEV Dual n_cbrt(Dual a) { return dcbrt(a); }
DQC_DEV Dual n_pow(Dual a, double e) { return dpow(a, e); }
DQC_DEV Dual n_xasinhx(Dual a) { return dxasinhx(a); }
DQC_DEV Dual n_floor(Dual a, double lo) { return a.v < lo ? Dual{lo, 0.0, 0.0} : a; }

// ---------------------------------------------------------------------------------------------
// Exchange GGAs as a TABLE of enhancement factors (round 4): e_x = -(3/4)(3/pi)^(1/3) rho^(4/3) F(s^2), s = |grad rho| / (2 k_F rho);
// spin-polarised by the exact scaling E_x[rho_u, rho_d] = (E_x[2 rho_u] + E_x[2 rho_d]) / 2.  Adding one is one case below (+ its id
// in include/dqc_amd.h and a name in dqc_amd/xc.py).  x = s / X2S is the spin-channel reduced gradient |grad rho_s| / rho_s^(4/3) the
// original papers are written in.  Constants as libxc parametrises them (from memory of its published sources -- there is no
// libxc here: see DESIGN.md, pin table).
//   gga_x_pw91  Perdew, Wang (1991/92): F = 1 + [(c + d e^(-alpha s^2)) s^2 - f s^4] / [1 + a s asinh(b s) + f s^4], constants derived
//               from bt = 0.0042, alpha = 100, expo = 4 (a = 6 bt / X2S = 0.19645, b = 1 / X2S = 7.7956, c = 0.2743, -d = 0.1508, f = 0.004)
//   gga_x_b86   Becke, JCP 84, 4524 (1986): F = 1 + (0.0036 / C_x) x^2 / (1 + 0.004 x^2)
//   gga_x_g96   Gill, Mol. Phys. 89, 433 (1996): F = 1 + x^(3/2) / (137 C_x)
//   gga_x_pw86  Perdew, Wang, PRB 33, 8800 (1986): F = (1 + 1.296 s^2 + 14 s^4 + 0.2 s^6)^(1/15)
//   gga_x_optx  Handy, Cohen, Mol. Phys. 99, 403 (2001): F = 1.05151 + (1.43169 / C_x) u^2, u = 0.006 x^2 / (1 + 0.006 x^2)
//   gga_x_wc    Wu, Cohen, PRB 73, 235116 (2006): PBE form with x = 10/81 s^2 + (mu - 10/81) s^2 e^(-s^2) + ln(1 + c s^4)
// ---------------------------------------------------------------------------------------------
constexpr double kX2S = 0.1282782438530421943003109254455883701296;   // 1 / (2 (6 pi^2)^(1/3))
constexpr double kXFactorC = 0.9305257363491000250020102180716672510262;  // (3/8) (3/pi)^(1/3) 4^(2/3)
__host__ __device__ inline bool xc_id_is_x_enh(int id) {
    return id == DQC_XC_GGA_X_PW91 || id == DQC_XC_GGA_X_B86 || id == DQC_XC_GGA_X_G96 || id == DQC_XC_GGA_X_PW86 ||
           id == DQC_XC_GGA_X_OPTX || id == DQC_XC_GGA_X_WC;
}
template <class T>
DQC_DEV T x_enhancement(int id, T s2) {
    const double ix2 = 1.0 / (kX2S * kX2S);  // x^2 = s^2 / X2S^2
    switch (id) {
    case DQC_XC_GGA_X_PW91: {
        const double bt = 0.0042, alpha = 100.0, beta = 0.0018903811666999256;  // beta = 5 (36 pi)^(-5/3)
        const double a = 6.0 * bt / kX2S, b = 1.0 / kX2S, c = bt / (kXFactorC * kX2S * kX2S), d = -(bt - beta) / (kXFactorC * kX2S * kX2S),
                     f = 1.0e-6 / (kXFactorC * kX2S * kX2S * kX2S * kX2S);
        T s4 = s2 * s2;
        T sas = (a / b) * n_xasinhx((b * b) * s2);
        return 1.0 + ((c + d * n_exp((-alpha) * s2)) * s2 - f * s4) / (1.0 + sas + f * s4);
    }
    case DQC_XC_GGA_X_B86: {
        T x2 = ix2 * s2;
        return 1.0 + (0.0036 / kXFactorC) * x2 / (1.0 + 0.004 * x2);
    }
    case DQC_XC_GGA_X_G96:
        return 1.0 + (1.0 / (137.0 * kXFactorC)) * n_pow(ix2 * n_floor(s2, 1e-40), 0.75);
    case DQC_XC_GGA_X_PW86:
        return n_pow(1.0 + 1.296 * s2 + 14.0 * (s2 * s2) + 0.2 * (s2 * s2 * s2), 1.0 / 15.0);
    case DQC_XC_GGA_X_OPTX: {
        T gx2 = (0.006 * ix2) * s2;
        T u = gx2 / (1.0 + gx2);
        return 1.05151 + (1.43169 / kXFactorC) * (u * u);
    }
    default: {  // DQC_XC_GGA_X_WC
        const double kappa = kPbeKappa_, mu = kPbeMu_, c = (146.0 / 2025.0) * (4.0 / 9.0) - (73.0 / 405.0) * (2.0 / 3.0) + (mu - 10.0 / 81.0);
        T x = (10.0 / 81.0) * s2 + (mu - 10.0 / 81.0) * (s2 * n_exp((-1.0) * s2)) + n_log1p(c * (s2 * s2));
        return (1.0 + kappa) - kappa / (1.0 + x / kappa);
    }
    }
}
template <class T>
DQC_DEV T gga_x_by_enh(int id, T rho, T sigma) {  // unpolarised energy density; polarised callers use the spin scaling
    const double c2 = 4.0 * 9.5707800006273038;  // 4 (3 pi^2)^(2/3)
    T r43 = rho * n_cbrt(rho);
    T s2 = sigma / (c2 * (r43 * r43));
    return (-0.75 * 0.98474502184269641) * (r43 * x_enhancement(id, s2));
}

// ---------------------------------------------------------------------------------------------
// lda_c_pz (Perdew, Zunger, PRB 23, 5048 (1981), appendix C) and gga_c_p86 (Perdew, PRB 33, 8822 (1986)) on top of it:
//   eps_i(rs) = gamma_i / (1 + beta1_i sqrt(rs) + beta2_i rs)  (rs >= 1),   A_i ln rs + B_i + C_i rs ln rs + D_i rs  (rs < 1),   i = para, ferro
//   eps = eps_P + (eps_F - eps_P) f(zeta),  f = ((1 + zeta)^(4/3) + (1 - zeta)^(4/3) - 2) / (2^(4/3) - 2)
//   P86: e = n eps_PZ + exp(-Phi) C(n) |grad n|^2 / (d n^(4/3)),  Phi = 1.745 f~ (C(inf) / C(n)) |grad n| / n^(7/6),  f~ = 0.11,
//        C(n) = 0.001667 + (0.002568 + a rs + b rs^2) / (1 + g rs + d rs^2 + 10^4 b rs^3),  d(zeta) = 2^(1/3) sqrt(((1+zeta)/2)^(5/3) + ((1-zeta)/2)^(5/3))
// fz, dz: f(zeta) and d(zeta) (0 and 1 for the closed-shell forms)
// ---------------------------------------------------------------------------------------------
template <class T>
DQC_DEV T pz81_channel(T rs, int i) {
    const double gam[2] = {-0.1423, -0.0843}, b1[2] = {1.0529, 1.3981}, b2[2] = {0.3334, 0.2611};
    const double A[2] = {0.0311, 0.01555}, B[2] = {-0.048, -0.0269}, C[2] = {0.0020, 0.0007}, D[2] = {-0.0116, -0.0048};
    if (rs.v >= 1.0) return gam[i] / (1.0 + b1[i] * n_sqrt(rs) + b2[i] * rs);
    T lr = n_log(rs);
    return A[i] * lr + B[i] + C[i] * (rs * lr) + D[i] * rs;
}
template <class T>
DQC_DEV T pz81_eps(T rho, T fz, bool pol) {
    T rs = n_cbrt((3.0 / (4.0 * kPi)) / rho);
    T eP = pz81_channel(rs, 0);
    if (!pol) return eP;
    return eP + (pz81_channel(rs, 1) - eP) * fz;
}
template <class T>
DQC_DEV T p86_gradient_term(T rho, T sigma, T dz, bool pol) {
    const double a = 0.023266, b = 7.389e-6, g = 8.723, d = 0.472, cinf = 0.001667 + 0.002568, ft = 0.11;
    T rs = n_cbrt((3.0 / (4.0 * kPi)) / rho);
    T rs2 = rs * rs;
    T Cn = 0.001667 + (0.002568 + a * rs + b * rs2) / (1.0 + g * rs + d * rs2 + (1.0e4 * b) * (rs2 * rs));
    T sg = n_floor(sigma, 1e-40);
    T r16 = n_pow(rho, 1.0 / 6.0);
    T phi = (1.745 * ft * cinf) * n_sqrt(sg) / (Cn * (rho * r16));
    T H = n_exp((-1.0) * phi) * Cn * sg / (rho * n_cbrt(rho));
    return pol ? H / dz : H;
}

DQC_DEV Dual f_lda_x(Dual rho) {
    const double c = -0.75 * 0.98474502184269641;  // -(3/4) (3/pi)^(1/3)
    Dual r13 = dcbrt(rho);
    return c * (rho * r13);
}

// PW92 correlation energy per particle, unpolarised; a = 0.0310907 (lda_c_pw) or (1-ln2)/pi^2 (pw_mod)
DQC_DEV Dual pw92_eps(Dual rho, double a) {
    const double alpha1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    Dual rs = dcbrt(mk(3.0 / (4.0 * kPi)) / rho);
    Dual sq = dsqrt(rs);
    Dual q1 = (2.0 * a) * (b1 * sq + b2 * rs + b3 * (rs * sq) + b4 * (rs * rs));
    return (-2.0 * a) * (1.0 + alpha1 * rs) * dlog1p(1.0 / q1);
}

DQC_DEV Dual f_lda_c_pw(Dual rho) { return rho * pw92_eps(rho, 0.0310907); }
DQC_DEV Dual f_lda_c_pw_mod(Dual rho) { return rho * pw92_eps(rho, 0.031090690869654895); }

// the PBE exchange family: enhancement factor F(s^2) = 1 + kappa - kappa / (1 + mu s^2 / kappa) with (kappa, mu) =
// (0.804, 0.21951) PBE, (1.245, 0.21951) revPBE, (0.804, 10/81) PBEsol; RPBE: F = 1 + kappa (1 - exp(-mu s^2 / kappa))
constexpr double kPbeKappa = 0.8040, kPbeMu = 0.2195149727645171, kPbeBeta = 0.06672455060314922;
DQC_DEV Dual f_gga_x_pbe(Dual rho, Dual sigma, double kappa = kPbeKappa, double mu = kPbeMu, bool rpbe = false) {
    const double c2 = 4.0 * 9.5707800006273038;  // 4 (3 pi^2)^(2/3)
    Dual r13 = dcbrt(rho);
    Dual r43 = rho * r13;
    Dual s2 = sigma / (c2 * (r43 * r43));
    Dual F = rpbe ? (1.0 + kappa) - kappa * dexp(mk(0.0) - (mu / kappa) * s2) : (1.0 + kappa) - kappa / (1.0 + (mu / kappa) * s2);
    const double c = -0.75 * 0.98474502184269641;
    return c * (r43 * F);
}

// PBE correlation; beta = 0.066725 (PBE) or 0.046 (PBEsol)
DQC_DEV Dual f_gga_c_pbe(Dual rho, Dual sigma, double beta = kPbeBeta) {
    const double gamma = 0.031090690869654895;  // (1 - ln 2)/pi^2
    Dual eps = pw92_eps(rho, gamma);
    Dual kf = dcbrt((3.0 * kPi * kPi) * rho);
    Dual ks2 = (4.0 / kPi) * kf;
    Dual t2 = sigma / (4.0 * (ks2 * (rho * rho)));
    Dual A = mk(beta / gamma) / dexpm1(mk(0.0) - eps / gamma);
    Dual At2 = A * t2;
    Dual X = (beta / gamma) * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2);
    Dual H = gamma * dlog1p(X);
    return rho * (eps + H);
}

// VWN5 correlation energy per particle of one spin channel fit (Vosko, Wilk, Nusair, Can. J. Phys. 58, 1200 (1980), eq. 4.4), x = sqrt(rs)
template <class T, class FL, class FA>
DQC_DEV T vwn_fit(T x, double A, double b, double c, double x0, FL flog, FA fatan) {
    const double Q = sqrt(4.0 * c - b * b), X0 = x0 * x0 + b * x0 + c;
    T X = x * x + b * x + c;
    T at = fatan(Q / (2.0 * x + b));
    T xm = x - x0;
    return A * (flog(x * x / X) + (2.0 * b / Q) * at - (b * x0 / X0) * (flog(xm * xm / X) + (2.0 * (b + 2.0 * x0) / Q) * at));
}

// lda_c_vwn (libxc id 7 = VWN5), unpolarised: the paramagnetic fit
DQC_DEV Dual f_lda_c_vwn(Dual rho) {
    Dual x = dsqrt(dcbrt(mk(3.0 / (4.0 * kPi)) / rho));
    return rho * vwn_fit(x, 0.0310907, 3.72744, 12.9352, -0.10498, [](Dual a) { return dlog(a); }, [](Dual a) { return datan(a); });
}

// gga_x_b88 (Becke, PRA 38, 3098 (1988)): e = sum_s -rho_s^(4/3) [Cx + beta x_s^2 / (1 + 6 beta x_s asinh x_s)], x_s = |grad rho_s| / rho_s^(4/3);
// unpolarised: rho_s = rho / 2, sigma_ss = sigma / 4
DQC_DEV Dual f_gga_x_b88(Dual rho, Dual sigma) {
    const double beta = 0.0042, cx = 0.9305257363491;  // (3/2) (3 / (4 pi))^(1/3)
    Dual rs_ = 0.5 * rho;
    Dual r43 = rs_ * dcbrt(rs_);
    Dual y = (0.25 * sigma) / (r43 * r43);  // x_s^2
    return -2.0 * (r43 * (cx + beta * y / (1.0 + (6.0 * beta) * dxasinhx(y))));
}

// gga_c_lyp (Lee, Yang, Parr, PRB 37, 785 (1988) in the gradient-only form of Miehlich et al., CPL 157, 200 (1989)), closed shell:
//   e = -a rho / (1 + d rho^(-1/3)) - a b omega [ C_F rho^(14/3) - rho^2 sigma (1/24 + 7 delta / 72) ]
//   omega = exp(-c rho^(-1/3)) / (1 + d rho^(-1/3)) rho^(-11/3),  delta = c rho^(-1/3) + d rho^(-1/3) / (1 + d rho^(-1/3))
DQC_DEV Dual f_gga_c_lyp(Dual rho, Dual sigma) {
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349, CF = 2.8712340001881915;  // (3/10) (3 pi^2)^(2/3)
    Dual r13 = dcbrt(rho);
    Dual ir13 = 1.0 / r13;
    Dual den = 1.0 + d * ir13;
    Dual delta = c * ir13 + d * ir13 / den;
    Dual r2 = rho * rho;
    Dual r113 = r2 * rho * r13 * r13;                 // rho^(11/3)
    Dual omega = dexp(mk(0.0) - c * ir13) / (den * r113);
    Dual bracket = CF * (r113 * rho) - r2 * sigma * ((1.0 / 24.0) + (7.0 / 72.0) * delta);
    return mk(0.0) - a * (rho / den) - (a * b) * (omega * bracket);
}

__host__ __device__ inline bool xc_id_is_lda(int id) {
    return id == DQC_XC_LDA_X || id == DQC_XC_LDA_C_PW || id == DQC_XC_LDA_C_PW_MOD || id == DQC_XC_LDA_C_VWN || id == DQC_XC_LDA_C_PZ;
}
__host__ __device__ inline bool xc_id_is_gga(int id) {
    return id == DQC_XC_GGA_X_PBE || id == DQC_XC_GGA_C_PBE || id == DQC_XC_GGA_X_B88 || id == DQC_XC_GGA_C_LYP || id == DQC_XC_GGA_X_PBE_R ||
           id == DQC_XC_GGA_X_PBE_SOL || id == DQC_XC_GGA_X_RPBE || id == DQC_XC_GGA_C_PBE_SOL || id == DQC_XC_GGA_C_P86 || xc_id_is_x_enh(id);
}
inline bool xc_host_is_lda(int id) { return xc_id_is_lda(id); }
inline bool xc_host_is_gga(int id) { return xc_id_is_gga(id); }

// the functionals of round 4 (enhancement-factor exchange, PZ81, P86): kernels that may see one are compiled as their own
// instantiation (EXT) -- folded into the one switch they cost the PBE / LDA kernels their registers (xc_kernel 34 -> 80 us on a
// 20-atom grid) although no such term was asked for
__host__ __device__ inline bool xc_id_is_ext(int id) { return xc_id_is_x_enh(id) || id == DQC_XC_LDA_C_PZ || id == DQC_XC_GGA_C_P86; }

// one LDA / GGA functional of the kernel set at (rho, sigma) with its first derivatives
template <bool EXT>
DQC_DEV Dual f_lda_gga(int id, Dual dr, Dual ds) {
    if constexpr (EXT) {
        if (xc_id_is_x_enh(id)) return gga_x_by_enh(id, dr, ds);
        if (id == DQC_XC_LDA_C_PZ) return dr * pz81_eps(dr, dr, false);
        if (id == DQC_XC_GGA_C_P86) return dr * pz81_eps(dr, dr, false) + p86_gradient_term(dr, ds, dr, false);
    }
    switch (id) {
    case DQC_XC_LDA_X: return f_lda_x(dr);
    case DQC_XC_LDA_C_PW: return f_lda_c_pw(dr);
    case DQC_XC_LDA_C_VWN: return f_lda_c_vwn(dr);
    case DQC_XC_LDA_C_PW_MOD: return f_lda_c_pw_mod(dr);
    case DQC_XC_GGA_X_PBE: return f_gga_x_pbe(dr, ds);
    case DQC_XC_GGA_X_PBE_R: return f_gga_x_pbe(dr, ds, 1.245);
    case DQC_XC_GGA_X_PBE_SOL: return f_gga_x_pbe(dr, ds, kPbeKappa, 10.0 / 81.0);
    case DQC_XC_GGA_X_RPBE: return f_gga_x_pbe(dr, ds, kPbeKappa, kPbeMu, true);
    case DQC_XC_GGA_C_PBE_SOL: return f_gga_c_pbe(dr, ds, 0.046);
    case DQC_XC_GGA_X_B88: return f_gga_x_b88(dr, ds);
    case DQC_XC_GGA_C_LYP: return f_gga_c_lyp(dr, ds);
    default: return f_gga_c_pbe(dr, ds);
    }
}

struct XcTerms {
    int n;
    int id[8];
    double c[8];
};

// value and derivatives of sum_t c_t f_t(rho, sigma) at one point (libxc-style density threshold: all zero below 1e-15)
template <bool EXT = false>
DQC_DEV void xc_point(const XcTerms &terms, double r, double sigma, double &e, double &vr, double &vs) {
    e = vr = vs = 0.0;
    if (r > 1e-15) {
        const Dual dr = mk(r, 1.0, 0.0), ds = mk(sigma, 0.0, 1.0);
        for (int t = 0; t < terms.n; t++) {
            const Dual f = f_lda_gga<EXT>(terms.id[t], dr, ds);
            e += terms.c[t] * f.v;
            vr += terms.c[t] * f.r;
            vs += terms.c[t] * f.s;
        }
    }
}

// the same for the two functional pairs the BASELINE configs run, with the functionals fixed at compile time: the generic kernel
// carries the register budget of the widest branch of the switch (240 VGPRs, two waves per SIMD); PAIR 1 = gga_x_pbe + gga_c_pbe,
// PAIR 2 = lda_x + lda_c_pw.  Same functions, same order of the additions: bit-identical to xc_point.
template <int PAIR>
DQC_DEV void xc_point_pair(const XcTerms &terms, double r, double sigma, double &e, double &vr, double &vs) {
    e = vr = vs = 0.0;
    if (r > 1e-15) {
        const Dual dr = mk(r, 1.0, 0.0), ds = mk(sigma, 0.0, 1.0);
        const Dual f0 = PAIR == 1 ? f_gga_x_pbe(dr, ds) : f_lda_x(dr);
        e += terms.c[0] * f0.v;
        vr += terms.c[0] * f0.r;
        vs += terms.c[0] * f0.s;
        const Dual f1 = PAIR == 1 ? f_gga_c_pbe(dr, ds) : f_lda_c_pw(dr);
        e += terms.c[1] * f1.v;
        vr += terms.c[1] * f1.r;
        vs += terms.c[1] * f1.s;
    }
}

}  // namespace dqc
