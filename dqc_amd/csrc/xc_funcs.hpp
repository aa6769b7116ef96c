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
    return id == DQC_XC_LDA_X || id == DQC_XC_LDA_C_PW || id == DQC_XC_LDA_C_PW_MOD || id == DQC_XC_LDA_C_VWN;
}
__host__ __device__ inline bool xc_id_is_gga(int id) {
    return id == DQC_XC_GGA_X_PBE || id == DQC_XC_GGA_C_PBE || id == DQC_XC_GGA_X_B88 || id == DQC_XC_GGA_C_LYP || id == DQC_XC_GGA_X_PBE_R ||
           id == DQC_XC_GGA_X_PBE_SOL || id == DQC_XC_GGA_X_RPBE || id == DQC_XC_GGA_C_PBE_SOL;
}
inline bool xc_host_is_lda(int id) { return xc_id_is_lda(id); }
inline bool xc_host_is_gga(int id) { return xc_id_is_gga(id); }

// one LDA / GGA functional of the kernel set at (rho, sigma) with its first derivatives
DQC_DEV Dual f_lda_gga(int id, Dual dr, Dual ds) {
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
DQC_DEV void xc_point(const XcTerms &terms, double r, double sigma, double &e, double &vr, double &vs) {
    e = vr = vs = 0.0;
    if (r > 1e-15) {
        const Dual dr = mk(r, 1.0, 0.0), ds = mk(sigma, 0.0, 1.0);
        for (int t = 0; t < terms.n; t++) {
            const Dual f = f_lda_gga(terms.id[t], dr, ds);
            e += terms.c[t] * f.v;
            vr += terms.c[t] * f.r;
            vs += terms.c[t] * f.s;
        }
    }
}

}  // namespace dqc
