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

DQC_DEV Dual f_gga_x_pbe(Dual rho, Dual sigma) {
    const double kappa = 0.8040, mu = 0.2195149727645171;
    const double c2 = 4.0 * 9.5707800006273038;  // 4 (3 pi^2)^(2/3)
    Dual r13 = dcbrt(rho);
    Dual r43 = rho * r13;
    Dual s2 = sigma / (c2 * (r43 * r43));
    Dual F = (1.0 + kappa) - kappa / (1.0 + (mu / kappa) * s2);
    const double c = -0.75 * 0.98474502184269641;
    return c * (r43 * F);
}

DQC_DEV Dual f_gga_c_pbe(Dual rho, Dual sigma) {
    const double beta = 0.06672455060314922, gamma = 0.031090690869654895;  // (1 - ln 2)/pi^2
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
            Dual f;
            switch (terms.id[t]) {
            case DQC_XC_LDA_X: f = f_lda_x(dr); break;
            case DQC_XC_LDA_C_PW: f = f_lda_c_pw(dr); break;
            case DQC_XC_GGA_X_PBE: f = f_gga_x_pbe(dr, ds); break;
            default: f = f_gga_c_pbe(dr, ds); break;
            }
            e += terms.c[t] * f.v;
            vr += terms.c[t] * f.r;
            vs += terms.c[t] * f.s;
        }
    }
}

}  // namespace dqc
