// xc.hip -- exchange-correlation functionals on the grid (unpolarised), value + first derivatives.
// Replaces pylibxc.LibXCFunctional(...).compute as used by dqc/xc/libxc.py:40-85 and
// dqc/xc/libxc_wrapper.py:380-413; conventions: edens = zk*rho (energy per unit volume,
// libxc_wrapper.py:400-411), vrho, and vgrad = 2*vsigma*grad rho (libxc.py:239).
//
// Functional forms: lda_x and PW92 as pinned by dqc/test/test_xc.py:390-417; gga_x_pbe as
// test_xc.py:419-425 with libxc's kappa/mu; gga_c_pbe from Perdew-Burke-Ernzerhof (PRL 77, 3865)
// with libxc's beta, gamma and its modified-PW92 LDA part.  Derivatives are obtained by
// forward-mode differentiation (dual numbers in (rho, sigma)) of the energy expression -- a
// different route from the hand-derived formulas in the oracle, so the two check each other.
#include "common.hpp"
#include "xc_funcs.hpp"

namespace dqc {

template <bool EXT, int PAIR = 0>
__global__ __launch_bounds__(256) void xc_kernel(double *__restrict__ edens, double *__restrict__ vrho, double *__restrict__ vgrad,
                          const double *__restrict__ rho, const double *__restrict__ grho, int n, XcTerms terms,
                          int gga, const double *__restrict__ w, double *__restrict__ exc) {
    double equad = 0.0;  // this lane's share of the quadrature E_xc = sum_i w_i e_i (hcgto.py:320-328), when asked for
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double r = rho[i];
        double gx = 0, gy = 0, gz = 0;
        if (gga) { gx = grho[i]; gy = grho[(size_t)n + i]; gz = grho[2 * (size_t)n + i]; }
        double e, vr, vs;
        if constexpr (PAIR != 0) xc_point_pair<PAIR>(terms, r, gx * gx + gy * gy + gz * gz, e, vr, vs);
        else xc_point<EXT>(terms, r, gx * gx + gy * gy + gz * gz, e, vr, vs);
        if (edens) edens[i] = e;
        if (exc) equad += w[i] * e;
        if (vrho) vrho[i] = vr;
        if (vgrad && gga) {
            vgrad[i] = 2.0 * vs * gx;
            vgrad[(size_t)n + i] = 2.0 * vs * gy;
            vgrad[2 * (size_t)n + i] = 2.0 * vs * gz;
        }
    }
    if (exc) {  // wavefront reduction of the grid quadrature; one partial per block, summed in order by xc_quad_sum_kernel
        __shared__ double part[4];
        for (int o = 32; o > 0; o >>= 1) equad += __shfl_xor(equad, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = equad;
        __syncthreads();
        if (threadIdx.x == 0) exc[1 + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

// exc[0] = sum of the per-block partials exc[1 .. nb] in a fixed order (no atomics: the quadrature is reproducible bit for bit)
__global__ __launch_bounds__(64) void xc_quad_sum_kernel(double *__restrict__ exc, int nb) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += exc[1 + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) exc[0] = s;
}


// ---------------------------------------------------------------------------------------------
// spin-polarised functionals (polarised branches of dqc/xc/libxc.py:124-242): five inputs
// (rho_u, rho_d, sigma_uu, sigma_ud, sigma_dd), forward-mode duals with five derivative slots.
// lda_x / gga_x_pbe by exact spin scaling; PW92 with the zeta interpolation (constants of
// dqc/test/test_xc.py:399-414); PBE correlation with phi(zeta) and the modified-PW92 LDA part.
// ---------------------------------------------------------------------------------------------
struct D5 {
    double v, d[5];
};
DQC_DEV D5 c5(double v) { D5 r; r.v = v; for (int i = 0; i < 5; i++) r.d[i] = 0.0; return r; }
DQC_DEV D5 var5(double v, int k) { D5 r = c5(v); r.d[k] = 1.0; return r; }
DQC_DEV D5 operator+(D5 a, D5 b) { D5 r; r.v = a.v + b.v; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
DQC_DEV D5 operator-(D5 a, D5 b) { D5 r; r.v = a.v - b.v; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
DQC_DEV D5 operator*(D5 a, D5 b) { D5 r; r.v = a.v * b.v; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
DQC_DEV D5 operator/(D5 a, D5 b) {
    D5 r; const double ib = 1.0 / b.v; r.v = a.v * ib;
    for (int i = 0; i < 5; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
DQC_DEV D5 operator*(double a, D5 b) { D5 r; r.v = a * b.v; for (int i = 0; i < 5; i++) r.d[i] = a * b.d[i]; return r; }
DQC_DEV D5 operator+(double a, D5 b) { b.v += a; return b; }
DQC_DEV D5 operator-(double a, D5 b) { D5 r; r.v = a - b.v; for (int i = 0; i < 5; i++) r.d[i] = -b.d[i]; return r; }
DQC_DEV D5 operator/(double a, D5 b) { return c5(a) / b; }
DQC_DEV D5 chain(D5 a, double f, double df) { D5 r; r.v = f; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] * df; return r; }
DQC_DEV D5 p5(D5 a, double e) { const double f = pow(a.v, e); return chain(a, f, e * f / a.v); }
DQC_DEV D5 cbrt5(D5 a) { const double f = cbrt(a.v); return chain(a, f, f / (3.0 * a.v)); }
DQC_DEV D5 sqrt5(D5 a) { const double f = sqrt(a.v); return chain(a, f, 0.5 / f); }
DQC_DEV D5 log1p5(D5 a) { return chain(a, log1p(a.v), 1.0 / (1.0 + a.v)); }
DQC_DEV D5 expm15(D5 a) { return chain(a, expm1(a.v), exp(a.v)); }

DQC_DEV D5 log5(D5 a) { return chain(a, log(a.v), 1.0 / a.v); }
DQC_DEV D5 atan5(D5 a) { return chain(a, atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
DQC_DEV D5 exp5(D5 a) { const double e = exp(a.v); return chain(a, e, e); }
DQC_DEV D5 xasinhx5(D5 y) { double dg; const double v = xasinhx_val(y.v, dg); return chain(y, v, dg); }
DQC_DEV D5 n_exp(D5 a) { return exp5(a); }
DQC_DEV D5 n_log(D5 a) { return log5(a); }
DQC_DEV D5 n_log1p(D5 a) { return log1p5(a); }
DQC_DEV D5 n_sqrt(D5 a) { return sqrt5(a); }
DQC_DEV D5 n_cbrt(D5 a) { return cbrt5(a); }
DQC_DEV D5 n_pow(D5 a, double e) { return p5(a, e); }
DQC_DEV D5 n_xasinhx(D5 a) { return xasinhx5(a); }
DQC_DEV D5 n_floor(D5 a, double lo) { return a.v < lo ? c5(lo) : a; }
DQC_DEV D5 operator*(D5 a, double b) { return b * a; }
DQC_DEV D5 operator+(D5 a, double b) { return b + a; }
DQC_DEV D5 operator-(D5 a, double b) { a.v -= b; return a; }
DQC_DEV D5 operator/(D5 a, double b) { return (1.0 / b) * a; }

// lda_c_vwn (VWN5), spin-polarised: eps = eps_P + alpha_c f(zeta) (1 - zeta^4) / f''(0) + (eps_F - eps_P) f(zeta) zeta^4
DQC_DEV D5 vwn_pol_eps(D5 rho, D5 zeta) {
    const double fz20 = 1.709920934161365617563962776245, Aalpha = -1.0 / (6.0 * kPi * kPi);
    D5 x = sqrt5(cbrt5((3.0 / (4.0 * kPi)) / rho));
    auto lg = [](D5 a) { return log5(a); };
    auto at = [](D5 a) { return atan5(a); };
    D5 eP = vwn_fit(x, 0.0310907, 3.72744, 12.9352, -0.10498, lg, at);
    D5 eF = vwn_fit(x, 0.01554535, 7.06042, 18.0578, -0.32500, lg, at);
    D5 aC = vwn_fit(x, Aalpha, 1.13107, 13.0045, -0.0047584, lg, at);
    D5 fz = (p5(1.0 + zeta, 4.0 / 3.0) + p5(1.0 - zeta, 4.0 / 3.0) - c5(2.0)) / c5(0.51984209978974632953);
    D5 z2 = zeta * zeta, z4 = z2 * z2;
    return eP + aC * fz * (c5(1.0) - z4) / c5(fz20) + (eF - eP) * fz * z4;
}

// one spin channel of gga_x_b88: -rho_s^(4/3) [Cx + beta x^2 / (1 + 6 beta x asinh x)]
DQC_DEV D5 b88_spin5(D5 rs_, D5 sss) {
    const double beta = 0.0042, cx = 0.9305257363491;
    D5 r43 = rs_ * cbrt5(rs_);
    D5 y = sss / (r43 * r43);
    return c5(0.0) - r43 * (cx + beta * y / (1.0 + (6.0 * beta) * xasinhx5(y)));
}

// gga_c_lyp, general spin form (Miehlich, Savin, Stoll, Preuss, CPL 157, 200 (1989), eq. 2)
DQC_DEV D5 lyp_pol5(D5 ra, D5 rb, D5 saa, D5 sab, D5 sbb) {
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349, CF = 2.8712340001881915;
    D5 rho = ra + rb;
    D5 ir13 = 1.0 / cbrt5(rho);
    D5 den = 1.0 + d * ir13;
    D5 delta = c * ir13 + d * ir13 / den;
    D5 r2 = rho * rho;
    D5 r113 = r2 * rho / (ir13 * ir13);
    D5 omega = exp5(c5(0.0) - c * ir13) / (den * r113);
    D5 sig = saa + 2.0 * sab + sbb;
    D5 ra83 = ra * ra * cbrt5(ra) * cbrt5(ra), rb83 = rb * rb * cbrt5(rb) * cbrt5(rb);
    D5 t1 = (12.699208415745595 * CF) * (ra83 + rb83);  // 2^(11/3) C_F (ra^(8/3) + rb^(8/3))
    D5 t2 = ((47.0 / 18.0) - (7.0 / 18.0) * delta) * sig;
    D5 t3 = ((5.0 / 2.0) - (1.0 / 18.0) * delta) * (saa + sbb);
    D5 t4 = ((delta - 11.0) / 9.0) * ((ra / rho) * saa + (rb / rho) * sbb);
    D5 br = ra * rb * (t1 + t2 - t3 - t4) - (2.0 / 3.0) * r2 * sig + ((2.0 / 3.0) * r2 - ra * ra) * sbb + ((2.0 / 3.0) * r2 - rb * rb) * saa;
    return c5(0.0) - (4.0 * a) * (ra * rb / (rho * den)) - (a * b) * (omega * br);
}

DQC_DEV D5 pw92_pol_eps(D5 rho, D5 zeta, const double *a3) {
    const double alpha1[3] = {0.21370, 0.20548, 0.11125}, b1[3] = {7.5957, 14.1189, 10.357},
                 b2[3] = {3.5876, 6.1977, 3.6231}, b3[3] = {1.6382, 3.3662, 0.88026}, b4[3] = {0.49294, 0.62517, 0.49671};
    const double fz20 = 1.709920934161365617563962776245;
    D5 rs = cbrt5((3.0 / (4.0 * kPi)) / rho);
    D5 sq = sqrt5(rs);
    D5 g[3];
    for (int i = 0; i < 3; i++) {
        D5 q1 = (2.0 * a3[i]) * (b1[i] * sq + b2[i] * rs + b3[i] * (rs * sq) + b4[i] * (rs * rs));
        g[i] = (-2.0 * a3[i]) * (1.0 + alpha1[i] * rs) * log1p5(1.0 / q1);
    }
    D5 fz = (p5(1.0 + zeta, 4.0 / 3.0) + p5(1.0 - zeta, 4.0 / 3.0) - c5(2.0)) / c5(0.51984209978974632953);  // 2^(4/3)-2
    D5 z2 = zeta * zeta, z4 = z2 * z2;
    return g[0] + z4 * fz * (g[1] - g[0] + g[2] / c5(fz20)) - fz * g[2] / c5(fz20);
}

DQC_DEV D5 pbe_x_unpol5(D5 r, D5 s, double kappa = kPbeKappa, double mu = kPbeMu, bool rpbe = false) {
    const double c2 = 4.0 * 9.5707800006273038;
    D5 r43 = r * cbrt5(r);
    D5 s2 = s / (c2 * (r43 * r43));
    D5 F = rpbe ? (1.0 + kappa) - kappa * exp5(c5(0.0) - (mu / kappa) * s2) : (1.0 + kappa) - kappa / (1.0 + (mu / kappa) * s2);
    return (-0.75 * 0.98474502184269641) * (r43 * F);
}


template <bool EXT, int PAIR = 0>
__global__ __launch_bounds__(256) void xc_pol_kernel(double *__restrict__ edens, double *__restrict__ vru, double *__restrict__ vrd,
                              double *__restrict__ vgu, double *__restrict__ vgd, const double *__restrict__ ru_,
                              const double *__restrict__ rd_, const double *__restrict__ gu_,
                              const double *__restrict__ gd_, int n, XcTerms terms, int gga) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double ru = ru_[i], rd = rd_[i];
        double gu[3] = {0, 0, 0}, gd[3] = {0, 0, 0};
        if (gga)
            for (int d = 0; d < 3; d++) { gu[d] = gu_[(size_t)d * n + i]; gd[d] = gd_[(size_t)d * n + i]; }
        double e = 0, dv[5] = {0, 0, 0, 0, 0};
        if (ru + rd > 1e-15) {
            ru = fmax(ru, 0.5e-15);
            rd = fmax(rd, 0.5e-15);
            const D5 u = var5(ru, 0), d = var5(rd, 1);
            const D5 suu = var5(gu[0] * gu[0] + gu[1] * gu[1] + gu[2] * gu[2], 2);
            const D5 sud = var5(gu[0] * gd[0] + gu[1] * gd[1] + gu[2] * gd[2], 3);
            const D5 sdd = var5(gd[0] * gd[0] + gd[1] * gd[1] + gd[2] * gd[2], 4);
            const D5 rho = u + d;
            D5 zeta = (u - d) / rho;
            zeta.v = fmin(fmax(zeta.v, -1.0 + 1e-10), 1.0 - 1e-10);
            // one term of the functional at this point; called with a run-time id in the generic kernel and with compile-time ids in
            // the PBE-pair instantiation (PAIR = 1: the switch folds away -- the generic kernel holds 506 VGPRs, one wave per SIMD)
            auto term = [&](const int tid_) -> D5 {
                D5 f;
                bool ext_done = false;
                if constexpr (EXT) {  // the round-4 functionals live in their own instantiation (see xc_funcs.hpp: xc_id_is_ext)
                    const int id_ = tid_;
                    if (xc_id_is_x_enh(id_)) {  // enhancement-factor exchange: exact spin scaling of the unpolarised form
                        f = 0.5 * (gga_x_by_enh(id_, 2.0 * u, 4.0 * suu) + gga_x_by_enh(id_, 2.0 * d, 4.0 * sdd));
                        ext_done = true;
                    } else if (id_ == DQC_XC_LDA_C_PZ || id_ == DQC_XC_GGA_C_P86) {
                        D5 fz = (p5(1.0 + zeta, 4.0 / 3.0) + p5(1.0 - zeta, 4.0 / 3.0) - c5(2.0)) / c5(0.51984209978974632953);
                        f = rho * pz81_eps(rho, fz, true);
                        if (id_ == DQC_XC_GGA_C_P86) {
                            D5 dz = 1.2599210498948732 * sqrt5(p5(0.5 * (1.0 + zeta), 5.0 / 3.0) + p5(0.5 * (1.0 - zeta), 5.0 / 3.0));
                            f = f + p86_gradient_term(rho, suu + 2.0 * sud + sdd, dz, true);
                        }
                        ext_done = true;
                    }
                }
                if (!ext_done)
                switch (tid_) {
                case DQC_XC_LDA_X:
                    f = (-0.75 * 0.98474502184269641 * 1.2599210498948732) * (u * cbrt5(u) + d * cbrt5(d));
                    break;
                case DQC_XC_LDA_C_PW: {
                    const double a3[3] = {0.0310907, 0.01554535, 0.0168869};
                    f = rho * pw92_pol_eps(rho, zeta, a3);
                } break;
                case DQC_XC_LDA_C_PW_MOD: {
                    const double a3[3] = {0.0310906908696548950, 0.01554534543482745, 0.0168868639403896};
                    f = rho * pw92_pol_eps(rho, zeta, a3);
                } break;
                case DQC_XC_GGA_X_PBE: case DQC_XC_GGA_X_PBE_R: case DQC_XC_GGA_X_PBE_SOL: case DQC_XC_GGA_X_RPBE: {
                    // exchange: exact spin scaling of the unpolarised form (kappa, mu by member of the family)
                    const int id_ = tid_;
                    const double ka = id_ == DQC_XC_GGA_X_PBE_R ? 1.245 : kPbeKappa, mu_ = id_ == DQC_XC_GGA_X_PBE_SOL ? 10.0 / 81.0 : kPbeMu;
                    const bool rp = id_ == DQC_XC_GGA_X_RPBE;
                    f = 0.5 * (pbe_x_unpol5(2.0 * u, 4.0 * suu, ka, mu_, rp) + pbe_x_unpol5(2.0 * d, 4.0 * sdd, ka, mu_, rp));
                } break;
                case DQC_XC_LDA_C_VWN: f = rho * vwn_pol_eps(rho, zeta); break;
                case DQC_XC_GGA_X_B88: f = b88_spin5(u, suu) + b88_spin5(d, sdd); break;
                case DQC_XC_GGA_C_LYP: f = lyp_pol5(u, d, suu, sud, sdd); break;
                default: {
                    const double a3[3] = {0.0310906908696548950, 0.01554534543482745, 0.0168868639403896};
                    const double beta = tid_ == DQC_XC_GGA_C_PBE_SOL ? 0.046 : kPbeBeta, gamma = 0.031090690869654895;
                    D5 eps = pw92_pol_eps(rho, zeta, a3);
                    D5 phi = 0.5 * (p5(1.0 + zeta, 2.0 / 3.0) + p5(1.0 - zeta, 2.0 / 3.0));
                    D5 phi3 = phi * phi * phi;
                    D5 sig = suu + 2.0 * sud + sdd;
                    D5 kf = cbrt5((3.0 * kPi * kPi) * rho);
                    D5 t2 = sig / (4.0 * (phi * phi) * ((4.0 / kPi) * kf) * (rho * rho));
                    D5 A = c5(beta / gamma) / expm15(c5(0.0) - eps / (gamma * phi3));
                    D5 At2 = A * t2;
                    D5 X = (beta / gamma) * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2);
                    f = rho * (eps + gamma * phi3 * log1p5(X));
                } break;
                }
                return f;
            };
            if constexpr (PAIR == 1) {
                const D5 f0 = term(DQC_XC_GGA_X_PBE);
                e += terms.c[0] * f0.v;
                for (int k = 0; k < 5; k++) dv[k] += terms.c[0] * f0.d[k];
                const D5 f1 = term(DQC_XC_GGA_C_PBE);
                e += terms.c[1] * f1.v;
                for (int k = 0; k < 5; k++) dv[k] += terms.c[1] * f1.d[k];
            } else {
                for (int t = 0; t < terms.n; t++) {
                    const D5 f = term(terms.id[t]);
                    e += terms.c[t] * f.v;
                    for (int k = 0; k < 5; k++) dv[k] += terms.c[t] * f.d[k];
                }
            }
        }
        if (edens) edens[i] = e;
        if (vru) { vru[i] = dv[0]; vrd[i] = dv[1]; }
        if (vgu && gga)
            for (int k = 0; k < 3; k++) {  // libxc.py:205-215
                vgu[(size_t)k * n + i] = 2.0 * dv[2] * gu[k] + dv[3] * gd[k];
                vgd[(size_t)k * n + i] = 2.0 * dv[4] * gd[k] + dv[3] * gu[k];
            }
    }
}

}  // namespace dqc

extern "C" int dqc_xc_eval_quad(double *d_exc, double *d_edens, double *d_vrho, double *d_vgrad, const double *d_rho,
                                const double *d_grho, const double *d_w, int n, const int *ids, const double *coefs, int nterm,
                                void *stream) {
    // dqc_xc_eval plus the quadrature of the energy density in the same pass: d_exc[0] = sum_i w_i e_i (wavefront reductions,
    // one partial per block in d_exc[1 ..], summed in a fixed order: no atomics).  d_exc: DQC_XC_QUAD_DOUBLES doubles.
    // d_exc / d_w NULL: no quadrature.
    using namespace dqc;
    if (nterm < 0 || nterm > 8) { set_error("dqc_xc_eval: at most 8 functional terms"); return DQC_EINVAL; }
    if (d_exc && !d_w) { set_error("dqc_xc_eval_quad: the quadrature needs the grid weights"); return DQC_EINVAL; }
    XcTerms t;
    t.n = nterm;
    bool need_grad = false;
    for (int i = 0; i < nterm; i++) {
        t.id[i] = ids[i];
        t.c[i] = coefs[i];
        if (xc_host_is_gga(ids[i])) need_grad = true;
        else if (!xc_host_is_lda(ids[i])) { set_error("dqc_xc_eval: unknown functional id"); return DQC_EINVAL; }
    }
    if (need_grad && !d_grho) { set_error("dqc_xc_eval: GGA functional needs the density gradient"); return DQC_EINVAL; }
    if (n <= 0) {
        if (d_exc) DQC_HIP(hipMemsetAsync(d_exc, 0, sizeof(double), (hipStream_t)stream));
        return DQC_OK;
    }
    int blocks = (n + 255) / 256;
    const int cap = d_exc ? DQC_XC_QUAD_DOUBLES - 1 : 4096;
    if (blocks > cap) blocks = cap;
    bool ext = false;
    for (int i = 0; i < nterm; i++) ext = ext || xc_id_is_ext(ids[i]);
    // the two pairs the BASELINE configs run have their own instantiations (xc_point_pair: fewer registers, more waves per SIMD)
    const int pair = (nterm == 2 && ids[0] == DQC_XC_GGA_X_PBE && ids[1] == DQC_XC_GGA_C_PBE && d_grho) ? 1
                   : (nterm == 2 && ids[0] == DQC_XC_LDA_X && ids[1] == DQC_XC_LDA_C_PW) ? 2 : 0;
    static const bool no_pair = getenv("DQC_XC_GENERIC") != nullptr;  // (A/B runs)
    if (pair == 1 && !no_pair) hipLaunchKernelGGL((xc_kernel<false, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad,
                                                  d_rho, d_grho, n, t, 1, d_w, d_exc);
    else if (pair == 2 && !no_pair) hipLaunchKernelGGL((xc_kernel<false, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad,
                                                       d_rho, d_grho, n, t, d_grho ? 1 : 0, d_w, d_exc);
    else if (ext) hipLaunchKernelGGL(xc_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad, d_rho,
                                d_grho, n, t, d_grho ? 1 : 0, d_w, d_exc);
    else hipLaunchKernelGGL(xc_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad, d_rho,
                            d_grho, n, t, d_grho ? 1 : 0, d_w, d_exc);
    DQC_CHECK_LAUNCH();
    if (d_exc) {
        hipLaunchKernelGGL(xc_quad_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_exc, blocks);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}

extern "C" int dqc_xc_eval(double *d_edens, double *d_vrho, double *d_vgrad, const double *d_rho,
                           const double *d_grho, int n, const int *ids, const double *coefs, int nterm,
                           void *stream) {
    return dqc_xc_eval_quad(nullptr, d_edens, d_vrho, d_vgrad, d_rho, d_grho, nullptr, n, ids, coefs, nterm, stream);
}

namespace dqc {

__host__ __device__ inline bool xc_id_is_mgga(int id) {
    return id == DQC_XC_MGGA_X_SCAN || id == DQC_XC_MGGA_C_SCAN || id == DQC_XC_MGGA_X_TPSS || id == DQC_XC_MGGA_C_TPSS;
}

DQC_DEV D5 exp5c(D5 a) {  // exp with the argument clipped at 50 (only active in a discarded branch)
    if (a.v >= 50.0) return c5(5.184705528587072e21);
    const double f = exp(a.v);
    return chain(a, f, f);
}

// SCAN exchange, unpolarised: slots 0 = rho, 1 = sigma, 2 = tau  (closed form of dqc/test/test_xc.py:427-455)
DQC_DEV D5 f_mgga_x_scan(D5 r, D5 sg, D5 ta) {
    const double a1 = 4.9479, c1x = 0.667, c2x = 0.8, dx = 1.24, mu_ak = 10.0 / 81.0;
    const double b2 = 0.12083045973594572, b1 = 0.15663207743548518, b3 = 0.5, k1 = 0.065, h0 = 1.174;
    const double b4 = mu_ak * mu_ak / k1 - 1606.0 / 18225.0 - b1 * b1;
    D5 kf2 = p5((3.0 * kPi * kPi) * r, 2.0 / 3.0);
    D5 s2 = sg / (4.0 * (r * r) * kf2);
    D5 tau_w = sg / (8.0 * r);
    D5 tau_unif = 0.3 * kf2 * r;
    D5 alpha = (ta - tau_w) / tau_unif;
    D5 oma = 1.0 - alpha;
    D5 t1 = b1 * s2 + b2 * oma * exp5c((-b3) * (oma * oma));
    D5 x = mu_ak * s2 * (1.0 + (b4 / mu_ak) * s2 * exp5c((-fabs(b4) / mu_ak) * s2)) + t1 * t1;
    D5 h1 = 1.0 + k1 * (1.0 - k1 / (k1 + x));
    D5 gs = 1.0 - exp5c(c5(-a1) / p5(s2, 0.25));
    D5 fa = c5(0.0);
    if (fabs(oma.v) >= 1e-12) {
        if (alpha.v < 1.0) fa = exp5c((-c1x) * alpha / oma);
        else fa = (-dx) * exp5c(c2x / oma);
    }
    D5 Fx = (h1 + fa * (h0 - h1)) * gs;
    return (-0.75 * 0.98474502184269641) * (r * cbrt5(r)) * Fx;
}

// TPSS exchange (mgga_x_tpss): Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003), eqs. (5)-(10); slots 0 = rho, 1 = sigma,
// 2 = tau.  No formula or literal in the reference (it reaches it through pylibxc): pinned by the uniform-gas limit and by the
// exact exchange energy of the hydrogen atom, -0.3125 Ha, which the paper's constants c, e were fixed to (oracle/xc.py, tests).
DQC_DEV D5 f_mgga_x_tpss(D5 r, D5 sg, D5 ta) {
    const double kappa = 0.804, b = 0.40, c = 1.59096, e = 1.537, mu = 0.21951, se = 1.2397580409095961;  // sqrt(e)
    D5 kf2 = p5((3.0 * kPi * kPi) * r, 2.0 / 3.0);
    D5 p = sg / (4.0 * (r * r) * kf2);
    D5 tau_w = sg / (8.0 * r);
    D5 z = tau_w / ta;
    D5 alpha = (ta - tau_w) / (0.3 * kf2 * r);
    if (z.v > 1.0) {  // tau < tau_W: round-off of a one-orbital region on the grid; z = 1, alpha = 0 as constants
        z = c5(1.0);
        alpha = c5(0.0);
    }
    D5 am1 = alpha - 1.0;
    D5 qb = 0.45 * am1 / sqrt5(1.0 + b * alpha * am1) + (2.0 / 3.0) * p;
    D5 z2 = z * z;
    D5 opz = 1.0 + z2;
    D5 t35 = 0.36 * z2;  // (3 z / 5)^2
    D5 num = (10.0 / 81.0 + c * z2 / (opz * opz)) * p + (146.0 / 2025.0) * (qb * qb)
             - (73.0 / 405.0) * qb * sqrt5(0.5 * t35 + 0.5 * (p * p)) + ((10.0 / 81.0) * (10.0 / 81.0) / kappa) * (p * p)
             + (2.0 * se * 10.0 / 81.0) * t35 + (e * mu) * (p * p * p);
    D5 den = 1.0 + se * p;
    D5 x = num / (den * den);
    D5 Fx = (1.0 + kappa) - kappa / (1.0 + x / kappa);
    return (-0.75 * 0.98474502184269641) * (r * cbrt5(r)) * Fx;
}

// ---------------------------------------------------------------------------------------------
// SCAN correlation (mgga_c_scan): Sun, Ruzsinszky, Perdew, PRL 115, 036402, eqs. (9)-(17) + supplementary material, libxc's
// constants.  eps_c = eps_c^1 + f_c(alpha) (eps_c^0 - eps_c^1):
//   eps_c^1 = eps_LSDA + gamma phi^3 ln[1 + w1 (1 - (1 + 4 A t^2)^(-1/4))],  w1 = exp(-eps_LSDA / (gamma phi^3)) - 1,
//             A = beta(rs) / (gamma w1),  beta(rs) = 0.066725 (1 + 0.1 rs) / (1 + 0.1778 rs),  LSDA = modified PW92
//   eps_c^0 = (eps_LDA0 + b1c ln[1 + w0 (1 - (1 + 4 chi_inf s^2)^(-1/4))]) Gc(zeta),  eps_LDA0 = -b1c / (1 + b2c sqrt(rs) + b3c rs)
//   alpha = (tau - tau_W) / (tau_unif ds(zeta)),  f_c = exp(-c1c alpha / (1 - alpha)) (alpha < 1), -dc exp(c2c / (1 - alpha)) (alpha > 1)
// Two separate codings: the closed zeta = 0 form (slots 0 = rho, 1 = sigma, 2 = tau) and the general spin-polarised form
// (slots 0 = rho_u, 1 = rho_d, 2 = sigma_total, 3 = tau_total); the oracle codes the general form once more in numpy.
// No literal of this functional exists in the reference: parity against libxc is UNPINNED (like gga_c_pbe).
// ---------------------------------------------------------------------------------------------
struct ScanC {
    static constexpr double b1c = 0.0285764, b2c = 0.0889, b3c = 0.125541, c1c = 0.64, c2c = 1.5, dc = 0.7;
    static constexpr double chi_inf = 0.12802585262625815, gcnst = 2.3631, gamma = 0.031090690869654895;
};

DQC_DEV D5 scan_c_switch(D5 alpha) {
    D5 oma = 1.0 - alpha;
    if (fabs(oma.v) < 1e-12) return c5(0.0);
    if (alpha.v < 1.0) return exp5c((-ScanC::c1c) * alpha / oma);
    return (-ScanC::dc) * exp5c(ScanC::c2c / oma);
}

DQC_DEV D5 pw92_mod_unpol5(D5 rho) {  // modified-PW92 paramagnetic branch
    const double a = 0.0310906908696548950, alpha1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    D5 rs = cbrt5((3.0 / (4.0 * kPi)) / rho);
    D5 sq = sqrt5(rs);
    D5 q1 = (2.0 * a) * (b1 * sq + b2 * rs + b3 * (rs * sq) + b4 * (rs * rs));
    return (-2.0 * a) * (1.0 + alpha1 * rs) * log1p5(1.0 / q1);
}

// unpolarised closed form (zeta = 0: phi = dx = ds = Gc = 1)
DQC_DEV D5 f_mgga_c_scan(D5 r, D5 sg, D5 ta) {
    D5 rs = cbrt5((3.0 / (4.0 * kPi)) / r);
    D5 kf = cbrt5((3.0 * kPi * kPi) * r);
    D5 kf2 = kf * kf;
    D5 s2 = sg / (4.0 * kf2 * (r * r));
    D5 alpha = (ta - sg / (8.0 * r)) / (0.3 * kf2 * r);
    D5 fc = scan_c_switch(alpha);
    D5 eps = pw92_mod_unpol5(r);
    D5 beta = 0.066725 * (1.0 + 0.1 * rs) / (1.0 + 0.1778 * rs);
    D5 t2 = sg / (4.0 * ((4.0 / kPi) * kf) * (r * r));
    D5 w1 = expm15(c5(0.0) - eps / ScanC::gamma);
    D5 A = beta / (ScanC::gamma * w1);
    D5 g = 1.0 / p5(1.0 + 4.0 * (A * t2), 0.25);
    D5 eps1 = eps + ScanC::gamma * log1p5(w1 * (1.0 - g));
    D5 e0 = c5(-ScanC::b1c) / (1.0 + ScanC::b2c * sqrt5(rs) + ScanC::b3c * rs);
    D5 w0 = expm15(c5(0.0) - e0 / ScanC::b1c);
    D5 ginf = 1.0 / p5(1.0 + (4.0 * ScanC::chi_inf) * s2, 0.25);
    D5 eps0 = e0 + ScanC::b1c * log1p5(w0 * (1.0 - ginf));
    return r * (eps1 + fc * (eps0 - eps1));
}

// general spin-polarised form
DQC_DEV D5 f_mgga_c_scan_pol(D5 u, D5 d, D5 sg, D5 ta) {
    const double a3[3] = {0.0310906908696548950, 0.01554534543482745, 0.0168868639403896};
    D5 rho = u + d;
    D5 zeta = (u - d) / rho;
    zeta.v = fmin(fmax(zeta.v, -1.0 + 1e-10), 1.0 - 1e-10);
    D5 zp = 1.0 + zeta, zm = 1.0 - zeta;
    D5 rs = cbrt5((3.0 / (4.0 * kPi)) / rho);
    D5 kf = cbrt5((3.0 * kPi * kPi) * rho);
    D5 kf2 = kf * kf;
    D5 s2 = sg / (4.0 * kf2 * (rho * rho));
    D5 phi = 0.5 * (p5(zp, 2.0 / 3.0) + p5(zm, 2.0 / 3.0));
    D5 phi3 = phi * phi * phi;
    D5 dxz = 0.5 * (p5(zp, 4.0 / 3.0) + p5(zm, 4.0 / 3.0));
    D5 dsz = 0.5 * (p5(zp, 5.0 / 3.0) + p5(zm, 5.0 / 3.0));
    D5 alpha = (ta - sg / (8.0 * rho)) / (0.3 * kf2 * rho * dsz);
    D5 fc = scan_c_switch(alpha);
    D5 eps = pw92_pol_eps(rho, zeta, a3);
    D5 beta = 0.066725 * (1.0 + 0.1 * rs) / (1.0 + 0.1778 * rs);
    D5 t2 = sg / (4.0 * (phi * phi) * ((4.0 / kPi) * kf) * (rho * rho));
    D5 w1 = expm15(c5(0.0) - eps / (ScanC::gamma * phi3));
    D5 A = beta / (ScanC::gamma * w1);
    D5 g = 1.0 / p5(1.0 + 4.0 * (A * t2), 0.25);
    D5 eps1 = eps + ScanC::gamma * phi3 * log1p5(w1 * (1.0 - g));
    D5 e0 = c5(-ScanC::b1c) / (1.0 + ScanC::b2c * sqrt5(rs) + ScanC::b3c * rs);
    D5 w0 = expm15(c5(0.0) - e0 / ScanC::b1c);
    D5 ginf = 1.0 / p5(1.0 + (4.0 * ScanC::chi_inf) * s2, 0.25);
    D5 z2 = zeta * zeta, z6 = z2 * z2 * z2;
    D5 gc = (1.0 - ScanC::gcnst * (dxz - 1.0)) * (1.0 - z6 * z6);
    D5 eps0 = (e0 + ScanC::b1c * log1p5(w0 * (1.0 - ginf))) * gc;
    return rho * (eps1 + fc * (eps0 - eps1));
}

// ---------------------------------------------------------------------------------------------
// TPSS correlation (mgga_c_tpss): Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003), eqs. (11)-(14) -- formulas in
// oracle/xc.py.  Unlike SCAN it depends on sigma_uu, sigma_ud, sigma_dd separately (|grad zeta|, the one-spin PBE terms): six
// variables with tau.  The five-slot dual carries (rho_u, rho_d, sigma_uu, sigma_ud, sigma_dd); tau enters ONLY through
// z = tau_W / tau, so the core takes z as an argument: evaluated once with z = sigma / (8 n tau) as a dual of the five (tau fixed)
// and once with z alone as the variable -- d e / d tau = (d e / d z)(-z / tau).  The closed-shell kernel has three variables
// (rho, sigma, tau) and evaluates the same core once at rho_u = rho_d = rho / 2, sigma_ss' = sigma / 4.
// No formula or literal in the reference: pinned by its construction (no correlation for a one-electron density, PBE where z = 0,
// uniform-gas limit) -- parity against libxc UNPINNED.
// ---------------------------------------------------------------------------------------------
DQC_DEV D5 pbe_c_eps5(D5 rho, D5 eps, D5 phi2, D5 phi3, D5 sig) {
    const double beta = kPbeBeta, gamma = 0.031090690869654895;
    D5 kf = cbrt5((3.0 * kPi * kPi) * rho);
    D5 t2 = sig / (4.0 * phi2 * ((4.0 / kPi) * kf) * (rho * rho));
    D5 A = c5(beta / gamma) / expm15(c5(0.0) - eps / (gamma * phi3));
    D5 At2 = A * t2;
    D5 X = (beta / gamma) * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2);
    return eps + gamma * phi3 * log1p5(X);
}

DQC_DEV D5 pbe_c_eps5_ferro(D5 rs_, D5 sss) {  // eps_PBE(n_s, 0, grad n_s, 0): zeta = 1 in closed form (phi = 2^(-1/3), PW92's ferromagnetic fit)
    const double a = 0.01554534543482745, alpha1 = 0.20548, b1 = 14.1189, b2 = 6.1977, b3 = 3.3662, b4 = 0.62517;
    D5 rs = cbrt5((3.0 / (4.0 * kPi)) / rs_);
    D5 sq = sqrt5(rs);
    D5 q1 = (2.0 * a) * (b1 * sq + b2 * rs + b3 * (rs * sq) + b4 * (rs * rs));
    D5 eps = (-2.0 * a) * (1.0 + alpha1 * rs) * log1p5(1.0 / q1);
    return pbe_c_eps5(rs_, eps, c5(0.62996052494743658), c5(0.5), sss);  // 2^(-2/3), 2^(-1)
}

DQC_DEV D5 dmax5(D5 a, D5 b) { return a.v >= b.v ? a : b; }

DQC_DEV D5 tpss_c_core(D5 u, D5 d, D5 suu, D5 sud, D5 sdd, D5 z) {
    const double a3[3] = {0.0310906908696548950, 0.01554534543482745, 0.0168868639403896};
    D5 rho = u + d;
    D5 zeta = (u - d) / rho;
    zeta.v = fmin(fmax(zeta.v, -1.0 + 1e-10), 1.0 - 1e-10);
    D5 sig = suu + 2.0 * sud + sdd;
    sig.v = fmax(sig.v, 1e-40);
    D5 opz = 1.0 + zeta, omz = 1.0 - zeta;
    D5 phi = 0.5 * (p5(opz, 2.0 / 3.0) + p5(omz, 2.0 / 3.0));
    D5 eps = pbe_c_eps5(rho, pw92_pol_eps(rho, zeta, a3), phi * phi, phi * phi * phi, sig);
    D5 et_u = dmax5(pbe_c_eps5_ferro(u, suu), eps);
    D5 et_d = dmax5(pbe_c_eps5_ferro(d, sdd), eps);
    D5 z2_ = zeta * zeta;
    D5 c0 = 0.53 + 0.87 * z2_ + 0.50 * (z2_ * z2_) + 2.26 * (z2_ * z2_ * z2_);
    D5 gz2 = (omz * omz * suu - 2.0 * (omz * opz) * sud + opz * opz * sdd) / (rho * rho);
    D5 xi2 = gz2 / (4.0 * p5((3.0 * kPi * kPi) * rho, 2.0 / 3.0));
    D5 cd = 1.0 + 0.5 * xi2 * (p5(opz, -4.0 / 3.0) + p5(omz, -4.0 / 3.0));
    D5 cd2 = cd * cd;
    D5 C = c0 / (cd2 * cd2);
    D5 zz = z * z;
    D5 erev = eps * (1.0 + C * zz) - (1.0 + C) * zz * ((u * et_u + d * et_d) / rho);
    return rho * erev * (1.0 + 2.8 * erev * (zz * z));
}

// polarised meta-GGA correlation with a gradient potential PER SPIN (dqc_xc_eval_mgga_pol2): mgga_c_scan (depends on the total
// gradient: both potentials are 2 v_sigma grad n) and mgga_c_tpss (v_grad,u = 2 v_uu grad n_u + v_ud grad n_d, libxc.py:205-215)
__global__ __launch_bounds__(256) void xc_mgga_pol2_kernel(double *__restrict__ edens, double *__restrict__ vru, double *__restrict__ vrd,
                                                           double *__restrict__ vgu, double *__restrict__ vgd, double *__restrict__ vtau,
                                                           const double *__restrict__ ru_, const double *__restrict__ rd_,
                                                           const double *__restrict__ gu_, const double *__restrict__ gd_,
                                                           const double *__restrict__ tu_, const double *__restrict__ td_, int n,
                                                           XcTerms terms) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double ru = ru_[i], rd = rd_[i];
        double gu[3], gd[3];
        for (int k = 0; k < 3; k++) { gu[k] = gu_[(size_t)k * n + i]; gd[k] = gd_[(size_t)k * n + i]; }
        double e = 0, dv[5] = {0, 0, 0, 0, 0}, vt = 0;  // d/d (rho_u, rho_d, sigma_uu, sigma_ud, sigma_dd), d/d tau
        if (ru + rd > 1e-15) {
            ru = fmax(ru, 0.5e-15);
            rd = fmax(rd, 0.5e-15);
            const double suu = fmax(gu[0] * gu[0] + gu[1] * gu[1] + gu[2] * gu[2], 1e-40);
            const double sdd = fmax(gd[0] * gd[0] + gd[1] * gd[1] + gd[2] * gd[2], 1e-40);
            const double sud = gu[0] * gd[0] + gu[1] * gd[1] + gu[2] * gd[2];
            const double tk = fmax(tu_[i] + td_[i], 1e-20);
            for (int t = 0; t < terms.n; t++) {
                if (terms.id[t] == DQC_XC_MGGA_C_SCAN) {
                    const double sig = fmax(suu + 2.0 * sud + sdd, 1e-40);
                    const D5 f = f_mgga_c_scan_pol(var5(ru, 0), var5(rd, 1), var5(sig, 2), var5(tk, 3));
                    e += terms.c[t] * f.v;
                    dv[0] += terms.c[t] * f.d[0]; dv[1] += terms.c[t] * f.d[1];
                    dv[2] += terms.c[t] * f.d[2]; dv[3] += terms.c[t] * 2.0 * f.d[2]; dv[4] += terms.c[t] * f.d[2];
                    vt += terms.c[t] * f.d[3];
                } else {  // DQC_XC_MGGA_C_TPSS
                    const D5 u = var5(ru, 0), d = var5(rd, 1), a = var5(suu, 2), b = var5(sud, 3), c = var5(sdd, 4);
                    D5 sg = a + 2.0 * b + c;
                    sg.v = fmax(sg.v, 1e-40);
                    D5 z = sg / ((8.0 * tk) * (u + d));
                    const bool over = z.v > 1.0;
                    if (over) z = c5(1.0);
                    const D5 f = tpss_c_core(u, d, a, b, c, z);
                    e += terms.c[t] * f.v;
                    for (int k = 0; k < 5; k++) dv[k] += terms.c[t] * f.d[k];
                    if (!over) {
                        const D5 fz = tpss_c_core(c5(ru), c5(rd), c5(suu), c5(sud), c5(sdd), var5(z.v, 0));
                        vt += terms.c[t] * fz.d[0] * (-z.v / tk);
                    }
                }
            }
        }
        if (edens) edens[i] = e;
        if (vru) { vru[i] = dv[0]; vrd[i] = dv[1]; }
        if (vgu)
            for (int k = 0; k < 3; k++) {
                vgu[(size_t)k * n + i] = 2.0 * dv[2] * gu[k] + dv[3] * gd[k];
                vgd[(size_t)k * n + i] = 2.0 * dv[4] * gd[k] + dv[3] * gu[k];
            }
        if (vtau) vtau[i] = vt;
    }
}

__global__ __launch_bounds__(256) void xc_mgga_pol_kernel(double *__restrict__ edens, double *__restrict__ vru, double *__restrict__ vrd,
                                                          double *__restrict__ vgrad, double *__restrict__ vtau,
                                                          const double *__restrict__ ru_, const double *__restrict__ rd_,
                                                          const double *__restrict__ gu_, const double *__restrict__ gd_,
                                                          const double *__restrict__ tu_, const double *__restrict__ td_, int n,
                                                          XcTerms terms) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double ru = ru_[i], rd = rd_[i];
        double gt[3];
        for (int k = 0; k < 3; k++) gt[k] = gu_[(size_t)k * n + i] + gd_[(size_t)k * n + i];
        double e = 0, dv[4] = {0, 0, 0, 0};
        if (ru + rd > 1e-15) {
            ru = fmax(ru, 0.5e-15);
            rd = fmax(rd, 0.5e-15);
            const double sig = fmax(gt[0] * gt[0] + gt[1] * gt[1] + gt[2] * gt[2], 1e-40), tk = fmax(tu_[i] + td_[i], 1e-20);
            for (int t = 0; t < terms.n; t++) {
                const D5 f = f_mgga_c_scan_pol(var5(ru, 0), var5(rd, 1), var5(sig, 2), var5(tk, 3));
                e += terms.c[t] * f.v;
                for (int k = 0; k < 4; k++) dv[k] += terms.c[t] * f.d[k];
            }
        }
        if (edens) edens[i] = e;
        if (vru) { vru[i] = dv[0]; vrd[i] = dv[1]; }
        if (vgrad)
            for (int k = 0; k < 3; k++) vgrad[(size_t)k * n + i] = 2.0 * dv[2] * gt[k];
        if (vtau) vtau[i] = dv[3];
    }
}

// PAIR: 0 generic (run-time ids), 1 mgga_x_scan alone, 2 mgga_x_scan + mgga_c_scan (compile-time ids: the branches fold away)
template <bool EXT, int PAIR = 0>
__global__ __launch_bounds__(256) void xc_mgga_kernel(double *__restrict__ edens, double *__restrict__ vrho, double *__restrict__ vgrad,
                               double *__restrict__ vtau, const double *__restrict__ rho,
                               const double *__restrict__ grho, const double *__restrict__ tau, int n, XcTerms terms) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double r = rho[i];
        const double gx = grho[i], gy = grho[(size_t)n + i], gz = grho[2 * (size_t)n + i];
        double e = 0, vr = 0, vs = 0, vt = 0;
        if (r > 1e-15) {
            const double sig = fmax(gx * gx + gy * gy + gz * gz, 1e-40), tk = fmax(tau[i], 1e-20);
            const Dual dr = mk(r, 1.0, 0.0), ds = mk(sig, 0.0, 1.0);
            auto term = [&](const int tid_, double &fv, double &fr, double &fs, double &ft) {
                ft = 0.0;
                if (xc_id_is_mgga(tid_)) {
                    D5 f;
                    if (tid_ == DQC_XC_MGGA_C_TPSS) {  // closed shell: the general form at rho_u = rho_d, sigma_ss' = sigma / 4
                        const D5 h = 0.5 * var5(r, 0), q = 0.25 * var5(sig, 1);
                        D5 z = var5(sig, 1) / (8.0 * var5(r, 0) * var5(tk, 2));
                        if (z.v > 1.0) z = c5(1.0);
                        f = tpss_c_core(h, h, q, q, q, z);
                    } else {
                        if (tid_ == DQC_XC_MGGA_X_SCAN) f = f_mgga_x_scan(var5(r, 0), var5(sig, 1), var5(tk, 2));
                        else if (tid_ == DQC_XC_MGGA_X_TPSS) f = f_mgga_x_tpss(var5(r, 0), var5(sig, 1), var5(tk, 2));
                        else f = f_mgga_c_scan(var5(r, 0), var5(sig, 1), var5(tk, 2));
                    }
                    fv = f.v; fr = f.d[0]; fs = f.d[1]; ft = f.d[2];
                } else {
                    const Dual f = f_lda_gga<EXT>(tid_, dr, ds);
                    fv = f.v; fr = f.r; fs = f.s;
                }
            };
            constexpr int NFIX = PAIR == 1 ? 1 : (PAIR == 2 ? 2 : 0);
            if constexpr (NFIX > 0) {
                double fv, fr, fs, ft;
                term(DQC_XC_MGGA_X_SCAN, fv, fr, fs, ft);
                e += terms.c[0] * fv; vr += terms.c[0] * fr; vs += terms.c[0] * fs; vt += terms.c[0] * ft;
                if constexpr (NFIX > 1) {
                    term(DQC_XC_MGGA_C_SCAN, fv, fr, fs, ft);
                    e += terms.c[1] * fv; vr += terms.c[1] * fr; vs += terms.c[1] * fs; vt += terms.c[1] * ft;
                }
            } else {
                for (int t = 0; t < terms.n; t++) {
                    double fv, fr, fs, ft;
                    term(terms.id[t], fv, fr, fs, ft);
                    e += terms.c[t] * fv; vr += terms.c[t] * fr; vs += terms.c[t] * fs; vt += terms.c[t] * ft;
                }
            }
        }
        if (edens) edens[i] = e;
        if (vrho) vrho[i] = vr;
        if (vgrad) {
            vgrad[i] = 2.0 * vs * gx;
            vgrad[(size_t)n + i] = 2.0 * vs * gy;
            vgrad[2 * (size_t)n + i] = 2.0 * vs * gz;
        }
        if (vtau) vtau[i] = vt;
    }
}

}  // namespace dqc

extern "C" int dqc_xc_eval_mgga(double *d_edens, double *d_vrho, double *d_vgrad, double *d_vtau, const double *d_rho,
                                const double *d_grho, const double *d_tau, int n, const int *ids, const double *coefs,
                                int nterm, void *stream) {
    using namespace dqc;
    if (nterm < 0 || nterm > 8) { set_error("dqc_xc_eval_mgga: at most 8 functional terms"); return DQC_EINVAL; }
    if (!d_grho || !d_tau) { set_error("dqc_xc_eval_mgga: needs the density gradient and tau"); return DQC_EINVAL; }
    XcTerms t;
    t.n = nterm;
    for (int i = 0; i < nterm; i++) {
        t.id[i] = ids[i];
        t.c[i] = coefs[i];
        if (!(xc_host_is_lda(ids[i]) || xc_host_is_gga(ids[i]) || xc_id_is_mgga(ids[i]))) {
            set_error("dqc_xc_eval_mgga: unknown functional id");
            return DQC_EINVAL;
        }
    }
    if (n <= 0) return DQC_OK;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    bool ext = false;
    for (int i = 0; i < nterm; i++) ext = ext || xc_id_is_ext(ids[i]);
    static const bool no_pair = getenv("DQC_XC_GENERIC") != nullptr;  // (A/B runs)
    const int pair = no_pair ? 0 : (nterm == 1 && ids[0] == DQC_XC_MGGA_X_SCAN) ? 1
                   : (nterm == 2 && ids[0] == DQC_XC_MGGA_X_SCAN && ids[1] == DQC_XC_MGGA_C_SCAN) ? 2 : 0;
    if (pair == 1) hipLaunchKernelGGL((xc_mgga_kernel<false, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad, d_vtau,
                                      d_rho, d_grho, d_tau, n, t);
    else if (pair == 2) hipLaunchKernelGGL((xc_mgga_kernel<false, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad, d_vtau,
                                           d_rho, d_grho, d_tau, n, t);
    else if (ext) hipLaunchKernelGGL(xc_mgga_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad, d_vtau,
                                d_rho, d_grho, d_tau, n, t);
    else hipLaunchKernelGGL(xc_mgga_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho, d_vgrad, d_vtau,
                            d_rho, d_grho, d_tau, n, t);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_xc_eval_pol(double *d_edens, double *d_vrho_u, double *d_vrho_d, double *d_vgrad_u, double *d_vgrad_d,
                               const double *d_rho_u, const double *d_rho_d, const double *d_grho_u,
                               const double *d_grho_d, int n, const int *ids, const double *coefs, int nterm,
                               void *stream) {
    using namespace dqc;
    if (nterm < 0 || nterm > 8) { set_error("dqc_xc_eval_pol: at most 8 functional terms"); return DQC_EINVAL; }
    XcTerms t;
    t.n = nterm;
    bool need_grad = false;
    for (int i = 0; i < nterm; i++) {
        t.id[i] = ids[i];
        t.c[i] = coefs[i];
        if (xc_host_is_gga(ids[i])) need_grad = true;
        else if (!xc_host_is_lda(ids[i])) { set_error("dqc_xc_eval_pol: unknown functional id"); return DQC_EINVAL; }
    }
    const bool gga = d_grho_u && d_grho_d;
    if (need_grad && !gga) { set_error("dqc_xc_eval_pol: GGA functional needs both density gradients"); return DQC_EINVAL; }
    if ((d_vrho_u == nullptr) != (d_vrho_d == nullptr)) { set_error("dqc_xc_eval_pol: give both vrho outputs or none"); return DQC_EINVAL; }
    if (n <= 0) return DQC_OK;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    bool ext = false;
    for (int i = 0; i < nterm; i++) ext = ext || xc_id_is_ext(ids[i]);
    static const bool no_pair = getenv("DQC_XC_GENERIC") != nullptr;  // (A/B runs)
    if (!no_pair && gga && nterm == 2 && ids[0] == DQC_XC_GGA_X_PBE && ids[1] == DQC_XC_GGA_C_PBE)  // compile-time PBE pair
        hipLaunchKernelGGL((xc_pol_kernel<false, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho_u, d_vrho_d, d_vgrad_u,
                           d_vgrad_d, d_rho_u, d_rho_d, d_grho_u, d_grho_d, n, t, 1);
    else if (ext) hipLaunchKernelGGL(xc_pol_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho_u, d_vrho_d,
                                d_vgrad_u, d_vgrad_d, d_rho_u, d_rho_d, d_grho_u, d_grho_d, n, t, gga ? 1 : 0);
    else hipLaunchKernelGGL(xc_pol_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho_u, d_vrho_d,
                            d_vgrad_u, d_vgrad_d, d_rho_u, d_rho_d, d_grho_u, d_grho_d, n, t, gga ? 1 : 0);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_xc_eval_mgga_pol(double *d_edens, double *d_vrho_u, double *d_vrho_d, double *d_vgrad, double *d_vtau,
                                    const double *d_rho_u, const double *d_rho_d, const double *d_grho_u, const double *d_grho_d,
                                    const double *d_tau_u, const double *d_tau_d, int n, const int *ids, const double *coefs,
                                    int nterm, void *stream) {
    using namespace dqc;
    if (nterm < 0 || nterm > 8) { set_error("dqc_xc_eval_mgga_pol: at most 8 functional terms"); return DQC_EINVAL; }
    if (!d_grho_u || !d_grho_d || !d_tau_u || !d_tau_d) { set_error("dqc_xc_eval_mgga_pol: needs both density gradients and both tau"); return DQC_EINVAL; }
    if ((d_vrho_u == nullptr) != (d_vrho_d == nullptr)) { set_error("dqc_xc_eval_mgga_pol: give both vrho outputs or none"); return DQC_EINVAL; }
    XcTerms t;
    t.n = nterm;
    for (int i = 0; i < nterm; i++) {
        t.id[i] = ids[i];
        t.c[i] = coefs[i];
        if (ids[i] != DQC_XC_MGGA_C_SCAN) { set_error("dqc_xc_eval_mgga_pol: only meta-GGA correlation ids (DQC_XC_MGGA_C_SCAN)"); return DQC_EINVAL; }
    }
    if (n <= 0) return DQC_OK;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(xc_mgga_pol_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho_u, d_vrho_d, d_vgrad,
                       d_vtau, d_rho_u, d_rho_d, d_grho_u, d_grho_d, d_tau_u, d_tau_d, n, t);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_xc_eval_mgga_pol2(double *d_edens, double *d_vrho_u, double *d_vrho_d, double *d_vgrad_u, double *d_vgrad_d,
                                     double *d_vtau, const double *d_rho_u, const double *d_rho_d, const double *d_grho_u,
                                     const double *d_grho_d, const double *d_tau_u, const double *d_tau_d, int n, const int *ids,
                                     const double *coefs, int nterm, void *stream) {
    // dqc_xc_eval_mgga_pol with one gradient potential per spin: DQC_XC_MGGA_C_SCAN and DQC_XC_MGGA_C_TPSS (the latter depends on
    // sigma_uu, sigma_ud, sigma_dd separately).  d_vtau is shared by the spins (both functionals see tau_u + tau_d only).
    using namespace dqc;
    if (nterm < 0 || nterm > 8) { set_error("dqc_xc_eval_mgga_pol2: at most 8 functional terms"); return DQC_EINVAL; }
    if (!d_grho_u || !d_grho_d || !d_tau_u || !d_tau_d) { set_error("dqc_xc_eval_mgga_pol2: needs both density gradients and both tau"); return DQC_EINVAL; }
    if ((d_vrho_u == nullptr) != (d_vrho_d == nullptr) || (d_vgrad_u == nullptr) != (d_vgrad_d == nullptr)) {
        set_error("dqc_xc_eval_mgga_pol2: give both spin outputs or none");
        return DQC_EINVAL;
    }
    XcTerms t;
    t.n = nterm;
    for (int i = 0; i < nterm; i++) {
        t.id[i] = ids[i];
        t.c[i] = coefs[i];
        if (ids[i] != DQC_XC_MGGA_C_SCAN && ids[i] != DQC_XC_MGGA_C_TPSS) {
            set_error("dqc_xc_eval_mgga_pol2: only meta-GGA correlation ids (DQC_XC_MGGA_C_SCAN, DQC_XC_MGGA_C_TPSS)");
            return DQC_EINVAL;
        }
    }
    if (n <= 0) return DQC_OK;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(xc_mgga_pol2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_edens, d_vrho_u, d_vrho_d, d_vgrad_u,
                       d_vgrad_d, d_vtau, d_rho_u, d_rho_d, d_grho_u, d_grho_d, d_tau_u, d_tau_d, n, t);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}
