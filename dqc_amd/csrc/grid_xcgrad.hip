// grid_xcgrad.hip -- the grid sums of the XC part of a nuclear gradient (GGA / meta-GGA), one pass over the AO derivative arrays.
//
// Reference: the reference differentiates E_xc = sum_g w_g e(rho(r_g)) by autograd through eval_gradgto / _dm2densinfo
// (dqc/hamilton/intor/gtoeval.py:173-193, hcgto.py:371-443); dqc_amd/gradient.py writes the same derivative out (its docstring):
// with b = Phi D, c_i = d_i Phi D, u = the gradient part of the potential, S_j = sum_i u_i d_i d_j Phi,
//   (ii)  points riding on their atom:   q[g][j]      = w [ v_rho d_j rho + 2 sum_mu (b S_j + c_j (u . grad phi)) ]
//   (iii) centres of the AOs:            per_ao[mu][j] = sum_g w [ d_j phi (v_rho b + u . c) + b S_j ]
//   meta-GGA: b S_j gains 1/2 v_tau sum_d (d_d d_j phi) c_d in both.
// Round 4 formed these with ~40 torch element-wise passes over (ngrid, nao) arrays (62 ms of a 184 ms gradient on a 20-atom
// molecule).  Here: one wave per grid point at a time, lanes along the AO index (coalesced rows of the ten derivative arrays and
// of b, c), the per-point sums by a wave reduction, the per-AO sums in registers across the wave's points and one atomic per
// (AO, direction) and wave at the end.  HBM-bound: 14 arrays of ngrid x lda doubles are read once (8.9 GB for the C5 molecule).
#include "common.hpp"

namespace dqc {

// component of d2/(d i d j) in the deriv-3 AO array (value, 3 gradients, xx xy xz yy yz zz)
__device__ __forceinline__ constexpr int hess_comp(int i, int j) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    return a == 0 ? 4 + b : (a == 1 ? 6 + b : 9);
}

constexpr int XG_PTS = 32;  // grid points per wave
constexpr int XG_MAXT = 8;  // AO columns per lane: lda <= 512

template <int NT>
__global__ __launch_bounds__(256) void xc_grad_terms_kernel(double *__restrict__ q, double *__restrict__ perao, const double *__restrict__ ao,
                                                            int ngrid, int nao, int lda, const double *__restrict__ b,
                                                            const double *__restrict__ c0, const double *__restrict__ c1,
                                                            const double *__restrict__ c2, int ldb, const double *__restrict__ w,
                                                            const double *__restrict__ vrho, const double *__restrict__ u,
                                                            const double *__restrict__ grho, const double *__restrict__ vtau) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long g0 = ((long long)blockIdx.x * 4 + wave) * XG_PTS;
    const size_t cs = (size_t)ngrid * lda;
    double pa[NT][3];
#pragma unroll
    for (int t = 0; t < NT; t++) pa[t][0] = pa[t][1] = pa[t][2] = 0.0;
    for (int ip = 0; ip < XG_PTS; ip++) {
        const long long g = g0 + ip;
        if (g >= ngrid) break;  // (wave-uniform)
        const double wg = w[g], vr = vrho[g];
        const double ug[3] = {u[g], u[(size_t)ngrid + g], u[2 * (size_t)ngrid + g]};
        const double vt = vtau ? 0.5 * vtau[g] : 0.0;
        double qs[3] = {0.0, 0.0, 0.0};
        const double *row = ao + (size_t)g * lda;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int mu = lane + 64 * t;
            if (mu >= nao) continue;
            double a[10];
#pragma unroll
            for (int k = 1; k < 10; k++) a[k] = row[k * cs + mu];
            const double bb = b[(size_t)g * ldb + mu];
            const double cc[3] = {c0[(size_t)g * ldb + mu], c1[(size_t)g * ldb + mu], c2[(size_t)g * ldb + mu]};
            const double ugphi = ug[0] * a[1] + ug[1] * a[2] + ug[2] * a[3];
            const double t1 = vr * bb + ug[0] * cc[0] + ug[1] * cc[1] + ug[2] * cc[2];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double h0 = a[hess_comp(0, j)], h1 = a[hess_comp(1, j)], h2 = a[hess_comp(2, j)];
                const double bsj = bb * (ug[0] * h0 + ug[1] * h1 + ug[2] * h2) + vt * (h0 * cc[0] + h1 * cc[1] + h2 * cc[2]);
                qs[j] += bsj + cc[j] * ugphi;
                pa[t][j] += wg * (a[1 + j] * t1 + bsj);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double v = qs[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) q[(size_t)g * 3 + j] = wg * (vr * grho[(size_t)j * ngrid + g] + 2.0 * v);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int mu = lane + 64 * t;
        if (mu < nao)
#pragma unroll
            for (int j = 0; j < 3; j++) atomicAdd(&perao[(size_t)mu * 3 + j], pa[t][j]);
    }
}

}  // namespace dqc

extern "C" {

int dqc_grid_xc_gradient_terms(double *d_q, double *d_perao, const double *d_ao, int ngrid, int nao, const double *d_b,
                               const double *d_c0, const double *d_c1, const double *d_c2, int ldb, const double *d_w,
                               const double *d_vrho, const double *d_u, const double *d_grho, const double *d_vtau, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0 || nao <= 0) return DQC_OK;
    const int lda = dqc_ao_stride(nao);
    if (nao > 64 * XG_MAXT) { set_error("dqc_grid_xc_gradient_terms: more than 512 basis functions"); return DQC_EINVAL; }
    if (ldb < nao) { set_error("dqc_grid_xc_gradient_terms: ldb < nao"); return DQC_EINVAL; }
    DQC_HIP(hipMemsetAsync(d_perao, 0, sizeof(double) * (size_t)nao * 3, st));
    const unsigned nblk = (unsigned)(((long long)ngrid + 4 * XG_PTS - 1) / (4 * XG_PTS));
    const int nt = (nao + 63) / 64;
#define DQC_XG_CASE(N)                                                                                                        \
    case N:                                                                                                                   \
        hipLaunchKernelGGL((xc_grad_terms_kernel<N>), dim3(nblk), dim3(256), 0, st, d_q, d_perao, d_ao, ngrid, nao, lda, d_b, d_c0, \
                           d_c1, d_c2, ldb, d_w, d_vrho, d_u, d_grho, d_vtau);                                                 \
        break;
    switch (nt) {
        DQC_XG_CASE(1) DQC_XG_CASE(2) DQC_XG_CASE(3) DQC_XG_CASE(4) DQC_XG_CASE(5) DQC_XG_CASE(6) DQC_XG_CASE(7) DQC_XG_CASE(8)
    }
#undef DQC_XG_CASE
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
