// grid.hip -- density on the grid and the Vxc matrix: the two GEMM-shaped passes over the cached
// AO matrix (reference: HamiltonCGTO._dm2densinfo hcgto.py:371-443 and _get_vxc_from_potinfo
// hcgto.py:445-495, which run them as chunked torch.matmul + einsum on the CPU).
//
// Both kernels are fp64 MFMA (v_mfma_f64_16x16x4_f64) GEMMs whose operands stream from HBM exactly
// once per pass; the element-wise parts of the reference (row dots, v*phi, symmetrisation) are
// fused into the prologue/epilogue so nothing of size (ngrid, nao) is ever written back.
//
//   density:  A = Phi[32 pts x n] . D[n x n] per wave, accumulators stay in registers, epilogue
//             rho_g = sum_j A_gj Phi_gj , grad rho_g = 2 sum_j A_gj dPhi_gj   (16-lane DPP reduce)
//   vxc:      M = Phi^T . Psi,  Psi = w (vrho Phi + sum_d 2 vgrad_d dPhi_d), split-K over point slabs,
//             16-point chunks staged once in LDS (Psi is formed on the way in), every wave owns an equal
//             share of the 16x16 output tiles, partial sums reduced with fp64 atomics; V = (M + M^T)/2.
//
// f64 MFMA fragment layout (gfx950): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// C[row = (lane>>4) + 4*reg][col = lane&15].
#include "common.hpp"

namespace dqc {

typedef double v4d __attribute__((ext_vector_type(4)));

DQC_DEV v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------
// density
// ---------------------------------------------------------------------------------------------
template <int NCT, bool GGA>
__global__ __launch_bounds__(256, 1) void density_kernel(double *__restrict__ rho, double *__restrict__ grho,
                                                         const double *__restrict__ ao, int ngrid, int ld,
                                                         const double *__restrict__ dm, int ntile) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * 4 + wave) * 32;
    if (g0 >= ngrid) return;
    const int lr = lane & 15, lk = lane >> 4;
    const size_t cs = (size_t)ngrid * ld;  // component stride of ao
    const double *a0p = ao + (size_t)min(g0 + lr, ngrid - 1) * ld + lk;
    const double *a1p = ao + (size_t)min(g0 + 16 + lr, ngrid - 1) * ld + lk;

    double p[2][4][GGA ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < (GGA ? 4 : 1); q++) p[i][r][q] = 0.0;

    for (int jc = 0; jc < ntile; jc += NCT) {
        v4d acc[2][NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) { acc[0][ct] = v4d{0, 0, 0, 0}; acc[1][ct] = v4d{0, 0, 0, 0}; }
        const double *bp = dm + (size_t)lk * ld + jc * 16 + lr;
        const int nvalid = min(NCT, ntile - jc);
#pragma unroll 2
        for (int k = 0; k < ld; k += 4) {
            const double a0 = a0p[k], a1 = a1p[k];
            const double *bk = bp + (size_t)k * ld;
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
                if (ct < nvalid) {  // wave-uniform
                    const double b = bk[ct * 16];
                    acc[0][ct] = mfma_f64(a0, b, acc[0][ct]);
                    acc[1][ct] = mfma_f64(a1, b, acc[1][ct]);
                }
            }
        }
        // epilogue: row dots with Phi (and its gradient) in the accumulator layout
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = min(g0 + rt * 16 + lk + 4 * r, ngrid - 1);
                const double *ap = ao + (size_t)row * ld + jc * 16 + lr;
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) {
                    if (ct < nvalid) {
                        const double v = acc[rt][ct][r];
                        p[rt][r][0] += v * ap[ct * 16];
                        if (GGA) {
                            p[rt][r][1] += v * ap[cs + ct * 16];
                            p[rt][r][2] += v * ap[2 * cs + ct * 16];
                            p[rt][r][3] += v * ap[3 * cs + ct * 16];
                        }
                    }
                }
            }
    }
    // reduce over the 16 lanes that share a row
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < (GGA ? 4 : 1); q++) {
                double v = p[rt][r][q];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                p[rt][r][q] = v;
            }
    if (lr == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g0 + rt * 16 + lk + 4 * r;
                if (row < ngrid) {
                    rho[row] = p[rt][r][0];
                    if (GGA) {
                        grho[row] = 2.0 * p[rt][r][1];
                        grho[(size_t)ngrid + row] = 2.0 * p[rt][r][2];
                        grho[2 * (size_t)ngrid + row] = 2.0 * p[rt][r][3];
                    }
                }
            }
    }
}

template <bool GGA>
static int launch_density(int nct, dim3 grid, hipStream_t st, double *rho, double *grho, const double *ao,
                          int ngrid, int ld, const double *dm, int ntile) {
#define DQC_DENS_CASE(N)                                                                                  \
    case N:                                                                                               \
        hipLaunchKernelGGL((density_kernel<N, GGA>), grid, dim3(256), 0, st, rho, grho, ao, ngrid, ld, dm, ntile); \
        break;
    switch (nct) {
        DQC_DENS_CASE(1) DQC_DENS_CASE(2) DQC_DENS_CASE(3) DQC_DENS_CASE(4) DQC_DENS_CASE(5) DQC_DENS_CASE(6)
        DQC_DENS_CASE(7) DQC_DENS_CASE(8) DQC_DENS_CASE(9) DQC_DENS_CASE(10) DQC_DENS_CASE(11) DQC_DENS_CASE(12)
        DQC_DENS_CASE(13) DQC_DENS_CASE(14) DQC_DENS_CASE(15) DQC_DENS_CASE(16)
    default:
        set_error("density: internal tile-count dispatch error");
        return DQC_EINVAL;
    }
#undef DQC_DENS_CASE
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Vxc
// ---------------------------------------------------------------------------------------------
constexpr int VXC_KC = 16;      // points per LDS chunk
constexpr int VXC_WAVES = 8;    // waves per block

DQC_DEV int lds_stride(int ld) {  // stride == 16 (mod 32) doubles -> conflict-free ds_read_b64 fragments
    int s = ld;
    while ((s & 31) != 16) s += 16;
    return s;
}

template <int MAXT, bool GGA>
__global__ __launch_bounds__(512, 2) void vxc_kernel(double *__restrict__ vmat, const double *__restrict__ ao,
                                                     int ngrid, int ld, const double *__restrict__ w,
                                                     const double *__restrict__ vrho, const double *__restrict__ vgrad,
                                                     int slab, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int LS = lds_stride(ld);
    double *sphi = lds, *spsi = lds + VXC_KC * LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int T = ld >> 4, ttot = T * T;
    const size_t cs = (size_t)ngrid * ld;

    // this block: point slab blockIdx.x, output tile chunk blockIdx.y
    const int gs = blockIdx.x * slab, ge = min(gs + slab, ngrid);
    const int tc0 = blockIdx.y * tiles_per_chunk;
    const int tc1 = min(tc0 + tiles_per_chunk, ttot);
    const int per_wave = (tc1 - tc0 + VXC_WAVES - 1) / VXC_WAVES;
    const int t0 = tc0 + wave * per_wave;
    const int nt = max(0, min(per_wave, tc1 - t0));

    v4d acc[MAXT];
    unsigned offab[MAXT];  // LDS offsets of the A (low 16 bits) and B (high 16 bits) fragments
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        const int id = min(t0 + t, ttot - 1);
        offab[t] = (unsigned)(lk * LS + (id / T) * 16 + lr) | ((unsigned)(lk * LS + (id % T) * 16 + lr) << 16);
    }

    const int half = ld >> 1;  // double2 columns per row
    for (int gc = gs; gc < ge; gc += VXC_KC) {
        __syncthreads();  // previous chunk fully consumed
        for (int e = tid; e < VXC_KC * half; e += 512) {
            const int row = e / half, c2 = (e - row * half) * 2;
            const int g = gc + row;
            double2 phi = make_double2(0.0, 0.0), psi = make_double2(0.0, 0.0);
            if (g < ge) {
                const double *src = ao + (size_t)g * ld + c2;
                phi = *reinterpret_cast<const double2 *>(src);
                const double wg = w[g];
                const double c0 = wg * vrho[g];
                psi.x = c0 * phi.x;
                psi.y = c0 * phi.y;
                if (GGA) {
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        const double cd = 2.0 * wg * vgrad[(size_t)d * ngrid + g];
                        const double2 dp = *reinterpret_cast<const double2 *>(src + (d + 1) * cs);
                        psi.x += cd * dp.x;
                        psi.y += cd * dp.y;
                    }
                }
            }
            *reinterpret_cast<double2 *>(sphi + row * LS + c2) = phi;
            *reinterpret_cast<double2 *>(spsi + row * LS + c2) = psi;
        }
        __syncthreads();
#pragma unroll 1
        for (int kk = 0; kk < VXC_KC / 4; kk++) {
            const int ko = kk * 4 * LS;
#pragma unroll
            for (int t = 0; t < MAXT; t++) {
                if (t < nt) {  // wave-uniform
                    const double a = sphi[ko + (offab[t] & 0xffffu)];
                    const double b = spsi[ko + (offab[t] >> 16)];
                    acc[t] = mfma_f64(a, b, acc[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (t < nt) {
            const int id = t0 + t;
            const int ia = (id / T) * 16 + lk, ib = (id % T) * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; r++) atomicAdd(&vmat[(size_t)(ia + 4 * r) * ld + ib], acc[t][r]);
        }
    }
}

__global__ void symmetrize_kernel(double *m, int ld) {
    const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
    if (i < ld && j < i) {
        const double v = 0.5 * (m[(size_t)i * ld + j] + m[(size_t)j * ld + i]);
        m[(size_t)i * ld + j] = v;
        m[(size_t)j * ld + i] = v;
    }
}

template <bool GGA>
static int launch_vxc(int maxt, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid,
                      int ld, const double *w, const double *vrho, const double *vgrad, int slab, int tpc) {
#define DQC_VXC_CASE(N)                                                                                          \
    case N:                                                                                                      \
        (void)hipFuncSetAttribute((const void *)vxc_kernel<N, GGA>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)shmem);                                                                   \
        hipLaunchKernelGGL((vxc_kernel<N, GGA>), grid, dim3(512), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, \
                           slab, tpc);                                                                           \
        break;
    switch (maxt) {
        DQC_VXC_CASE(2) DQC_VXC_CASE(4) DQC_VXC_CASE(8) DQC_VXC_CASE(12) DQC_VXC_CASE(16) DQC_VXC_CASE(22)
    default:
        set_error("vxc: internal tile-count dispatch error");
        return DQC_EINVAL;
    }
#undef DQC_VXC_CASE
    return 0;
}

}  // namespace dqc

extern "C" {

int dqc_grid_density(double *d_rho, double *d_grho, const double *d_ao, int ncomp, int ngrid, int nao,
                     const double *d_dm, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0) return DQC_OK;
    const bool gga = d_grho != nullptr;
    if (gga && ncomp < 4) { set_error("dqc_grid_density: gradient requested but ao has < 4 components"); return DQC_EINVAL; }
    const int ld = dqc_padded_nao(nao), ntile = ld / 16;
    const int nchunk = (ntile + 15) / 16;
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + 127) / 128);
    int rc = gga ? launch_density<true>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_dm, ntile)
                 : launch_density<false>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_dm, ntile);
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_grid_vxc(double *d_vmat, const double *d_ao, int ncomp, int ngrid, int nao, const double *d_w,
                 const double *d_vrho, const double *d_vgrad, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    const bool gga = d_vgrad != nullptr;
    if (gga && ncomp < 4) { set_error("dqc_grid_vxc: vgrad given but ao has < 4 components"); return DQC_EINVAL; }
    const int ld = dqc_padded_nao(nao), T = ld / 16, ttot = T * T;
    DQC_HIP(hipMemsetAsync(d_vmat, 0, sizeof(double) * (size_t)ld * ld, st));
    if (ngrid > 0) {
        static const int sizes[] = {2, 4, 8, 12, 16, 22};
        const int cap = 22 * VXC_WAVES;
        const int nchunk = (ttot + cap - 1) / cap;
        const int tpc = (ttot + nchunk - 1) / nchunk;
        const int need = (tpc + VXC_WAVES - 1) / VXC_WAVES;
        int maxt = 22;
        for (int s : sizes)
            if (s >= need) { maxt = s; break; }
        // one block per CU and chunk (register budget admits one 8-wave block per CU)
        int nslab = (2 * 256 + nchunk - 1) / nchunk;
        int slab = (ngrid + nslab - 1) / nslab;
        slab = (slab + VXC_KC - 1) / VXC_KC * VXC_KC;
        nslab = (ngrid + slab - 1) / slab;
        int LS = ld;
        while ((LS & 31) != 16) LS += 16;
        const size_t shmem = sizeof(double) * 2 * VXC_KC * LS;
        dim3 grid(nslab, nchunk);
        int rc;
        if (gga) {
            rc = launch_vxc<true>(maxt, grid, shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, tpc);
        } else {
            rc = launch_vxc<false>(maxt, grid, shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, tpc);
        }
        if (rc) return rc;
        DQC_CHECK_LAUNCH();
        hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}

}  // extern "C"
