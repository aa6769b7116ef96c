// fock.hip -- the small-matrix ends of a restricted Fock build: six tile-parallel launches instead of ~22 torch launches (round 6)
//
// Around the tile stream (and the grid pass) a Fock build of the reference does, per density matrix, in torch.matmul / einsum calls
// (hcgto.py:204-241 J / K with the orbital-converter round trips of orbconverter.py:126-163; hcgto.py:272-281 ao_orb2dm;
// hf.py:182-201, ks.py:176-187):
//     L = X (C sqrt(w)),  D_ao = L L^T   or   D_ao = X D X^T      (AO density from the orbitals / the orthogonal-basis density)
//     J, K from the accumulators, symmetrised                     (jk_finish_kernel)
//     E_J = 1/2 tr D_ao J,  E_K = -1/4 tr D_ao K
//     F = X^T (J - K / 2 + V_xc,ao) X, symmetrised, + the core Hamiltonian
// Each of these is microseconds of work and was one launch (or three) of its own: at nao = 114 (benzene / cc-pVDZ) the RHF build
// took 0.15 ms as a hipGraph around a 58 us tile pass.
//
// Every product here is ONE 4-wave block per 16 x 16 output tile, K split over the four waves, operand fragments straight from L2
// in batches of independent loads, partial tiles added through LDS in a fixed order (deterministic).  A first version gave every
// 16-row PANEL of the result one 16-wave block that kept its intermediate in LDS (two launches per build): each of the nao / 16
// blocks then pulls whole matrices through one CU, and a CU moves ~40 GB/s -- 39 us for the finish at nao 208, 85 us per build
// (rocprofv3).  With tiles the same bytes are spread over ~170 CUs.
#include "grid_common.hpp"

namespace dqc {

constexpr int FT_NT = 256, FT_WAVES = 4;

// One 16 x 16 tile  C = sum_k A[m][k] B[k][n]  by a 256-thread block: wave w takes the k-steps w, w + 4, ... in batches of U
// (all operand loads of a batch are issued before its first MFMA: af / bf load unconditionally from clamped addresses and select);
// the four partial tiles meet in `sred` (4 x 256 doubles) and every thread of wave 0 returns its 4 elements of the sum
// (C layout: row = kq + 4 reg, col = lr); the other waves return zeros.  K4: K rounded up to a multiple of 4.
template <class AF, class BF>
DQC_DEV v4d ft_tile(int K4, AF af, BF bf, double *sred) {
    constexpr int U = 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    v4d acc{0, 0, 0, 0};
    for (int k0 = 4 * wave; k0 < K4; k0 += 4 * FT_WAVES * U) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + 4 * FT_WAVES * u + kq;
            a[u] = af(lr, k);
            b[u] = bf(k, lr);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) acc = mfma_f64(a[u], b[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) sred[wave * 256 + (kq + 4 * r) * 16 + lr] = acc[r];
    __syncthreads();
    v4d out{0, 0, 0, 0};
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int e = (kq + 4 * r) * 16 + lr;
            out[r] = ((sred[e] + sred[256 + e]) + sred[512 + e]) + sred[768 + e];
        }
    }
    return out;
}

// upper-triangular tile index u -> (ti <= tj) of a T x T tile grid
DQC_DEV void ft_upper(int u, int T, int &ti, int &tj) {
    int i = 0, rem = u;
    while (rem >= T - i) { rem -= T - i; i++; }
    ti = i;
    tj = i + rem;
}

// ---------------------------------------------------------------------------------------------
// factor.  L = X (C sqrt(w)) (nao x r), written zero padded as orb (ld x rp) and its transpose orbt (rp x ld): the operand pair of
// dqc_grid_density_lr and of fock_dao_kernel.  grid (ld / 16, rp / 16).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FT_NT) void fock_factor_kernel(double *__restrict__ orb, double *__restrict__ orbt, const double *__restrict__ x,
                                                            const double *__restrict__ c, int ldc, const double *__restrict__ w, int nao,
                                                            int north, int r, int ld, int rp) {
    __shared__ double sred[4 * 256];
    const int lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4;
    const int i0 = 16 * blockIdx.x, c0 = 16 * blockIdx.y;
    const int col = c0 + lr;
    const double sw = col < r ? sqrt(w[min(col, r - 1)]) : 0.0;
    const v4d t = ft_tile((north + 3) & ~3,
        [&](int m, int k) { const double v = x[(size_t)min(i0 + m, nao - 1) * north + min(k, north - 1)]; return (i0 + m < nao && k < north) ? v : 0.0; },
        [&](int k, int n) { const double v = c[(size_t)min(k, north - 1) * ldc + min(c0 + n, r - 1)]; return k < north ? v * sw : 0.0; }, sred);
    if (threadIdx.x < 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + kq + 4 * q;
            orb[(size_t)i * rp + col] = t[q];
            orbt[(size_t)col * ld + i] = t[q];
        }
    }
}

// ao_orb2dm and its factor in ONE launch (hcgto.py:272-281): the first nfac blocks are the tiles of L = X (C sqrt(w)) (as
// fock_factor_kernel), the others the tiles on and right of the diagonal of D = C diag(w) C^T (north x north, orthogonal basis),
// written with their mirror images.
__global__ __launch_bounds__(FT_NT) void fock_orb2dm_kernel(double *__restrict__ dm, double *__restrict__ orb, double *__restrict__ orbt,
                                                            const double *__restrict__ x, const double *__restrict__ c, int ldc,
                                                            const double *__restrict__ w, int nao, int north, int r, int ld, int rp) {
    __shared__ double sred[4 * 256];
    __shared__ double sd[16 * 17];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int nfx = ld / 16, nfac = nfx * (rp / 16);
    if ((int)blockIdx.x < nfac) {
        const int i0 = 16 * ((int)blockIdx.x % nfx), c0 = 16 * ((int)blockIdx.x / nfx);
        const int col = c0 + lr;
        const double sw = col < r ? sqrt(w[min(col, r - 1)]) : 0.0;
        const v4d t = ft_tile((north + 3) & ~3,
            [&](int m, int k) { const double v = x[(size_t)min(i0 + m, nao - 1) * north + min(k, north - 1)]; return (i0 + m < nao && k < north) ? v : 0.0; },
            [&](int k, int n) { const double v = c[(size_t)min(k, north - 1) * ldc + min(c0 + n, r - 1)]; return k < north ? v * sw : 0.0; }, sred);
        if (tid < 64) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + kq + 4 * q;
                orb[(size_t)i * rp + col] = t[q];
                orbt[(size_t)col * ld + i] = t[q];
            }
        }
        return;
    }
    const int T = (north + 15) / 16;
    int ti, tj;
    ft_upper((int)blockIdx.x - nfac, T, ti, tj);
    const int a0 = 16 * ti, b0 = 16 * tj;
    const v4d t = ft_tile((r + 3) & ~3,
        [&](int m, int k) { const int kc = min(k, r - 1); const double v = c[(size_t)min(a0 + m, north - 1) * ldc + kc] * w[kc]; return (a0 + m < north && k < r) ? v : 0.0; },
        [&](int k, int n) { const double v = c[(size_t)min(b0 + n, north - 1) * ldc + min(k, r - 1)]; return (b0 + n < north && k < r) ? v : 0.0; }, sred);
    if (ti != tj) {
        if (tid < 64) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int a = a0 + kq + 4 * q, b = b0 + lr;
                if (a < north && b < north) {
                    dm[(size_t)a * north + b] = t[q];
                    dm[(size_t)b * north + a] = t[q];
                }
            }
        }
        return;
    }
    if (tid < 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) sd[(kq + 4 * q) * 17 + lr] = t[q];
    }
    __syncthreads();
    {
        const int p = tid >> 4, q = tid & 15, a = a0 + p, b = a0 + q;
        if (a < north && b < north) dm[(size_t)a * north + b] = 0.5 * (sd[p * 17 + q] + sd[q * 17 + p]);
    }
}

// ---------------------------------------------------------------------------------------------
// AO density into the work buffer of the tile pass: work[0 : n2] <- D_ao (npad x npad, zero padded, BITWISE symmetric: the tiles on
// and right of the diagonal are computed and written with their mirror images), work[n2 : (2 | 3) n2] <- 0 (J / K accumulators).
//   MODE 0: D_ao = L L^T, L = orb (>= nao rows, rp columns, row-major, rows >= nao zero)
//   MODE 1: D_ao = T X^T with T = X Ds (fock_xd_kernel) in scratch (nao x north, row stride north)
// grid: T (T + 1) / 2 upper tiles of the npad-padded matrix.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(FT_NT) void fock_dao_kernel(double *__restrict__ work, const double *__restrict__ a, const double *__restrict__ b,
                                                         int K, int nao, int npad, int with_k) {
    __shared__ double sred[4 * 256];
    __shared__ double sd[16 * 17];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const size_t n2 = (size_t)npad * npad;
    {
        const size_t tot = (size_t)(with_k ? 2 : 1) * n2;
        for (size_t e = (size_t)blockIdx.x * FT_NT + tid; e < tot; e += (size_t)gridDim.x * FT_NT) work[n2 + e] = 0.0;
        if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<unsigned *>(work + 3 * n2 + 1) = 0u;  // the ticket of fock_combine_kernel
    }
    const int T = (npad + 15) / 16;
    int ti, tj;
    ft_upper(blockIdx.x, T, ti, tj);
    const int i0 = 16 * ti, j0 = 16 * tj;
    // A[m][k] = a[(i0 + m) K + k],  B[k][n] = b[(j0 + n) K + k]   (a = b = L in MODE 0;  a = T, b = X in MODE 1)
    const v4d t = ft_tile((K + 3) & ~3,
        [&](int m, int k) { const double v = a[(size_t)min(i0 + m, nao - 1) * K + min(k, K - 1)]; return (i0 + m < nao && k < K) ? v : 0.0; },
        [&](int k, int n) { const double v = b[(size_t)min(j0 + n, nao - 1) * K + min(k, K - 1)]; return (j0 + n < nao && k < K) ? v : 0.0; }, sred);
    if (ti != tj) {
        if (tid < 64) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + kq + 4 * q, j = j0 + lr;
                if (i < npad && j < npad) {
                    work[(size_t)i * npad + j] = t[q];
                    work[(size_t)j * npad + i] = t[q];
                }
            }
        }
        return;
    }
    if (tid < 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) sd[(kq + 4 * q) * 17 + lr] = t[q];
    }
    __syncthreads();
    {
        const int p = tid >> 4, q = tid & 15, i = i0 + p, j = i0 + q;
        if (i < npad && j < npad) work[(size_t)i * npad + j] = 0.5 * (sd[p * 17 + q] + sd[q * 17 + p]);
    }
}

// T = X Ds (nao x north), Ds = (D + D^T) / 2: the first half of X D X^T for a density that comes without its orbital factor.
// grid (ceil(nao / 16), ceil(north / 16)).
__global__ __launch_bounds__(FT_NT) void fock_xd_kernel(double *__restrict__ t_out, const double *__restrict__ x, const double *__restrict__ dm,
                                                        int nao, int north) {
    __shared__ double sred[4 * 256];
    const int lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4;
    const int i0 = 16 * blockIdx.x, c0 = 16 * blockIdx.y;
    const v4d t = ft_tile((north + 3) & ~3,
        [&](int m, int k) { const double v = x[(size_t)min(i0 + m, nao - 1) * north + min(k, north - 1)]; return (i0 + m < nao && k < north) ? v : 0.0; },
        [&](int k, int n) {
            const int kc = min(k, north - 1), cc = min(c0 + n, north - 1);
            const double v = 0.5 * (dm[(size_t)kc * north + cc] + dm[(size_t)cc * north + kc]);
            return (k < north && c0 + n < north) ? v : 0.0;
        }, sred);
    if (threadIdx.x < 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + kq + 4 * q, cidx = c0 + lr;
            if (i < nao && cidx < north) t_out[(size_t)i * north + cidx] = t[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// combine.  M = (Wj + Wj^T) - (Wk + Wk^T) / 2 + V  (AO basis) -> scratch m (npad-strided); the two energy traces
// en[0] = 1/2 sum D_ao J, en[1] = -1/4 sum D_ao K: per-block partial sums in a fixed order, the LAST block to arrive (atomic ticket)
// adds the partials up in block order -- bit-reproducible.  jout (nao x nao, optional) <- J.
// D_ao is bitwise symmetric (fock_dao_kernel), so sum_ij D_ij (W_ij + W_ji) = 2 sum_ij D_ij W_ij: no transposed read for the traces.
// ---------------------------------------------------------------------------------------------
constexpr int FC_NB = 64;  // blocks

// HAS_V: 0 no V, 1 a symmetric AO matrix, 2 the RAW cross-block sums of dqc_grid_vxc_raw (V = (M + M^T) / 2; fixed-point integers of
// scale vscale in deterministic mode)
template <bool DET, bool WITH_K, int HAS_V>
__global__ __launch_bounds__(FT_NT) void fock_combine_kernel(double *__restrict__ m, double *__restrict__ en, double *__restrict__ jout,
                                                             const double *__restrict__ work, const double *__restrict__ v, int ldv, int nao,
                                                             int npad, const double *__restrict__ dscp, double *__restrict__ part,
                                                             unsigned *__restrict__ ticket, double vscale) {
    __shared__ double red[2][FT_WAVES];
    __shared__ bool last;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t n2 = (size_t)npad * npad;
    // deterministic mode: the accumulators hold 64-bit integers in units of 1 / scale, scale = 2^k (jk_det_scale_kernel): the inverse
    // is exact.  Compile-time DET / WITH_K / HAS_V: no branch between the loads
    const double dinv = DET ? 1.0 / *dscp : 0.0;
    const double *wj = work + n2, *wk = work + 2 * n2;
    auto dval = [&](double stored) { return DET ? (double)__double_as_longlong(stored) * dinv : stored; };
    const int tot = nao * nao;
    constexpr int U = 4;
    double sj = 0.0, sk = 0.0;
    for (int e0 = blockIdx.x * FT_NT + tid; e0 < tot; e0 += FC_NB * FT_NT * U) {
        double d[U], ja[U], jb[U], ka[U], kb[U], vv[U], vt[U];
        const double vinv = (HAS_V == 2 && DET) ? 1.0 / vscale : 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = min(e0 + u * FC_NB * FT_NT, tot - 1), i = e / nao, j = e - i * nao;
            d[u] = work[(size_t)i * npad + j];
            ja[u] = wj[(size_t)i * npad + j];
            jb[u] = wj[(size_t)j * npad + i];
            ka[u] = WITH_K ? wk[(size_t)i * npad + j] : 0.0;
            kb[u] = WITH_K ? wk[(size_t)j * npad + i] : 0.0;
            vv[u] = HAS_V ? v[(size_t)i * ldv + j] : 0.0;
            vt[u] = HAS_V == 2 ? v[(size_t)j * ldv + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = e0 + u * FC_NB * FT_NT;
            if (e < tot) {
                const int i = e / nao, j = e - i * nao;
                const double j1 = dval(ja[u]), jv = j1 + dval(jb[u]);
                double mv = jv;
                sj += d[u] * j1;
                if (WITH_K) {
                    const double k1 = dval(ka[u]);
                    mv -= 0.5 * (k1 + dval(kb[u]));
                    sk += d[u] * k1;
                }
                if (HAS_V == 1) mv += vv[u];
                if (HAS_V == 2) {
                    const double va = DET ? (double)__double_as_longlong(vv[u]) * vinv : vv[u];
                    const double vb = DET ? (double)__double_as_longlong(vt[u]) * vinv : vt[u];
                    mv += 0.5 * (va + vb);
                }
                m[(size_t)i * npad + j] = mv;
                if (jout) jout[e] = jv;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sj += __shfl_down(sj, o);
        sk += __shfl_down(sk, o);
    }
    if (lane == 0) { red[0][wave] = sj; red[1][wave] = sk; }
    __syncthreads();
    if (tid == 0) {
        part[2 * blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        part[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        __threadfence();
        last = atomicAdd(ticket, 1u) == FC_NB - 1;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        double a = 0.0, b = 0.0;
        for (int q = 0; q < FC_NB; q++) {
            a += __hip_atomic_load(&part[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b += __hip_atomic_load(&part[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        en[0] = a;          // = 1/2 . 2 sum D W
        en[1] = -0.5 * b;   // = -1/4 . 2 sum D W
        *ticket = 0u;       // (ready for the next launch on this work buffer)
    }
}

// ---------------------------------------------------------------------------------------------
// finish.  T = X^T M (north x nao) -> scratch (fock_xtm_kernel), then F = sym(T X) + core: the tiles on and right of the diagonal,
// written with their mirror images (fock_out_kernel).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FT_NT) void fock_xtm_kernel(double *__restrict__ t_out, const double *__restrict__ x, const double *__restrict__ m,
                                                         int nao, int north, int npad) {
    __shared__ double sred[4 * 256];
    const int lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4;
    const int a0 = 16 * blockIdx.x, j0 = 16 * blockIdx.y;
    // A[mm][k] = X[k][a0 + mm] (16 consecutive doubles per k: coalesced over lr),  B[k][n] = M[k][j0 + n]
    const v4d t = ft_tile((nao + 3) & ~3,
        [&](int mm, int k) { const double v = x[(size_t)min(k, nao - 1) * north + min(a0 + mm, north - 1)]; return (k < nao && a0 + mm < north) ? v : 0.0; },
        [&](int k, int n) { const double v = m[(size_t)min(k, nao - 1) * npad + min(j0 + n, nao - 1)]; return (k < nao && j0 + n < nao) ? v : 0.0; }, sred);
    if (threadIdx.x < 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int a = a0 + kq + 4 * q, j = j0 + lr;
            if (a < north && j < nao) t_out[(size_t)a * nao + j] = t[q];
        }
    }
}

__global__ __launch_bounds__(FT_NT) void fock_out_kernel(double *__restrict__ fock, const double *__restrict__ tm, const double *__restrict__ x,
                                                         const double *__restrict__ core, int nao, int north) {
    __shared__ double sred[4 * 256];
    __shared__ double sd[16 * 17];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int T = (north + 15) / 16;
    int ti, tj;
    ft_upper(blockIdx.x, T, ti, tj);
    const int a0 = 16 * ti, b0 = 16 * tj;
    const v4d t = ft_tile((nao + 3) & ~3,
        [&](int mm, int k) { const double v = tm[(size_t)min(a0 + mm, north - 1) * nao + min(k, nao - 1)]; return (a0 + mm < north && k < nao) ? v : 0.0; },
        [&](int k, int n) { const double v = x[(size_t)min(k, nao - 1) * north + min(b0 + n, north - 1)]; return (k < nao && b0 + n < north) ? v : 0.0; }, sred);
    if (ti != tj) {
        if (tid < 64) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int a = a0 + kq + 4 * q, b = b0 + lr;
                if (a < north && b < north) {
                    fock[(size_t)a * north + b] = t[q] + (core ? core[(size_t)a * north + b] : 0.0);
                    fock[(size_t)b * north + a] = t[q] + (core ? core[(size_t)b * north + a] : 0.0);
                }
            }
        }
        return;
    }
    if (tid < 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) sd[(kq + 4 * q) * 17 + lr] = t[q];
    }
    __syncthreads();
    {
        const int p = tid >> 4, q = tid & 15, a = a0 + p, b = a0 + q;
        if (a < north && b < north) fock[(size_t)a * north + b] = 0.5 * (sd[p * 17 + q] + sd[q * 17 + p]) + (core ? core[(size_t)a * north + b] : 0.0);
    }
}

}  // namespace dqc

extern "C" {

int dqc_fock_max_nao(void) { return 1024; }

// work buffer layout (dqc_jk_work_doubles): [0, n2) AO density | [n2, 2 n2) J accumulator | [2 n2, 3 n2) K accumulator |
// [3 n2, 3 n2 + 8): slot 0 the fixed-point scale of the deterministic mode, slot 1 the ticket of fock_combine_kernel |
// [3 n2 + 8, 4 n2 + 8) M | [4 n2 + 8, 5 n2 + 8) T | then 2 FC_NB partial sums

int dqc_fock_factor(double *d_orb, double *d_orbt, const double *d_x, const double *d_c, int ldc, const double *d_w, int nao, int north,
                    int r, int ld, int rp, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    if (nao > dqc_fock_max_nao() || north > nao || north <= 0 || r <= 0 || r > rp || rp % 16 || ld < nao || ld % 16 || ldc < r) {
        set_error("dqc_fock_factor: needs 0 < north <= nao <= 1024, 0 < r <= rp (a multiple of 16), ld >= nao (a multiple of 16), ldc >= r");
        return DQC_EINVAL;
    }
    hipLaunchKernelGGL(fock_factor_kernel, dim3(ld / 16, rp / 16), dim3(FT_NT), 0, (hipStream_t)stream, d_orb, d_orbt, d_x, d_c, ldc, d_w, nao,
                       north, r, ld, rp);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_fock_orb2dm(double *d_dm, double *d_orb, double *d_orbt, const double *d_x, const double *d_c, int ldc, const double *d_w, int nao,
                    int north, int r, int ld, int rp, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    if (nao > dqc_fock_max_nao() || north > nao || north <= 0 || r <= 0 || r > rp || rp % 16 || ld < nao || ld % 16 || ldc < r) {
        set_error("dqc_fock_orb2dm: needs 0 < north <= nao <= 1024, 0 < r <= rp (a multiple of 16), ld >= nao (a multiple of 16), ldc >= r");
        return DQC_EINVAL;
    }
    const int T = (north + 15) / 16;
    hipLaunchKernelGGL(fock_orb2dm_kernel, dim3((ld / 16) * (rp / 16) + T * (T + 1) / 2), dim3(FT_NT), 0, (hipStream_t)stream, d_dm, d_orb, d_orbt,
                       d_x, d_c, ldc, d_w, nao, north, r, ld, rp);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_fock_prep(double *d_work, const double *d_dm, const double *d_x, const double *d_orb, int rp, int nao, int north, int with_k,
                  void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    if (nao > dqc_fock_max_nao() || north > nao || north <= 0) { set_error("dqc_fock_prep: needs 0 < north <= nao <= 1024"); return DQC_EINVAL; }
    if (!d_orb && (!d_dm || !d_x)) { set_error("dqc_fock_prep: either the AO-basis factor or (dm, x)"); return DQC_EINVAL; }
    if (d_orb && rp <= 0) { set_error("dqc_fock_prep: factor width must be positive"); return DQC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const size_t n2 = (size_t)npad * npad;
    const int T = (npad + 15) / 16;
    const dim3 grid(T * (T + 1) / 2);
    if (d_orb) {
        hipLaunchKernelGGL(fock_dao_kernel<0>, grid, dim3(FT_NT), 0, st, d_work, d_orb, d_orb, rp, nao, npad, with_k);
    } else {
        double *d_t = d_work + 4 * n2 + 8;
        hipLaunchKernelGGL(fock_xd_kernel, dim3((nao + 15) / 16, (north + 15) / 16), dim3(FT_NT), 0, st, d_t, d_x, d_dm, nao, north);
        DQC_CHECK_LAUNCH();
        hipLaunchKernelGGL(fock_dao_kernel<1>, grid, dim3(FT_NT), 0, st, d_work, d_t, d_x, north, nao, npad, with_k);
    }
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

static int fock_finish_impl(double *d_fock, double *d_energies, double *d_j_ao, double *d_work, const double *d_vxc_ao, int ldv,
                            const double *d_core, const double *d_x, int nao, int north, int with_k, int vraw, double vscale, void *stream);

int dqc_fock_finish(double *d_fock, double *d_energies, double *d_j_ao, double *d_work, const double *d_vxc_ao, int ldv,
                    const double *d_core, const double *d_x, int nao, int north, int with_k, void *stream) {
    return fock_finish_impl(d_fock, d_energies, d_j_ao, d_work, d_vxc_ao, ldv, d_core, d_x, nao, north, with_k, 0, 0.0, stream);
}

int dqc_fock_finish_vraw(double *d_fock, double *d_energies, double *d_work, const double *d_vxc_raw, int ldv, double vscale,
                         const double *d_core, const double *d_x, int nao, int north, void *stream) {
    // the Kohn-Sham finish on the raw sums of dqc_grid_vxc_raw (vscale: what that call returned)
    if (!d_vxc_raw) { dqc::set_error("dqc_fock_finish_vraw: null matrix"); return DQC_EINVAL; }
    return fock_finish_impl(d_fock, d_energies, nullptr, d_work, d_vxc_raw, ldv, d_core, d_x, nao, north, 0, 1, vscale, stream);
}

static int fock_finish_impl(double *d_fock, double *d_energies, double *d_j_ao, double *d_work, const double *d_vxc_ao, int ldv,
                            const double *d_core, const double *d_x, int nao, int north, int with_k, int vraw, double vscale, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    if (nao > dqc_fock_max_nao() || north > nao || north <= 0) { set_error("dqc_fock_finish: needs 0 < north <= nao <= 1024"); return DQC_EINVAL; }
    if (!d_fock || !d_energies || !d_work || !d_x) { set_error("dqc_fock_finish: null argument"); return DQC_EINVAL; }
    if (d_vxc_ao && ldv < nao) { set_error("dqc_fock_finish: ldv < nao"); return DQC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const size_t n2 = (size_t)npad * npad;
    const double *dscp = deterministic_mode() ? d_work + 3 * n2 : nullptr;
    unsigned *ticket = reinterpret_cast<unsigned *>(d_work + 3 * n2 + 1);
    double *d_m = d_work + 3 * n2 + 8, *d_t = d_work + 4 * n2 + 8, *d_part = d_work + 5 * n2 + 8;
    // (the ticket is zero when the kernel starts: dqc_fock_prep, which every build runs first on this buffer, resets it)
#define DQC_FK_COMBINE(D, K, V) \
    hipLaunchKernelGGL((fock_combine_kernel<D, K, V>), dim3(FC_NB), dim3(FT_NT), 0, st, d_m, d_energies, d_j_ao, d_work, d_vxc_ao, ldv, nao, npad, dscp, d_part, ticket, vscale);
    const bool det = dscp != nullptr, wk_ = with_k != 0, hv = d_vxc_ao != nullptr;
    if (vraw && (vscale != 0.0) != det) { set_error("dqc_fock_finish_vraw: the scale does not match the deterministic mode"); return DQC_EINVAL; }
    if (vraw) {
        if (det) DQC_FK_COMBINE(true, false, 2) else DQC_FK_COMBINE(false, false, 2)
    } else if (det) {
        if (wk_) { if (hv) DQC_FK_COMBINE(true, true, 1) else DQC_FK_COMBINE(true, true, 0) }
        else { if (hv) DQC_FK_COMBINE(true, false, 1) else DQC_FK_COMBINE(true, false, 0) }
    } else {
        if (wk_) { if (hv) DQC_FK_COMBINE(false, true, 1) else DQC_FK_COMBINE(false, true, 0) }
        else { if (hv) DQC_FK_COMBINE(false, false, 1) else DQC_FK_COMBINE(false, false, 0) }
    }
#undef DQC_FK_COMBINE
    DQC_CHECK_LAUNCH();
    hipLaunchKernelGGL(fock_xtm_kernel, dim3((north + 15) / 16, (nao + 15) / 16), dim3(FT_NT), 0, st, d_t, d_x, d_m, nao, north, npad);
    DQC_CHECK_LAUNCH();
    const int T = (north + 15) / 16;
    hipLaunchKernelGGL(fock_out_kernel, dim3(T * (T + 1) / 2), dim3(FT_NT), 0, st, d_fock, d_t, d_x, d_core, nao, north);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
