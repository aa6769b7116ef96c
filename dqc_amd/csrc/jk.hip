// jk.hip -- Coulomb (J) and exchange (K) matrices from the tile-stored ERIs.
// Replaces  einsum("ij,ijkl->kl", dm, el_mat)  and  einsum("il,ijkl->ijk", dm, el_mat).sum(-3)
// (reference: HamiltonCGTO.get_elrep / get_exchange, dqc/hamilton/hcgto.py:204-241), which stream the
// dense nao^4 tensor twice on the CPU.  Here the 8-fold-unique tiles are streamed from HBM exactly once
// per call (J and K fused); this is a pure bandwidth kernel: 4096 doubles per tile, 2 (J) or 6 (J+K)
// FMAs per integral.
//
// A tile (I>=J, K>=L, IJ>=KL) holds g[i][j][k][l]; with f = (I==J ? 1/2 : 1)(K==L ? 1/2 : 1)(IJ==KL ? 1/2 : 1)
//   Jacc[I,J] += 2f g.D[K,L]       Jacc[K,L] += 2f g.D[I,J]
//   Kacc[I,L] +=  f g.D[J,K]       Kacc[J,L] +=  f g.D[I,K]
//   Kacc[I,K] +=  f g.D[J,L]       Kacc[J,K] +=  f g.D[I,L]
// and finally J = Jacc + Jacc^T, K = Kacc + Kacc^T (the 8 index permutations of every unique integral).
//
// Mapping: one 256-thread workgroup per tile, viewed as a 64x64 matrix M[(ij)][(kl)]; thread t owns the
// 4x4 patch rows 4(t/16).., cols 4(t%16).. (16-byte loads, 512-byte rows per 16 lanes).  J needs row sums
// (16-lane DPP reduce) and column sums (2 shuffles + a 4-wave LDS combine).  For K the patch is parked in
// LDS (row stride 65) and the four contractions re-read it with output-major lane mappings.
#include "common.hpp"

namespace dqc {

DQC_DEV void decode_tri(long long t, int &a, int &b) {  // t = a(a+1)/2 + b, b <= a
    long long r = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > t) r--;
    while ((r + 1) * (r + 2) / 2 <= t) r++;
    a = (int)r;
    b = (int)(t - r * (r + 1) / 2);
}

__global__ void jk_prep_kernel(double *__restrict__ work, const double *__restrict__ dm, int nao, int npad, int with_k) {
    const size_t n2 = (size_t)npad * npad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (size_t)gridDim.x * blockDim.x) {
        const int i = e / npad, j = e % npad;
        double v = 0.0;
        if (i < nao && j < nao) v = 0.5 * (dm[(size_t)i * nao + j] + dm[(size_t)j * nao + i]);
        work[e] = v;
        work[n2 + e] = 0.0;
        if (with_k) work[2 * n2 + e] = 0.0;
    }
}

__global__ void jk_finish_kernel(double *__restrict__ J, double *__restrict__ K, const double *__restrict__ work,
                                 int nao, int npad) {
    const size_t n2 = (size_t)npad * npad;
    const size_t tot = (size_t)nao * nao;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int i = e / nao, j = e % nao;
        J[e] = work[n2 + (size_t)i * npad + j] + work[n2 + (size_t)j * npad + i];
        if (K) K[e] = work[2 * n2 + (size_t)i * npad + j] + work[2 * n2 + (size_t)j * npad + i];
    }
}

template <bool WITH_K>
__global__ __launch_bounds__(256, WITH_K ? 4 : 1) void jk_tiles_kernel(const double *__restrict__ tiles, double *__restrict__ work,
                                                       int npad, long long ntiles) {
    constexpr int LDT = 68;  // row stride of the tile parked in LDS: 16-byte aligned rows, bank = 4 row + col (mod 32)
    __shared__ double s_col[4][64];
    __shared__ __attribute__((aligned(16))) double s_g[WITH_K ? 64 * LDT : 2];
    __shared__ double s_d[WITH_K ? 4 : 1][72];  // D[J,K], D[I,K], D[J,L], D[I,L]; element (a, v) at a * 9 + v
    const size_t n2 = (size_t)npad * npad;
    const double *Dp = work;
    double *Jacc = work + n2, *Kacc = work + 2 * n2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = 4 * (t >> 4), c0 = 4 * (t & 15);

    for (long long T = blockIdx.x; T < ntiles; T += gridDim.x) {
        int IJ, KL, I, J, K, L;
        decode_tri(T, IJ, KL);
        decode_tri(IJ, I, J);
        decode_tri(KL, K, L);
        const double f = (I == J ? 0.5 : 1.0) * (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const double *tp = tiles + (size_t)T * DQC_TILE_SZ;
        // D[K,L](k,l) for the 4 columns, D[I,J](i,j) for the 4 rows of this thread's patch
        const int kk = c0 >> 3, l0 = c0 & 7, ii = r0 >> 3, j0 = r0 & 7;
        const double *dklp = Dp + (size_t)(K * 8 + kk) * npad + L * 8 + l0;
        const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
        double dkl[4], dij[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { dkl[q] = dklp[q]; dij[q] = dijp[q]; }
        double rs[4] = {0, 0, 0, 0}, cs[4] = {0, 0, 0, 0};
        double g[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double2 a = *reinterpret_cast<const double2 *>(tp + (r0 + r) * 64 + c0);
            const double2 b = *reinterpret_cast<const double2 *>(tp + (r0 + r) * 64 + c0 + 2);
            g[r][0] = a.x; g[r][1] = a.y; g[r][2] = b.x; g[r][3] = b.y;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                rs[r] += g[r][c] * dkl[c];
                cs[c] += g[r][c] * dij[r];
            }
        if (WITH_K) {
            // (the barrier at the end of the previous iteration has retired that tile's LDS readers)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                *reinterpret_cast<double2 *>(&s_g[(r0 + r) * LDT + c0]) = make_double2(g[r][0], g[r][1]);
                *reinterpret_cast<double2 *>(&s_g[(r0 + r) * LDT + c0 + 2]) = make_double2(g[r][2], g[r][3]);
            }
            // D blocks: thread t loads element (t&63) of block (t>>6)
            {
                const int blk = t >> 6, e = t & 63, x = e >> 3, y = e & 7;
                const int R = (blk & 1) ? I : J, Cb = (blk & 2) ? L : K;
                s_d[blk][x * 9 + y] = Dp[(size_t)(R * 8 + x) * npad + Cb * 8 + y];
            }
        }
        // ---- J: row sums over the 16 lanes of a row group, column sums over the 16 row groups ----
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = rs[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            rs[r] = v;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double v = cs[c];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            cs[c] = v;
        }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                atomicAdd(&Jacc[(size_t)(I * 8 + ii) * npad + J * 8 + j0 + r], 2.0 * f * rs[r]);
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_col[wave][c0 + c] = cs[c];
        }
        __syncthreads();
        if (t < 64) {
            const double v = s_col[0][t] + s_col[1][t] + s_col[2][t] + s_col[3][t];
            atomicAdd(&Jacc[(size_t)(K * 8 + (t >> 3)) * npad + L * 8 + (t & 7)], 2.0 * f * v);
        }
        if (WITH_K) {
            // four contractions; thread = output o (64) x partial group pg (4), 16 of the 64 terms each.  Which 16 is chosen
            // per contraction so that the 32 lanes of a half-wave (x fixed, y = 0..7, pg = 0..3) hit 32 different LDS banks at
            // every step (row stride 68: bank = 4 row + col mod 32; yh = y >> 2):
            //   K1  row = 8x + a, col = 8v + y : v = pg + 4u, all a               -> bank = 8 pg + y + const
            //   K2  row = 8a + x, col = 8v + y : v = pg + 4u, all a               -> bank = 8 pg + y + const
            //   K3  row = 8x + a, col = 8y + v : v = pg + 4 (u ^ yh), all a       -> 8 (y & 3) + pg + 4 (u ^ yh) + const
            //   K4  row = 8a + x, col = 8y + v : v = pg + 4 (u ^ yh), all a       -> same
            // (with the straightforward split every read had 3- to 4-way conflicts and the K part cost as much as the stream)
            const int o = t >> 2, pg = t & 3, x = o >> 3, y = o & 7, yh = y >> 2;
            double k1 = 0, k2 = 0, k3 = 0, k4 = 0;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = pg + 4 * u, q4 = pg + 4 * (u ^ yh);
#pragma unroll
                for (int a = 0; a < 8; a++) {
                    k1 += s_g[(x * 8 + a) * LDT + q * 8 + y] * s_d[0][a * 9 + q];    // g[x][a][v=q][y]  D[J,K](a,v)
                    k2 += s_g[(a * 8 + x) * LDT + q * 8 + y] * s_d[1][a * 9 + q];    // g[a][x][v=q][y]  D[I,K](a,v)
                    k3 += s_g[(x * 8 + a) * LDT + y * 8 + q4] * s_d[2][a * 9 + q4];  // g[x][a][y][v=q4] D[J,L](a,v)
                    k4 += s_g[(a * 8 + x) * LDT + y * 8 + q4] * s_d[3][a * 9 + q4];  // g[a][x][y][v=q4] D[I,L](a,v)
                }
            }
            k1 += __shfl_xor(k1, 1); k1 += __shfl_xor(k1, 2);
            k2 += __shfl_xor(k2, 1); k2 += __shfl_xor(k2, 2);
            k3 += __shfl_xor(k3, 1); k3 += __shfl_xor(k3, 2);
            k4 += __shfl_xor(k4, 1); k4 += __shfl_xor(k4, 2);
            if (pg == 0) {
                atomicAdd(&Kacc[(size_t)(I * 8 + x) * npad + L * 8 + y], f * k1);
                atomicAdd(&Kacc[(size_t)(J * 8 + x) * npad + L * 8 + y], f * k2);
                atomicAdd(&Kacc[(size_t)(I * 8 + x) * npad + K * 8 + y], f * k3);
                atomicAdd(&Kacc[(size_t)(J * 8 + x) * npad + K * 8 + y], f * k4);
            }
        }
        __syncthreads();  // s_col / s_g reuse
    }
}

}  // namespace dqc

extern "C" {

size_t dqc_jk_work_doubles(int nao) {
    const size_t npad = (size_t)(nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    return 3 * npad * npad;
}

int dqc_jk_from_tiles(double *d_J, double *d_K, const double *d_tiles, const double *d_dm, int nao,
                      double *d_work, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const long long ntiles = (long long)dqc_eri_tile_count(nao);
    const int with_k = d_K != nullptr;
    hipLaunchKernelGGL(jk_prep_kernel, dim3(64), dim3(256), 0, st, d_work, d_dm, nao, npad, with_k);
    DQC_CHECK_LAUNCH();
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 256 * 16);
    if (with_k)
        hipLaunchKernelGGL(jk_tiles_kernel<true>, dim3(grid), dim3(256), 0, st, d_tiles, d_work, npad, ntiles);
    else
        hipLaunchKernelGGL(jk_tiles_kernel<false>, dim3(grid), dim3(256), 0, st, d_tiles, d_work, npad, ntiles);
    DQC_CHECK_LAUNCH();
    hipLaunchKernelGGL(jk_finish_kernel, dim3(64), dim3(256), 0, st, d_J, d_K, d_work, nao, npad);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
