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
#include <cstdlib>

#include "common.hpp"

namespace dqc {

DQC_DEV void decode_tri(long long t, int &a, int &b) {  // t = a(a+1)/2 + b, b <= a
    long long r = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > t) r--;
    while ((r + 1) * (r + 2) / 2 <= t) r++;
    a = (int)r;
    b = (int)(t - r * (r + 1) / 2);
}

// this thread's 4 x 4 patch -- rows r0 .. r0 + 3, columns c0 .. c0 + 3 of the FULL 64 x 64 view g[(i,j)][(k,l)] -- of tile
// (IJ, KL) of the packed store (common.hpp).  A plain tile is read as four pairs of 16-byte loads.  A tile with a diagonal block
// pair keeps only its a >= b rows / columns: every element is fetched through its packed index, the a < b ones from their
// a > b twins -- the same cache lines, so HBM delivers the packed size -- and the arithmetic downstream (the 1/2 weights of
// the diagonal pairs) is that of the full view, unchanged.
DQC_DEV void tile_load_patch(const double *__restrict__ tiles, int IJ, int KL, int I, int J, int K, int L, int r0, int c0, double2 &a0, double2 &b0, double2 &a1, double2 &b1, double2 &a2,
                             double2 &b2, double2 &a3, double2 &b3, const TileLay &ly) {
    const bool dr = I == J, dc = K == L;
    const double *tp = tiles + tile_base(I, J, K, KL, ly);
    // the last block row is stored at its true width (common.hpp: TileLay): rows of the view past the prefix that belongs to AOs
    // do not exist in the store -- they read as zeros.  (8 wl is a multiple of the patch height: a patch is in or out as a whole.)
    const bool trunc = I == ly.last && ly.wl < 8;
    if (!dr && !dc) {
        if (trunc && r0 >= 8 * ly.wl) {
            a0 = b0 = a1 = b1 = a2 = b2 = a3 = b3 = make_double2(0.0, 0.0);
            return;
        }
        const double *q = tp + r0 * 64 + c0;
        a0 = *reinterpret_cast<const double2 *>(q);       b0 = *reinterpret_cast<const double2 *>(q + 2);
        a1 = *reinterpret_cast<const double2 *>(q + 64);  b1 = *reinterpret_cast<const double2 *>(q + 66);
        a2 = *reinterpret_cast<const double2 *>(q + 128); b2 = *reinterpret_cast<const double2 *>(q + 130);
        a3 = *reinterpret_cast<const double2 *>(q + 192); b3 = *reinterpret_cast<const double2 *>(q + 194);
        return;
    }
    const int il = r0 >> 3, j0 = r0 & 7, kl = c0 >> 3, l0 = c0 & 7;
    const double2 z2 = make_double2(0.0, 0.0);
    // row x of the patch (view row r0 + x = local pair (il, j0 + x)) exists unless the pair holds a padding AO of the last block
    const bool ok0 = !trunc || (dr ? max(il, j0) < ly.wl : il < ly.wl), ok1 = !trunc || (dr ? max(il, j0 + 1) < ly.wl : il < ly.wl);
    const bool ok2 = !trunc || (dr ? max(il, j0 + 2) < ly.wl : il < ly.wl), ok3 = !trunc || (dr ? max(il, j0 + 3) < ly.wl : il < ly.wl);
    if (!dc) {  // only the rows are packed: the four columns stay contiguous and 16-byte aligned
        const double *q0 = tp + tile_pidx(true, il, j0) * 64 + c0, *q1 = tp + tile_pidx(true, il, j0 + 1) * 64 + c0;
        const double *q2 = tp + tile_pidx(true, il, j0 + 2) * 64 + c0, *q3 = tp + tile_pidx(true, il, j0 + 3) * 64 + c0;
        a0 = b0 = a1 = b1 = a2 = b2 = a3 = b3 = z2;
        if (ok0) { a0 = *reinterpret_cast<const double2 *>(q0); b0 = *reinterpret_cast<const double2 *>(q0 + 2); }
        if (ok1) { a1 = *reinterpret_cast<const double2 *>(q1); b1 = *reinterpret_cast<const double2 *>(q1 + 2); }
        if (ok2) { a2 = *reinterpret_cast<const double2 *>(q2); b2 = *reinterpret_cast<const double2 *>(q2 + 2); }
        if (ok3) { a3 = *reinterpret_cast<const double2 *>(q3); b3 = *reinterpret_cast<const double2 *>(q3 + 2); }
        return;
    }
    const int C = 36;
    const int pc0 = tile_pidx(dc, kl, l0), pc1 = tile_pidx(dc, kl, l0 + 1), pc2 = tile_pidx(dc, kl, l0 + 2), pc3 = tile_pidx(dc, kl, l0 + 3);
    const double *q0 = tp + tile_pidx(dr, il, j0) * C, *q1 = tp + tile_pidx(dr, il, j0 + 1) * C;
    const double *q2 = tp + tile_pidx(dr, il, j0 + 2) * C, *q3 = tp + tile_pidx(dr, il, j0 + 3) * C;
    a0 = b0 = a1 = b1 = a2 = b2 = a3 = b3 = z2;
    if (ok0) { a0 = make_double2(q0[pc0], q0[pc1]); b0 = make_double2(q0[pc2], q0[pc3]); }
    if (ok1) { a1 = make_double2(q1[pc0], q1[pc1]); b1 = make_double2(q1[pc2], q1[pc3]); }
    if (ok2) { a2 = make_double2(q2[pc0], q2[pc1]); b2 = make_double2(q2[pc2], q2[pc3]); }
    if (ok3) { a3 = make_double2(q3[pc0], q3[pc1]); b3 = make_double2(q3[pc2], q3[pc3]); }
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
                                 int nao, int npad, const double *__restrict__ dscp) {
    const double dsc = dscp ? *dscp : 0.0;
    const size_t n2 = (size_t)npad * npad;
    const size_t tot = (size_t)nao * nao;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int i = e / nao, j = e % nao;
        J[e] = det_value(work[n2 + (size_t)i * npad + j], dsc) + det_value(work[n2 + (size_t)j * npad + i], dsc);
        if (K) K[e] = det_value(work[2 * n2 + (size_t)i * npad + j], dsc) + det_value(work[2 * n2 + (size_t)j * npad + i], dsc);
    }
}

// deterministic mode: the fixed-point scale of the J / K accumulators of one call, 2^k with 2 gmax sum|D| 2^k < 2^61.
// |(ij|kl)| <= max_i (ii|ii) = gmax (Cauchy-Schwarz twice), so no accumulator -- each a partial sum of g D terms with
// weights <= 2 -- can exceed 2 gmax sum_ij |D_ij|.  One block, fixed summation order: the scale itself is reproducible.
// dms: nmat symmetrised, padded matrices (n2 doubles apart) as the prep kernels leave them in the work buffer.
__global__ __launch_bounds__(256) void jk_det_scale_kernel(double *__restrict__ slot, const double *__restrict__ dms, int nmat,
                                                           size_t n2, const double *__restrict__ tiles, int nao) {
    __shared__ double red[256];
    const int t = threadIdx.x;
    double smax = 0.0;
    for (int q = 0; q < nmat; q++) {
        double sacc = 0.0;
        for (size_t e = t; e < n2; e += 256) sacc += fabs(dms[q * n2 + e]);
        red[t] = sacc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) red[t] += red[t + o];
            __syncthreads();
        }
        smax = fmax(smax, red[0]);
        __syncthreads();
    }
    double g = 0.0;
    for (int i = t; i < nao; i += 256) {
        const long long b = i >> 3, IJ = b * (b + 1) / 2 + b;  // (ii|ii): tile (IJ, IJ) of the diagonal block pair (b, b)
        const int pa = tile_pidx(true, i & 7, i & 7);
        g = fmax(g, fabs(tiles[tile_base((int)b, (int)b, (int)b, (int)IJ, TileLay(nao)) + pa * 36 + pa]));
    }
    red[t] = g;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] = fmax(red[t], red[t + o]);
        __syncthreads();
    }
    if (t == 0) {
        const double bound = 2.0 * fmax(red[0], 1e-300) * fmax(smax, 1e-300);
        slot[0] = exp2(floor(61.0 - log2(bound)));
    }
}

template <bool WITH_K>
__global__ __launch_bounds__(256, WITH_K ? 4 : 1) void jk_tiles_kernel(const double *__restrict__ dscp, const double *__restrict__ tiles,
                                                       double *__restrict__ work, int npad, long long ntiles, int nao) {
    const TileLay ly(nao);
    const double dsc = dscp ? *dscp : 0.0;  // deterministic mode: fixed-point scale of the accumulators (common.hpp: acc_add)
    constexpr int LDT = 68;  // row stride of the tile parked in LDS: 16-byte aligned rows, bank = 4 row + col (mod 32)
    __shared__ double s_col[4][64];
    __shared__ __attribute__((aligned(16))) double s_g[WITH_K ? 64 * LDT : 2];
    // D[J,K], D[I,K] | D[J,L], D[I,L]; element (a, v) of a block at a * 9 + v, the two blocks of a pair interleaved so that
    // one ds_read_b128 fetches both (the exchange part is LDS-read-bound)
    __shared__ __attribute__((aligned(16))) double s_d[WITH_K ? 2 : 1][72][2];
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
        // D[K,L](k,l) for the 4 columns, D[I,J](i,j) for the 4 rows of this thread's patch
        const int kk = c0 >> 3, l0 = c0 & 7, ii = r0 >> 3, j0 = r0 & 7;
        const double *dklp = Dp + (size_t)(K * 8 + kk) * npad + L * 8 + l0;
        const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
        double dkl[4], dij[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { dkl[q] = dklp[q]; dij[q] = dijp[q]; }
        double rs[4] = {0, 0, 0, 0}, cs[4] = {0, 0, 0, 0};
        double g[4][4];
        {
            double2 ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3;
            tile_load_patch(tiles, IJ, KL, I, J, K, L, r0, c0, ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3, ly);
            g[0][0] = ta0.x; g[0][1] = ta0.y; g[0][2] = tb0.x; g[0][3] = tb0.y;
            g[1][0] = ta1.x; g[1][1] = ta1.y; g[1][2] = tb1.x; g[1][3] = tb1.y;
            g[2][0] = ta2.x; g[2][1] = ta2.y; g[2][2] = tb2.x; g[2][3] = tb2.y;
            g[3][0] = ta3.x; g[3][1] = ta3.y; g[3][2] = tb3.x; g[3][3] = tb3.y;
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
                s_d[blk >> 1][x * 9 + y][blk & 1] = Dp[(size_t)(R * 8 + x) * npad + Cb * 8 + y];
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
                acc_add(&Jacc[(size_t)(I * 8 + ii) * npad + J * 8 + j0 + r], 2.0 * f * rs[r], dsc);
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_col[wave][c0 + c] = cs[c];
        }
        __syncthreads();
        if (t < 64) {
            const double v = s_col[0][t] + s_col[1][t] + s_col[2][t] + s_col[3][t];
            acc_add(&Jacc[(size_t)(K * 8 + (t >> 3)) * npad + L * 8 + (t & 7)], 2.0 * f * v, dsc);
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
#pragma unroll 1
            for (int u = 0; u < 2; u++) {  // rolled: fully unrolled, the hoisted LDS reads spill (128-VGPR budget)
                const int q = pg + 4 * u, q4 = pg + 4 * (u ^ yh);
#pragma unroll
                for (int a = 0; a < 8; a++) {
                    typedef double vd2_ __attribute__((ext_vector_type(2)));
                    const vd2_ d12 = *reinterpret_cast<const vd2_ *>(&s_d[0][a * 9 + q][0]);
                    const vd2_ d34 = *reinterpret_cast<const vd2_ *>(&s_d[1][a * 9 + q4][0]);
                    k1 += s_g[(x * 8 + a) * LDT + q * 8 + y] * d12.x;    // g[x][a][v=q][y]  D[J,K](a,v)
                    k2 += s_g[(a * 8 + x) * LDT + q * 8 + y] * d12.y;    // g[a][x][v=q][y]  D[I,K](a,v)
                    k3 += s_g[(x * 8 + a) * LDT + y * 8 + q4] * d34.x;   // g[x][a][y][v=q4] D[J,L](a,v)
                    k4 += s_g[(a * 8 + x) * LDT + y * 8 + q4] * d34.y;   // g[a][x][y][v=q4] D[I,L](a,v)
                }
            }
            k1 += __shfl_xor(k1, 1); k1 += __shfl_xor(k1, 2);
            k2 += __shfl_xor(k2, 1); k2 += __shfl_xor(k2, 2);
            k3 += __shfl_xor(k3, 1); k3 += __shfl_xor(k3, 2);
            k4 += __shfl_xor(k4, 1); k4 += __shfl_xor(k4, 2);
            if (pg == 0) {
                acc_add(&Kacc[(size_t)(I * 8 + x) * npad + L * 8 + y], f * k1, dsc);
                acc_add(&Kacc[(size_t)(J * 8 + x) * npad + L * 8 + y], f * k2, dsc);
                acc_add(&Kacc[(size_t)(I * 8 + x) * npad + K * 8 + y], f * k3, dsc);
                acc_add(&Kacc[(size_t)(J * 8 + x) * npad + K * 8 + y], f * k4, dsc);
            }
        }
        __syncthreads();  // s_col / s_g reuse
    }
}

// ---------------------------------------------------------------------------------------------
// Several right-hand sides in ONE pass over the tiles.  The reference's callers ask for J and K of different density
// matrices back to back -- unrestricted Hartree-Fock: J[D_u + D_d], K[2 D_u], K[2 D_d] (hcgto.py:238-241, hf.py:93-103,
// 198-199); batched density matrices (base_hamilton.py:92-93) -- and each call streamed the whole tile store again.
// Here a tile is loaded once (non-temporal: it is not read again this pass), its 4 x 4 patch stays in registers / LDS, and
//   * the Coulomb part loops over `nj` densities (row / column sums per density, registers reused),
//   * the exchange part contracts NK (1 or 2) densities at once: the tile reads from LDS are shared between them.
// work layout (n2 = npad^2 doubles each):  Dj[nj] | Dk[NK] | Jacc[nj] | Kacc[NK].
// ---------------------------------------------------------------------------------------------
template <int NK>
__global__ __launch_bounds__(256, NK ? 3 : 1) void jk_multi_kernel(const double *__restrict__ dscp, const double *__restrict__ tiles,
                                                                   double *__restrict__ work, int npad, long long ntiles, int nj, int nao) {
    const TileLay ly(nao);
    const double dsc = dscp ? *dscp : 0.0;
    constexpr int LDT = 68;
    constexpr int NKD = NK ? NK : 1;
    __shared__ double s_col[2][4][64];
    __shared__ __attribute__((aligned(16))) double s_g[NK ? 64 * LDT : 2];
    __shared__ __attribute__((aligned(16))) double s_d[4][72][NKD];  // density blocks, the NK exchange densities interleaved: one ds_read_b128 serves both
    const size_t n2 = (size_t)npad * npad;
    const double *Dj = work, *Dk = work + (size_t)nj * n2;
    double *Jacc = work + (size_t)(nj + NK) * n2, *Kacc = Jacc + (size_t)nj * n2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = 4 * (t >> 4), c0 = 4 * (t & 15);
    typedef double vd2 __attribute__((ext_vector_type(2)));

    // the NEXT tile's 8 loads are issued before this tile's work (scalar register set: arrays that live across the
    // back-edge end up in scratch), so a block overlaps its own HBM latency with its LDS / atomic phase
    double2 na0, na1, na2, na3, nb0, nb1, nb2, nb3;
#define JKM_LOAD(TT)                                                                     \
    {                                                                                    \
        int ij_, kl_, i_, j_, k_, l_;                                                    \
        decode_tri((TT), ij_, kl_);                                                      \
        decode_tri(ij_, i_, j_);                                                         \
        decode_tri(kl_, k_, l_);                                                         \
        tile_load_patch(tiles, ij_, kl_, i_, j_, k_, l_, r0, c0, na0, nb0, na1, nb1, na2, nb2, na3, nb3, ly); \
    }
    if ((long long)blockIdx.x < ntiles) JKM_LOAD(blockIdx.x)
    for (long long T = blockIdx.x; T < ntiles; T += gridDim.x) {
        int IJ, KL, I, J, K, L;
        decode_tri(T, IJ, KL);
        decode_tri(IJ, I, J);
        decode_tri(KL, K, L);
        const double f = (I == J ? 0.5 : 1.0) * (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const int kk = c0 >> 3, l0 = c0 & 7, ii = r0 >> 3, j0 = r0 & 7;
        double g[4][4];
        g[0][0] = na0.x; g[0][1] = na0.y; g[0][2] = nb0.x; g[0][3] = nb0.y;
        g[1][0] = na1.x; g[1][1] = na1.y; g[1][2] = nb1.x; g[1][3] = nb1.y;
        g[2][0] = na2.x; g[2][1] = na2.y; g[2][2] = nb2.x; g[2][3] = nb2.y;
        g[3][0] = na3.x; g[3][1] = na3.y; g[3][2] = nb3.x; g[3][3] = nb3.y;
        if (T + gridDim.x < ntiles) JKM_LOAD(T + gridDim.x)
        if (NK) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                *reinterpret_cast<double2 *>(&s_g[(r0 + r) * LDT + c0]) = make_double2(g[r][0], g[r][1]);
                *reinterpret_cast<double2 *>(&s_g[(r0 + r) * LDT + c0 + 2]) = make_double2(g[r][2], g[r][3]);
            }
            const int blk = t >> 6, e = t & 63, x = e >> 3, y = e & 7;
            const int R = (blk & 1) ? I : J, Cb = (blk & 2) ? L : K;
#pragma unroll
            for (int q = 0; q < NK; q++) s_d[blk][x * 9 + y][q] = Dk[q * n2 + (size_t)(R * 8 + x) * npad + Cb * 8 + y];
        }
        // ---- Coulomb: one density at a time
        for (int q = 0; q < nj; q++) {
            const double *dklp = Dj + q * n2 + (size_t)(K * 8 + kk) * npad + L * 8 + l0;
            const double *dijp = Dj + q * n2 + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
            double dkl[4], dij[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { dkl[c] = dklp[c]; dij[c] = dijp[c]; }
            double rs[4] = {0, 0, 0, 0}, cs[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    rs[r] += g[r][c] * dkl[c];
                    cs[c] += g[r][c] * dij[r];
                }
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
            double *Jq = Jacc + q * n2;
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc_add(&Jq[(size_t)(I * 8 + ii) * npad + J * 8 + j0 + r], 2.0 * f * rs[r], dsc);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 4; c++) s_col[q & 1][wave][c0 + c] = cs[c];
            }
            __syncthreads();  // (also publishes s_g / s_d on the first round; s_col is double-buffered across densities)
            if (t < 64) {
                const double v = s_col[q & 1][0][t] + s_col[q & 1][1][t] + s_col[q & 1][2][t] + s_col[q & 1][3][t];
                acc_add(&Jq[(size_t)(K * 8 + (t >> 3)) * npad + L * 8 + (t & 7)], 2.0 * f * v, dsc);
            }
        }
        if (NK) {
            if (nj == 0) __syncthreads();
            // the four exchange contractions (lane mapping and bank analysis: jk_tiles_kernel), NK densities per tile read
            const int o = t >> 2, pg = t & 3, x = o >> 3, y = o & 7, yh = y >> 2;
            double k1[NKD], k2[NKD], k3[NKD], k4[NKD];
#pragma unroll
            for (int q = 0; q < NK; q++) k1[q] = k2[q] = k3[q] = k4[q] = 0.0;
            // (rolled outer loop, inner loop in groups of 4: fully unrolled the compiler hoists all 64 + 64 NK LDS reads and spills)
#pragma unroll 1
            for (int u = 0; u < 2; u++) {
                const int qq = pg + 4 * u, q4 = pg + 4 * (u ^ yh);
#pragma unroll 4
                for (int a = 0; a < 8; a++) {
                    const double g1 = s_g[(x * 8 + a) * LDT + qq * 8 + y], g2 = s_g[(a * 8 + x) * LDT + qq * 8 + y];
                    const double g3 = s_g[(x * 8 + a) * LDT + y * 8 + q4], g4 = s_g[(a * 8 + x) * LDT + y * 8 + q4];
                    if constexpr (NK == 2) {
                        const vd2 d1 = *reinterpret_cast<const vd2 *>(&s_d[0][a * 9 + qq][0]);
                        const vd2 d2 = *reinterpret_cast<const vd2 *>(&s_d[1][a * 9 + qq][0]);
                        const vd2 d3 = *reinterpret_cast<const vd2 *>(&s_d[2][a * 9 + q4][0]);
                        const vd2 d4 = *reinterpret_cast<const vd2 *>(&s_d[3][a * 9 + q4][0]);
                        k1[0] += g1 * d1.x; k1[1] += g1 * d1.y;
                        k2[0] += g2 * d2.x; k2[1] += g2 * d2.y;
                        k3[0] += g3 * d3.x; k3[1] += g3 * d3.y;
                        k4[0] += g4 * d4.x; k4[1] += g4 * d4.y;
                    } else {
#pragma unroll
                        for (int q = 0; q < NK; q++) {
                            k1[q] += g1 * s_d[0][a * 9 + qq][q];
                            k2[q] += g2 * s_d[1][a * 9 + qq][q];
                            k3[q] += g3 * s_d[2][a * 9 + q4][q];
                            k4[q] += g4 * s_d[3][a * 9 + q4][q];
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NK; q++) {
                double a1 = k1[q], a2 = k2[q], a3 = k3[q], a4 = k4[q];
                a1 += __shfl_xor(a1, 1); a1 += __shfl_xor(a1, 2);
                a2 += __shfl_xor(a2, 1); a2 += __shfl_xor(a2, 2);
                a3 += __shfl_xor(a3, 1); a3 += __shfl_xor(a3, 2);
                a4 += __shfl_xor(a4, 1); a4 += __shfl_xor(a4, 2);
                if (pg == 0) {
                    double *Kq = Kacc + q * n2;
                    acc_add(&Kq[(size_t)(I * 8 + x) * npad + L * 8 + y], f * a1, dsc);
                    acc_add(&Kq[(size_t)(J * 8 + x) * npad + L * 8 + y], f * a2, dsc);
                    acc_add(&Kq[(size_t)(I * 8 + x) * npad + K * 8 + y], f * a3, dsc);
                    acc_add(&Kq[(size_t)(J * 8 + x) * npad + K * 8 + y], f * a4, dsc);
                }
            }
        }
        __syncthreads();  // s_col / s_g / s_d reuse by the next tile
    }
}

#undef JKM_LOAD
// nmat matrices in / out: symmetrise + pad the densities, clear the accumulators
__global__ void jk_multi_prep_kernel(double *__restrict__ work, const double *__restrict__ dmj, int nj, const double *__restrict__ dmk,
                                     int nk, int nao, int npad) {
    const size_t n2 = (size_t)npad * npad, nn = (size_t)nao * nao;
    const int nd = nj + nk;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2 * nd; e += (size_t)gridDim.x * blockDim.x) {
        const int q = e / n2;
        const size_t r = e - (size_t)q * n2;
        const int i = r / npad, j = r % npad;
        const double *dm = q < nj ? dmj + (size_t)q * nn : dmk + (size_t)(q - nj) * nn;
        double v = 0.0;
        if (i < nao && j < nao) v = 0.5 * (dm[(size_t)i * nao + j] + dm[(size_t)j * nao + i]);
        work[e] = v;
        work[n2 * nd + e] = 0.0;
    }
}

__global__ void jk_multi_finish_kernel(double *__restrict__ J, int nj, double *__restrict__ K, int nk, const double *__restrict__ work,
                                       int nao, int npad, const double *__restrict__ dscp) {
    const double dsc = dscp ? *dscp : 0.0;
    const size_t n2 = (size_t)npad * npad, nn = (size_t)nao * nao;
    const double *acc = work + (size_t)(nj + nk) * n2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn * (nj + nk); e += (size_t)gridDim.x * blockDim.x) {
        const int q = e / nn;
        const size_t r = e - (size_t)q * nn;
        const int i = r / nao, j = r % nao;
        const double v = det_value(acc[q * n2 + (size_t)i * npad + j], dsc) + det_value(acc[q * n2 + (size_t)j * npad + i], dsc);
        if (q < nj) J[e] = v;
        else K[e - (size_t)nj * nn] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Coulomb only (the Kohn-Sham Fock build), one density: every block streams a CONTIGUOUS range of tiles.  Tiles are ordered
// T = IJ (IJ + 1) / 2 + KL, so consecutive tiles share the bra block pair IJ: the row sums  J[I,J] += g . D[K,L]  stay in
// registers (per thread, un-reduced) until IJ changes and only then take the 16-lane reduce and their 64 atomics -- half
// of the kernel's fp64 atomics (7.9 M per 20-atom molecule, device scope: each one is a trip to the memory side) and the
// same-address contention of concurrently running neighbouring blocks (grid-stride order made them all hit J[I,J] at once)
// disappear; D[I,J] is reloaded only when IJ changes.  The column sums J[K,L] += g . D[I,J] change target every tile and
// keep the 2-shuffle + 4-wave LDS combine of jk_tiles_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void j_stream_kernel(const double *__restrict__ dscp, const double *__restrict__ tiles,
                                                         double *__restrict__ work, int npad, long long ntiles, long long per_block,
                                                         long long tbeg, int nao) {
    const TileLay ly(nao);
    // (tiles [tbeg, ntiles): the whole store, or one rank's slice of it -- dqc_jk_from_tiles_part)
    const double dsc = dscp ? *dscp : 0.0;
    __shared__ double s_col[2][4][64];
    const size_t n2 = (size_t)npad * npad;
    const double *Dp = work;
    double *Jacc = work + n2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = 4 * (t >> 4), c0 = 4 * (t & 15);
    const int kk = c0 >> 3, l0 = c0 & 7, ii = r0 >> 3, j0 = r0 & 7;
    const long long T0 = tbeg + (long long)blockIdx.x * per_block, T1 = min(T0 + per_block, ntiles);
    if (T0 >= T1) return;
    int IJ, KL, I, J, K, L;
    decode_tri(T0, IJ, KL);
    decode_tri(IJ, I, J);
    double rsacc[4] = {0, 0, 0, 0}, dij[4];
    {
        const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
#pragma unroll
        for (int q = 0; q < 4; q++) dij[q] = dijp[q];
    }
    auto flush_rows = [&]() {
        const double fI = (I == J ? 0.5 : 1.0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = rsacc[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if ((lane & 15) == 0) acc_add(&Jacc[(size_t)(I * 8 + ii) * npad + J * 8 + j0 + r], 2.0 * fI * v, dsc);
            rsacc[r] = 0.0;
        }
    };
    int par = 0;
    for (long long T = T0; T < T1; T++, KL++) {
        if (KL > IJ) {  // next bra block pair
            flush_rows();
            IJ++;
            KL = 0;
            decode_tri(IJ, I, J);
            const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
#pragma unroll
            for (int q = 0; q < 4; q++) dij[q] = dijp[q];
        }
        decode_tri(KL, K, L);
        // factors of the unique-tile weights: (I == J) is applied at the flush, (K == L) and (IJ == KL) here
        const double fk = (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const double *dklp = Dp + (size_t)(K * 8 + kk) * npad + L * 8 + l0;
        double dkl[4];
#pragma unroll
        for (int q = 0; q < 4; q++) dkl[q] = fk * dklp[q];
        double cs[4] = {0, 0, 0, 0};
        {
            double2 ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3;
            tile_load_patch(tiles, IJ, KL, I, J, K, L, r0, c0, ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3, ly);
#define DQC_JS_ROW(R_, A_, B_)                                                                           \
    rsacc[R_] += A_.x * dkl[0] + A_.y * dkl[1] + B_.x * dkl[2] + B_.y * dkl[3];                          \
    cs[0] += A_.x * dij[R_]; cs[1] += A_.y * dij[R_]; cs[2] += B_.x * dij[R_]; cs[3] += B_.y * dij[R_];
            DQC_JS_ROW(0, ta0, tb0) DQC_JS_ROW(1, ta1, tb1) DQC_JS_ROW(2, ta2, tb2) DQC_JS_ROW(3, ta3, tb3)
#undef DQC_JS_ROW
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double v = cs[c];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            cs[c] = v;
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_col[par][wave][c0 + c] = cs[c];
        }
        __syncthreads();  // one barrier per tile: s_col is double-buffered
        if (t < 64) {
            const double v = s_col[par][0][t] + s_col[par][1][t] + s_col[par][2][t] + s_col[par][3][t];
            acc_add(&Jacc[(size_t)(K * 8 + (t >> 3)) * npad + L * 8 + (t & 7)], 2.0 * (I == J ? 0.5 : 1.0) * fk * v, dsc);
        }
        par ^= 1;
    }
    flush_rows();
}


// ---------------------------------------------------------------------------------------------
// Coulomb AND exchange of one density (restricted Hartree-Fock, hf.py:198-199) over CONTIGUOUS tile ranges: j_stream_kernel's
// scheme -- row sums J[I,J] in registers until IJ changes -- plus the four exchange contractions of jk_tiles_kernel on the
// tile parked in LDS.  Consecutive tiles of a range share (I, J) and, for K + 1 tiles in a row, K: the exchange blocks
// K[I,K] and K[J,K] keep accumulating in registers until K (or IJ) changes, the other two, K[I,L] and K[J,L], change every
// tile.  Per tile 64 + 128 fp64 atomics instead of 128 + 256 (ablation of the grid-stride kernel on a 20-atom molecule: the J
// atomics cost 12 % of its time, the K atomics 6 %, the LDS contraction 16 %).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void jk_stream_kernel(const double *__restrict__ dscp, const double *__restrict__ tiles,
                                                          double *__restrict__ work, int npad, long long ntiles, long long per_block,
                                                          long long tbeg, int nao) {
    const TileLay ly(nao);
    const double dsc = dscp ? *dscp : 0.0;
    constexpr int LDT = 68;
    __shared__ double s_col[4][64];  // (single buffer: two barriers per tile separate its writers and readers anyway)
    __shared__ __attribute__((aligned(16))) double s_g[64 * LDT];
    __shared__ __attribute__((aligned(16))) double s_d[2][72][2];
    const size_t n2 = (size_t)npad * npad;
    const double *Dp = work;
    double *Jacc = work + n2, *Kacc = work + 2 * n2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = 4 * (t >> 4), c0 = 4 * (t & 15);
    const int kk = c0 >> 3, l0 = c0 & 7, ii = r0 >> 3, j0 = r0 & 7;
    const int o = t >> 2, pg = t & 3, x = o >> 3, y = o & 7, yh = y >> 2;  // exchange part: output (x, y), partial group pg
    const long long T0 = tbeg + (long long)blockIdx.x * per_block, T1 = min(T0 + per_block, ntiles);
    if (T0 >= T1) return;
    int IJ, KL, I, J, K, L;
    decode_tri(T0, IJ, KL);
    decode_tri(IJ, I, J);
    decode_tri(KL, K, L);
    double rsacc[4] = {0, 0, 0, 0}, dij[4];
    double k3acc = 0.0, k4acc = 0.0;  // K[I,K](x, y), K[J,K](x, y): this lane's share, carried over the tiles of one K
    {
        const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
#pragma unroll
        for (int q = 0; q < 4; q++) dij[q] = dijp[q];
    }
    auto flush_rows = [&]() {
        const double fI = (I == J ? 0.5 : 1.0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = rsacc[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if ((lane & 15) == 0) acc_add(&Jacc[(size_t)(I * 8 + ii) * npad + J * 8 + j0 + r], 2.0 * fI * v, dsc);
            rsacc[r] = 0.0;
        }
    };
    auto flush_k34 = [&]() {  // (weights were applied tile by tile)
        double a3 = k3acc, a4 = k4acc;
        a3 += __shfl_xor(a3, 1); a3 += __shfl_xor(a3, 2);
        a4 += __shfl_xor(a4, 1); a4 += __shfl_xor(a4, 2);
        if (pg == 0) {
            acc_add(&Kacc[(size_t)(I * 8 + x) * npad + K * 8 + y], a3, dsc);
            acc_add(&Kacc[(size_t)(J * 8 + x) * npad + K * 8 + y], a4, dsc);
        }
        k3acc = k4acc = 0.0;
    };
    for (long long T = T0; T < T1; T++) {
        const double f = (I == J ? 0.5 : 1.0) * (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const double fk = (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const double *dklp = Dp + (size_t)(K * 8 + kk) * npad + L * 8 + l0;
        double dkl[4];
#pragma unroll
        for (int q = 0; q < 4; q++) dkl[q] = fk * dklp[q];
        double cs[4] = {0, 0, 0, 0};
        {
            double2 ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3;
            tile_load_patch(tiles, IJ, KL, I, J, K, L, r0, c0, ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3, ly);
#define DQC_JS_ROW(R_, A_, B_)                                                                           \
    rsacc[R_] += A_.x * dkl[0] + A_.y * dkl[1] + B_.x * dkl[2] + B_.y * dkl[3];                          \
    cs[0] += A_.x * dij[R_]; cs[1] += A_.y * dij[R_]; cs[2] += B_.x * dij[R_]; cs[3] += B_.y * dij[R_]; \
    *reinterpret_cast<double2 *>(&s_g[(r0 + R_) * LDT + c0]) = A_;                                       \
    *reinterpret_cast<double2 *>(&s_g[(r0 + R_) * LDT + c0 + 2]) = B_;
            // (the barrier at the end of the previous iteration has retired that tile's LDS readers)
            DQC_JS_ROW(0, ta0, tb0) DQC_JS_ROW(1, ta1, tb1) DQC_JS_ROW(2, ta2, tb2) DQC_JS_ROW(3, ta3, tb3)
#undef DQC_JS_ROW
        }
        {   // D blocks of the exchange part: thread t loads element (t & 63) of block (t >> 6)
            const int blk = t >> 6, e = t & 63, xx = e >> 3, yy = e & 7;
            const int R = (blk & 1) ? I : J, Cb = (blk & 2) ? L : K;
            s_d[blk >> 1][xx * 9 + yy][blk & 1] = Dp[(size_t)(R * 8 + xx) * npad + Cb * 8 + yy];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double v = cs[c];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            cs[c] = v;
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 4; c++) s_col[wave][c0 + c] = cs[c];
        }
        __syncthreads();
        if (t < 64) {
            const double v = s_col[0][t] + s_col[1][t] + s_col[2][t] + s_col[3][t];
            acc_add(&Jacc[(size_t)(K * 8 + (t >> 3)) * npad + L * 8 + (t & 7)], 2.0 * (I == J ? 0.5 : 1.0) * fk * v, dsc);
        }
        {   // exchange: lane mapping and bank analysis as in jk_tiles_kernel
            double k1 = 0, k2 = 0, k3 = 0, k4 = 0;
#pragma unroll 1
            for (int u = 0; u < 2; u++) {
                const int q = pg + 4 * u, q4 = pg + 4 * (u ^ yh);
#pragma unroll
                for (int a = 0; a < 8; a++) {
                    typedef double vd2_ __attribute__((ext_vector_type(2)));
                    const vd2_ d12 = *reinterpret_cast<const vd2_ *>(&s_d[0][a * 9 + q][0]);
                    const vd2_ d34 = *reinterpret_cast<const vd2_ *>(&s_d[1][a * 9 + q4][0]);
                    k1 += s_g[(x * 8 + a) * LDT + q * 8 + y] * d12.x;    // g[x][a][v=q][y]  D[J,K](a,v)  -> K[I,L]
                    k2 += s_g[(a * 8 + x) * LDT + q * 8 + y] * d12.y;    // g[a][x][v=q][y]  D[I,K](a,v)  -> K[J,L]
                    k3 += s_g[(x * 8 + a) * LDT + y * 8 + q4] * d34.x;   // g[x][a][y][v=q4] D[J,L](a,v)  -> K[I,K]
                    k4 += s_g[(a * 8 + x) * LDT + y * 8 + q4] * d34.y;   // g[a][x][y][v=q4] D[I,L](a,v)  -> K[J,K]
                }
            }
            k1 += __shfl_xor(k1, 1); k1 += __shfl_xor(k1, 2);
            k2 += __shfl_xor(k2, 1); k2 += __shfl_xor(k2, 2);
            if (pg == 0) {
                acc_add(&Kacc[(size_t)(I * 8 + x) * npad + L * 8 + y], f * k1, dsc);
                acc_add(&Kacc[(size_t)(J * 8 + x) * npad + L * 8 + y], f * k2, dsc);
            }
            k3acc += f * k3;
            k4acc += f * k4;
        }
        // ---- next tile of the range
        if (T + 1 < T1) {
            if (KL == IJ) {  // next bra block pair: everything carried over goes out
                flush_rows();
                flush_k34();
                IJ++;
                KL = 0; K = 0; L = 0;
                decode_tri(IJ, I, J);
                const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
#pragma unroll
                for (int q = 0; q < 4; q++) dij[q] = dijp[q];
            } else {
                KL++;
                if (L == K) { flush_k34(); K++; L = 0; }
                else L++;
            }
        }
        __syncthreads();  // s_g / s_d reuse by the next tile
    }
    flush_rows();
    flush_k34();
}

// ---------------------------------------------------------------------------------------------
// The same stream scheme for the unrestricted Hartree-Fock trio (hf.py:93-103): J of NJ (0 or 1) densities and K of NK (1 or 2)
// exchange densities from ONE pass over contiguous tile ranges; work layout as jk_multi_kernel: Dj[NJ] | Dk[NK] | Jacc[NJ] |
// Kacc[NK].  (More Coulomb densities than one -- batched density matrices -- keep the grid-stride jk_multi_kernel.)
// ---------------------------------------------------------------------------------------------
template <int NJ, int NK>
__global__ __launch_bounds__(256, 3) void jk_multi_stream_kernel(const double *__restrict__ dscp, const double *__restrict__ tiles,
                                                                double *__restrict__ work, int npad, long long ntiles,
                                                                long long per_block, int nao) {
    const TileLay ly(nao);
    const double dsc = dscp ? *dscp : 0.0;
    constexpr int LDT = 68;
    __shared__ double s_col[4][64];  // (single buffer: two barriers per tile separate its writers and readers anyway)
    __shared__ __attribute__((aligned(16))) double s_g[64 * LDT];
    __shared__ __attribute__((aligned(16))) double s_d[4][72][NK];  // D[J,K], D[I,K], D[J,L], D[I,L] of the NK exchange densities
    const size_t n2 = (size_t)npad * npad;
    const double *Dp = work, *Dk = work + (size_t)NJ * n2;
    double *Jacc = work + (size_t)(NJ + NK) * n2, *Kacc = Jacc + (size_t)NJ * n2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = 4 * (t >> 4), c0 = 4 * (t & 15);
    const int kk = c0 >> 3, l0 = c0 & 7, ii = r0 >> 3, j0 = r0 & 7;
    const int o = t >> 2, pg = t & 3, x = o >> 3, y = o & 7, yh = y >> 2;  // exchange part: output (x, y), partial group pg
    const long long T0 = (long long)blockIdx.x * per_block, T1 = min(T0 + per_block, ntiles);
    if (T0 >= T1) return;
    int IJ, KL, I, J, K, L;
    decode_tri(T0, IJ, KL);
    decode_tri(IJ, I, J);
    decode_tri(KL, K, L);
    double rsacc[4] = {0, 0, 0, 0}, dij[4];
    double k3acc[NK], k4acc[NK];  // K[I,K](x, y), K[J,K](x, y) per exchange density: this lane's share, carried over the tiles of one K
#pragma unroll
    for (int q = 0; q < NK; q++) k3acc[q] = k4acc[q] = 0.0;
    if (NJ) {
        const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
#pragma unroll
        for (int q = 0; q < 4; q++) dij[q] = dijp[q];
    }
    auto flush_rows = [&]() {
        if (!NJ) return;
        const double fI = (I == J ? 0.5 : 1.0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = rsacc[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if ((lane & 15) == 0) acc_add(&Jacc[(size_t)(I * 8 + ii) * npad + J * 8 + j0 + r], 2.0 * fI * v, dsc);
            rsacc[r] = 0.0;
        }
    };
    auto flush_k34 = [&]() {  // (weights were applied tile by tile)
#pragma unroll
        for (int q = 0; q < NK; q++) {
            double a3 = k3acc[q], a4 = k4acc[q];
            a3 += __shfl_xor(a3, 1); a3 += __shfl_xor(a3, 2);
            a4 += __shfl_xor(a4, 1); a4 += __shfl_xor(a4, 2);
            if (pg == 0) {
                acc_add(&Kacc[q * n2 + (size_t)(I * 8 + x) * npad + K * 8 + y], a3, dsc);
                acc_add(&Kacc[q * n2 + (size_t)(J * 8 + x) * npad + K * 8 + y], a4, dsc);
            }
            k3acc[q] = k4acc[q] = 0.0;
        }
    };
    for (long long T = T0; T < T1; T++) {
        const double f = (I == J ? 0.5 : 1.0) * (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const double fk = (K == L ? 0.5 : 1.0) * (IJ == KL ? 0.5 : 1.0);
        const double *dklp = Dp + (size_t)(K * 8 + kk) * npad + L * 8 + l0;
        double dkl[4];
#pragma unroll
        for (int q = 0; q < 4; q++) dkl[q] = NJ ? fk * dklp[q] : 0.0;
        double cs[4] = {0, 0, 0, 0};
        {
            double2 ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3;
            tile_load_patch(tiles, IJ, KL, I, J, K, L, r0, c0, ta0, tb0, ta1, tb1, ta2, tb2, ta3, tb3, ly);
#define DQC_JS_ROW(R_, A_, B_)                                                                           \
    if (NJ) {                                                                                            \
        rsacc[R_] += A_.x * dkl[0] + A_.y * dkl[1] + B_.x * dkl[2] + B_.y * dkl[3];                      \
        cs[0] += A_.x * dij[R_]; cs[1] += A_.y * dij[R_]; cs[2] += B_.x * dij[R_]; cs[3] += B_.y * dij[R_]; \
    }                                                                                                    \
    *reinterpret_cast<double2 *>(&s_g[(r0 + R_) * LDT + c0]) = A_;                                       \
    *reinterpret_cast<double2 *>(&s_g[(r0 + R_) * LDT + c0 + 2]) = B_;
            // (the barrier at the end of the previous iteration has retired that tile's LDS readers)
            DQC_JS_ROW(0, ta0, tb0) DQC_JS_ROW(1, ta1, tb1) DQC_JS_ROW(2, ta2, tb2) DQC_JS_ROW(3, ta3, tb3)
#undef DQC_JS_ROW
        }
        {   // D blocks of the exchange part: thread t loads element (t & 63) of block (t >> 6)
            const int blk = t >> 6, e = t & 63, xx = e >> 3, yy = e & 7;
            const int R = (blk & 1) ? I : J, Cb = (blk & 2) ? L : K;
#pragma unroll
            for (int q = 0; q < NK; q++) s_d[blk][xx * 9 + yy][q] = Dk[q * n2 + (size_t)(R * 8 + xx) * npad + Cb * 8 + yy];
        }
        if (NJ) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double v = cs[c];
                v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                cs[c] = v;
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 4; c++) s_col[wave][c0 + c] = cs[c];
            }
        }
        __syncthreads();
        if (NJ && t < 64) {
            const double v = s_col[0][t] + s_col[1][t] + s_col[2][t] + s_col[3][t];
            acc_add(&Jacc[(size_t)(K * 8 + (t >> 3)) * npad + L * 8 + (t & 7)], 2.0 * (I == J ? 0.5 : 1.0) * fk * v, dsc);
        }
        {   // exchange: lane mapping and bank analysis as in jk_tiles_kernel; the tile reads are shared by the NK densities
            double k1[NK], k2[NK], k3[NK], k4[NK];
#pragma unroll
            for (int q = 0; q < NK; q++) k1[q] = k2[q] = k3[q] = k4[q] = 0.0;
#pragma unroll 1
            for (int u = 0; u < 2; u++) {
                const int qq = pg + 4 * u, q4 = pg + 4 * (u ^ yh);
#pragma unroll 4
                for (int a = 0; a < 8; a++) {
                    const double g1 = s_g[(x * 8 + a) * LDT + qq * 8 + y], g2 = s_g[(a * 8 + x) * LDT + qq * 8 + y];
                    const double g3 = s_g[(x * 8 + a) * LDT + y * 8 + q4], g4 = s_g[(a * 8 + x) * LDT + y * 8 + q4];
#pragma unroll
                    for (int q = 0; q < NK; q++) {
                        k1[q] += g1 * s_d[0][a * 9 + qq][q];   // D[J,K](a,v) -> K[I,L]
                        k2[q] += g2 * s_d[1][a * 9 + qq][q];   // D[I,K](a,v) -> K[J,L]
                        k3[q] += g3 * s_d[2][a * 9 + q4][q];   // D[J,L](a,v) -> K[I,K]
                        k4[q] += g4 * s_d[3][a * 9 + q4][q];   // D[I,L](a,v) -> K[J,K]
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NK; q++) {
                double a1 = k1[q], a2 = k2[q];
                a1 += __shfl_xor(a1, 1); a1 += __shfl_xor(a1, 2);
                a2 += __shfl_xor(a2, 1); a2 += __shfl_xor(a2, 2);
                if (pg == 0) {
                    acc_add(&Kacc[q * n2 + (size_t)(I * 8 + x) * npad + L * 8 + y], f * a1, dsc);
                    acc_add(&Kacc[q * n2 + (size_t)(J * 8 + x) * npad + L * 8 + y], f * a2, dsc);
                }
                k3acc[q] += f * k3[q];
                k4acc[q] += f * k4[q];
            }
        }
        // ---- next tile of the range
        if (T + 1 < T1) {
            if (KL == IJ) {  // next bra block pair: everything carried over goes out
                flush_rows();
                flush_k34();
                IJ++;
                KL = 0; K = 0; L = 0;
                decode_tri(IJ, I, J);
                if (NJ) {
                    const double *dijp = Dp + (size_t)(I * 8 + ii) * npad + J * 8 + j0;
#pragma unroll
                    for (int q = 0; q < 4; q++) dij[q] = dijp[q];
                }
            } else {
                KL++;
                if (L == K) { flush_k34(); K++; L = 0; }
                else L++;
            }
        }
        __syncthreads();  // s_g / s_d reuse by the next tile
    }
    flush_rows();
    flush_k34();
}


}  // namespace dqc

extern "C" {

size_t dqc_jk_work_doubles(int nao) {
    const size_t npad = (size_t)(nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    // D, J, K accumulators | 8 slots (the fixed-point scale of the deterministic mode, the ticket of fock_combine_kernel) | the combined
    // AO matrix and the half-transformed matrix of dqc_fock_finish / dqc_fock_prep | 2 x 64 partial sums (csrc/fock.hip)
    return 5 * npad * npad + 8 + 128;
}

long long dqc_eri_tile_offset(int nao, long long tile) {
    // doubles in front of tile `tile` (0 ... tile count) of the packed store of a basis with nao functions: tiles follow each
    // other in the order (IJ, KL <= IJ), tile = IJ (IJ + 1) / 2 + KL
    using namespace dqc;
    const long long nt = (long long)dqc_eri_tile_count(nao);
    if (nao <= 0 || tile <= 0) return 0;
    if (tile >= nt) return eri_store_data_doubles(nao);
    auto tri = [](long long t, long long &a, long long &b) {
        long long r = (long long)((std::sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (r * (r + 1) / 2 > t) r--;
        while ((r + 1) * (r + 2) / 2 <= t) r++;
        a = r;
        b = t - r * (r + 1) / 2;
    };
    long long IJ, KL, I, J, K, L;
    tri(tile, IJ, KL);
    tri(IJ, I, J);
    tri(KL, K, L);
    return tile_base((int)I, (int)J, (int)K, (int)KL, TileLay(nao));
}

int dqc_jk_from_tiles(double *d_J, double *d_K, const double *d_tiles, const double *d_dm, int nao,
                      double *d_work, void *stream) {
    return dqc_jk_from_tiles_part(d_J, d_K, d_tiles, d_dm, nao, d_work, 0, (long long)dqc_eri_tile_count(nao), stream);
}

static int jk_from_tiles_impl(double *d_J, double *d_K, const double *d_tiles_part, const double *d_dm, int nao, double *d_work,
                              long long tile_begin, long long tile_end, int with_k_prepared, void *stream);

int dqc_jk_from_tiles_part(double *d_J, double *d_K, const double *d_tiles_part, const double *d_dm, int nao, double *d_work,
                           long long tile_begin, long long tile_end, void *stream) {
    return jk_from_tiles_impl(d_J, d_K, d_tiles_part, d_dm, nao, d_work, tile_begin, tile_end, 0, stream);
}

int dqc_jk_stream_prepared(const double *d_tiles, int nao, double *d_work, int with_k, void *stream) {
    // the tile pass alone, on a work buffer dqc_fock_prep has filled (AO density, zeroed accumulators); the accumulators stay in the
    // work buffer for dqc_fock_finish
    return jk_from_tiles_impl(nullptr, nullptr, d_tiles, nullptr, nao, d_work, 0, (long long)dqc_eri_tile_count(nao), with_k ? 1 : 0, stream);
}

static int jk_from_tiles_impl(double *d_J, double *d_K, const double *d_tiles_part, const double *d_dm, int nao, double *d_work,
                              long long tile_begin, long long tile_end, int with_k_prepared, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const long long nt_all = (long long)dqc_eri_tile_count(nao);
    if (tile_begin < 0 || tile_end > nt_all || tile_begin > tile_end) { set_error("dqc_jk_from_tiles_part: tile range outside the store"); return DQC_EINVAL; }
    const bool whole = tile_begin == 0 && tile_end == nt_all;
    if (!whole && deterministic_mode()) {
        set_error("dqc_jk_from_tiles_part: the deterministic mode needs the whole store (its scale reads the diagonal tiles)");
        return DQC_EINVAL;
    }
    // the kernels address tiles by their offset in the WHOLE store: a slice is handed over with its virtual origin
    const double *d_tiles = d_tiles_part - dqc_eri_tile_offset(nao, tile_begin);
    const long long ntiles = tile_end, tbeg = tile_begin, nrun = tile_end - tile_begin;
    const int with_k = d_dm ? (d_K != nullptr) : with_k_prepared;
    // d_dm == nullptr: the work buffer was prepared by dqc_fock_prep (symmetric AO density in place, accumulators zeroed);
    // d_J == nullptr: the accumulators are left for dqc_fock_finish (dqc_jk_stream_prepared)
    if (d_dm) {
        hipLaunchKernelGGL(jk_prep_kernel, dim3(64), dim3(256), 0, st, d_work, d_dm, nao, npad, with_k);
        DQC_CHECK_LAUNCH();
    }
    double *dscp = nullptr;
    if (deterministic_mode()) {
        dscp = d_work + 3 * (size_t)npad * npad;
        hipLaunchKernelGGL(jk_det_scale_kernel, dim3(1), dim3(256), 0, st, dscp, d_work, 1, (size_t)npad * npad, d_tiles, nao);
        DQC_CHECK_LAUNCH();
    }
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 256 * 16);
    static const char *jimpl = getenv("DQC_J_IMPL");  // "stride": the grid-stride kernel of round 1 (A/B runs)
    if (nrun == 0) {
        // (an empty slice: nothing to add)
    } else if (!whole && jimpl && jimpl[0] == 's') {
        set_error("dqc_jk_from_tiles_part: DQC_J_IMPL=stride streams the whole store only");
        return DQC_EINVAL;
    } else
    if (with_k && !(jimpl && jimpl[0] == 's')) {
        // contiguous tile ranges of >= 8 tiles, 1024 ... 6144 blocks (4 resident per CU; sweep 1024 ... 8192 on benzene, 20-atom
        // cc-pVDZ and naphthalene / cc-pVTZ: flat within 3 % inside this window)
        static const long long jk_nblk_env = [] { const char *e = getenv("DQC_JK_NBLK"); return e ? atoll(e) : 0LL; }();
        const long long nblk = jk_nblk_env > 0 ? std::min<long long>(nrun, jk_nblk_env) : std::min<long long>(nrun, std::max<long long>(1024, std::min<long long>(6144, nrun / 8)));
        const long long per = (nrun + nblk - 1) / nblk;
        hipLaunchKernelGGL(jk_stream_kernel, dim3((unsigned)((nrun + per - 1) / per)), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, per, tbeg, nao);
    } else if (with_k) {
        hipLaunchKernelGGL(jk_tiles_kernel<true>, dim3(grid), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, nao);
    } else if (jimpl && jimpl[0] == 's') {
        hipLaunchKernelGGL(jk_tiles_kernel<false>, dim3(grid), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, nao);
    } else {
        // contiguous tile ranges, ~6 resident blocks per CU x 2 rounds.  (Tried: 8 x 4 tile rectangles with the column sums in
        // LDS, 24 atomics per tile and no barrier -- 0.396 ms against 0.37 ms for this form: shorter contiguous runs.)
        static const long long j_nblk_env = [] { const char *e = getenv("DQC_J_NBLK"); return e ? atoll(e) : 0LL; }();
        // Ranges of >= 4 tiles, at most 65536 blocks (DQC_J_NBLK overrides the cap).  Sweep of the cap, same box
        // (profiles/r05s_j_nblk_sweep.txt): 3072 blocks (rounds 2-4) benzene 0.0366 ms, C5 0.368, naphthalene / cc-pVTZ 5.57;
        // 16384: 0.0364 / 0.341 / 5.27; 65536: 0.0363 / 0.341 / 5.20 -- shorter ranges even out the tail of the launch; ranges
        // of 2 tiles (benzene with 4096 blocks) lose it again to the per-range prologue and flush: 0.043 ms.
        const long long nblk = std::max<long long>(1, std::min<long long>((nrun + 3) / 4, j_nblk_env > 0 ? j_nblk_env : 65536));
        const long long per = (nrun + nblk - 1) / nblk;
        hipLaunchKernelGGL(j_stream_kernel, dim3((unsigned)((nrun + per - 1) / per)), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, per, tbeg, nao);
    }
    DQC_CHECK_LAUNCH();
    if (d_J) {
        hipLaunchKernelGGL(jk_finish_kernel, dim3(64), dim3(256), 0, st, d_J, d_K, d_work, nao, npad, dscp);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}

size_t dqc_jk_multi_work_doubles(int nao, int nj, int nk) {
    const size_t npad = (size_t)(nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    return 2 * (size_t)(nj + (nk > 2 ? 2 : nk)) * npad * npad + 8;
}

int dqc_jk_from_tiles_multi(double *d_J, const double *d_dmJ, int nj, double *d_K, const double *d_dmK, int nk,
                            const double *d_tiles, int nao, double *d_work, void *stream) {
    using namespace dqc;
    if (nao <= 0 || (nj <= 0 && nk <= 0)) return DQC_OK;
    if (nj < 0 || nk < 0) { set_error("dqc_jk_from_tiles_multi: negative matrix count"); return DQC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const size_t nn = (size_t)nao * nao;
    const long long ntiles = (long long)dqc_eri_tile_count(nao);
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 256 * 16);
    // exchange densities go two per pass (their tile reads from LDS are shared); the Coulomb ones all ride on the first pass
    int kdone = 0, first = 1;
    while (first || kdone < nk) {
        const int njp = first ? nj : 0, nkp = std::min(2, nk - kdone);
        hipLaunchKernelGGL(jk_multi_prep_kernel, dim3(64), dim3(256), 0, st, d_work, d_dmJ, njp, d_dmK + (size_t)kdone * nn, nkp, nao, npad);
        DQC_CHECK_LAUNCH();
        double *dscp = nullptr;
        if (deterministic_mode()) {
            dscp = d_work + 2 * (size_t)(njp + nkp) * npad * npad;
            hipLaunchKernelGGL(jk_det_scale_kernel, dim3(1), dim3(256), 0, st, dscp, d_work, njp + nkp, (size_t)npad * npad, d_tiles, nao);
            DQC_CHECK_LAUNCH();
        }
        // at most one Coulomb density in the pass: the stream form (contiguous ranges, sums carried in registers)
        const long long nblk_s = std::min<long long>(ntiles, std::max<long long>(1024, std::min<long long>(6144, ntiles / 8)));
        const long long per_s = (ntiles + nblk_s - 1) / nblk_s;
        const dim3 grid_s((unsigned)((ntiles + per_s - 1) / per_s));
        static const char *mimpl = getenv("DQC_JK_MULTI_IMPL");  // "grid": the grid-stride kernel (A/B runs)
        const bool stream_ok = njp <= 1 && nkp >= 1 && !(mimpl && mimpl[0] == 'g');
        if (stream_ok && njp == 1 && nkp == 2) hipLaunchKernelGGL((jk_multi_stream_kernel<1, 2>), grid_s, dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, per_s, nao);
        else if (stream_ok && njp == 1 && nkp == 1) hipLaunchKernelGGL((jk_multi_stream_kernel<1, 1>), grid_s, dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, per_s, nao);
        else if (stream_ok && njp == 0 && nkp == 2) hipLaunchKernelGGL((jk_multi_stream_kernel<0, 2>), grid_s, dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, per_s, nao);
        else if (stream_ok && njp == 0 && nkp == 1) hipLaunchKernelGGL((jk_multi_stream_kernel<0, 1>), grid_s, dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, per_s, nao);
        else if (nkp == 2) hipLaunchKernelGGL(jk_multi_kernel<2>, dim3(grid), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, njp, nao);
        else if (nkp == 1) hipLaunchKernelGGL(jk_multi_kernel<1>, dim3(grid), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, njp, nao);
        else hipLaunchKernelGGL(jk_multi_kernel<0>, dim3(grid), dim3(256), 0, st, dscp, d_tiles, d_work, npad, ntiles, njp, nao);
        DQC_CHECK_LAUNCH();
        hipLaunchKernelGGL(jk_multi_finish_kernel, dim3(64), dim3(256), 0, st, d_J, njp, d_K + (size_t)kdone * nn, nkp, d_work, nao, npad, dscp);
        DQC_CHECK_LAUNCH();
        kdone += nkp;
        first = 0;
    }
    return DQC_OK;
}

}  // extern "C"
