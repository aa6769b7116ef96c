// grid_density.hip -- density (and its gradient) on the grid: the first of the two GEMM-shaped passes over the cached AO matrix
// (the file was one with grid_vxc.hip until round 3; the shared header comment follows)
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
#include "grid_common.hpp"

namespace dqc {

// ---------------------------------------------------------------------------------------------
// density:  C[64 pts x n] = Phi_blk . D, K-chunks of 16 staged in double-buffered LDS, 4 waves per block,
// wave w owns points 16w..16w+15 and all column tiles (accumulators in registers), fused row-dot epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int DEN_BM = 64;    // points per block (4 waves x 16); two blocks share a CU so that one block's
                              // memory-bound epilogue overlaps the other's MFMA main loop
constexpr int DEN_NT = DEN_BM * 4;  // threads per block
constexpr int DEN_KC = 16;    // K chunk
constexpr int DEN_SA = DEN_KC + 2;  // LDS row stride of the A chunk (conflict-free ds_read_b64 fragments)

#ifdef DEN_TRACE  // per-block timeline for tools/ubench/den_trace*.hip: CU slot, start / MFMA-end / end (100 MHz ticks)
__device__ long long g_den_trace[4 * 16384];
DQC_DEV void den_trace(int k) {
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        if (k == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_den_trace[4 * blockIdx.x + 3] = (long long)(((xcc & 7u) << 8) | ((hw >> 8) & 0xffu));
        }
        g_den_trace[4 * blockIdx.x + k] = wall_clock64();
    }
}
#define DEN_TRACE_POINT(k) den_trace(k)
#else
#define DEN_TRACE_POINT(k)
#endif

#ifdef DEN_EXP_UNPAIRED  // A/B builds: 8-byte epilogue loads
constexpr bool DEN_PAIRED = false;
#else
constexpr bool DEN_PAIRED = true;
#endif

// LDS row stride of the staged B panel (D columns / L^T).  The permuted fragment reads (lane lr at double 2 lr of row lk) are conflict-free
// when the stride is ODD (rows lk and lk + 1 of a half-wave then take the even and the odd doubles); the plain reads want
// stride == 16 (mod 32)
constexpr int lr_panel_stride(int nct) {
    return (DEN_PAIRED && nct >= 2) ? nct * 16 + 1 : (((nct * 16) & 31) == 16 ? nct * 16 : nct * 16 + 16);
}

// row-dot epilogue shared by the density kernels: p[r][q] += sum_ct acc[ct][r] * Phi_q[row_r][col0 + 16 ct].
// All loads of a batch (NCT tiles x 2 components) are issued before the first FMA and there is no per-tile guard
// (the callers make every column panel a full one), so 2 NCT loads per lane are in flight instead of 4 -- the
// epilogue is a latency-bound HBM read otherwise.
template <int NCT, bool GGA, int Q0 = 0>
DQC_DEV void rowdot_epilogue(const v4d (&acc)[NCT], double (&p)[4][GGA ? 4 : 1], const double *__restrict__ blk0,
                             const double *__restrict__ blkg, size_t cs, const int (&roff)[4], int col0) {
    // Q0 = 1 skips the value component (the factor-form kernel gets rho from |A'|^2 and never re-reads Phi)
    // blk0 / blkg: uniform pointers to this block's first row of the value / gradient-carrying AO array;
    // roff[r]: block-local element offset (row * ld + lane column) of accumulator row r.
    // Software pipeline over the 4 x (1|4) (row, component) batches of NCT loads: batch b+1 is issued before batch
    // b is consumed, so NCT..2 NCT loads per lane are always in flight.
    constexpr int NQ = (GGA ? 4 : 1) - Q0, NB = 4 * NQ;
    if (NB == 0) return;
    constexpr int DP = 2;  // batches in flight (three: 0.588 instead of 0.563 ms on the C5 shape -- registers, not latency)
    double t[DP][NCT];
    auto issue = [&](int bt, double (&dst)[NCT]) {
        const int r = bt / NQ, q = bt % NQ + Q0;
        const double *base = (q == 0 ? blk0 : blkg + q * cs) + col0;  // uniform; tiles at immediate offsets
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) dst[ct] = base[roff[r] + ct * 16];
    };
#pragma unroll
    for (int b0 = 0; b0 < DP - 1 && b0 < NB; b0++) issue(b0, t[b0]);
#pragma unroll
    for (int bt = 0; bt < NB; bt++) {
        if (bt + DP - 1 < NB) issue(bt + DP - 1, t[(bt + DP - 1) % DP]);
        const int r = bt / NQ, q = bt % NQ + Q0;
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) p[r][q] += acc[ct][r] * t[bt % DP][ct];
        __builtin_amdgcn_sched_barrier(0);  // pin the pipeline: later batches must not be hoisted (spills)
    }
}

// The same row dots with 16-byte loads (density_lr_kernel): phase 2 reads its B fragments with the panel's columns PERMUTED,
// so that lane lr of the accumulator tiles (2 m, 2 m + 1) holds the ADJACENT AO columns 32 m + 2 lr and 32 m + 2 lr + 1 -- one
// double2 load per tile pair instead of two 8-byte loads; an odd last tile keeps its plain layout.  C5 shape: 0.54 ms against
// 0.564 ms (tools/gpu_den_time.py; the staged panel gets an odd row stride so that the permuted ds_read_b64 pattern stays
// conflict-free).  Tried instead: trading accumulators between neighbouring lanes with DPP swaps so that the LDS layout stays
// plain -- 0.572 ms.
template <int NCT, bool GGA, int Q0 = 0>
DQC_DEV void rowdot_epilogue_paired(const v4d (&acc)[NCT], double (&p)[4][GGA ? 4 : 1], const double *__restrict__ blk0,
                                    const double *__restrict__ blkg, size_t cs, const int (&roff)[4], int lr, int col0) {
    // roff[r] = row * ld + 2 lr (the lane's first column of a tile pair).  A batch = the NP double2 loads (+ the odd tile) of one (row, component); two batches in flight
    constexpr int NQ = (GGA ? 4 : 1) - Q0, NB = 4 * NQ, NP = NCT / 2, ODD = NCT & 1, DP = 2;
    if (NB == 0) return;
    double2 t2[DP][NP > 0 ? NP : 1];
    double t1[DP];
    auto issue = [&](int bt, double2 (&d2)[NP > 0 ? NP : 1], double &d1) {
        const int r = bt / NQ, q = bt % NQ + Q0;
        const double *base = (q == 0 ? blk0 : blkg + q * cs) + col0;  // uniform; tile pairs at immediate offsets
#pragma unroll
        for (int m = 0; m < NP; m++) d2[m] = *reinterpret_cast<const double2 *>(base + roff[r] + m * 32);
        if (ODD) d1 = base[(roff[r] - lr) + (NCT - 1) * 16];
    };
    issue(0, t2[0], t1[0]);
#pragma unroll
    for (int bt = 0; bt < NB; bt++) {
        if (bt + 1 < NB) issue(bt + 1, t2[(bt + 1) % DP], t1[(bt + 1) % DP]);
        const int r = bt / NQ, q = bt % NQ + Q0;
#pragma unroll
        for (int m = 0; m < NP; m++) p[r][q] += acc[2 * m][r] * t2[bt % DP][m].x + acc[2 * m + 1][r] * t2[bt % DP][m].y;
        if (ODD) p[r][q] += acc[NCT - 1][r] * t1[bt % DP];
        __builtin_amdgcn_sched_barrier(0);  // pin the pipeline: later batches must not be hoisted (spills)
    }
}

// the paired row dots for TWO accumulator sets that share their loads (spin-fused factor-form density: the gradient arrays of
// the AO matrix are read once for both spins); gradient components only (Q0 = 1)
template <int NCT>
DQC_DEV void rowdot_epilogue_paired2(const v4d (&acc)[2][NCT], double (&p)[2][4][4], const double *__restrict__ blkg, size_t cs,
                                     const int (&roff)[4], int lr, int col0) {
    constexpr int NQ = 3, NB = 4 * NQ, NP = NCT / 2, ODD = NCT & 1, DP = 2;
    double2 t2[DP][NP > 0 ? NP : 1];
    double t1[DP];
    auto issue = [&](int bt, double2 (&d2)[NP > 0 ? NP : 1], double &d1) {
        const int r = bt / NQ, q = bt % NQ + 1;
        const double *base = blkg + q * cs + col0;
#pragma unroll
        for (int m = 0; m < NP; m++) d2[m] = *reinterpret_cast<const double2 *>(base + roff[r] + m * 32);
        if (ODD) d1 = base[(roff[r] - lr) + (NCT - 1) * 16];
    };
    issue(0, t2[0], t1[0]);
#pragma unroll
    for (int bt = 0; bt < NB; bt++) {
        if (bt + 1 < NB) issue(bt + 1, t2[(bt + 1) % DP], t1[(bt + 1) % DP]);
        const int r = bt / NQ, q = bt % NQ + 1;
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
#pragma unroll
            for (int m = 0; m < NP; m++) p[sp][r][q] += acc[sp][2 * m][r] * t2[bt % DP][m].x + acc[sp][2 * m + 1][r] * t2[bt % DP][m].y;
            if (ODD) p[sp][r][q] += acc[sp][NCT - 1][r] * t1[bt % DP];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NCT, bool GGA>
__global__ __launch_bounds__(256, 2) void density_kernel(double *__restrict__ rho, double *__restrict__ grho,
                                                         const double *__restrict__ ao, int ngrid, int ld,
                                                         const double *__restrict__ dm, int ntile,
                                                         const double *__restrict__ aoe, int lda) {
    // aoe: array the row dots are taken with (== ao except for the "pair" form rowdot(ao . D, aoe), LDA mode only)
    // lda: row stride of the AO arrays in HBM (dqc_ao_stride); ld = 16 ntile: rows / columns of the zero-padded D.  The K loop and
    // the epilogue run over 16 ntile columns of an AO row: columns lda .. ld - 1 are the first doubles of the next row (the
    // arrays carry ld - lda doubles of slack at their end), finite values that meet zero rows / columns of D
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int LSB = NCT * 16;                 // width of the staged D column panel
    constexpr int LSBP = lr_panel_stride(NCT);    // odd: the B fragments are read with permuted columns (rowdot_epilogue_paired)
    constexpr int A_SZ = DEN_BM * DEN_SA, B_SZ = DEN_KC * LSBP;
    constexpr int NB2 = (DEN_KC * LSB / 2 + DEN_NT - 1) / DEN_NT;  // double2 loads of the B chunk per thread
    constexpr int NKK = DEN_KC / 4;
    double *sA = lds, *sB = lds + 2 * A_SZ;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int g0 = blockIdx.x * DEN_BM;
    const size_t cs = (size_t)ngrid * lda;
    // all global addresses are (uniform 64-bit base) + (small 32-bit lane offset)
    const double *aoblk = ao + (size_t)g0 * lda;   // this block's 64 rows of Phi
    const double *aoeblk = aoe + (size_t)g0 * lda;
    const int rmax = ngrid - 1 - g0;               // last valid block-local row
    // staging roles: A chunk = 64 rows x 16 doubles -> thread (row = tid/4, 4 doubles at seg = tid%4)
    const int arow = tid >> 2, aseg = (tid & 3) * 4;
    const int aoff = min(arow, rmax) * lda + aseg;

    double p[4][GGA ? 4 : 1];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < (GGA ? 4 : 1); q++) p[r][q] = 0.0;
    int roff[4];  // block-local element offsets of this lane's four accumulator rows
#pragma unroll
    for (int r = 0; r < 4; r++) roff[r] = min(wave * 16 + lk + 4 * r, rmax) * lda + (DEN_PAIRED ? 2 : 1) * lr;

    DEN_TRACE_POINT(0);
    const int nk = ld / DEN_KC;
    // every panel is a full one: the last panel is shifted back to end at ntile and the tiles it shares with its
    // predecessor (tile index < jnew) get zero D columns, so nothing is counted twice and nothing is read past ld
    for (int jnew = 0; jnew < ntile; jnew += NCT) {
        const int jc = min(jnew, ntile - NCT);
        v4d acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[ct] = v4d{0, 0, 0, 0};
        // TWO register sets: the loads of chunk kc + 2 are issued (in slices between the MFMA groups: waves issue in order, a wave
        // that first pushes a whole prefetch through the address pipe starts its MFMAs late) while chunk kc is multiplied and
        // chunk kc + 1 -- loaded one chunk period earlier -- waits in the other set to be staged.  With ONE set (rounds 1-3) a chunk
        // was staged right after the MFMAs it was loaded under: 0.894 -> 0.865 ms on the C5 shape.  tools/ubench/den_trace_dense.hip:
        // a block's K loop takes 50 us (was 58) for 20 us of MFMAs (676 x 64 cycles), two blocks per CU, a wave of either in its
        // K loop 98 % of the time -- the loop itself keeps the matrix pipe half busy (fragment reads are not pipelined ahead of the
        // MFMAs, one barrier per chunk); the consumer / producer split of the Vxc kernels is what this kernel still lacks.
        // Scalars / token-pasted names, not arrays indexed by the set: those would live in scratch across the back-edge.
        double2 paA0 = make_double2(0.0, 0.0), paA1 = paA0, paB0 = paA0, paB1 = paA0, pbA[NB2], pbB[NB2];
        int boff[NB2];
        bool bzero[NB2];
#pragma unroll
        for (int i = 0; i < NB2; i++) {
            const int e = min((tid + i * DEN_NT) * 2, DEN_KC * LSB - 2);
            const int row = e / LSB, col = e - row * LSB;
            boff[i] = row * ld + jc * 16 + col;
            bzero[i] = jc * 16 + col < jnew * 16;
            pbA[i] = pbB[i] = make_double2(0.0, 0.0);
        }
#define DQC_DEN_PREF(KC, S, PART)                                                                          \
    {                                                                                                      \
        const int kq_ = min((KC), nk - 1); /* past-the-end chunks re-read the last one, never staged */     \
        if ((PART) == 0) {                                                                                 \
            const double *s_ = aoblk + kq_ * DEN_KC + aoff;                                                \
            pa##S##0 = *reinterpret_cast<const double2 *>(s_);                                             \
            pa##S##1 = *reinterpret_cast<const double2 *>(s_ + 2);                                         \
        }                                                                                                  \
        const double *d_ = dm + (size_t)kq_ * DEN_KC * ld;                                                 \
        _Pragma("unroll") for (int i = 0; i < NB2; i++) {                                                  \
            if (i % NKK != (PART)) continue;                                                               \
            pb##S[i] = *reinterpret_cast<const double2 *>(d_ + boff[i]);                                   \
            if (bzero[i]) pb##S[i] = make_double2(0.0, 0.0);                                               \
        }                                                                                                  \
    }
#define DQC_DEN_STAGE(KC, S)                                                                               \
    if ((KC) < nk) {                                                                                       \
        const int buf_ = (KC) & 1;                                                                         \
        double *a_ = sA + buf_ * A_SZ + arow * DEN_SA + aseg;                                              \
        *reinterpret_cast<double2 *>(a_) = pa##S##0;                                                       \
        *reinterpret_cast<double2 *>(a_ + 2) = pa##S##1;                                                   \
        _Pragma("unroll") for (int i = 0; i < NB2; i++) {                                                  \
            const int e = (tid + i * DEN_NT) * 2;                                                          \
            const int row = e / LSB, col = e - row * LSB;                                                  \
            if (row < DEN_KC) { /* two 8-byte stores: odd row stride */                                    \
                double *d = sB + buf_ * B_SZ + row * LSBP + col;                                           \
                d[0] = pb##S[i].x;                                                                         \
                d[1] = pb##S[i].y;                                                                         \
            }                                                                                              \
        }                                                                                                  \
    }
#define DQC_DEN_MFMAS(KC, S)                                                                               \
    {                                                                                                      \
        const int buf_ = (KC) & 1;                                                                         \
        const double *a = sA + buf_ * A_SZ + (wave * 16 + lr) * DEN_SA + lk;                               \
        const double *b = sB + buf_ * B_SZ + lk * LSBP + lr;                                               \
        const double *b2 = b + lr; /* permuted columns: tile 2 m + h, lane lr <- column 32 m + 2 lr + h */ \
        _Pragma("unroll") for (int kk = 0; kk < NKK; kk++) {                                               \
            DQC_DEN_PREF((KC) + 2, S, kk) /* global loads in flight during the MFMAs */                    \
            const double av = a[kk * 4];                                                                   \
            _Pragma("unroll") for (int ct = 0; ct < NCT; ct++) {                                           \
                const double bv = (DEN_PAIRED && ct < 2 * (NCT / 2)) ? b2[kk * 4 * LSBP + 32 * (ct >> 1) + (ct & 1)] \
                                                                    : b[kk * 4 * LSBP + ct * 16];          \
                acc[ct] = mfma_f64(av, bv, acc[ct]);                                                       \
            }                                                                                              \
        }                                                                                                  \
    }
        __syncthreads();  // buffers free (previous column panel fully consumed)
#pragma unroll
        for (int part = 0; part < NKK; part++) DQC_DEN_PREF(0, A, part)
#pragma unroll
        for (int part = 0; part < NKK; part++) DQC_DEN_PREF(1, B, part)
        DQC_DEN_STAGE(0, A)
        __syncthreads();
        int kc = 0;
        for (; kc + 1 < nk; kc += 2) {
            DQC_DEN_MFMAS(kc, A)          // (set A <- chunk kc + 2)
            DQC_DEN_STAGE(kc + 1, B)
            __syncthreads();
            DQC_DEN_MFMAS(kc + 1, B)      // (set B <- chunk kc + 3)
            DQC_DEN_STAGE(kc + 2, A)
            __syncthreads();
        }
        if (kc < nk) {  // odd chunk count: the last chunk was staged by the loop's second half
            DQC_DEN_MFMAS(kc, A)
            __syncthreads();
        }
#undef DQC_DEN_PREF
#undef DQC_DEN_STAGE
#undef DQC_DEN_MFMAS
        // epilogue: row dots with Phi (and its gradient) in the accumulator layout, straight from global
        DEN_TRACE_POINT(1);
#ifdef ABL_DEN_NO_EPI
        if (ngrid < 0)
#endif
        {
            if constexpr (DEN_PAIRED) rowdot_epilogue_paired<NCT, GGA>(acc, p, GGA ? aoblk : aoeblk, aoblk, cs, roff, lr, jc * 16);
            else rowdot_epilogue<NCT, GGA>(acc, p, GGA ? aoblk : aoeblk, aoblk, cs, roff, jc * 16);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < (GGA ? 4 : 1); q++) {
            double v = p[r][q];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            p[r][q] = v;
        }
    DEN_TRACE_POINT(2);
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = g0 + wave * 16 + lk + 4 * r;
            if (row < ngrid) {
                rho[row] = p[r][0];
                if (GGA) {
                    grho[row] = 2.0 * p[r][1];
                    grho[(size_t)ngrid + row] = 2.0 * p[r][2];
                    grho[2 * (size_t)ngrid + row] = 2.0 * p[r][3];
                }
            }
        }
    }
}

template <int NCT>
static constexpr size_t density_lds_bytes() {
    constexpr int LSBP = lr_panel_stride(NCT);
    return sizeof(double) * 2 * (DEN_BM * DEN_SA + DEN_KC * LSBP);
}

template <bool GGA>
static int launch_density(int nct, dim3 grid, hipStream_t st, double *rho, double *grho, const double *ao,
                          int ngrid, int ld, const double *dm, int ntile, const double *aoe, int lda) {

#define DQC_DENS_CASE(N)                                                                                           \
    case N:                                                                                                        \
        if constexpr (!GGA || N <= 14) { /* GGA panels of 15 / 16 tiles would spill: never instantiated */          \
            (void)hipFuncSetAttribute((const void *)density_kernel<N, GGA>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)density_lds_bytes<N>());                                                \
            hipLaunchKernelGGL((density_kernel<N, GGA>), grid, dim3(DEN_NT), density_lds_bytes<N>(), st, rho, grho, ao, \
                               ngrid, ld, dm, ntile, aoe, lda);                                                     \
            return 0;                                                                                              \
        }                                                                                                          \
        break;
    switch (nct) {
        DQC_DENS_CASE(1) DQC_DENS_CASE(2) DQC_DENS_CASE(3) DQC_DENS_CASE(4) DQC_DENS_CASE(5) DQC_DENS_CASE(6)
        DQC_DENS_CASE(7) DQC_DENS_CASE(8) DQC_DENS_CASE(9) DQC_DENS_CASE(10) DQC_DENS_CASE(11) DQC_DENS_CASE(12)
        DQC_DENS_CASE(13) DQC_DENS_CASE(14) DQC_DENS_CASE(15) DQC_DENS_CASE(16)
    default:
        break;
    }
#undef DQC_DENS_CASE
    set_error("density: internal tile-count dispatch error");
    return DQC_EINVAL;
}

// ---------------------------------------------------------------------------------------------
// density from the orbital factor.  Every density matrix an SCF iteration feeds to the grid pass is
// D = C_occ diag(n) C_occ^T (reference: HamiltonCGTO.ao_orb2dm hcgto.py:272-281), i.e. D = L L^T with
// L = C_occ sqrt(n) of r = n_occ columns.  Two chained MFMA GEMMs replace Phi . D:
//     phase 1   A'^T[r x 16 pts] = L^T . Phi_blk^T     (K = nao)
//     phase 2   B[16 pts x nao]  = A' . L^T            (K = r)
// 2 * 2*G*n*r flops instead of 2*G*n^2 (0.46x for the 20-atom cc-pVDZ molecules).  Phase 1 is computed
// TRANSPOSED so that its accumulators C[row = r-index (lane>>4)+4*reg][col = point lane&15] are, register by
// register, exactly the A-operand fragments A[i = point][k = r-index] phase 2 needs: the intermediate never
// leaves the VGPRs.  Phase 2 and the row-dot epilogue are those of density_kernel.
//   orb  (ld x RP)  row-major, zero padded;  orbt (RP x ld) its transpose;  RP = 16 * NRT.
// ---------------------------------------------------------------------------------------------

template <int NRT>
struct LrGeom {
    static constexpr int RP = NRT * 16;
    static constexpr int RPS = (RP & 31) == 16 ? RP : RP + 16;  // LDS row stride of the L chunk
};

// NS = 2: the factor holds TWO spin channels of NRT / 2 tiles each, [L_u | L_d] (orbt: their transposes stacked).  Phase 1 is the
// same GEMM; rho_s comes from the s-th half of its accumulators; phase 2 keeps one accumulator set per spin (chunk kc of the
// stacked L^T belongs to spin kc / (NRT / 2)) and the row-dot epilogue forms both spins' gradients from ONE read of the gradient
// arrays.  Outputs: rho (NS, ngrid), grho (NS, 3, ngrid).  An unrestricted Kohn-Sham build read the AO matrix twice before.
template <int NRT, int NCT, bool GGA, int NS = 1>
__global__ __launch_bounds__(256, 2) void density_lr_kernel(double *__restrict__ rho, double *__restrict__ grho,
                                                            const double *__restrict__ ao, int ngrid, int ld,
                                                            const double *__restrict__ orb,
                                                            const double *__restrict__ orbt, int ntile, int lda) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RP = LrGeom<NRT>::RP, RPS = LrGeom<NRT>::RPS;
    constexpr int LSB = NCT * 16;
    constexpr int LSBP = lr_panel_stride(NCT);
    constexpr int A_SZ = DEN_BM * DEN_SA;
    constexpr int B_SZ = DEN_KC * (LSBP > RPS ? LSBP : RPS);
    constexpr int NL2 = (DEN_KC * RP / 2 + DEN_NT - 1) / DEN_NT;   // double2 loads of the L chunk per thread
    constexpr int NB2 = (DEN_KC * LSB / 2 + DEN_NT - 1) / DEN_NT;  // double2 loads of the L^T chunk per thread
    double *sA = lds, *sB = lds + 2 * A_SZ;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int g0 = blockIdx.x * DEN_BM;
    const size_t cs = (size_t)ngrid * lda;       // lda: row stride of the AO arrays; ld = 16 ntile (see density_kernel)
    // all global addresses are (uniform 64-bit base) + (small 32-bit lane offset): one VGPR per address
    const double *aoblk = ao + (size_t)g0 * lda; // this block's 64 rows of Phi
    const int rmax = ngrid - 1 - g0;             // last valid block-local row
    const int arow = tid >> 2, aseg = (tid & 3) * 4;
    const int aoff = min(arow, rmax) * lda + aseg;

    DEN_TRACE_POINT(0);
    // ---- phase 1: a1[ct][reg] = A'[pt = lr][r = 16 ct + 4 reg + lk]
    v4d a1[NRT];
#pragma unroll
    for (int ct = 0; ct < NRT; ct++) a1[ct] = v4d{0, 0, 0, 0};
    {
        // two register sets: the loads of chunk kc+2 are issued before the MFMAs of chunk kc, i.e. two chunks ahead
        // (a phase-1 chunk is only 4 x NRT MFMAs per wave -- one chunk of lead does not cover the HBM latency).
        // Macros, not lambdas: register sets passed by reference end up in scratch.
        // scalars, not arrays: register sets that live across the loop back-edge are otherwise left in scratch
        static_assert(NL2 <= 4, "L chunk wider than 4 double2 per thread");
        double2 pa0a, pa0b, pa1a, pa1b, pl0a, pl0b, pl0c, pl0d, pl1a, pl1b, pl1c, pl1d;
        pl0a = pl0b = pl0c = pl0d = pl1a = pl1b = pl1c = pl1d = make_double2(0.0, 0.0);
        const int nk = ld / DEN_KC;
        const int lo0 = min((tid + 0 * DEN_NT) * 2, DEN_KC * RP - 2), lo1 = min((tid + 1 * DEN_NT) * 2, DEN_KC * RP - 2);
        const int lo2 = min((tid + 2 * DEN_NT) * 2, DEN_KC * RP - 2), lo3 = min((tid + 3 * DEN_NT) * 2, DEN_KC * RP - 2);
#define DQC_LR_PREFETCH(KC, S)                                                                            \
    {                                                                                                     \
        const int kq = min((KC), nk - 1); /* past-the-end chunks re-read the last one, never staged */     \
        const double *s_ = aoblk + kq * DEN_KC + aoff;                                                    \
        pa##S##a = *reinterpret_cast<const double2 *>(s_);                                                \
        pa##S##b = *reinterpret_cast<const double2 *>(s_ + 2);                                            \
        const double *l_ = orb + (size_t)kq * DEN_KC * RP; /* the (KC x RP) chunk is contiguous in orb */ \
        pl##S##a = *reinterpret_cast<const double2 *>(l_ + lo0);                                          \
        if (NL2 > 1) pl##S##b = *reinterpret_cast<const double2 *>(l_ + lo1);                             \
        if (NL2 > 2) pl##S##c = *reinterpret_cast<const double2 *>(l_ + lo2);                             \
        if (NL2 > 3) pl##S##d = *reinterpret_cast<const double2 *>(l_ + lo3);                             \
    }
#define DQC_LR_PUT(I, V)                                                                                  \
    {                                                                                                     \
        const int e_ = (tid + (I) * DEN_NT) * 2;                                                          \
        const int row_ = e_ / RP, col_ = e_ - row_ * RP;                                                  \
        if (row_ < DEN_KC) *reinterpret_cast<double2 *>(sB + buf_ * B_SZ + row_ * RPS + col_) = V;        \
    }
#define DQC_LR_STAGE(KC, S)                                                                               \
    if ((KC) < nk) {                                                                                      \
        const int buf_ = (KC) & 1;                                                                        \
        double *a_ = sA + buf_ * A_SZ + arow * DEN_SA + aseg;                                             \
        *reinterpret_cast<double2 *>(a_) = pa##S##a;                                                      \
        *reinterpret_cast<double2 *>(a_ + 2) = pa##S##b;                                                  \
        DQC_LR_PUT(0, pl##S##a)                                                                           \
        if (NL2 > 1) DQC_LR_PUT(1, pl##S##b)                                                              \
        if (NL2 > 2) DQC_LR_PUT(2, pl##S##c)                                                              \
        if (NL2 > 3) DQC_LR_PUT(3, pl##S##d)                                                              \
    }
#define DQC_LR_MFMAS(KC)                                                                                  \
    {                                                                                                     \
        const int buf_ = (KC) & 1;                                                                        \
        const double *b_ = sA + buf_ * A_SZ + (wave * 16 + lr) * DEN_SA + lk; /* Phi[pt][ao]  (B operand) */ \
        const double *a_ = sB + buf_ * B_SZ + lk * RPS + lr;                  /* L[ao][r]     (A operand) */ \
        _Pragma("unroll") for (int kk = 0; kk < DEN_KC / 4; kk++) {                                       \
            const double bv = b_[kk * 4];                                                                 \
            _Pragma("unroll") for (int ct = 0; ct < NRT; ct++)                                            \
                a1[ct] = mfma_f64(a_[kk * 4 * RPS + ct * 16], bv, a1[ct]);                                \
        }                                                                                                 \
    }
        DQC_LR_PREFETCH(0, 0)
        DQC_LR_PREFETCH(1, 1)
        DQC_LR_STAGE(0, 0)
        __syncthreads();
        int kc = 0;
        for (; kc + 1 < nk; kc += 2) {
            DQC_LR_PREFETCH(kc + 2, 0)
            DQC_LR_MFMAS(kc)
            DQC_LR_STAGE(kc + 1, 1)
            __syncthreads();
            DQC_LR_PREFETCH(kc + 3, 1)
            DQC_LR_MFMAS(kc + 1)
            DQC_LR_STAGE(kc + 2, 0)
            __syncthreads();
        }
        if (kc < nk) {  // odd chunk count: the last chunk was staged by the loop's second half
            DQC_LR_MFMAS(kc)
            __syncthreads();
        }
#undef DQC_LR_PREFETCH
#undef DQC_LR_STAGE
#undef DQC_LR_PUT
#undef DQC_LR_MFMAS
    }

    // rho_g = sum_r A'[g][r]^2 straight from the phase-1 accumulators (lane (lr = point, lk) holds r = 16 ct + 4 reg + lk):
    // Phi is never read a second time
    static_assert(NS == 1 || (NS == 2 && NRT % 2 == 0), "two spin channels of equal padded width");
    constexpr int NRU = NRT / NS;  // factor tiles per spin channel
#pragma unroll
    for (int sp = 0; sp < NS; sp++) {
        double rs = 0.0;
#pragma unroll
        for (int ct = sp * NRU; ct < (sp + 1) * NRU; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) rs += a1[ct][q] * a1[ct][q];
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        const int row = g0 + wave * 16 + lr;
        if (lk == 0 && row < ngrid) rho[(size_t)sp * ngrid + row] = rs;
    }
    if (!GGA) return;

    double p[NS][4][GGA ? 4 : 1];
#pragma unroll
    for (int sp = 0; sp < NS; sp++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < (GGA ? 4 : 1); q++) p[sp][r][q] = 0.0;
    int roff[4];  // block-local element offsets of this lane's four accumulator rows
#pragma unroll
    for (int r = 0; r < 4; r++) roff[r] = min(wave * 16 + lk + 4 * r, rmax) * lda + (DEN_PAIRED ? 2 : 1) * lr;

    // ---- phase 2 + epilogue, one column panel of NCT tiles at a time.  Every panel is a full one: the last panel
    // is shifted back to end at ntile and the tiles it shares with its predecessor (tile index < jnew) get zero L^T
    // columns, so nothing is counted twice and nothing is read past ld.
    for (int jnew = 0; jnew < ntile; jnew += NCT) {
        const int jc = min(jnew, ntile - NCT);
        v4d acc[NS][NCT];
#pragma unroll
        for (int sp = 0; sp < NS; sp++)
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[sp][ct] = v4d{0, 0, 0, 0};
        double2 pb[NB2];
        int boff[NB2];
        bool bzero[NB2];
#pragma unroll
        for (int i = 0; i < NB2; i++) {
            const int e = min((tid + i * DEN_NT) * 2, DEN_KC * LSB - 2);
            const int row = e / LSB, col = e - row * LSB;
            boff[i] = row * ld + jc * 16 + col;
            bzero[i] = jc * 16 + col < jnew * 16;
        }
        auto prefetch = [&](int kc) {
            const double *l = orbt + (size_t)kc * DEN_KC * ld;
#pragma unroll
            for (int i = 0; i < NB2; i++) {
                pb[i] = *reinterpret_cast<const double2 *>(l + boff[i]);
                if (bzero[i]) pb[i] = make_double2(0.0, 0.0);
            }
        };
        auto stage = [&](int buf) {
#pragma unroll
            for (int i = 0; i < NB2; i++) {
                const int e = (tid + i * DEN_NT) * 2;
                const int row = e / LSB, col = e - row * LSB;
                if (row < DEN_KC) {  // two 8-byte stores: an odd row stride leaves every other row 8-byte aligned only
                    double *d = sB + buf * B_SZ + row * LSBP + col;
                    d[0] = pb[i].x;
                    d[1] = pb[i].y;
                }
            }
        };
        __syncthreads();
        prefetch(0);
        stage(0);
        __syncthreads();
#pragma unroll 1
        for (int kc = 0; kc < NRT; kc++) {  // rolled (one live set of prefetch registers); a1[kc] by uniform select
            const int buf = kc & 1;
            if (kc + 1 < NRT) prefetch(kc + 1);
            v4d a4 = a1[0];
#pragma unroll
            for (int c = 1; c < NRT; c++)
                if (kc == c) a4 = a1[c];
            // B fragments with the panel's columns permuted (rowdot_epilogue_paired): tile 2 m + h, lane lr <- column
            // 32 m + 2 lr + h (immediate offsets from one more base address)
            const double *b = sB + buf * B_SZ + lk * LSBP + lr;
            const double *b2 = b + lr;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const double av = a4[kk];
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) {
                    const double bv = (DEN_PAIRED && ct < 2 * (NCT / 2)) ? b2[kk * 4 * LSBP + 32 * (ct >> 1) + (ct & 1)]
                                                                        : b[kk * 4 * LSBP + ct * 16];
                    if (NS == 1 || kc < NRU) acc[0][ct] = mfma_f64(av, bv, acc[0][ct]);  // (kc is wave-uniform)
                    else acc[NS - 1][ct] = mfma_f64(av, bv, acc[NS - 1][ct]);
                }
            }
            if (kc + 1 < NRT) stage(buf ^ 1);
            __syncthreads();
        }
        DEN_TRACE_POINT(1);
        if constexpr (NS == 2) {
            static_assert(NS == 1 || (DEN_PAIRED && GGA), "spin-fused form: paired epilogue only");
            rowdot_epilogue_paired2<NCT>(acc, p, aoblk, cs, roff, lr, jc * 16);
        } else if constexpr (!DEN_PAIRED) rowdot_epilogue<NCT, GGA, 1>(acc[0], p[0], aoblk, aoblk, cs, roff, jc * 16);
        else if constexpr (GGA) rowdot_epilogue_paired<NCT, true, 1>(acc[0], p[0], aoblk, aoblk, cs, roff, lr, jc * 16);
        DEN_TRACE_POINT(2);
    }
#pragma unroll
    for (int sp = 0; sp < NS; sp++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < (GGA ? 4 : 1); q++) {
            double v = p[sp][r][q];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            p[sp][r][q] = v;
        }
    if (lr == 0) {
#pragma unroll
        for (int sp = 0; sp < NS; sp++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = g0 + wave * 16 + lk + 4 * r;
            if (row < ngrid) {
                double *gs = grho + (size_t)sp * 3 * ngrid;
                gs[row] = 2.0 * p[sp][r][1];
                gs[(size_t)ngrid + row] = 2.0 * p[sp][r][2];
                gs[2 * (size_t)ngrid + row] = 2.0 * p[sp][r][3];
            }
        }
    }
}

template <int NRT, int NCT>
static constexpr size_t density_lr_lds_bytes() {
    constexpr int LSBP = lr_panel_stride(NCT);
    constexpr int RPS = LrGeom<NRT>::RPS;
    return sizeof(double) * 2 * (DEN_BM * DEN_SA + DEN_KC * (LSBP > RPS ? LSBP : RPS));
}

// widest phase-2 column panel per factor width (NRT tiles) that compiles without VGPR spills in GGA mode: 8 NRT phase-1 +
// 8 NCT phase-2 accumulator registers + the epilogue's 4 NCT load-batch registers share 256; wider bases take more panels
constexpr int lr_max_nct(int nrt) { return nrt <= 3 ? 14 : (nrt <= 4 ? 12 : 10); }

template <int NRT, bool GGA>
static int launch_density_lr_n(int nct, dim3 grid, hipStream_t st, double *rho, double *grho, const double *ao, int ngrid,
                               int ld, const double *orb, const double *orbt, int ntile, int lda) {
#define DQC_DLR_CASE(N)                                                                                          \
    case N:                                                                                                      \
        if constexpr (!GGA || N <= lr_max_nct(NRT)) { /* wider panels would spill: never instantiated */         \
            constexpr size_t shm = density_lr_lds_bytes<NRT, N>();                                               \
            auto kern = density_lr_kernel<NRT, N, GGA>;                                                          \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
            hipLaunchKernelGGL(kern, grid, dim3(DEN_NT), shm, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda); \
            return 0;                                                                                            \
        }                                                                                                        \
        break;
    switch (nct) {
        DQC_DLR_CASE(1) DQC_DLR_CASE(2) DQC_DLR_CASE(3) DQC_DLR_CASE(4) DQC_DLR_CASE(5) DQC_DLR_CASE(6) DQC_DLR_CASE(7)
        DQC_DLR_CASE(8) DQC_DLR_CASE(9) DQC_DLR_CASE(10) DQC_DLR_CASE(11) DQC_DLR_CASE(12) DQC_DLR_CASE(13) DQC_DLR_CASE(14)
        DQC_DLR_CASE(15) DQC_DLR_CASE(16)
    default:
        break;
    }
#undef DQC_DLR_CASE
    set_error("density_lr: internal tile-count dispatch error");
    return DQC_EINVAL;
}

// ---------------------------------------------------------------------------------------------
// meta-GGA densities from the orbital factor in ONE pass (round 4): psi = Phi L and d_d psi = (d_d Phi) L for the three gradient
// components -- four phase-1 GEMMs of density_lr_kernel, no phase 2 and no row-dot epilogue --
//     rho = sum_r psi_r^2,   grad_d rho = 2 sum_r psi_r d_d psi_r,   tau = 1/2 sum_d sum_r (d_d psi_r)^2        (hcgto.py:398-438)
// The AO components are read once, as GEMM operands staged through LDS (coalesced, two chunks of prefetch), instead of once by
// density_lr_kernel (value: GEMM operand; gradients: latency-bound row dots) plus once more by three value-only passes for tau.
// Two components per sweep over K (LDS: 2 x 2 A chunks + 2 L chunks = 49 KB per block), psi kept in registers for the second sweep.
// ---------------------------------------------------------------------------------------------
template <int NRT>
__global__ __launch_bounds__(256, 2) void density_lr_tau_kernel(double *__restrict__ rho, double *__restrict__ grho, double *__restrict__ tau,
                                                                const double *__restrict__ ao, int ngrid, int ld,
                                                                const double *__restrict__ orb, int lda) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RP = LrGeom<NRT>::RP, RPS = LrGeom<NRT>::RPS;
    constexpr int A_SZ = DEN_BM * DEN_SA, B_SZ = DEN_KC * RPS;
    constexpr int NL2 = (DEN_KC * RP / 2 + DEN_NT - 1) / DEN_NT;
    static_assert(NL2 <= 4, "L chunk wider than 4 double2 per thread");
    double *sA = lds, *sB = lds + 4 * A_SZ;  // sA: [buffer][component of the pair][A_SZ]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int g0 = blockIdx.x * DEN_BM;
    const size_t cs = (size_t)ngrid * lda;
    const double *aoblk = ao + (size_t)g0 * lda;
    const int rmax = ngrid - 1 - g0;
    const int arow = tid >> 2, aseg = (tid & 3) * 4;
    const int aoff = min(arow, rmax) * lda + aseg;
    const int nk = ld / DEN_KC;
    const int lo0 = min((tid + 0 * DEN_NT) * 2, DEN_KC * RP - 2), lo1 = min((tid + 1 * DEN_NT) * 2, DEN_KC * RP - 2);
    const int lo2 = min((tid + 2 * DEN_NT) * 2, DEN_KC * RP - 2), lo3 = min((tid + 3 * DEN_NT) * 2, DEN_KC * RP - 2);

    v4d psi[NRT];
    double rs = 0.0, ts = 0.0, gs[3] = {0.0, 0.0, 0.0};
#pragma unroll 1
    for (int pair = 0; pair < 2; pair++) {
        const double *blk0 = aoblk + (size_t)(2 * pair) * cs, *blk1 = blk0 + cs;
        v4d a0[NRT], a1[NRT];
#pragma unroll
        for (int ct = 0; ct < NRT; ct++) a0[ct] = a1[ct] = v4d{0, 0, 0, 0};
        double2 x0a, x0b, y0a, y0b, x1a, x1b, y1a, y1b, pl0a, pl0b, pl0c, pl0d, pl1a, pl1b, pl1c, pl1d;
        pl0a = pl0b = pl0c = pl0d = pl1a = pl1b = pl1c = pl1d = make_double2(0.0, 0.0);
#define DQC_LT_PREFETCH(KC, S)                                                                            \
    {                                                                                                     \
        const int kq = min((KC), nk - 1); /* past-the-end chunks re-read the last one, never staged */     \
        const double *s0_ = blk0 + kq * DEN_KC + aoff, *s1_ = blk1 + kq * DEN_KC + aoff;                  \
        x##S##a = *reinterpret_cast<const double2 *>(s0_);                                                \
        x##S##b = *reinterpret_cast<const double2 *>(s0_ + 2);                                            \
        y##S##a = *reinterpret_cast<const double2 *>(s1_);                                                \
        y##S##b = *reinterpret_cast<const double2 *>(s1_ + 2);                                            \
        const double *l_ = orb + (size_t)kq * DEN_KC * RP;                                                \
        pl##S##a = *reinterpret_cast<const double2 *>(l_ + lo0);                                          \
        if (NL2 > 1) pl##S##b = *reinterpret_cast<const double2 *>(l_ + lo1);                             \
        if (NL2 > 2) pl##S##c = *reinterpret_cast<const double2 *>(l_ + lo2);                             \
        if (NL2 > 3) pl##S##d = *reinterpret_cast<const double2 *>(l_ + lo3);                             \
    }
#define DQC_LT_PUT(I, V)                                                                                  \
    {                                                                                                     \
        const int e_ = (tid + (I) * DEN_NT) * 2;                                                          \
        const int row_ = e_ / RP, col_ = e_ - row_ * RP;                                                  \
        if (row_ < DEN_KC) *reinterpret_cast<double2 *>(sB + buf_ * B_SZ + row_ * RPS + col_) = V;        \
    }
#define DQC_LT_STAGE(KC, S)                                                                               \
    if ((KC) < nk) {                                                                                      \
        const int buf_ = (KC) & 1;                                                                        \
        double *a_ = sA + buf_ * 2 * A_SZ + arow * DEN_SA + aseg;                                         \
        *reinterpret_cast<double2 *>(a_) = x##S##a;                                                       \
        *reinterpret_cast<double2 *>(a_ + 2) = x##S##b;                                                   \
        *reinterpret_cast<double2 *>(a_ + A_SZ) = y##S##a;                                                \
        *reinterpret_cast<double2 *>(a_ + A_SZ + 2) = y##S##b;                                            \
        DQC_LT_PUT(0, pl##S##a)                                                                           \
        if (NL2 > 1) DQC_LT_PUT(1, pl##S##b)                                                              \
        if (NL2 > 2) DQC_LT_PUT(2, pl##S##c)                                                              \
        if (NL2 > 3) DQC_LT_PUT(3, pl##S##d)                                                              \
    }
#define DQC_LT_MFMAS(KC)                                                                                  \
    {                                                                                                     \
        const int buf_ = (KC) & 1;                                                                        \
        const double *b_ = sA + buf_ * 2 * A_SZ + (wave * 16 + lr) * DEN_SA + lk; /* Phi_c[pt][ao] (B operand) */ \
        const double *l_ = sB + buf_ * B_SZ + lk * RPS + lr;                      /* L[ao][r]      (A operand) */ \
        _Pragma("unroll") for (int kk = 0; kk < DEN_KC / 4; kk++) {                                       \
            const double bv0 = b_[kk * 4], bv1 = b_[A_SZ + kk * 4];                                       \
            _Pragma("unroll") for (int ct = 0; ct < NRT; ct++) {                                          \
                const double lv = l_[kk * 4 * RPS + ct * 16];                                             \
                a0[ct] = mfma_f64(lv, bv0, a0[ct]);                                                       \
                a1[ct] = mfma_f64(lv, bv1, a1[ct]);                                                       \
            }                                                                                             \
        }                                                                                                 \
    }
        __syncthreads();  // (the previous sweep's last chunk fully consumed)
        DQC_LT_PREFETCH(0, 0)
        DQC_LT_PREFETCH(1, 1)
        DQC_LT_STAGE(0, 0)
        __syncthreads();
        int kc = 0;
        for (; kc + 1 < nk; kc += 2) {
            DQC_LT_PREFETCH(kc + 2, 0)
            DQC_LT_MFMAS(kc)
            DQC_LT_STAGE(kc + 1, 1)
            __syncthreads();
            DQC_LT_PREFETCH(kc + 3, 1)
            DQC_LT_MFMAS(kc + 1)
            DQC_LT_STAGE(kc + 2, 0)
            __syncthreads();
        }
        if (kc < nk) {
            DQC_LT_MFMAS(kc)
            __syncthreads();
        }
#undef DQC_LT_PREFETCH
#undef DQC_LT_STAGE
#undef DQC_LT_PUT
#undef DQC_LT_MFMAS
        // lane (lr = point, lk) holds r = 16 ct + 4 q + lk of both components
        if (pair == 0) {  // a0 = psi, a1 = d_x psi
#pragma unroll
            for (int ct = 0; ct < NRT; ct++) {
                psi[ct] = a0[ct];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    rs += a0[ct][q] * a0[ct][q];
                    gs[0] += a0[ct][q] * a1[ct][q];
                    ts += a1[ct][q] * a1[ct][q];
                }
            }
        } else {  // a0 = d_y psi, a1 = d_z psi
#pragma unroll
            for (int ct = 0; ct < NRT; ct++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    gs[1] += psi[ct][q] * a0[ct][q];
                    gs[2] += psi[ct][q] * a1[ct][q];
                    ts += a0[ct][q] * a0[ct][q] + a1[ct][q] * a1[ct][q];
                }
        }
    }
    double out[5] = {rs, 2.0 * gs[0], 2.0 * gs[1], 2.0 * gs[2], 0.5 * ts};
#pragma unroll
    for (int q = 0; q < 5; q++) {
        out[q] += __shfl_xor(out[q], 16);
        out[q] += __shfl_xor(out[q], 32);
    }
    const int row = g0 + wave * 16 + lr;
    if (lk == 0 && row < ngrid) {
        rho[row] = out[0];
        grho[row] = out[1];
        grho[(size_t)ngrid + row] = out[2];
        grho[2 * (size_t)ngrid + row] = out[3];
        tau[row] = out[4];
    }
}

// spin-fused form: two accumulator sets in phase 2 -- panels of at most this many tiles (8 NRT + 16 NCT + 4 NCT + 48 registers)
constexpr int lr2_max_nct(int nrt) { return nrt <= 4 ? 6 : (nrt <= 6 ? 5 : 4); }

template <int NRT>
static int launch_density_lr2_n(int nct, dim3 grid, hipStream_t st, double *rho, double *grho, const double *ao, int ngrid, int ld,
                                const double *orb, const double *orbt, int ntile, int lda) {
#define DQC_DLR2_CASE(N)                                                                                         \
    case N:                                                                                                      \
        if constexpr (N <= lr2_max_nct(NRT)) {                                                                   \
            constexpr size_t shm = density_lr_lds_bytes<NRT, N>();                                               \
            auto kern = density_lr_kernel<NRT, N, true, 2>;                                                      \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
            hipLaunchKernelGGL(kern, grid, dim3(DEN_NT), shm, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda); \
            return 0;                                                                                            \
        }                                                                                                        \
        break;
    switch (nct) {
        DQC_DLR2_CASE(1) DQC_DLR2_CASE(2) DQC_DLR2_CASE(3) DQC_DLR2_CASE(4) DQC_DLR2_CASE(5) DQC_DLR2_CASE(6) DQC_DLR2_CASE(7)
    default:
        break;
    }
#undef DQC_DLR2_CASE
    set_error("density_lr (two spins): internal tile-count dispatch error");
    return DQC_EINVAL;
}

template <bool GGA>
static int launch_density_lr(int nrt, int nct, dim3 grid, hipStream_t st, double *rho, double *grho, const double *ao,
                             int ngrid, int ld, const double *orb, const double *orbt, int ntile, int lda) {
    switch (nrt) {
    case 1: return launch_density_lr_n<1, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda);
    case 2: return launch_density_lr_n<2, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda);
    case 3: return launch_density_lr_n<3, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda);
    case 4: return launch_density_lr_n<4, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda);
    case 6: return launch_density_lr_n<6, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda);
    case 8: return launch_density_lr_n<8, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile, lda);
    default:
        set_error("density_lr: internal factor-width dispatch error");
        return DQC_EINVAL;
    }
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
    const int ld = dqc_padded_nao(nao), ntile = ld / 16, lda = dqc_ao_stride(nao);
    // column panels: <= 16 tiles (LDA) / <= 14 (GGA: 15 and 16 tiles of accumulators + the epilogue's load batches spill)
    const int nchunk = (ntile + (gga ? 13 : 15)) / (gga ? 14 : 16);
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc = gga ? launch_density<true>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_dm, ntile, d_ao, lda)
                 : launch_density<false>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_dm, ntile, d_ao, lda);
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_padded_norb(int norb) {
    static const int sizes[] = {16, 32, 48, 64, 96, 128};
    for (int sz : sizes)
        if (norb <= sz) return sz;
    return 0;  // wider factors: use dqc_grid_density with the full matrix
}

int dqc_grid_density_lr(double *d_rho, double *d_grho, const double *d_ao, int ncomp, int ngrid, int nao,
                        const double *d_orb, const double *d_orbt, int norb_pad, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0) return DQC_OK;
    const bool gga = d_grho != nullptr;
    if (gga && ncomp < 4) { set_error("dqc_grid_density_lr: gradient requested but ao has < 4 components"); return DQC_EINVAL; }
    if (norb_pad <= 0 || dqc_padded_norb(norb_pad) != norb_pad) {
        set_error("dqc_grid_density_lr: norb_pad must be a value returned by dqc_padded_norb");
        return DQC_EINVAL;
    }
    const int ld = dqc_padded_nao(nao), ntile = ld / 16, lda = dqc_ao_stride(nao);
    // (narrower panels -- fewer registers and less LDS, 3 blocks per CU instead of 2 -- change nothing: 0.57 ms for 5, 7, 9 or 13 tiles)
    const int lim = gga ? dqc::lr_max_nct(norb_pad / 16) : 16;
    const int nchunk = (ntile + lim - 1) / lim;
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc = gga ? launch_density_lr<true>(norb_pad / 16, nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile, lda)
                 : launch_density_lr<false>(norb_pad / 16, nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile, lda);
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_grid_density_lr_pol(double *d_rho, double *d_grho, const double *d_ao, int ncomp, int ngrid, int nao,
                            const double *d_orb, const double *d_orbt, int norb_pad_spin, void *stream) {
    // both spin densities of an unrestricted calculation from ONE pass over the AO matrix: d_orb (ld, 2 norb_pad_spin) = [L_u | L_d]
    // row-major, d_orbt (2 norb_pad_spin, ld) its transpose; d_rho (2, ngrid), d_grho (2, 3, ngrid) -- GGA only (d_grho != NULL);
    // norb_pad_spin in {16, 32, 48, 64}.  Enqueues only.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0) return DQC_OK;
    if (!d_grho || ncomp < 4) { set_error("dqc_grid_density_lr_pol: needs the four AO components and the gradient output"); return DQC_EINVAL; }
    if (norb_pad_spin != 16 && norb_pad_spin != 32 && norb_pad_spin != 48 && norb_pad_spin != 64) {
        set_error("dqc_grid_density_lr_pol: norb_pad_spin must be 16, 32, 48 or 64");
        return DQC_EINVAL;
    }
    const int ld = dqc_padded_nao(nao), ntile = ld / 16, lda = dqc_ao_stride(nao);
    const int nrt = 2 * norb_pad_spin / 16;
    const int lim = lr2_max_nct(nrt);
    const int nchunk = (ntile + lim - 1) / lim;
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc;
    switch (nrt) {
    case 2: rc = launch_density_lr2_n<2>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile, lda); break;
    case 4: rc = launch_density_lr2_n<4>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile, lda); break;
    case 6: rc = launch_density_lr2_n<6>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile, lda); break;
    default: rc = launch_density_lr2_n<8>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile, lda); break;
    }
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_grid_density_pair(double *d_out, const double *d_ao_a, const double *d_ao_b, int ngrid, int nao,
                          const double *d_dm, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0) return DQC_OK;
    const int ld = dqc_padded_nao(nao), ntile = ld / 16, lda = dqc_ao_stride(nao);
    const int nchunk = (ntile + 15) / 16;
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc = launch_density<false>(nct, grid, st, d_out, nullptr, d_ao_a, ngrid, ld, d_dm, ntile, d_ao_b, lda);
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_grid_density_lr_tau(double *d_rho, double *d_grho, double *d_tau, const double *d_ao, int ncomp, int ngrid, int nao,
                            const double *d_orb, int norb_pad, void *stream) {
    // rho (ngrid), grad rho (3, ngrid) and tau = 1/2 sum |grad psi|^2 (ngrid) of D = L L^T from ONE pass over the four AO components
    // (density_lr_tau_kernel); d_orb: the padded factor as for dqc_grid_density_lr, norb_pad <= 96.  Enqueues only.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0) return DQC_OK;
    if (ncomp < 4 || !d_grho || !d_tau) { set_error("dqc_grid_density_lr_tau: needs the four AO components and all three outputs"); return DQC_EINVAL; }
    if (norb_pad <= 0 || dqc_padded_norb(norb_pad) != norb_pad) {
        set_error("dqc_grid_density_lr_tau: norb_pad must be a value returned by dqc_padded_norb");
        return DQC_EINVAL;
    }
    const int ld = dqc_padded_nao(nao), lda = dqc_ao_stride(nao);
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
#define DQC_LT_CASE(N)                                                                                                         \
    case N: {                                                                                                                  \
        constexpr size_t bytes = sizeof(double) * (4 * DEN_BM * DEN_SA + 2 * DEN_KC * LrGeom<N>::RPS);                         \
        (void)hipFuncSetAttribute((const void *)density_lr_tau_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        hipLaunchKernelGGL((density_lr_tau_kernel<N>), grid, dim3(DEN_NT), bytes, st, d_rho, d_grho, d_tau, d_ao, ngrid, ld, d_orb, lda); \
    } break;
    switch (norb_pad / 16) {
        DQC_LT_CASE(1) DQC_LT_CASE(2) DQC_LT_CASE(3) DQC_LT_CASE(4) DQC_LT_CASE(6)
    default:  // (128 orbitals: three accumulator sets of 8 tiles spill -- the caller takes the separate passes)
        set_error("dqc_grid_density_lr_tau: factors wider than 96 columns are not supported");
        return DQC_EINVAL;
    }
#undef DQC_LT_CASE
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
