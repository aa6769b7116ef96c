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
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "xc_funcs.hpp"

namespace dqc {

typedef double v4d __attribute__((ext_vector_type(4)));

DQC_DEV v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------
// density:  C[64 pts x n] = Phi_blk . D, K-chunks of 16 staged in double-buffered LDS, 4 waves per block,
// wave w owns points 16w..16w+15 and all column tiles (accumulators in registers), fused row-dot epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int DEN_BM = 64;    // points per block (4 waves x 16); two blocks share a CU so that one block's
                              // memory-bound epilogue overlaps the other's MFMA main loop
constexpr int DEN_NT = DEN_BM * 4;  // threads per block
constexpr int DEN_KC = 16;    // K chunk
constexpr int DEN_SA = DEN_KC + 2;  // LDS row stride of the A chunk (conflict-free ds_read_b64 fragments)

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

template <int NCT, bool GGA>
__global__ __launch_bounds__(256, 2) void density_kernel(double *__restrict__ rho, double *__restrict__ grho,
                                                         const double *__restrict__ ao, int ngrid, int ld,
                                                         const double *__restrict__ dm, int ntile,
                                                         const double *__restrict__ aoe) {
    // aoe: array the row dots are taken with (== ao except for the "pair" form rowdot(ao . D, aoe), LDA mode only)
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
    const size_t cs = (size_t)ngrid * ld;
    // all global addresses are (uniform 64-bit base) + (small 32-bit lane offset)
    const double *aoblk = ao + (size_t)g0 * ld;    // this block's 64 rows of Phi
    const double *aoeblk = aoe + (size_t)g0 * ld;
    const int rmax = ngrid - 1 - g0;               // last valid block-local row
    // staging roles: A chunk = 64 rows x 16 doubles -> thread (row = tid/4, 4 doubles at seg = tid%4)
    const int arow = tid >> 2, aseg = (tid & 3) * 4;
    const int aoff = min(arow, rmax) * ld + aseg;

    double p[4][GGA ? 4 : 1];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < (GGA ? 4 : 1); q++) p[r][q] = 0.0;
    int roff[4];  // block-local element offsets of this lane's four accumulator rows
#pragma unroll
    for (int r = 0; r < 4; r++) roff[r] = min(wave * 16 + lk + 4 * r, rmax) * ld + (DEN_PAIRED ? 2 : 1) * lr;

    const int nk = ld / DEN_KC;
    // every panel is a full one: the last panel is shifted back to end at ntile and the tiles it shares with its
    // predecessor (tile index < jnew) get zero D columns, so nothing is counted twice and nothing is read past ld
    for (int jnew = 0; jnew < ntile; jnew += NCT) {
        const int jc = min(jnew, ntile - NCT);
        v4d acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[ct] = v4d{0, 0, 0, 0};
        double2 pa0 = make_double2(0.0, 0.0), pa1 = pa0, pb[NB2];  // scalars: an array would be left in scratch
        int boff[NB2];
        bool bzero[NB2];
#pragma unroll
        for (int i = 0; i < NB2; i++) {
            const int e = min((tid + i * DEN_NT) * 2, DEN_KC * LSB - 2);
            const int row = e / LSB, col = e - row * LSB;
            boff[i] = row * ld + jc * 16 + col;
            bzero[i] = jc * 16 + col < jnew * 16;
        }
        // the next chunk's loads are issued in slices between the MFMA groups of the current one (waves issue in
        // order: a wave that first pushes its whole prefetch through the address pipe starts its MFMAs late)
        auto prefetch_part = [&](int kc, int part) {
            if (part == 0) {
                const double *s_ = aoblk + kc * DEN_KC + aoff;
                pa0 = *reinterpret_cast<const double2 *>(s_);
                pa1 = *reinterpret_cast<const double2 *>(s_ + 2);
            }
            const double *d_ = dm + (size_t)kc * DEN_KC * ld;
#pragma unroll
            for (int i = 0; i < NB2; i++) {
                if (i % NKK != part) continue;
                pb[i] = *reinterpret_cast<const double2 *>(d_ + boff[i]);
                if (bzero[i]) pb[i] = make_double2(0.0, 0.0);
            }
        };
        auto stage = [&](int buf) {
            double *a = sA + buf * A_SZ + arow * DEN_SA + aseg;
            *reinterpret_cast<double2 *>(a) = pa0;
            *reinterpret_cast<double2 *>(a + 2) = pa1;
#pragma unroll
            for (int i = 0; i < NB2; i++) {
                const int e = (tid + i * DEN_NT) * 2;
                const int row = e / LSB, col = e - row * LSB;
                if (row < DEN_KC) {  // two 8-byte stores: odd row stride
                    double *d = sB + buf * B_SZ + row * LSBP + col;
                    d[0] = pb[i].x;
                    d[1] = pb[i].y;
                }
            }
        };
        __syncthreads();  // buffers free (previous column panel fully consumed)
#pragma unroll
        for (int part = 0; part < NKK; part++) prefetch_part(0, part);
        stage(0);
        __syncthreads();
        for (int kc = 0; kc < nk; kc++) {
            const int buf = kc & 1;
            const bool more = kc + 1 < nk;
            const double *a = sA + buf * A_SZ + (wave * 16 + lr) * DEN_SA + lk;
            const double *b = sB + buf * B_SZ + lk * LSBP + lr;
            const double *b2 = b + lr;  // permuted columns: tile 2 m + h, lane lr <- column 32 m + 2 lr + h
#pragma unroll
            for (int kk = 0; kk < NKK; kk++) {
                if (more) prefetch_part(kc + 1, kk);  // global loads in flight during the MFMAs
                const double av = a[kk * 4];
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) {
                    const double bv = (DEN_PAIRED && ct < 2 * (NCT / 2)) ? b2[kk * 4 * LSBP + 32 * (ct >> 1) + (ct & 1)]
                                                                        : b[kk * 4 * LSBP + ct * 16];
#ifndef ABL_DEN_NO_MFMA
                    acc[ct] = mfma_f64(av, bv, acc[ct]);
#else
                    acc[ct][0] += av * bv;
#endif
                }
            }
            if (more) stage(buf ^ 1);
            __syncthreads();
        }
        // epilogue: row dots with Phi (and its gradient) in the accumulator layout, straight from global
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
                          int ngrid, int ld, const double *dm, int ntile, const double *aoe) {
#define DQC_DENS_CASE(N)                                                                                           \
    case N:                                                                                                        \
        if constexpr (!GGA || N <= 14) { /* GGA panels of 15 / 16 tiles would spill: never instantiated */          \
            (void)hipFuncSetAttribute((const void *)density_kernel<N, GGA>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)density_lds_bytes<N>());                                                \
            hipLaunchKernelGGL((density_kernel<N, GGA>), grid, dim3(DEN_NT), density_lds_bytes<N>(), st, rho, grho, ao, \
                               ngrid, ld, dm, ntile, aoe);                                                          \
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
#ifdef DEN_TRACE  // per-block timeline for tools/ubench/den_trace.hip: CU slot, start / MFMA-end / end (100 MHz ticks)
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

template <int NRT>
struct LrGeom {
    static constexpr int RP = NRT * 16;
    static constexpr int RPS = (RP & 31) == 16 ? RP : RP + 16;  // LDS row stride of the L chunk
};

template <int NRT, int NCT, bool GGA>
__global__ __launch_bounds__(256, 2) void density_lr_kernel(double *__restrict__ rho, double *__restrict__ grho,
                                                            const double *__restrict__ ao, int ngrid, int ld,
                                                            const double *__restrict__ orb,
                                                            const double *__restrict__ orbt, int ntile) {
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
    const size_t cs = (size_t)ngrid * ld;
    // all global addresses are (uniform 64-bit base) + (small 32-bit lane offset): one VGPR per address
    const double *aoblk = ao + (size_t)g0 * ld;  // this block's 64 rows of Phi
    const int rmax = ngrid - 1 - g0;             // last valid block-local row
    const int arow = tid >> 2, aseg = (tid & 3) * 4;
    const int aoff = min(arow, rmax) * ld + aseg;

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
    {
        double rs = 0.0;
#pragma unroll
        for (int ct = 0; ct < NRT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) rs += a1[ct][q] * a1[ct][q];
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        const int row = g0 + wave * 16 + lr;
        if (lk == 0 && row < ngrid) rho[row] = rs;
    }
    if (!GGA) return;

    double p[4][GGA ? 4 : 1];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < (GGA ? 4 : 1); q++) p[r][q] = 0.0;
    int roff[4];  // block-local element offsets of this lane's four accumulator rows
#pragma unroll
    for (int r = 0; r < 4; r++) roff[r] = min(wave * 16 + lk + 4 * r, rmax) * ld + (DEN_PAIRED ? 2 : 1) * lr;

    // ---- phase 2 + epilogue, one column panel of NCT tiles at a time.  Every panel is a full one: the last panel
    // is shifted back to end at ntile and the tiles it shares with its predecessor (tile index < jnew) get zero L^T
    // columns, so nothing is counted twice and nothing is read past ld.
    for (int jnew = 0; jnew < ntile; jnew += NCT) {
        const int jc = min(jnew, ntile - NCT);
        v4d acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[ct] = v4d{0, 0, 0, 0};
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
                    acc[ct] = mfma_f64(av, bv, acc[ct]);
                }
            }
            if (kc + 1 < NRT) stage(buf ^ 1);
            __syncthreads();
        }
        DEN_TRACE_POINT(1);
        if constexpr (!DEN_PAIRED) rowdot_epilogue<NCT, GGA, 1>(acc, p, aoblk, aoblk, cs, roff, jc * 16);
        else if constexpr (GGA) rowdot_epilogue_paired<NCT, true, 1>(acc, p, aoblk, aoblk, cs, roff, lr, jc * 16);
        DEN_TRACE_POINT(2);
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
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = g0 + wave * 16 + lk + 4 * r;
            if (row < ngrid) {
                grho[row] = 2.0 * p[r][1];
                grho[(size_t)ngrid + row] = 2.0 * p[r][2];
                grho[2 * (size_t)ngrid + row] = 2.0 * p[r][3];
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
                               int ld, const double *orb, const double *orbt, int ntile) {
#define DQC_DLR_CASE(N)                                                                                          \
    case N:                                                                                                      \
        if constexpr (!GGA || N <= lr_max_nct(NRT)) { /* wider panels would spill: never instantiated */         \
            constexpr size_t shm = density_lr_lds_bytes<NRT, N>();                                               \
            auto kern = density_lr_kernel<NRT, N, GGA>;                                                          \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
            hipLaunchKernelGGL(kern, grid, dim3(DEN_NT), shm, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);   \
            return 0;                                                                                            \
        }                                                                                                        \
        break;
    switch (nct) {  // ld / 16 is odd; panels of a split matrix may be even
        DQC_DLR_CASE(1) DQC_DLR_CASE(3) DQC_DLR_CASE(5) DQC_DLR_CASE(7) DQC_DLR_CASE(9) DQC_DLR_CASE(10)
        DQC_DLR_CASE(11) DQC_DLR_CASE(12) DQC_DLR_CASE(13) DQC_DLR_CASE(14) DQC_DLR_CASE(15) DQC_DLR_CASE(16)
    default:
        break;
    }
#undef DQC_DLR_CASE
    set_error("density_lr: internal tile-count dispatch error");
    return DQC_EINVAL;
}

template <bool GGA>
static int launch_density_lr(int nrt, int nct, dim3 grid, hipStream_t st, double *rho, double *grho, const double *ao,
                             int ngrid, int ld, const double *orb, const double *orbt, int ntile) {
    switch (nrt) {
    case 1: return launch_density_lr_n<1, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);
    case 2: return launch_density_lr_n<2, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);
    case 3: return launch_density_lr_n<3, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);
    case 4: return launch_density_lr_n<4, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);
    case 6: return launch_density_lr_n<6, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);
    case 8: return launch_density_lr_n<8, GGA>(nct, grid, st, rho, grho, ao, ngrid, ld, orb, orbt, ntile);
    default:
        set_error("density_lr: internal factor-width dispatch error");
        return DQC_EINVAL;
    }
}

// ---------------------------------------------------------------------------------------------
// Vxc:  M = Phi^T . Psi, split-K over point slabs.  16-point chunks of Phi and Psi live in double-buffered
// LDS; the next chunk's four AO components are prefetched into registers while the MFMAs run, combined into
// Psi and stored to the other buffer afterwards.  The 16x16 output tiles are dealt evenly to the 8 waves of
// NSPLIT co-scheduled blocks (same XCD, so the slab is fetched from HBM once).
// ---------------------------------------------------------------------------------------------
constexpr int VXC_KC = 16;      // points per LDS chunk
constexpr int VXC_WAVES = 8;    // waves per block

template <int MAXT, int NL, int KCH, bool GGA>
__global__ __launch_bounds__(512, 2) void vxc_kernel(double *__restrict__ vmat, const double *__restrict__ ao,
                                                     int ngrid, int ld, const double *__restrict__ w,
                                                     const double *__restrict__ vrho, const double *__restrict__ vgrad,
                                                     int slab, int nsplit, int tiles_per_split,
                                                     const double *__restrict__ aob) {
    // aob: LDA mode only -- array the Psi operand is built from (== ao except for the "pair" form ao^T diag(w v) aob)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int LS = ld;  // ld == 16 (mod 32): conflict-free fragment reads without extra padding
    const int BUF = 2 * KCH * LS;  // phi + psi
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int T = ld >> 4, ttot = T * T;
    const size_t cs = (size_t)ngrid * ld;

    // XCD-aware decode: the nsplit blocks that share a slab get ids 8 apart -> same XCD, dispatched together
    const int id = blockIdx.x;
    const int grp = id / (8 * nsplit), rem = id - grp * 8 * nsplit;
    const int split = rem / 8, sl = grp * 8 + (rem & 7);
    const int gs = sl * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int tc0 = split * tiles_per_split;
    const int tc1 = min(tc0 + tiles_per_split, ttot);
    const int per_wave = (tc1 - tc0 + VXC_WAVES - 1) / VXC_WAVES;
    const int t0 = tc0 + wave * per_wave;
    const int nt = max(0, min(per_wave, tc1 - t0));

    v4d acc[MAXT];
    unsigned offab[MAXT];  // LDS offsets of the A (low 16 bits) and B (high 16 bits) fragments
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        const int tid2 = min(t0 + t, ttot - 1);
        offab[t] = (unsigned)(lk * LS + (tid2 / T) * 16 + lr) | ((unsigned)(KCH * LS + lk * LS + (tid2 % T) * 16 + lr) << 16);
    }

    // staging roles: a thread serves ONE row of the chunk (row = tid / TPR) and up to NL double2 columns of it, so
    // its four Psi coefficients are loaded once; the raw AO loads stay in flight during the MFMA phase and are
    // only combined into Psi when they are written to LDS afterwards.
    constexpr int TPR = 512 / KCH;       // threads per row
    const int prow = tid / TPR, pcol = tid % TPR;
    double2 raw[NL][GGA ? 4 : 2];  // LDA mode: [0] = phi (A operand), [1] = the array Psi is built from
    double cf[GGA ? 4 : 1], wg = 0.0;  // RAW loads here; the products are formed in stage() so that prefetch() never
    bool rowok = false;                // waits on memory (a wait here idles the matrix pipe at every chunk start)
    // the next chunk's loads are issued in KCH/4 slices BETWEEN the MFMA groups of the current chunk: waves issue in
    // order, and a wave that first has to push its whole 16 KB prefetch through the CU's address pipe (128 KB per chunk
    // for the 8 waves) starts its MFMAs thousands of cycles late
    const double *src = ao;
    const double *srcb = aob;
    auto prefetch_meta = [&](int gc) {
        const int g = gc + prow;
        rowok = g < ge;
        const int gg = rowok ? g : gs;
        wg = w[gg];
        cf[0] = vrho[gg];
        if (GGA) {
#pragma unroll
            for (int d = 0; d < 3; d++) cf[d + 1] = vgrad[(size_t)d * ngrid + gg];
        }
        src = ao + (size_t)gg * ld;
        srcb = aob + (size_t)gg * ld;
    };
    auto prefetch_cols = [&](int part) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            if (i % (KCH / 4) != part) continue;
            const int c2 = (pcol + i * TPR) * 2;
            const int cc = c2 < ld ? c2 : 0;
#pragma unroll
            for (int d = 0; d < (GGA ? 4 : 1); d++)
#ifndef ABL_VXC_NO_LOAD
                raw[i][d] = *reinterpret_cast<const double2 *>(src + d * cs + cc);
#else
                raw[i][d] = make_double2(1e-3 * cc, 2e-3 * d);
#endif
            if (!GGA) raw[i][1] = *reinterpret_cast<const double2 *>(srcb + cc);
        }
    };
    auto stage = [&](int buf) {
        const double ww = rowok ? wg : 0.0;
        cf[0] *= ww;
        if (GGA) {
#pragma unroll
            for (int d = 1; d < 4; d++) cf[d] *= 2.0 * ww;
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int c2 = (pcol + i * TPR) * 2;
            if (c2 < ld) {
                double2 ph = raw[i][0];
                const double2 pb = GGA ? ph : raw[i][1];
                double2 ps = make_double2(cf[0] * pb.x, cf[0] * pb.y);
                if (GGA) {
#pragma unroll
                    for (int d = 1; d < 4; d++) { ps.x += cf[d] * raw[i][d].x; ps.y += cf[d] * raw[i][d].y; }
                }
                if (!rowok) ph = make_double2(0.0, 0.0);
                *reinterpret_cast<double2 *>(lds + buf * BUF + prow * LS + c2) = ph;
                *reinterpret_cast<double2 *>(lds + buf * BUF + KCH * LS + prow * LS + c2) = ps;
            }
        }
    };

    prefetch_meta(gs);
#pragma unroll
    for (int part = 0; part < KCH / 4; part++) prefetch_cols(part);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int gc = gs; gc < ge; gc += KCH) {
        const bool more = gc + KCH < ge;
        if (more) prefetch_meta(gc + KCH);
        const double *base = lds + buf * BUF;
#pragma unroll
        for (int kk = 0; kk < KCH / 4; kk++) {
            if (more) prefetch_cols(kk);
            const int ko = kk * 4 * LS;
#pragma unroll
            for (int t = 0; t < MAXT; t++) {  // straight-line: tiles past nt are clamped duplicates, discarded later
                const double a = base[ko + (offab[t] & 0xffffu)];
                const double b = base[ko + (offab[t] >> 16)];
#ifndef ABL_VXC_NO_MFMA
                acc[t] = mfma_f64(a, b, acc[t]);
#else
                acc[t][0] += a * b;
#endif
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (t < nt) {
            const int tl = t0 + t;
            const int ia = (tl / T) * 16 + lk, ib = (tl % T) * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; r++)
#ifdef ABL_VXC_NO_ATOMIC
                if (acc[t][r] == 12345.678)
#endif
                    atomicAdd(&vmat[(size_t)(ia + 4 * r) * ld + ib], acc[t][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vxc, wave-specialised variant (the default).  Measured on MI355X: a SIMD reaches the fp64 MFMA peak (one
// 16x16x4 per 64.6 cycles, 78 TF chip-wide) only while TWO of its waves are issuing MFMAs; a single issuing wave
// gets one per 140 cycles (36 TF).  In vxc_kernel every wave alternates MFMA work with loads, the Psi combination
// and LDS writes, so for part of every chunk fewer than two waves per SIMD feed the matrix pipe.  Here a block is
// 16 waves: waves 0-7 (two per SIMD) are CONSUMERS that do nothing but fragment reads + MFMAs; waves 8-15 (two per
// SIMD) are PRODUCERS that fetch the next chunk's four AO components (buffer loads: no VALU address arithmetic), form
// Psi and write the (Phi, Psi) chunk to the other LDS buffer.  An fp64 MFMA occupies the SIMD's vector ALU, so the
// producers' VALU work cannot overlap the MFMAs: it runs in a window between two s_barriers per chunk during which the
// consumers wait; the loads fly during the MFMA phase (see the comments in the kernel and DESIGN.md, section 3).  Tile
// ownership, split-K over slabs, the XCD-aware block decode and the atomic epilogue are those of vxc_kernel.
// ---------------------------------------------------------------------------------------------
#ifndef VWS_PROD_THREADS
#define VWS_PROD_THREADS 512
#endif
constexpr int VWS_PROD = VWS_PROD_THREADS;    // producer threads (8 waves: two per SIMD)
constexpr int VWS_NT = 512 + VWS_PROD;        // threads per block
constexpr int VWS2_PROD = 256, VWS2_NT = 512 + VWS2_PROD;  // vxc_ws2_kernel: 4 producer waves

#ifdef VXC_TRACE  // per-chunk timeline of vxc_ws_kernel for tools/ubench/vxc_trace.hip (100 MHz ticks)
constexpr int VXC_TRACE_MAXC = 192;
__device__ long long g_vxc_trace[256 * 2 * (VXC_TRACE_MAXC + 2)];
#define VXC_TRACE_POINT(role, slot) \
    if (lane == 0 && blockIdx.x < 256 && (slot) < VXC_TRACE_MAXC + 2) g_vxc_trace[(blockIdx.x * 2 + (role)) * (VXC_TRACE_MAXC + 2) + (slot)] = wall_clock64()
#else
#define VXC_TRACE_POINT(role, slot)
#endif

// LDS layout of a (Phi, Psi) chunk in vxc_ws_kernel:  element (buffer b, component X, point k, column j) sits at
//     b * VWS_BUF + X * VWS_XS + (k >> 2) * VWS_GS + (k & 3) * ld + j        (doubles)
// with a FIXED stride VWS_GS between the 4-point k-groups.  A fragment read of k-group kk is then  ds_read_b64 v, addr
// offset:kk*VWS_GS*8  with one per-tile address register that does not change within a chunk: no VALU instruction at
// all between the MFMAs.  (With the natural stride 4 * ld, a run-time value, every read needs a v_add first; those two
// VALU instructions per MFMA cost 16 % of the MFMA rate -- tools/ubench/barrier_cost.hip: 60.6 vs 72.2 TF.)
constexpr int VWS_LSMAX = 256;                    // ld <= 208 reaches this kernel (larger bases: vxc_ws2_kernel)
constexpr int VWS_GS = 4 * VWS_LSMAX;             // doubles between k-groups
constexpr int VWS_XS = (16 / 4) * VWS_GS;         // doubles between Phi and Psi (16-point chunks)
constexpr int VWS_BUF = 2 * VWS_XS;               // doubles per buffer: 64 KB; two buffers = 128 KB of the 160 KB
typedef const __attribute__((address_space(3))) double lds_cdouble_t;
// vxc_ws2_kernel (rectangles of at most 8 x 11 tiles): Phi rows <= 8*16 (+16 pad), Psi rows <= 11*16 (+16 pad)
constexpr int WS2_GSA = 4 * 144, WS2_GSB = 4 * 208;       // k-group strides (doubles)
constexpr int WS2_XS = 4 * WS2_GSA;                       // Phi part -> Psi part
constexpr int WS2_BUF = WS2_XS + 4 * WS2_GSB;             // doubles per buffer (44 KB)

// One chunk of a consumer wave: (KCH / 4) k-steps x MAXT tiles of fragment reads + MFMAs, SOFTWARE-PIPELINED by hand.  Left
// to itself the compiler emits  ds_read a; ds_read b; s_waitcnt lgkmcnt(0); v_mfma  per tile (it minimises fragment
// registers), which exposes the LDS latency in front of every MFMA; here the fragments of step s + D are requested
// before the MFMA of step s and sched_barriers pin that order.  pa / pb: LDS byte addresses of the tile's A / B fragment
// in k-group 0 of the current buffer.
#ifndef WS_D
#define WS_D 2
#endif
template <int MAXT, int KCH, int D = WS_D, int GSA = VWS_GS, int GSB = VWS_GS>
__device__ __forceinline__ void ws_chunk(const unsigned (&pa)[MAXT], const unsigned (&pb)[MAXT], v4d (&acc)[MAXT]) {
    constexpr int NS = (KCH / 4) * MAXT;
    double fa[D + 1], fb[D + 1];
#pragma unroll
    for (int s = 0; s < D && s < NS; s++) {
        fa[s % (D + 1)] = *(lds_cdouble_t *)(pa[s % MAXT] + (s / MAXT) * GSA * 8);
        fb[s % (D + 1)] = *(lds_cdouble_t *)(pb[s % MAXT] + (s / MAXT) * GSB * 8);
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {  // tiles past the wave's count are clamped duplicates, discarded later
        if (s + D < NS) {
            const int s2 = s + D;
            fa[s2 % (D + 1)] = *(lds_cdouble_t *)(pa[s2 % MAXT] + (s2 / MAXT) * GSA * 8);
            fb[s2 % (D + 1)] = *(lds_cdouble_t *)(pb[s2 % MAXT] + (s2 / MAXT) * GSB * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[s % MAXT] = mfma_f64(fa[s % (D + 1)], fb[s % (D + 1)], acc[s % MAXT]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
#ifdef ABL_VWS_NO_LDSREAD
template <int MAXT, int KCH>
__device__ __forceinline__ void ws_chunk_nolds(double a, double b, v4d (&acc)[MAXT]) {
#pragma unroll
    for (int s = 0; s < (KCH / 4) * MAXT; s++) {
        acc[s % MAXT] = mfma_f64(a, b, acc[s % MAXT]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
#endif

template <int MAXT, int NLP, int KCH, bool GGA>
__global__ __launch_bounds__(VWS_NT, VWS_NT / 256) void vxc_ws_kernel(double *__restrict__ vmat, const double *__restrict__ ao,
                                                          int ngrid, int ld, const double *__restrict__ w,
                                                          const double *__restrict__ vrho,
                                                          const double *__restrict__ vgrad, int slab, int nsplit,
                                                          int tiles_per_split, const double *__restrict__ aob, int sym) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    static_assert(KCH == 16, "the fixed-stride chunk layout is laid out for 16-point chunks");
    const int LS = ld;
    constexpr int BUF = VWS_BUF;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t cs = (size_t)ngrid * ld;
    const int id = blockIdx.x;
    const int grp = id / (8 * nsplit), rem = id - grp * 8 * nsplit;
    const int split = rem / 8, sl = grp * 8 + (rem & 7);
    const int gs = sl * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;

    if (wave >= VXC_WAVES) {
        // ------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);  // the fp64 combine shares the DP pipe with the consumers' MFMAs: win arbitration
        // Measured (tools/ubench/vxc_trace.hip): an fp64 MFMA occupies the SIMD's vector ALU for its 64 cycles, and against two
        // waves issuing MFMAs back to back every VALU instruction of a third wave waits for a whole MFMA -- a producer that
        // needs ~250 VALU instructions per chunk (address arithmetic, selects, the combine) then takes twice the MFMA time.
        // The producer loop is therefore written to need almost no VALU work outside the combine:
        //   * global loads in  SGPR base + one loop-invariant VGPR offset + immediate  form (the chunk / component base
        //     advances on the scalar unit),
        //   * LDS writes as one address register + immediates,
        //   * the tail chunk (rows past the slab end) on a separate, wave-uniform path.
        constexpr int TPR = VWS_PROD / KCH;  // threads per chunk row
        const int pt = tid - 512;
        const int prow = pt / TPR, pcol = pt % TPR;
        const unsigned voff0 = 8u * (unsigned)(prow * ld + pcol * 2);  // bytes from the chunk's first row
        unsigned wlds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds +
                        8u * (unsigned)((prow >> 2) * VWS_GS + (prow & 3) * LS + pcol * 2);  // Phi slot in buffer 0
        typedef double vd2 __attribute__((ext_vector_type(2)));
        //     Buffer loads give exactly that addressing, and their range check (num_records = bytes left in the slab) returns
        //     zeros for rows past the slab end: Phi = Psi = w = 0 there, no per-lane row guard anywhere.
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        constexpr int BUF_FLAGS = 0x00020000;  // raw buffer, 32-bit data format (gfx9 dword 3)
        v4u raw[NLP][GGA ? 4 : 2];
        double cf[GGA ? 4 : 1], wg = 0.0;
        auto as_d = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
        auto prefetch = [&](int c) {
            const int g0 = gs + c * KCH;                 // uniform: everything below but voff0 / prow lives in SGPRs
            const int rows = ge - g0;                    // rows left in the slab (> 0)
            auto rsrc = [&](const double *base, size_t bytes) {
                return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, BUF_FLAGS);
            };
            const v2u xw = __builtin_amdgcn_raw_buffer_load_b64(rsrc(w + g0, (size_t)rows * 8), prow * 8, 0, 0);
            wg = as_d(xw[0], xw[1]);
            const v2u xr = __builtin_amdgcn_raw_buffer_load_b64(rsrc(vrho + g0, (size_t)rows * 8), prow * 8, 0, 0);
            cf[0] = as_d(xr[0], xr[1]);
            if (GGA) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const v2u xg = __builtin_amdgcn_raw_buffer_load_b64(rsrc(vgrad + (size_t)d * ngrid + g0, (size_t)rows * 8), prow * 8, 0, 0);
                    cf[d + 1] = as_d(xg[0], xg[1]);
                }
            }
            const size_t nb = (size_t)rows * ld * 8;     // < 2^31: a slab is a few MB
#pragma unroll
            for (int d = 0; d < (GGA ? 4 : 1); d++) {
                const auto r = rsrc(ao + d * cs + (size_t)g0 * ld, nb);
#pragma unroll
                for (int i = 0; i < NLP; i++)
                    if ((pcol + i * TPR) * 2 < ld) raw[i][d] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
            }
            if (!GGA && !sym) {  // (sym: the second operand is the first)
                const auto r = rsrc(aob + (size_t)g0 * ld, nb);
#pragma unroll
                for (int i = 0; i < NLP; i++)
                    if ((pcol + i * TPR) * 2 < ld) raw[i][1] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
            }
        };
        auto stage = [&]() {  // Psi from the raw registers; (Phi, Psi) -> the buffer wlds points into
            cf[0] *= wg;
            if (GGA) {
#pragma unroll
                for (int d = 1; d < 4; d++) cf[d] *= 2.0 * wg;
            }
#pragma unroll
            for (int i = 0; i < NLP; i++) {
                if ((pcol + i * TPR) * 2 < ld) {
                    const v4u pb = (GGA || sym) ? raw[i][0] : raw[i][1];
                    vd2 ps = {cf[0] * as_d(pb[0], pb[1]), cf[0] * as_d(pb[2], pb[3])};
                    if (GGA) {
#pragma unroll
                        for (int d = 1; d < 4; d++) {
                            ps.x += cf[d] * as_d(raw[i][d][0], raw[i][d][1]);
                            ps.y += cf[d] * as_d(raw[i][d][2], raw[i][d][3]);
                        }
                    }
                    *(__attribute__((address_space(3))) v4u *)(wlds + i * TPR * 16) = raw[i][0];
                    *(__attribute__((address_space(3))) vd2 *)(wlds + i * TPR * 16 + VWS_XS * 8) = ps;
                }
            }
        };
        prefetch(0);
        stage();
        if (nchunk > 1) prefetch(1);
        __syncthreads();
        if (wave == VXC_WAVES) VXC_TRACE_POINT(1, 0);
        for (int c = 0; c < nchunk; c++) {
            wlds += (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);  // buffer (c + 1) & 1
#ifndef VWS_ONE_BARRIER
            __syncthreads();  // the consumers have finished chunk c - 1 and wait: the vector ALUs are free for the combine
#endif
#ifndef ABL_VWS_NO_PROD
            if (c + 1 < nchunk) stage();              // chunk c + 1: its loads were issued a whole period ago
#endif
#ifndef VWS_ONE_BARRIER
            if (wave == VXC_WAVES) VXC_TRACE_POINT(1, c + 1);  // combine done
            __syncthreads();  // the consumers start the MFMAs of chunk c
#endif
#ifndef ABL_VWS_NO_PROD
            if (c + 2 < nchunk) prefetch(c + 2);      // VALU-free issue, in flight during the MFMAs
#endif
#ifdef VWS_ONE_BARRIER
            if (wave == VXC_WAVES) VXC_TRACE_POINT(1, c + 1);
            __syncthreads();
#endif
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int lr = lane & 15, lk = lane >> 4;
    // sym: A^T diag(w v) A with ONE operand (LDA Vxc, the tau terms of a meta-GGA) is symmetric -- only the tiles (i <= j) are
    // computed, off-diagonal ones doubled so that the final (M + M^T) / 2 restores both halves
    const int T = ld >> 4, ttot = sym ? T * (T + 1) / 2 : T * T;
    auto tile_ij = [&](int u, int &ti, int &tj) {
        if (!sym) { ti = u / T; tj = u - ti * T; return; }
        int i = 0, rem = u;
        while (rem >= T - i) { rem -= T - i; i++; }  // row i of the upper triangle holds T - i tiles
        ti = i;
        tj = i + rem;
    };
    const int tc0 = split * tiles_per_split;
    const int tc1 = min(tc0 + tiles_per_split, ttot);
    const int per_wave = (tc1 - tc0 + VXC_WAVES - 1) / VXC_WAVES;
    const int t0 = tc0 + wave * per_wave;
    const int nt = max(0, min(per_wave, tc1 - t0));
    v4d acc[MAXT];
    unsigned pa[MAXT], pb[MAXT];  // LDS byte addresses of the A / B fragments (k-group 0, current buffer)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        int ti, tj;
        tile_ij(min(t0 + t, ttot - 1), ti, tj);
        pa[t] = lds0 + 8u * (unsigned)(lk * LS + ti * 16 + lr);
        pb[t] = lds0 + 8u * (unsigned)(VWS_XS + lk * LS + tj * 16 + lr);
    }
    __syncthreads();
    if (wave == 0) VXC_TRACE_POINT(0, 0);
    for (int c = 0; c < nchunk; c++) {
#ifndef VWS_ONE_BARRIER
        __syncthreads();  // chunk c - 1 done: the producers' combine window opens ...
        __syncthreads();  // ... and closes
#endif
#ifdef ABL_VWS_NO_LDSREAD
        ws_chunk_nolds<MAXT, KCH>(1e-3 * lane, 2e-3 * lr, acc);
#elif !defined(ABL_VWS_NO_MFMA)
        ws_chunk<MAXT, KCH>(pa, pb, acc);
#endif
        const unsigned delta = (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);  // on to the other buffer
#pragma unroll
        for (int t = 0; t < MAXT; t++) { pa[t] += delta; pb[t] += delta; }
        if (wave == 0) VXC_TRACE_POINT(0, c + 1);  // MFMAs of this chunk issued
#ifdef VWS_ONE_BARRIER
        __syncthreads();
#endif
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (t < nt) {
            int ti, tj;
            tile_ij(t0 + t, ti, tj);
            const int ia = ti * 16 + lk, ib = tj * 16 + lr;
            const double sc = (sym && ti != tj) ? 2.0 : 1.0;
#pragma unroll
#ifndef ABL_VWS_NO_EPI
            for (int r = 0; r < 4; r++) atomicAdd(&vmat[(size_t)(ia + 4 * r) * ld + ib], sc * acc[t][r]);
#else
            for (int r = 0; r < 4; r++) if (acc[t][r] == 1.2345) vmat[0] = 1.0;
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vxc for larger bases: the wave-specialised kernel with RECTANGULAR output ownership.  vxc_ws_kernel splits the
// T x T output tiles linearly over `nsplit` blocks that each stage ALL columns of a slab; beyond two blocks per
// slab that re-reads the slab nsplit times (nsplit = 18 at nao = 624).  Here block (slab, i, j) owns the tile
// rectangle  rows [i T / NR, (i+1) T / NR)  x  cols [j T / NC, (j+1) T / NC)  (<= 8 x 11 tiles) and its producers
// stage only the Phi columns of those rows (A operand) and the four AO components of those columns (-> Psi, B
// operand): per-block loads are what vxc_ws_kernel loads at nao = 208, and a slab is re-read NC + 4 NR times in
// total instead of 5 nsplit.  Chunks are always 16 points (39 KB per LDS buffer).
// ---------------------------------------------------------------------------------------------
template <int MAXT, int NLA, int NLB, bool GGA>
__global__ __launch_bounds__(VWS2_NT, 3) void vxc_ws2_kernel(double *__restrict__ vmat, const double *__restrict__ ao,
                                                           int ngrid, int ld, const double *__restrict__ w,
                                                           const double *__restrict__ vrho,
                                                           const double *__restrict__ vgrad, int slab, int NR, int NC,
                                                           int LSA, int LSB, const double *__restrict__ aob) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16;
    // chunk layout as in vxc_ws_kernel: fixed strides between the 4-point k-groups (Phi part: WS2_GSA, Psi part: WS2_GSB), so that
    // the consumers' fragment reads are  address register + immediate;  rows inside a k-group at the run-time strides LSA / LSB
    constexpr int BUF = WS2_BUF;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t cs = (size_t)ngrid * ld;
    const int T = ld >> 4, nsplit = NR * NC;
    const int id = blockIdx.x;
    const int grp = id / (8 * nsplit), rem = id - grp * 8 * nsplit;
    const int split = rem / 8, sl = grp * 8 + (rem & 7);
    const int gs = sl * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;
    const int bi = split / NC, bj = split % NC;
    const int r0 = bi * T / NR, nr = (bi + 1) * T / NR - r0;   // tile rows of this block
    const int c0 = bj * T / NC, nc = (bj + 1) * T / NC - c0;   // tile columns
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;

    if (wave >= VXC_WAVES) {
        // ------------------------------------------------------------------ producers (see vxc_ws_kernel: buffer loads with
        // SGPR base + loop-invariant VGPR offset + immediate, range-checked against the slab end; the combine in its own window)
        __builtin_amdgcn_s_setprio(3);
        constexpr int TPR = VWS2_PROD / KCH;  // 16 threads per chunk row
        const int pt = tid - 512;
        const int prow = pt / TPR, pcol = pt % TPR;
        const int wa = nr * 16, wb = nc * 16;  // staged widths (doubles)
        const unsigned voff0 = 8u * (unsigned)(prow * ld + pcol * 2);
        unsigned wla = lds0 + 8u * (unsigned)((prow >> 2) * WS2_GSA + (prow & 3) * LSA + pcol * 2);
        unsigned wlb = lds0 + 8u * (unsigned)(WS2_XS + (prow >> 2) * WS2_GSB + (prow & 3) * LSB + pcol * 2);
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        typedef double vd2 __attribute__((ext_vector_type(2)));
        constexpr int BUF_FLAGS = 0x00020000;
        v4u ra[NLA], rb[NLB][GGA ? 4 : 1];
        double cf[GGA ? 4 : 1], wg = 0.0;
        auto as_d = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
        auto prefetch = [&](int c) {
            const int g0 = gs + c * KCH;
            const int rows = ge - g0;
            auto rsrc = [&](const double *base, size_t bytes) {
                return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, BUF_FLAGS);
            };
            const v2u xw = __builtin_amdgcn_raw_buffer_load_b64(rsrc(w + g0, (size_t)rows * 8), prow * 8, 0, 0);
            wg = as_d(xw[0], xw[1]);
            const v2u xr = __builtin_amdgcn_raw_buffer_load_b64(rsrc(vrho + g0, (size_t)rows * 8), prow * 8, 0, 0);
            cf[0] = as_d(xr[0], xr[1]);
            if (GGA) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const v2u xg = __builtin_amdgcn_raw_buffer_load_b64(rsrc(vgrad + (size_t)d * ngrid + g0, (size_t)rows * 8), prow * 8, 0, 0);
                    cf[d + 1] = as_d(xg[0], xg[1]);
                }
            }
            // bytes from the rectangle's first column of row g0 to the end of the slab (the loads of a row stop at its staged width)
            const size_t nba = (size_t)rows * ld * 8 - (size_t)r0 * 128, nbb = (size_t)rows * ld * 8 - (size_t)c0 * 128;
            {
                const auto r = rsrc(ao + (size_t)g0 * ld + r0 * 16, nba);
#pragma unroll
                for (int i = 0; i < NLA; i++)
                    if ((pcol + i * TPR) * 2 < wa) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
            }
#pragma unroll
            for (int d = 0; d < (GGA ? 4 : 1); d++) {
                const auto r = rsrc((GGA ? ao : aob) + d * cs + (size_t)g0 * ld + c0 * 16, nbb);
#pragma unroll
                for (int i = 0; i < NLB; i++)
                    if ((pcol + i * TPR) * 2 < wb) rb[i][d] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
            }
        };
        auto stage = [&]() {
            cf[0] *= wg;
            if (GGA) {
#pragma unroll
                for (int d = 1; d < 4; d++) cf[d] *= 2.0 * wg;
            }
#pragma unroll
            for (int i = 0; i < NLA; i++)
                if ((pcol + i * TPR) * 2 < wa) *(__attribute__((address_space(3))) v4u *)(wla + i * TPR * 16) = ra[i];
#pragma unroll
            for (int i = 0; i < NLB; i++) {
                if ((pcol + i * TPR) * 2 < wb) {
                    vd2 ps = {cf[0] * as_d(rb[i][0][0], rb[i][0][1]), cf[0] * as_d(rb[i][0][2], rb[i][0][3])};
                    if (GGA) {
#pragma unroll
                        for (int d = 1; d < 4; d++) {
                            ps.x += cf[d] * as_d(rb[i][d][0], rb[i][d][1]);
                            ps.y += cf[d] * as_d(rb[i][d][2], rb[i][d][3]);
                        }
                    }
                    *(__attribute__((address_space(3))) vd2 *)(wlb + i * TPR * 16) = ps;
                }
            }
        };
        prefetch(0);
        stage();
        if (nchunk > 1) prefetch(1);
        __syncthreads();
        for (int c = 0; c < nchunk; c++) {
            const unsigned delta = (c & 1) ? (unsigned)(-BUF * 8) : (unsigned)(BUF * 8);  // buffer (c + 1) & 1
            wla += delta;
            wlb += delta;
            __syncthreads();  // the consumers have finished chunk c - 1 and wait: the vector ALUs are free for the combine
            if (c + 1 < nchunk) stage();
            __syncthreads();  // the consumers start the MFMAs of chunk c
            if (c + 2 < nchunk) prefetch(c + 2);
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int lr = lane & 15, lk = lane >> 4;
    const int ttot = nr * nc;
    const int per_wave = (ttot + VXC_WAVES - 1) / VXC_WAVES;
    const int t0 = wave * per_wave;
    const int nt = max(0, min(per_wave, ttot - t0));
    v4d acc[MAXT];
    unsigned pa[MAXT], pb[MAXT];  // LDS byte addresses of the A / B fragments (k-group 0, current buffer)
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        const int tl = min(t0 + t, ttot - 1);
        pa[t] = lds0 + 8u * (unsigned)(lk * LSA + (tl / nc) * 16 + lr);
        pb[t] = lds0 + 8u * (unsigned)(WS2_XS + lk * LSB + (tl % nc) * 16 + lr);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
        __syncthreads();  // chunk c - 1 done: the producers' combine window opens ...
        __syncthreads();  // ... and closes
        ws_chunk<MAXT, KCH, 4, WS2_GSA, WS2_GSB>(pa, pb, acc);
        const unsigned delta = (c & 1) ? (unsigned)(-BUF * 8) : (unsigned)(BUF * 8);
#pragma unroll
        for (int t = 0; t < MAXT; t++) { pa[t] += delta; pb[t] += delta; }
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (t < nt) {
            const int tl = t0 + t;
            const int ia = (r0 + tl / nc) * 16 + lk, ib = (c0 + tl % nc) * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; r++) atomicAdd(&vmat[(size_t)(ia + 4 * r) * ld + ib], acc[t][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vxc, ONE block per slab (bases with 10 <= T <= 13 tile rows, i.e. 145 <= nao <= 208: the 20-atom cc-pVDZ molecules).
// vxc_ws_kernel needs two blocks per slab there (T^2 = 169 tiles > 8 waves x 11), so every chunk travels L2 -> CU twice
// and the chunk period is set by the producers' loads (2.7 us), not by the MFMAs (2.5 us).  Here the block owns the
// UPPER-TRIANGULAR tiles only (T (T + 1) / 2 = 91 <= 8 x 12) and accumulates the symmetrised matrix directly:
//     acc_ij = Phi_i^T Psi_j + Psi_i^T Phi_j = M_ij + (M_ji)^T = 2 V_ij         (two MFMAs per tile and k-group; GGA)
//     acc_ij = Phi_i^T Psi_j                 = M_ij = V_ij                      (one operand, no gradient term: M symmetric)
// -- 182 instead of 169 MFMAs per k-group (+8 %), but half the L2 -> CU traffic, half the producers (4 waves: one per
// SIMD, 12 waves per block => 168 VGPRs per wave for the 12 accumulator tiles) and one combine window per 2 x the MFMA
// work.  Chunk layout, buffer loads, the two-barrier combine window and the hand-pipelined fragment reads are those of
// vxc_ws_kernel; the fragment of tile row i is  Phi: pi[t] + kk GS 8,  Psi: pi[t] + (XS + kk GS) 8  (immediates).
// ---------------------------------------------------------------------------------------------
constexpr int VWU_PROD = 256, VWU_NT = 512 + VWU_PROD;

#ifdef VWU_TRACE  // per-chunk timeline of one block (100 MHz ticks): role 0 = consumer wave 0, role 1 = producer wave 8
constexpr int VWU_TRACE_N = 4 * 128;
__device__ long long g_vwu_trace[2 * VWU_TRACE_N];
#define VWU_STAMP(role, slot) \
    if (blockIdx.x == VWU_TRACE && (wave == 0 || wave == 8) && lane == 0 && (slot) < VWU_TRACE_N) g_vwu_trace[(role) * VWU_TRACE_N + (slot)] = wall_clock64()
#else
#define VWU_STAMP(role, slot)
#endif

template <int MAXT, int NTL, bool TWO, int D = 2, int KG0 = 0, int NKG = 4>
__device__ __forceinline__ void wsu_chunk(const unsigned (&pi)[MAXT], const unsigned (&pj)[MAXT], v4d (&acc)[MAXT]) {
    // NTL <= MAXT: tiles actually looped over (waves that own one tile fewer skip the dummy MFMAs);
    // k-groups KG0 .. KG0 + NKG - 1 of the 16-point chunk (the fused kernel runs a chunk in two halves)
    constexpr int H = TWO ? 2 : 1, NS = NKG * NTL * H;
    double fa[D + 1], fb[D + 1];
    auto rd = [&](int s) {
        const int kk = KG0 + s / (NTL * H), t = (s % (NTL * H)) / H, h = s % H;
        // h = 0: A = Phi_i, B = Psi_j;   h = 1: A = Psi_i, B = Phi_j
        fa[s % (D + 1)] = *(lds_cdouble_t *)(pi[t] + (kk * VWS_GS + (h ? VWS_XS : 0)) * 8);
        fb[s % (D + 1)] = *(lds_cdouble_t *)(pj[t] + (kk * VWS_GS + (h ? 0 : VWS_XS)) * 8);
    };
#pragma unroll
    for (int s = 0; s < D && s < NS; s++) rd(s);
#pragma unroll
    for (int s = 0; s < NS; s++) {  // tiles past the wave's count are clamped duplicates, discarded later
        if (s + D < NS) rd(s + D);
        __builtin_amdgcn_sched_barrier(0);
        const int t = (s % (NTL * H)) / H;
        acc[t] = mfma_f64(fa[s % (D + 1)], fb[s % (D + 1)], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the 4 producer waves of vxc_wsu_kernel / vxc_wsb_kernel (threads 512 .. 767): chunk c + 2 travels HBM -> registers while the
// consumers run the MFMAs of chunk c; chunk c + 1 is combined into (Phi, Psi) and written to LDS in the window between chunks
template <int NLP, bool GGA>
DQC_DEV void vwu_producer(double *lds, const double *__restrict__ ao, int ngrid, int ld, const double *__restrict__ w,
                          const double *__restrict__ vrho, const double *__restrict__ vgrad, int gs, int ge, int nchunk) {
    constexpr int KCH = 16;
    const int LS = ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    (void)wave; (void)lane;
    const size_t cs = (size_t)ngrid * ld;
    // ------------------------------------------------------------------ producers (see vxc_ws_kernel)
    __builtin_amdgcn_s_setprio(3);
    constexpr int TPR = VWU_PROD / KCH;  // 16 threads per chunk row
    const int pt = tid - 512;
    const int prow = pt / TPR, pcol = pt % TPR;
    const unsigned voff0 = 8u * (unsigned)(prow * ld + pcol * 2);
    unsigned wlds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds +
                    8u * (unsigned)((prow >> 2) * VWS_GS + (prow & 3) * LS + pcol * 2);
    typedef double vd2 __attribute__((ext_vector_type(2)));
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    constexpr int BUF_FLAGS = 0x00020000;
    v4u raw[NLP][GGA ? 4 : 1];
    double cf[GGA ? 4 : 1], wg = 0.0;
    auto as_d = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
    auto prefetch = [&](int c) {
        const int g0 = gs + c * KCH;
        const int rows = ge - g0;
        auto rsrc = [&](const double *base, size_t bytes) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, BUF_FLAGS);
        };
        const v2u xw = __builtin_amdgcn_raw_buffer_load_b64(rsrc(w + g0, (size_t)rows * 8), prow * 8, 0, 0);
        wg = as_d(xw[0], xw[1]);
        const v2u xr = __builtin_amdgcn_raw_buffer_load_b64(rsrc(vrho + g0, (size_t)rows * 8), prow * 8, 0, 0);
        cf[0] = as_d(xr[0], xr[1]);
        if (GGA) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const v2u xg = __builtin_amdgcn_raw_buffer_load_b64(rsrc(vgrad + (size_t)d * ngrid + g0, (size_t)rows * 8), prow * 8, 0, 0);
                cf[d + 1] = as_d(xg[0], xg[1]);
            }
        }
        const size_t nb = (size_t)rows * ld * 8;
#pragma unroll
        for (int d = 0; d < (GGA ? 4 : 1); d++) {
            const auto r = rsrc(ao + d * cs + (size_t)g0 * ld, nb);
#pragma unroll
            for (int i = 0; i < NLP; i++)
                if ((pcol + i * TPR) * 2 < ld) raw[i][d] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
        }
    };
    auto stage = [&]() {
        // GGA: acc = 2 V, so Psi = w (vrho Phi + sum_d 2 vgrad_d dPhi_d) as in vxc_ws_kernel and the epilogue halves
        cf[0] *= wg;
        if (GGA) {
#pragma unroll
            for (int d = 1; d < 4; d++) cf[d] *= 2.0 * wg;
        }
#pragma unroll
        for (int i = 0; i < NLP; i++) {
            if ((pcol + i * TPR) * 2 < ld) {
                vd2 ps = {cf[0] * as_d(raw[i][0][0], raw[i][0][1]), cf[0] * as_d(raw[i][0][2], raw[i][0][3])};
                if (GGA) {
#pragma unroll
                    for (int d = 1; d < 4; d++) {
                        ps.x += cf[d] * as_d(raw[i][d][0], raw[i][d][1]);
                        ps.y += cf[d] * as_d(raw[i][d][2], raw[i][d][3]);
                    }
                }
                *(__attribute__((address_space(3))) v4u *)(wlds + i * TPR * 16) = raw[i][0];
                *(__attribute__((address_space(3))) vd2 *)(wlds + i * TPR * 16 + VWS_XS * 8) = ps;
            }
        }
    };
    prefetch(0);
    stage();
    if (nchunk > 1) prefetch(1);
    __syncthreads();
#ifdef VWU_TRACE
    if (blockIdx.x == VWU_TRACE && tid == 512) { g_vwu_trace[VWU_TRACE_N - 4] = clock64(); g_vwu_trace[VWU_TRACE_N - 3] = wall_clock64(); }
#endif
    for (int c = 0; c < nchunk; c++) {
        wlds += (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);  // buffer (c + 1) & 1
        __syncthreads();  // the consumers have finished chunk c - 1 and wait: the vector ALUs are free for the combine
#ifdef VWU_TRACE_CHUNKS
        VWU_STAMP(1, 4 * c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VWU_STAMP(1, 4 * c + 1);
#endif
#ifndef VWU_EXP_NOSTAGE
        if (c + 1 < nchunk) stage();
#endif
        __syncthreads();  // the consumers start the MFMAs of chunk c
#ifndef VWU_EXP_NOLOAD
        if (c + 2 < nchunk) prefetch(c + 2);
#endif
    }
#ifdef VWU_TRACE
    if (blockIdx.x == VWU_TRACE && tid == 512) { g_vwu_trace[VWU_TRACE_N - 2] = clock64(); g_vwu_trace[VWU_TRACE_N - 1] = wall_clock64(); }
#endif
}

template <int MAXT, int NLP, bool GGA>
__global__ __launch_bounds__(VWU_NT, 3) void vxc_wsu_kernel(double *__restrict__ vmat, const double *__restrict__ ao, int ngrid,
                                                           int ld, const double *__restrict__ w, const double *__restrict__ vrho,
                                                           const double *__restrict__ vgrad, int slab) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16;
    const int LS = ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t cs = (size_t)ngrid * ld;
    const int gs = blockIdx.x * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;

    if (wave >= VXC_WAVES) {
        vwu_producer<NLP, GGA>(lds, ao, ngrid, ld, w, vrho, vgrad, gs, ge, nchunk);
        return;
    }

    // ---------------------------------------------------------------------- consumers: upper-triangular tiles
    const int lr = lane & 15, lk = lane >> 4;
    const int T = ld >> 4, ttot = T * (T + 1) / 2;
    auto tile_ij = [&](int u, int &ti, int &tj) {
        int i = 0, rem = u;
        while (rem >= T - i) { rem -= T - i; i++; }  // row i of the upper triangle holds T - i tiles
        ti = i;
        tj = i + rem;
    };
    // balanced deal: the first ttot % 8 waves own one tile more.  Waves w and w + 4 share a SIMD (a block's waves go to the
    // SIMDs cyclically), so for T = 13 the SIMDs carry 23, 23, 23, 22 tiles and no dummy MFMA is issued
    const int tbase = ttot / VXC_WAVES, trem = ttot % VXC_WAVES;
    const int nt = tbase + (wave < trem ? 1 : 0);
    const int t0 = wave * tbase + min(wave, trem);
    v4d acc[MAXT];
    unsigned pi[MAXT], pj[MAXT];  // LDS byte addresses of the row / column fragments in the Phi part (k-group 0, current buffer)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        int ti, tj;
        tile_ij(min(t0 + min(t, max(nt - 1, 0)), ttot - 1), ti, tj);
        pi[t] = lds0 + 8u * (unsigned)(lk * LS + ti * 16 + lr);
        pj[t] = lds0 + 8u * (unsigned)(lk * LS + tj * 16 + lr);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
#ifdef VWU_TRACE_CONS
        VWU_STAMP(0, 4 * c);
#endif
        __syncthreads();  // chunk c - 1 done: the producers' combine window opens ...
        __syncthreads();  // ... and closes
#ifdef VWU_TRACE_CONS
        VWU_STAMP(0, 4 * c + 1);
#endif
#ifndef VWU_EXP_NOMFMA
        if (nt == MAXT) wsu_chunk<MAXT, MAXT, GGA>(pi, pj, acc);
        else wsu_chunk<MAXT, MAXT - 1, GGA>(pi, pj, acc);
#endif
        const unsigned delta = (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);
#pragma unroll
        for (int t = 0; t < MAXT; t++) { pi[t] += delta; pj[t] += delta; }
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (t < nt) {
            int ti, tj;
            tile_ij(t0 + t, ti, tj);
            const int ia = ti * 16 + lk, ib = tj * 16 + lr;
            // symmetrize_kernel forms (m_ij + m_ji) / 2 over the whole matrix and the lower tiles stay zero:
            //   GGA: acc = 2 V -> off-diagonal tiles store acc (-> acc / 2 = V), diagonal tiles acc / 2 (already symmetric)
            //   one operand: acc = V -> off-diagonal tiles 2 acc, diagonal tiles acc
            const double sc = (GGA ? 1.0 : 2.0) * (ti != tj ? 1.0 : 0.5);
#ifdef VWU_EXP_NOEPI
            if (acc[t][0] != 1.2345e300) continue;
#endif
#pragma unroll
            for (int r = 0; r < 4; r++) atomicAdd(&vmat[(size_t)(ia + 4 * r) * ld + ib], sc * acc[t][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vxc, one block per slab, GGA, T = 11 or 13 tile rows: vxc_wsu_kernel with ONE MFMA on the diagonal tiles.
// vxc_wsu_kernel is 90 % MFMA-busy in cycles, but the chip sits at its power limit there (tools/gpu_vxc_trace.py: the shader
// clock is 1.96 GHz with the MFMAs and the HBM stream both running, 2.35 GHz with the MFMAs alone, 2.41 GHz with the stream
// alone), so what is left is the number of MFMAs.  A diagonal tile needs only M_ii = Phi_i^T Psi_i: symmetrize_kernel forms
// (M_ii + M_ii^T) / 2 = V_ii anyway.  T^2 = 169 MFMAs per k-group instead of T (T + 1) = 182 (-7 %).  Every wave owns NO
// off-diagonal tiles (two MFMAs each) followed by ND diagonal tiles (one each); the deal is fixed at compile time so that the
// SIMDs (waves w and w + 4) carry 42, 42, 42, 43 MFMAs per k-group for T = 13 and no wave more than 12 accumulator tiles.
// Measured (C5 shape, random data): 0.691 ms against 0.725 ms.  Not pursued: sharing fragment reads between the tiles of a
// row -- a build that issues a quarter of the ds_read_b64 (wrong results, timing only) is just 3-5 % faster.
// ---------------------------------------------------------------------------------------------
template <int T>
struct WsdDeal {
    int no[VXC_WAVES], nd[VXC_WAVES], o0[VXC_WAVES], d0[VXC_WAVES];
    constexpr WsdDeal() : no{}, nd{}, o0{}, d0{} {
        const int noff = T * (T - 1) / 2;
        int cost[4] = {0, 0, 0, 0};
        for (int w = 0; w < VXC_WAVES; w++) {
            no[w] = noff / VXC_WAVES + (w < noff % VXC_WAVES ? 1 : 0);
            cost[w & 3] += 2 * no[w];
        }
        for (int d = 0; d < T; d++) {  // greedy: the next diagonal tile goes to the lightest SIMD, there to the wave with fewer tiles
            int q = 0;
            for (int r = 1; r < 4; r++)
                if (cost[r] < cost[q]) q = r;
            const int w = no[q] + nd[q] <= no[q + 4] + nd[q + 4] ? q : q + 4;
            nd[w]++;
            cost[q]++;
        }
        for (int w = 1; w < VXC_WAVES; w++) {
            o0[w] = o0[w - 1] + no[w - 1];
            d0[w] = d0[w - 1] + nd[w - 1];
        }
    }
    constexpr int max_tiles() const {
        int m = 0;
        for (int w = 0; w < VXC_WAVES; w++) m = no[w] + nd[w] > m ? no[w] + nd[w] : m;
        return m;
    }
};

template <int NO, int ND, int D = 2>
DQC_DEV void wsd_chunk(const unsigned (&pi)[NO + ND], const unsigned (&pj)[NO + ND], v4d (&acc)[NO + ND]) {
    // step s of a k-group: s < 2 NO: tile s / 2, h = s % 2 (h = 0: Phi_i^T Psi_j, h = 1: Psi_i^T Phi_j); then the diagonal tiles, h = 0
    constexpr int PER = 2 * NO + ND, NS = 4 * PER;
    double fa[D + 1], fb[D + 1];
    auto rd = [&](int s) {
        const int kk = s / PER, u = s % PER;
        const int t = u < 2 * NO ? u / 2 : NO + (u - 2 * NO), h = u < 2 * NO ? u % 2 : 0;
        fa[s % (D + 1)] = *(lds_cdouble_t *)(pi[t] + (kk * VWS_GS + (h ? VWS_XS : 0)) * 8);
        fb[s % (D + 1)] = *(lds_cdouble_t *)(pj[t] + (kk * VWS_GS + (h ? 0 : VWS_XS)) * 8);
    };
#pragma unroll
    for (int s = 0; s < D && s < NS; s++) rd(s);
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (s + D < NS) rd(s + D);
        __builtin_amdgcn_sched_barrier(0);
        const int u = s % PER;
        const int t = u < 2 * NO ? u / 2 : NO + (u - 2 * NO);
        acc[t] = mfma_f64(fa[s % (D + 1)], fb[s % (D + 1)], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int T, int NO, int ND>
DQC_DEV void wsd_consumer(double *lds, double *__restrict__ vmat, int nchunk, int o0, int d0) {
    constexpr int NT = NO + ND, LS = 16 * T;
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    auto tile_ij = [&](int t, int &ti, int &tj) {
        if (t >= NO) { ti = tj = d0 + (t - NO); return; }
        int i = 0, rem = o0 + t;
        while (rem >= T - 1 - i) { rem -= T - 1 - i; i++; }  // row i of the strict upper triangle holds T - 1 - i tiles
        ti = i;
        tj = i + 1 + rem;
    };
    v4d acc[NT];
    unsigned pi[NT], pj[NT];  // LDS byte addresses of the row / column fragments in the Phi part (k-group 0, current buffer)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        int ti, tj;
        tile_ij(t, ti, tj);
        pi[t] = lds0 + 8u * (unsigned)(lk * LS + ti * 16 + lr);
        pj[t] = lds0 + 8u * (unsigned)(lk * LS + tj * 16 + lr);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
        __syncthreads();  // chunk c - 1 done: the producers' combine window opens ...
        __syncthreads();  // ... and closes
        wsd_chunk<NO, ND>(pi, pj, acc);
        const unsigned delta = (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);
#pragma unroll
        for (int t = 0; t < NT; t++) { pi[t] += delta; pj[t] += delta; }
    }
    // off-diagonal tiles hold 2 V_ij (symmetrize_kernel halves them against the zero lower tiles), diagonal tiles M_ii
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int ti, tj;
        tile_ij(t, ti, tj);
        const int ia = ti * 16 + lk, ib = tj * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; r++) atomicAdd(&vmat[(size_t)(ia + 4 * r) * LS + ib], acc[t][r]);
    }
}

template <int T, int NLP>
__global__ __launch_bounds__(VWU_NT, 3) void vxc_wsd_kernel(double *__restrict__ vmat, const double *__restrict__ ao, int ngrid,
                                                           const double *__restrict__ w, const double *__restrict__ vrho,
                                                           const double *__restrict__ vgrad, int slab) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16, ld = 16 * T;
    constexpr WsdDeal<T> DL{};
    static_assert(DL.max_tiles() <= 12, "more than 12 accumulator tiles per wave");
    const int wave = threadIdx.x >> 6;
    const int gs = blockIdx.x * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;
    if (wave >= VXC_WAVES) {
        vwu_producer<NLP, true>(lds, ao, ngrid, ld, w, vrho, vgrad, gs, ge, nchunk);
        return;
    }
#define DQC_WSD_CASE(W) case W: wsd_consumer<T, DL.no[W], DL.nd[W]>(lds, vmat, nchunk, DL.o0[W], DL.d0[W]); break;
    switch (wave) {  // wave-uniform; equal (NO, ND) pairs share one instantiation
        DQC_WSD_CASE(0) DQC_WSD_CASE(1) DQC_WSD_CASE(2) DQC_WSD_CASE(3)
        DQC_WSD_CASE(4) DQC_WSD_CASE(5) DQC_WSD_CASE(6) DQC_WSD_CASE(7)
    }
#undef DQC_WSD_CASE
}

// ---------------------------------------------------------------------------------------------
// FUSED grid pass (restricted GGA / LDA-type functionals, density matrix in factor form, 145 <= nao <= 208):
//     rho, grad rho  ->  XC potentials  ->  Vxc matrix      from ONE read of the AO matrix.
// Reference: HamiltonCGTO.get_vxc = _dm2densinfo -> xc.get_vxc -> _get_vxc_from_potinfo (hcgto.py:260-269, 371-495).
// The block is vxc_wsu_kernel's: 8 consumer waves own the upper-triangular tiles of V and run a chunk's MFMAs in two
// halves (k-groups 0-1, 2-3); all that is new lives in the 4 producer waves (one per SIMD), which used to idle during the
// MFMA phases.  Per 16-point chunk c (consumers on chunk c, producers one chunk ahead):
//   phase A  consumers: V += (chunk c, k-groups 0, 1)        producers: wait for the loads of chunk c+1, Phi(c+1) -> LDS
//   phase B  consumers: V += (chunk c, k-groups 2, 3)        producers: density GEMMs of chunk c+1 on the matrix pipe
//            D1  A'^T = L^T Phi^T   (K = nao split over the 4 producer waves, partial tiles summed with ds_add_f64)
//            D2  B = A' L^T         (column tiles dealt to the producers, written into the Psi slot of the next buffer)
//   window   consumers wait                                  producers: row dots rho = B.Phi, grad rho = 2 B.dPhi with the
//            gradient components still in their registers, the functional at the row's point (every lane of a 16-lane row
//            group evaluates it: no exchange), Psi(c+1) = w (vrho Phi + 4 vsigma grad rho . dPhi) over B, loads of chunk c+2
// L and L^T fragments come straight from L2 (80 KB each, shared by all blocks); the A' tiles pass between D1 and D2 through
// 2 KB x NRT of LDS in MFMA fragment order (the transposed-GEMM trick of density_lr_kernel: D1's accumulator registers ARE
// D2's A fragments).  Between D1 and D2 only the producers synchronise (an LDS counter; the consumers are mid-burst).
// An fp64 MFMA occupies the vector ALU of its SIMD, so the producers' VALU work is confined to the window.
// ---------------------------------------------------------------------------------------------
constexpr int FG_ABUF = 16 * 64;   // doubles: A' fragments, [4 NRT][64 lanes], NRT <= 4

#ifdef FG_TRACE  // per-phase timeline of block 0 (100 MHz ticks): role 0 = consumer wave 0, role 1 = producer wave 8
constexpr int FG_TRACE_N = 8 * 64;
__device__ long long g_fg_trace[2 * FG_TRACE_N];
#define FG_STAMP(role, slot) \
    if (blockIdx.x == 0 && lane == 0 && (slot) < FG_TRACE_N) g_fg_trace[(role) * FG_TRACE_N + (slot)] = wall_clock64()
#else
#define FG_STAMP(role, slot)
#endif

DQC_DEV void fg_prod_sync(int *cnt, int &epoch) {  // barrier among the 4 producer waves only
    epoch += 4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// XCS: functional set evaluated in the window -- 1: terms 0, 1 are (gga_x_pbe, gga_c_pbe), 2: (lda_x, lda_c_pw), 0: any list (the
// generic evaluator keeps all four functionals' duals live: 121 VGPRs on top of the chunk's gradient registers -> scratch)
template <int MAXT, int NLP, int NRT, int XCS>
__global__ __launch_bounds__(VWU_NT, 3) void fused_grid_kernel(double *__restrict__ vmat, const double *__restrict__ ao, int ngrid,
                                                              int ld, const double *__restrict__ w,
                                                              const double *__restrict__ orb, const double *__restrict__ orbt,
                                                              int slab, XcTerms terms, double *__restrict__ rho_out,
                                                              double *__restrict__ grho_out, double *__restrict__ exc_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16, RP = 16 * NRT;
    const int LS = ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const size_t cs = (size_t)ngrid * ld;
    const int gs = blockIdx.x * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;
    double *abuf = lds + 2 * VWS_BUF;
    int *pcnt = (int *)(abuf + FG_ABUF);
    if (tid == 0) *pcnt = 0;

    if (wave >= VXC_WAVES) {
        // ------------------------------------------------------------------ producers
        constexpr int TPR = VWU_PROD / KCH;  // 16 threads per chunk row
        const int pt = tid - 512, pw = wave - VXC_WAVES;
        const int prow = pt / TPR, pcol = pt % TPR;
        const unsigned voff0 = 8u * (unsigned)(prow * ld + pcol * 2);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
        unsigned wlds = lds0 + 8u * (unsigned)((prow >> 2) * VWS_GS + (prow & 3) * LS + pcol * 2);  // Phi slot, buffer 0
        typedef double vd2 __attribute__((ext_vector_type(2)));
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        constexpr int BUF_FLAGS = 0x00020000;
        v4u raw[NLP][3];   // the three gradient components of the chunk one ahead: held from their load to its Psi combine
        v4u rphi[NLP];     // its Phi: only from the load to the staging (the window re-reads Phi from LDS)
        double wg = 0.0, exc_acc = 0.0;
        int epoch = 0;
        auto as_d = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
        auto prefetch = [&](int c) __attribute__((always_inline)) {
            const int g0 = gs + c * KCH;
            const int rows = ge - g0;
            auto rsrc = [&](const double *base, size_t bytes) {
                return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, BUF_FLAGS);
            };
            const v2u xw = __builtin_amdgcn_raw_buffer_load_b64(rsrc(w + g0, (size_t)rows * 8), prow * 8, 0, 0);
            wg = as_d(xw[0], xw[1]);
            const size_t nb = (size_t)rows * ld * 8;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const auto r = rsrc(ao + d * cs + (size_t)g0 * ld, nb);
#pragma unroll
                for (int i = 0; i < NLP; i++)
                    if ((pcol + i * TPR) * 2 < ld) {
                        const v4u x = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
                        if (d == 0) rphi[i] = x;
                        else raw[i][d - 1] = x;
                    }
            }
        };
        auto stage_phi = [&]() __attribute__((always_inline)) {  // Phi of the chunk in the registers -> Phi slot of the buffer wlds points into
#pragma unroll
            for (int i = 0; i < NLP; i++)
                if ((pcol + i * TPR) * 2 < ld) *(__attribute__((address_space(3))) v4u *)(wlds + i * TPR * 16) = rphi[i];
        };
        auto zero_abuf = [&]() __attribute__((always_inline)) {
            for (int e = pt; e < 4 * NRT * 64; e += VWU_PROD) abuf[e] = 0.0;
        };
        // density GEMMs of the chunk whose Phi sits in buffer `nbuf`
        auto density = [&](int nbuf) __attribute__((always_inline)) {
            const double *phi = lds + nbuf * VWS_BUF;
            {   // D1: a1[t][q] = A'^T[r = 16 t + 4 q + lk][pt = lr], K range of this wave
                const int nk = ld >> 2, per = (nk + 3) >> 2;
                const int s0 = pw * per, s1 = min(s0 + per, nk);
                v4d a1[NRT];
#pragma unroll
                for (int t = 0; t < NRT; t++) a1[t] = v4d{0, 0, 0, 0};
                const double *bp = phi + (lr >> 2) * VWS_GS + (lr & 3) * LS + lk;          // Phi[pt = lr][ao = 4 s + lk]
                const double *ap = orb + (size_t)lk * RP + lr;                             // L[ao = 4 s + lk][r = 16 t + lr]
#pragma unroll 4
                for (int s_ = s0; s_ < s1; s_++) {
                    const double bv = bp[4 * s_];
#pragma unroll
                    for (int t = 0; t < NRT; t++) a1[t] = mfma_f64(ap[(size_t)s_ * 4 * RP + 16 * t], bv, a1[t]);
                }
#pragma unroll
                for (int t = 0; t < NRT; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++) atomicAdd(&abuf[(4 * t + q) * 64 + lane], a1[t][q]);  // ds_add_f64
            }
            fg_prod_sync(pcnt, epoch);
            {   // D2: B[pt][16 j + ..] = sum_r A'[pt][r] L[ao][r]; column tiles j = pw, pw + 4, ...
                double af[4 * NRT];
#pragma unroll
                for (int s_ = 0; s_ < 4 * NRT; s_++) af[s_] = abuf[s_ * 64 + lane];
                const int T = ld >> 4;
                double *bout = lds + nbuf * VWS_BUF + VWS_XS + lk * LS + lr;              // B[pt = lk + 4 q][col]: + q GS + 16 j
                const double *lp = orbt + (size_t)lk * ld + lr;                           // L^T[r = 4 s + lk][ao = 16 j + lr]
                for (int j = pw; j < T; j += 4) {
                    v4d acc = v4d{0, 0, 0, 0};
#pragma unroll
                    for (int s_ = 0; s_ < 4 * NRT; s_++) acc = mfma_f64(af[s_], lp[(size_t)s_ * 4 * ld + 16 * j], acc);
#pragma unroll
                    for (int q = 0; q < 4; q++) bout[q * VWS_GS + 16 * j] = acc[q];
                }
            }
            fg_prod_sync(pcnt, epoch);
        };
        // row dots + functional + Psi for the chunk in the registers (its B tiles in the Psi slot wlds + XS); g0: first point
        auto window = [&](int g0) __attribute__((always_inline)) {
            double s0_ = 0, s1_ = 0, s2_ = 0, s3_ = 0;
#pragma unroll
            for (int i = 0; i < NLP; i++) {
                if ((pcol + i * TPR) * 2 < ld) {
                    const vd2 b = *(__attribute__((address_space(3))) vd2 *)(wlds + i * TPR * 16 + VWS_XS * 8);
                    const vd2 ph = *(__attribute__((address_space(3))) vd2 *)(wlds + i * TPR * 16);
                    s0_ += b.x * ph.x + b.y * ph.y;
                    s1_ += b.x * as_d(raw[i][0][0], raw[i][0][1]) + b.y * as_d(raw[i][0][2], raw[i][0][3]);
                    s2_ += b.x * as_d(raw[i][1][0], raw[i][1][1]) + b.y * as_d(raw[i][1][2], raw[i][1][3]);
                    s3_ += b.x * as_d(raw[i][2][0], raw[i][2][1]) + b.y * as_d(raw[i][2][2], raw[i][2][3]);
                }
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                s0_ += __shfl_xor(s0_, m); s1_ += __shfl_xor(s1_, m); s2_ += __shfl_xor(s2_, m); s3_ += __shfl_xor(s3_, m);
            }
            const double rho = s0_, gx = 2.0 * s1_, gy = 2.0 * s2_, gz = 2.0 * s3_;
            const double sigma = gx * gx + gy * gy + gz * gz;
            double e = 0.0, vr = 0.0, vs = 0.0;
            if (XCS == 0) {
                xc_point(terms, rho, sigma, e, vr, vs);
            } else if (rho > 1e-15) {  // (the density threshold of xc_point)
                const Dual dr = mk(rho, 1.0, 0.0), ds = mk(sigma, 0.0, 1.0);
                const Dual f = XCS == 1 ? terms.c[0] * f_gga_x_pbe(dr, ds) + terms.c[1] * f_gga_c_pbe(dr, ds)
                                        : terms.c[0] * f_lda_x(dr) + terms.c[1] * f_lda_c_pw(dr);
                e = f.v; vr = f.r; vs = f.s;
            }
            if (pcol == 0 && g0 + prow < ge) {
                exc_acc += wg * e;
                if (rho_out) rho_out[g0 + prow] = rho;
                if (grho_out) {
                    grho_out[g0 + prow] = gx;
                    grho_out[(size_t)ngrid + g0 + prow] = gy;
                    grho_out[2 * (size_t)ngrid + g0 + prow] = gz;
                }
            }
            // Psi = w (vrho Phi + sum_d 2 vgrad_d dPhi_d), vgrad = 2 vsigma grad rho  (hcgto.py:466, libxc.py:239)
            const double c0 = wg * vr, c4 = 4.0 * wg * vs;
            const double c1 = c4 * gx, c2 = c4 * gy, c3 = c4 * gz;
#pragma unroll
            for (int i = 0; i < NLP; i++) {
                if ((pcol + i * TPR) * 2 < ld) {
                    const vd2 ph = *(__attribute__((address_space(3))) vd2 *)(wlds + i * TPR * 16);
                    vd2 ps = {c0 * ph.x, c0 * ph.y};
                    ps.x += c1 * as_d(raw[i][0][0], raw[i][0][1]); ps.y += c1 * as_d(raw[i][0][2], raw[i][0][3]);
                    ps.x += c2 * as_d(raw[i][1][0], raw[i][1][1]); ps.y += c2 * as_d(raw[i][1][2], raw[i][1][3]);
                    ps.x += c3 * as_d(raw[i][2][0], raw[i][2][1]); ps.y += c3 * as_d(raw[i][2][2], raw[i][2][3]);
                    *(__attribute__((address_space(3))) vd2 *)(wlds + i * TPR * 16 + VWS_XS * 8) = ps;
                }
            }
        };
        // ---- prologue: chunk 0 entirely (the consumers wait)
        prefetch(0);
        zero_abuf();
        stage_phi();
        __syncthreads();                 // (P0) pcnt = 0, abuf zeroed, Phi(0) staged -- for the producers themselves
        density(0);
        window(gs);
        if (nchunk > 1) prefetch(1);
        __syncthreads();                 // (P1) buffer 0 complete: the consumers start
        for (int c = 0; c < nchunk; c++) {
            const bool more = c + 1 < nchunk;
            wlds += (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);  // slots of buffer (c + 1) & 1
            if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 0);
            // phase A (consumers: first half of chunk c): Phi(c+1) -> LDS as soon as its loads have landed
            if (more) { zero_abuf(); stage_phi(); }
            if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 1);
            __syncthreads();             // (Ba)
            if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 2);
            // phase B (consumers: second half of chunk c): density GEMMs of chunk c+1 on the matrix pipe
            if (more) density((c + 1) & 1);
            if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 3);
            __syncthreads();             // (Bb) the consumers have finished chunk c and wait
            if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 4);
            if (more) {
                window(gs + (c + 1) * KCH);
                if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 5);
                if (c + 2 < nchunk) prefetch(c + 2);
            }
            if (wave == VXC_WAVES) FG_STAMP(1, 8 * c + 6);
            __syncthreads();             // (Bc)
        }
        if (exc_out) {
            double v = exc_acc;
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) v += __shfl_xor(v, m);
            if (lane == 0) atomicAdd(exc_out, v);
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers: upper-triangular tiles (vxc_wsu_kernel)
    const int T = ld >> 4, ttot = T * (T + 1) / 2;
    auto tile_ij = [&](int u, int &ti, int &tj) {
        int i = 0, rem = u;
        while (rem >= T - i) { rem -= T - i; i++; }
        ti = i;
        tj = i + rem;
    };
    const int tbase = ttot / VXC_WAVES, trem = ttot % VXC_WAVES;
    const int nt = tbase + (wave < trem ? 1 : 0);
    const int t0 = wave * tbase + min(wave, trem);
    v4d acc[MAXT];
    unsigned pi[MAXT], pj[MAXT];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        int ti, tj;
        tile_ij(min(t0 + min(t, max(nt - 1, 0)), ttot - 1), ti, tj);
        pi[t] = lds0 + 8u * (unsigned)(lk * LS + ti * 16 + lr);
        pj[t] = lds0 + 8u * (unsigned)(lk * LS + tj * 16 + lr);
    }
    __syncthreads();  // (P0)
    __syncthreads();  // (P1)
    for (int c = 0; c < nchunk; c++) {
        if (wave == 0) FG_STAMP(0, 8 * c + 0);
        if (nt == MAXT) wsu_chunk<MAXT, MAXT, true, 2, 0, 2>(pi, pj, acc);
        else wsu_chunk<MAXT, MAXT - 1, true, 2, 0, 2>(pi, pj, acc);
        if (wave == 0) FG_STAMP(0, 8 * c + 1);
        __syncthreads();  // (Ba)
        if (wave == 0) FG_STAMP(0, 8 * c + 2);
        if (nt == MAXT) wsu_chunk<MAXT, MAXT, true, 2, 2, 2>(pi, pj, acc);
        else wsu_chunk<MAXT, MAXT - 1, true, 2, 2, 2>(pi, pj, acc);
        if (wave == 0) FG_STAMP(0, 8 * c + 3);
        __syncthreads();  // (Bb)
        if (wave == 0) FG_STAMP(0, 8 * c + 4);
        __syncthreads();  // (Bc)
        if (wave == 0) FG_STAMP(0, 8 * c + 6);
        const unsigned delta = (c & 1) ? (unsigned)(-VWS_BUF * 8) : (unsigned)(VWS_BUF * 8);
#pragma unroll
        for (int t = 0; t < MAXT; t++) { pi[t] += delta; pj[t] += delta; }
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (t < nt) {
            int ti, tj;
            tile_ij(t0 + t, ti, tj);
            const int ia = ti * 16 + lk, ib = tj * 16 + lr;
            const double sc = (ti != tj ? 1.0 : 0.5);  // acc = 2 V: see vxc_wsu_kernel
#pragma unroll
            for (int r = 0; r < 4; r++) atomicAdd(&vmat[(size_t)(ia + 4 * r) * ld + ib], sc * acc[t][r]);
        }
    }
}

// V = (M + M^T) / 2 on the zero-padded (ld, ld) matrix
__global__ void symmetrize_kernel(double *m, int ld) {
    const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
    if (i < ld && j < i) {
        const double v = 0.5 * (m[(size_t)i * ld + j] + m[(size_t)j * ld + i]);
        m[(size_t)i * ld + j] = v;
        m[(size_t)j * ld + i] = v;
    }
}

template <int MAXT, int NL, int KCH, bool GGA>
static void launch_vxc_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                            const double *w, const double *vrho, const double *vgrad, int slab, int nsplit, int tps,
                            const double *aob) {
    (void)hipFuncSetAttribute((const void *)vxc_kernel<MAXT, NL, KCH, GGA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)shmem);
    hipLaunchKernelGGL((vxc_kernel<MAXT, NL, KCH, GGA>), grid, dim3(512), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab,
                       nsplit, tps, aob);
}

template <int MAXT, int NLP, int KCH, bool GGA>
static void launch_vxc_ws_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                               const double *w, const double *vrho, const double *vgrad, int slab, int nsplit, int tps,
                               const double *aob, int sym) {
    auto kern = vxc_ws_kernel<MAXT, NLP, KCH, GGA>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWS_NT), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob, sym);
}

template <bool GGA>
static int launch_vxc_ws(int maxt, int nlp, int kch, dim3 grid, size_t shmem, hipStream_t st, double *vmat,
                         const double *ao, int ngrid, int ld, const double *w, const double *vrho, const double *vgrad,
                         int slab, int nsplit, int tps, const double *aob, int sym) {
#define DQC_VWS_CASE(N, L)                                                                                          \
    if (maxt == N && nlp == L && kch == 16) {                                                                       \
        launch_vxc_ws_inst<N, L, 16, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob, sym); \
        return 0;                                                                                                   \
    }
    DQC_VWS_CASE(2, 1) DQC_VWS_CASE(4, 1) DQC_VWS_CASE(6, 1) DQC_VWS_CASE(8, 1) DQC_VWS_CASE(11, 1)
    DQC_VWS_CASE(2, 2) DQC_VWS_CASE(4, 2) DQC_VWS_CASE(6, 2) DQC_VWS_CASE(8, 2) DQC_VWS_CASE(11, 2)
    DQC_VWS_CASE(2, 4) DQC_VWS_CASE(4, 4) DQC_VWS_CASE(6, 4) DQC_VWS_CASE(8, 4) DQC_VWS_CASE(11, 4)
#undef DQC_VWS_CASE  // (ld <= 256 with 32 producer threads per row needs at most 4 pieces per thread)
    set_error("vxc_ws: internal dispatch error");
    return DQC_EINVAL;
}

template <int MAXT, int NLP, bool GGA>
static void launch_vxc_wsu_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                                const double *w, const double *vrho, const double *vgrad, int slab) {
    auto kern = vxc_wsu_kernel<MAXT, NLP, GGA>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWU_NT), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab);
}

template <int T>
static void launch_vxc_wsd(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, const double *w,
                           const double *vrho, const double *vgrad, int slab) {
    auto kern = vxc_wsd_kernel<T, 7>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWU_NT), shmem, st, vmat, ao, ngrid, w, vrho, vgrad, slab);
}

template <bool GGA>
static int launch_vxc_wsu(int maxt, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                          const double *w, const double *vrho, const double *vgrad, int slab) {
    if (maxt <= 9) launch_vxc_wsu_inst<9, 7, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab);
    else launch_vxc_wsu_inst<12, 7, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab);
    return 0;
}

template <int MAXT, int NLA, int NLB, bool GGA>
static void launch_vxc_ws2_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                                const double *w, const double *vrho, const double *vgrad, int slab, int NR, int NC, int LSA,
                                int LSB, const double *aob) {
    auto kern = vxc_ws2_kernel<MAXT, NLA, NLB, GGA>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWS2_NT), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, NR, NC, LSA, LSB, aob);
}

template <bool GGA>
static int launch_vxc_ws2(int maxt, int nla, int nlb, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao,
                          int ngrid, int ld, const double *w, const double *vrho, const double *vgrad, int slab, int NR,
                          int NC, int LSA, int LSB, const double *aob) {
#define DQC_VW2_CASE(N, A, B)                                                                                        \
    if (maxt == N && nla == A && nlb == B) {                                                                          \
        launch_vxc_ws2_inst<N, A, B, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, NR, NC, LSA, LSB, aob); \
        return 0;                                                                                                     \
    }
    DQC_VW2_CASE(8, 4, 4) DQC_VW2_CASE(8, 4, 6) DQC_VW2_CASE(11, 4, 4) DQC_VW2_CASE(11, 4, 6)
#undef DQC_VW2_CASE
    set_error("vxc_ws2: internal dispatch error");
    return DQC_EINVAL;
}

template <bool GGA>
static int launch_vxc(int maxt, int nl, int kch, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao,
                      int ngrid, int ld, const double *w, const double *vrho, const double *vgrad, int slab, int nsplit,
                      int tps, const double *aob) {
#define DQC_VXC_CASE(N, L)                                                                                        \
    if (maxt == N && nl == L && kch == 16) {                                                                      \
        launch_vxc_inst<N, L, 16, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob);  \
        return 0;                                                                                                 \
    }                                                                                                             \
    if (maxt == N && nl == L && kch == 8) {                                                                       \
        launch_vxc_inst<N, L, 8, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob);   \
        return 0;                                                                                                 \
    }
    DQC_VXC_CASE(2, 1) DQC_VXC_CASE(4, 1) DQC_VXC_CASE(8, 1) DQC_VXC_CASE(11, 1)
    DQC_VXC_CASE(2, 2) DQC_VXC_CASE(4, 2) DQC_VXC_CASE(8, 2) DQC_VXC_CASE(11, 2)
    DQC_VXC_CASE(2, 4) DQC_VXC_CASE(4, 4) DQC_VXC_CASE(8, 4) DQC_VXC_CASE(11, 4)
#undef DQC_VXC_CASE  // (this kernel only sees ld <= 208: wider bases take vxc_ws2_kernel)
    set_error("vxc: internal dispatch error");
    return DQC_EINVAL;
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
    // column panels: <= 16 tiles (LDA) / <= 14 (GGA: 15 and 16 tiles of accumulators + the epilogue's load batches spill)
    const int nchunk = (ntile + (gga ? 13 : 15)) / (gga ? 14 : 16);
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc = gga ? launch_density<true>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_dm, ntile, d_ao)
                 : launch_density<false>(nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_dm, ntile, d_ao);
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
    const int ld = dqc_padded_nao(nao), ntile = ld / 16;
    // (narrower panels -- fewer registers and less LDS, 3 blocks per CU instead of 2 -- change nothing: 0.57 ms for 5, 7, 9 or 13 tiles)
    const int lim = gga ? dqc::lr_max_nct(norb_pad / 16) : 16;
    const int nchunk = (ntile + lim - 1) / lim;
    int nct = (ntile + nchunk - 1) / nchunk;
    if (nct < 9 && (nct & 1) == 0) nct++;  // instantiated panel widths
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc = gga ? launch_density_lr<true>(norb_pad / 16, nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile)
                 : launch_density_lr<false>(norb_pad / 16, nct, grid, st, d_rho, d_grho, d_ao, ngrid, ld, d_orb, d_orbt, ntile);
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_grid_density_pair(double *d_out, const double *d_ao_a, const double *d_ao_b, int ngrid, int nao,
                          const double *d_dm, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ngrid <= 0) return DQC_OK;
    const int ld = dqc_padded_nao(nao), ntile = ld / 16;
    const int nchunk = (ntile + 15) / 16;
    const int nct = (ntile + nchunk - 1) / nchunk;
    dim3 grid((ngrid + DEN_BM - 1) / DEN_BM);
    int rc = launch_density<false>(nct, grid, st, d_out, nullptr, d_ao_a, ngrid, ld, d_dm, ntile, d_ao_b);
    if (rc) return rc;
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

static int grid_vxc_impl(double *d_vmat, const double *d_ao, const double *d_aob, int ncomp, int ngrid, int nao,
                         const double *d_w, const double *d_vrho, const double *d_vgrad, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    const bool gga = d_vgrad != nullptr;
    if (gga && ncomp < 4) { set_error("dqc_grid_vxc: vgrad given but ao has < 4 components"); return DQC_EINVAL; }
    const int ld = dqc_padded_nao(nao), T = ld / 16, ttot = T * T;
    DQC_HIP(hipMemsetAsync(d_vmat, 0, sizeof(double) * (size_t)ld * ld, st));
    if (ngrid > 0) {
        static const char *impl_env = getenv("DQC_VXC_IMPL");  // "reg": the unspecialised vxc_kernel (A/B runs)
        // one-operand forms without a gradient term (LDA Vxc, the tau terms of a meta-GGA) are symmetric matrices: the
        // wave-specialised kernel then computes the upper-triangular tiles only
        const bool ws_shape = ld <= VWS_LSMAX && !(impl_env && impl_env[0] == 'r');
        const bool sym = ws_shape && !gga && d_aob == d_ao;
        const int ttot_w = sym ? T * (T + 1) / 2 : ttot;
        if (ttot_w > 2 * 11 * VXC_WAVES && !(impl_env && impl_env[0] == 'r')) {
            // larger bases: rectangular ownership (vxc_ws2_kernel), rectangles of at most 8 x 11 tiles
            const int NR = (T + 7) / 8, NC = (T + 10) / 11;
            const int nrmax = (T + NR - 1) / NR, ncmax = (T + NC - 1) / NC;
            const int need2 = (nrmax * ncmax + VXC_WAVES - 1) / VXC_WAVES;
            const int maxt2 = need2 <= 8 ? 8 : 11;
            auto pad16 = [](int w_) { return (w_ & 31) == 16 ? w_ : w_ + 16; };  // == 16 (mod 32): conflict-free fragments
            const int LSA = pad16(nrmax * 16), LSB = pad16(ncmax * 16);
            const int nla = 4, nlb = (ncmax * 8 + 15) / 16 <= 4 ? 4 : 6;
            const int nsplit2 = NR * NC;
            int nslab = std::max(8, (512 / nsplit2) / 8 * 8);
            int slab = (ngrid + nslab - 1) / nslab;
            slab = (slab + 15) / 16 * 16;
            nslab = ((ngrid + slab - 1) / slab + 7) / 8 * 8;
            const size_t shmem2 = sizeof(double) * 2 * WS2_BUF;  // fixed-stride chunk layout, two buffers
            if (LSA > 144 || LSB > 208) { set_error("vxc_ws2: internal layout error"); return DQC_EINVAL; }
            dim3 grid2(nslab * nsplit2);
            int rc = gga ? launch_vxc_ws2<true>(maxt2, nla, nlb, grid2, shmem2, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, NR, NC, LSA, LSB, d_aob)
                         : launch_vxc_ws2<false>(maxt2, nla, nlb, grid2, shmem2, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, NR, NC, LSA, LSB, d_aob);
            if (rc) return rc;
            DQC_CHECK_LAUNCH();
            hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld);
            DQC_CHECK_LAUNCH();
            return DQC_OK;
        }
        // 145 <= nao <= 208 (10 <= T <= 13): one block per slab over the upper-triangular tiles (vxc_wsu_kernel) instead of two
        // blocks that each stage the whole slab.  One operand, symmetric result; DQC_VXC_IMPL=split keeps the two-block form.
        if (ws_shape && T >= 10 && T * (T + 1) / 2 <= 12 * VXC_WAVES && d_aob == d_ao && !(impl_env && impl_env[0] == 's')) {
            int ncu = 256;
            int nslab = ncu;
            int slab = (ngrid + nslab - 1) / nslab;
            slab = (slab + 15) / 16 * 16;
            nslab = (ngrid + slab - 1) / slab;
            const int need = (T * (T + 1) / 2 + VXC_WAVES - 1) / VXC_WAVES;
            const size_t shmem_u = sizeof(double) * 2 * VWS_BUF;
            int rc = 0;
            // GGA: one MFMA on the diagonal tiles (vxc_wsd_kernel); DQC_VXC_IMPL=upper keeps two on every tile (A/B runs)
            if (gga && (T == 13 || T == 11) && !(impl_env && impl_env[0] == 'u')) {
                if (T == 13) launch_vxc_wsd<13>(dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, d_w, d_vrho, d_vgrad, slab);
                else launch_vxc_wsd<11>(dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, d_w, d_vrho, d_vgrad, slab);
            } else {
                rc = gga ? launch_vxc_wsu<true>(need, dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab)
                         : launch_vxc_wsu<false>(need, dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab);
            }
            if (rc) return rc;
            DQC_CHECK_LAUNCH();
            hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld);
            DQC_CHECK_LAUNCH();
            return DQC_OK;
        }
        // tiles per block are capped at 11 per wave so that accumulators + prefetch registers fit 256 VGPRs
        static const int sizes_ws[] = {2, 4, 6, 8, 11}, sizes_reg[] = {2, 4, 8, 11, 11};
        const int *sizes_p = ws_shape ? sizes_ws : sizes_reg;
        const int cap = 11 * VXC_WAVES;
        const int nsplit = (ttot_w + cap - 1) / cap;
        const int tps = (ttot_w + nsplit - 1) / nsplit;
        const int need = (tps + VXC_WAVES - 1) / VXC_WAVES;
        int maxt = 11;
        for (int q = 0; q < 5; q++)
            if (sizes_p[q] >= need) { maxt = sizes_p[q]; break; }
        // chunk depth: 16 points while the double-buffered (phi, psi) chunk fits LDS, else 8
        const int kch = (sizeof(double) * 2 * 2 * 16 * (size_t)ld <= 150 * 1024) ? 16 : 8;
        const int tpr = 512 / kch;  // threads per chunk row
        const int nlneed = (ld / 2 + tpr - 1) / tpr;  // double2 columns per thread
        const int nl = nlneed <= 1 ? 1 : (nlneed <= 2 ? 2 : (nlneed <= 4 ? 4 : 8));
        if (nlneed > 8) { set_error("dqc_grid_vxc: nao above 1008 is not supported by this build"); return DQC_EINVAL; }
        // one 8-wave block per CU; slabs in multiples of 8 so that the XCD-aware decode is exact
        int nslab = std::max(8, (256 / nsplit) / 8 * 8);
        int slab = (ngrid + nslab - 1) / nslab;
        slab = (slab + kch - 1) / kch * kch;
        nslab = ((ngrid + slab - 1) / slab + 7) / 8 * 8;
        const size_t shmem = sizeof(double) * 2 * 2 * kch * ld;
        dim3 grid(nslab * nsplit);
        // default: wave-specialised kernel (8 MFMA waves + 4 producer waves); the producers hold a whole chunk in
        // registers, which bounds ld; DQC_VXC_IMPL=reg selects the unspecialised kernel
        const int tprp = VWS_PROD / kch;
        const int nlpneed = (ld / 2 + tprp - 1) / tprp;
        const int nlp = nlpneed <= 1 ? 1 : (nlpneed <= 2 ? 2 : (nlpneed <= 4 ? 4 : (nlpneed <= 7 ? 7 : 8)));
        if (nlpneed <= 7 && kch == 16 && ld <= VWS_LSMAX && !(impl_env && impl_env[0] == 'r')) {
            const size_t shmem_ws = sizeof(double) * 2 * VWS_BUF;  // fixed-stride chunk layout, two buffers
            int rc = gga ? launch_vxc_ws<true>(maxt, nlp, kch, grid, shmem_ws, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob, 0)
                         : launch_vxc_ws<false>(maxt, nlp, kch, grid, shmem_ws, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob, sym ? 1 : 0);
            if (rc) return rc;
            DQC_CHECK_LAUNCH();
            hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld);
            DQC_CHECK_LAUNCH();
            return DQC_OK;
        }
        int rc = gga ? launch_vxc<true>(maxt, nl, kch, grid, shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob)
                     : launch_vxc<false>(maxt, nl, kch, grid, shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob);
        if (rc) return rc;
        DQC_CHECK_LAUNCH();
        hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}

int dqc_grid_vxc(double *d_vmat, const double *d_ao, int ncomp, int ngrid, int nao, const double *d_w,
                 const double *d_vrho, const double *d_vgrad, void *stream) {
    return grid_vxc_impl(d_vmat, d_ao, d_ao, ncomp, ngrid, nao, d_w, d_vrho, d_vgrad, stream);
}

int dqc_grid_vxc_pair(double *d_vmat, const double *d_ao_a, const double *d_ao_b, int ngrid, int nao, const double *d_w,
                      const double *d_v, void *stream) {
    return grid_vxc_impl(d_vmat, d_ao_a, d_ao_b, 1, ngrid, nao, d_w, d_v, nullptr, stream);
}

#ifdef VWU_TRACE
int dqc_debug_vwu_trace(long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dqc::g_vwu_trace), sizeof(long long) * 2 * dqc::VWU_TRACE_N);
}
#endif
#ifdef FG_TRACE
int dqc_debug_fused_trace(long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dqc::g_fg_trace), sizeof(long long) * 2 * dqc::FG_TRACE_N);
}
#endif

/* fused density -> XC -> Vxc (see fused_grid_kernel).  Returns DQC_EUNSUPPORTED (no launch, no error text change) for shapes it
 * does not cover: the caller then runs dqc_grid_density_lr + dqc_xc_eval + dqc_grid_vxc. */
int dqc_grid_fused_supported(int nao, int norb_pad) {
    const int ld = dqc_padded_nao(nao), T = ld / 16;
    return (T == 11 || T == 13) && norb_pad >= 16 && norb_pad <= 64 && norb_pad % 16 == 0;
}

int dqc_grid_fused(double *d_vmat, double *d_rho, double *d_grho, double *d_exc, const double *d_ao, int ngrid, int nao,
                   const double *d_w, const double *d_orb, const double *d_orbt, int norb_pad, const int *ids,
                   const double *coefs, int nterm, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (!dqc_grid_fused_supported(nao, norb_pad)) { set_error("dqc_grid_fused: shape not covered (10 <= ld/16 <= 13, norb_pad <= 64)"); return DQC_EINVAL; }
    if (nterm < 0 || nterm > 8) { set_error("dqc_grid_fused: at most 8 functional terms"); return DQC_EINVAL; }
    XcTerms t;
    t.n = nterm;
    for (int i = 0; i < nterm; i++) {
        t.id[i] = ids[i];
        t.c[i] = coefs[i];
        switch (ids[i]) {
        case DQC_XC_LDA_X: case DQC_XC_LDA_C_PW: case DQC_XC_GGA_X_PBE: case DQC_XC_GGA_C_PBE: break;
        default: set_error("dqc_grid_fused: LDA / GGA functional ids only"); return DQC_EINVAL;
        }
    }
    const int ld = dqc_padded_nao(nao), T = ld / 16;
    DQC_HIP(hipMemsetAsync(d_vmat, 0, sizeof(double) * (size_t)ld * ld, st));
    if (d_exc) DQC_HIP(hipMemsetAsync(d_exc, 0, sizeof(double), st));
    if (ngrid <= 0) return DQC_OK;
    int nslab = 256;
    int slab = (ngrid + nslab - 1) / nslab;
    slab = (slab + 15) / 16 * 16;
    nslab = (ngrid + slab - 1) / slab;
    const size_t shmem = sizeof(double) * (2 * VWS_BUF + FG_ABUF) + 64;
    int xcs = 0;
    if (nterm == 2 && ids[0] == DQC_XC_GGA_X_PBE && ids[1] == DQC_XC_GGA_C_PBE) xcs = 1;
    if (nterm == 2 && ids[0] == DQC_XC_LDA_X && ids[1] == DQC_XC_LDA_C_PW) xcs = 2;
#define DQC_FG_CASE(M, R) DQC_FG_CASE2(M, R, 0) DQC_FG_CASE2(M, R, 1) DQC_FG_CASE2(M, R, 2)
#define DQC_FG_CASE2(M, R, X)                                                                                          \
    if ((T == 13 ? 12 : 9) == M && norb_pad / 16 == R && xcs == X) {                                                   \
        auto kern = fused_grid_kernel<M, 7, R, X>;                                                                     \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);         \
        hipLaunchKernelGGL(kern, dim3(nslab), dim3(VWU_NT), shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_orb, d_orbt, slab, t, \
                           d_rho, d_grho, d_exc);                                                                      \
    }
    DQC_FG_CASE(12, 1) DQC_FG_CASE(12, 2) DQC_FG_CASE(12, 3) DQC_FG_CASE(12, 4)
    DQC_FG_CASE(9, 1) DQC_FG_CASE(9, 2) DQC_FG_CASE(9, 3) DQC_FG_CASE(9, 4)
#undef DQC_FG_CASE
#undef DQC_FG_CASE2
    DQC_CHECK_LAUNCH();
    hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
