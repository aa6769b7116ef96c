// grid_vxc.hip -- the Vxc matrix M = Phi^T Psi: the second GEMM-shaped pass over the cached AO matrix
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

int vxc_cus_cap();  // host.hip: dqc_set_vxc_cus / DQC_VXC_CUS

// fixed-point scale of the split-K accumulation into V in deterministic mode (0: fp64 atomics); common.hpp: acc_add
__device__ double g_vxc_det_scale = 0.0;

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
                                                     const double *__restrict__ aob, int lda, int LS) {
    // aob: LDA mode only -- array the Psi operand is built from (== ao except for the "pair" form ao^T diag(w v) aob)
    // lda: row stride of the AO arrays in HBM (dqc_ao_stride); ld = 16 T: the tile-padded width that is staged (columns
    // lda .. ld - 1 of a row are the first doubles of the next row: finite values that only reach discarded rows / columns
    // of M); LS: LDS row stride, == 16 (mod 32) so that the fragment reads are conflict-free
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int BUF = 2 * KCH * LS;  // phi + psi
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int T = ld >> 4, ttot = T * T;
    const size_t cs = (size_t)ngrid * lda;

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
        src = ao + (size_t)gg * lda;
        srcb = aob + (size_t)gg * lda;
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
                    acc_add(&vmat[(size_t)(ia + 4 * r) * ld + ib], acc[t][r], g_vxc_det_scale);
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
template <int MAXT, int KCH, int D = WS_D, int GSA = VWS_GS, int GSB = VWS_GS, int NTL = MAXT>
__device__ __forceinline__ void ws_chunk(const unsigned (&pa)[MAXT], const unsigned (&pb)[MAXT], v4d (&acc)[MAXT]) {
    // NTL <= MAXT: the tiles this wave really owns (no dummy MFMAs: the deal below gives the waves of a block tile counts
    // that differ by at most one, and every count has its own straight-line body)
    constexpr int NS = (KCH / 4) * NTL;
    if constexpr (NTL > 0) {
        double fa[D + 1], fb[D + 1];
#pragma unroll
        for (int s = 0; s < D && s < NS; s++) {
            fa[s % (D + 1)] = *(lds_cdouble_t *)(pa[s % NTL] + (s / NTL) * GSA * 8);
            fb[s % (D + 1)] = *(lds_cdouble_t *)(pb[s % NTL] + (s / NTL) * GSB * 8);
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (s + D < NS) {
                const int s2 = s + D;
                fa[s2 % (D + 1)] = *(lds_cdouble_t *)(pa[s2 % NTL] + (s2 / NTL) * GSA * 8);
                fb[s2 % (D + 1)] = *(lds_cdouble_t *)(pb[s2 % NTL] + (s2 / NTL) * GSB * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[s % NTL] = mfma_f64(fa[s % (D + 1)], fb[s % (D + 1)], acc[s % NTL]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// A consumer wave of vxc_ws_kernel / vxc_ws2_kernel with NTL tiles (compile time: the accumulators, the fragment addresses and
// the straight-line MFMA stream are sized for exactly the tiles the wave owns).  Tile u = t0 + t of the block's list maps to
// local tile coordinates (li, lj) -- LDS columns 16 li of the A part, 16 lj of the B part -- and to the output tile
// (r0 + li, c0 + lj):  nc > 0: rectangle, (u / nc, u % nc);  nc == 0, sym == 0: (u / T, u % T);  sym: upper triangle of T rows.
struct WsTiles { int sym, T, nc, r0, c0, t0, nreal; };  // nreal: tiles of the block's list that exist (vxc_ws2_kernel pads lighter blocks)
__device__ __forceinline__ void ws_tile(const WsTiles &m, int u, int &li, int &lj) {
    if (m.nc) { li = u / m.nc; lj = u - li * m.nc; return; }
    if (!m.sym) { li = u / m.T; lj = u - li * m.T; return; }
    int i = 0, rem = u;
    while (rem >= m.T - i) { rem -= m.T - i; i++; }  // row i of the upper triangle holds T - i tiles
    li = i;
    lj = i + rem;
}
template <int NTL, int KCH, int D, int GSA, int GSB, int XS, int BUF>
__device__ __forceinline__ void ws_consumer(double *lds, double *__restrict__ vmat, int ld, int nchunk, const WsTiles m, int LSA, int LSB) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    constexpr int NA = NTL > 0 ? NTL : 1;
    v4d acc[NA];
    unsigned pa[NA], pb[NA];  // LDS byte addresses of the A / B fragments (k-group 0, current buffer)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
#pragma unroll
    for (int t = 0; t < NTL; t++) {
        acc[t] = v4d{0, 0, 0, 0};
        int li, lj;
        ws_tile(m, min(m.t0 + t, m.nreal - 1), li, lj);  // (a padding tile re-reads the block's last one; its sum is discarded)
        pa[t] = lds0 + 8u * (unsigned)(lk * LSA + li * 16 + lr);
        pb[t] = lds0 + 8u * (unsigned)(XS + lk * LSB + lj * 16 + lr);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
        __syncthreads();  // chunk c - 1 done: the producers' combine window opens ...
        __syncthreads();  // ... and closes
        if constexpr (NTL > 0) {
            ws_chunk<NA, KCH, D, GSA, GSB, NTL>(pa, pb, acc);
            const unsigned delta = (c & 1) ? (unsigned)(-BUF * 8) : (unsigned)(BUF * 8);  // on to the other buffer
#pragma unroll
            for (int t = 0; t < NTL; t++) { pa[t] += delta; pb[t] += delta; }
        }
    }
#pragma unroll
    for (int t = 0; t < NTL; t++) {
        if (m.t0 + t >= m.nreal) continue;
        int li, lj;
        ws_tile(m, m.t0 + t, li, lj);
        const int ia = (m.r0 + li) * 16 + lk, ib = (m.c0 + lj) * 16 + lr;
        const double sc = (m.sym && li != lj) ? 2.0 : 1.0;
#pragma unroll
        for (int r = 0; r < 4; r++) acc_add(&vmat[(size_t)(ia + 4 * r) * ld + ib], sc * acc[t][r], g_vxc_det_scale);
    }
}
// dispatch on the (wave-uniform) tile count nt in [0, MAXT]
template <int MAXT, int KCH, int D, int GSA, int GSB, int XS, int BUF, int N = MAXT>
__device__ __forceinline__ void ws_consumer_n(int nt, double *lds, double *__restrict__ vmat, int ld, int nchunk, const WsTiles m,
                                              int LSA, int LSB) {
    if (nt == N) ws_consumer<N, KCH, D, GSA, GSB, XS, BUF>(lds, vmat, ld, nchunk, m, LSA, LSB);
    else if constexpr (N > 0) ws_consumer_n<MAXT, KCH, D, GSA, GSB, XS, BUF, N - 1>(nt, lds, vmat, ld, nchunk, m, LSA, LSB);
}
// balanced deal of `n` tiles to the 8 consumer waves: counts differ by at most one (waves w and w + 4 share a SIMD)
__device__ __forceinline__ void ws_deal(int n, int wave, int &t0, int &nt) {
    const int tbase = n / VXC_WAVES, trem = n % VXC_WAVES;
    nt = tbase + (wave < trem ? 1 : 0);
    t0 = wave * tbase + min(wave, trem);
}

template <int MAXT, int NLP, int KCH, bool GGA>
__global__ __launch_bounds__(VWS_NT, VWS_NT / 256) void vxc_ws_kernel(double *__restrict__ vmat, const double *__restrict__ ao,
                                                          int ngrid, int ld, const double *__restrict__ w,
                                                          const double *__restrict__ vrho,
                                                          const double *__restrict__ vgrad, int slab, int nsplit,
                                                          int tiles_per_split, const double *__restrict__ aob, int sym, int lda, int LS) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    static_assert(KCH == 16, "the fixed-stride chunk layout is laid out for 16-point chunks");
    constexpr int BUF = VWS_BUF;  // (lda / ld / LS: see vxc_kernel)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t cs = (size_t)ngrid * lda;
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
        const unsigned voff0 = 8u * (unsigned)(prow * lda + pcol * 2);  // bytes from the chunk's first row
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
            const size_t nb = (size_t)rows * lda * 8;    // < 2^31: a slab is a few MB
#pragma unroll
            for (int d = 0; d < (GGA ? 4 : 1); d++) {
                const auto r = rsrc(ao + d * cs + (size_t)g0 * lda, nb);
#pragma unroll
                for (int i = 0; i < NLP; i++)
                    if ((pcol + i * TPR) * 2 < ld) raw[i][d] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
            }
            if (!GGA && !sym) {  // (sym: the second operand is the first)
                const auto r = rsrc(aob + (size_t)g0 * lda, nb);
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
            __syncthreads();  // the consumers have finished chunk c - 1 and wait: the vector ALUs are free for the combine
            if (c + 1 < nchunk) stage();              // chunk c + 1: its loads were issued a whole period ago
            if (wave == VXC_WAVES) VXC_TRACE_POINT(1, c + 1);  // combine done
            __syncthreads();  // the consumers start the MFMAs of chunk c
            if (c + 2 < nchunk) prefetch(c + 2);      // VALU-free issue, in flight during the MFMAs
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    // sym: A^T diag(w v) A with ONE operand (LDA Vxc, the tau terms of a meta-GGA) is symmetric -- only the tiles (i <= j) are
    // computed, off-diagonal ones doubled so that the final (M + M^T) / 2 restores both halves
    const int T = ld >> 4, ttot = sym ? T * (T + 1) / 2 : T * T;
    const int tc0 = split * tiles_per_split;
    const int tc1 = min(tc0 + tiles_per_split, ttot);
    int t0, nt;
    ws_deal(tc1 - tc0, wave, t0, nt);
    const WsTiles m{sym, T, 0, 0, 0, tc0 + t0, tc1};
    ws_consumer_n<MAXT, KCH, WS_D, VWS_GS, VWS_GS, VWS_XS, VWS_BUF>(nt, lds, vmat, ld, nchunk, m, LS, LS);
}

// ---------------------------------------------------------------------------------------------
// Vxc for larger bases: the wave-specialised kernel with RECTANGULAR output ownership.  vxc_ws_kernel splits the
// T x T output tiles linearly over `nsplit` blocks that each stage ALL columns of a slab; beyond two blocks per
// slab that re-reads the slab nsplit times (nsplit = 18 at nao = 624).  Here block (slab, i, j) owns the tile
// rectangle  rows [i T / NR, (i+1) T / NR)  x  cols [j T / NC, (j+1) T / NC)  (<= 9 x 12 tiles, <= 88 of them) and its producers
// stage only the Phi columns of those rows (A operand) and the four AO components of those columns (-> Psi, B
// operand): per-block loads are what vxc_ws_kernel loads at nao = 208, and a slab is re-read NC + 4 NR times in
// total instead of 5 nsplit.  Chunks are always 16 points (39 KB per LDS buffer).
// ---------------------------------------------------------------------------------------------
template <int MAXT, int NLA, int NLB, bool GGA>
__global__ __launch_bounds__(VWS2_NT, 3) void vxc_ws2_kernel(double *__restrict__ vmat, const double *__restrict__ ao,
                                                           int ngrid, int ld, const double *__restrict__ w,
                                                           const double *__restrict__ vrho,
                                                           const double *__restrict__ vgrad, int slab, int NR, int NC,
                                                           int LSA, int LSB, const double *__restrict__ aob, int lda) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16;
    // chunk layout as in vxc_ws_kernel: fixed strides between the 4-point k-groups (Phi part: WS2_GSA, Psi part: WS2_GSB), so that
    // the consumers' fragment reads are  address register + immediate;  rows inside a k-group at the run-time strides LSA / LSB
    constexpr int BUF = WS2_BUF;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t cs = (size_t)ngrid * lda;
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
        const unsigned voff0 = 8u * (unsigned)(prow * lda + pcol * 2);
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
            const size_t nba = (size_t)rows * lda * 8 - (size_t)r0 * 128, nbb = (size_t)rows * lda * 8 - (size_t)c0 * 128;
            {
                const auto r = rsrc(ao + (size_t)g0 * lda + r0 * 16, nba);
#pragma unroll
                for (int i = 0; i < NLA; i++)
                    if ((pcol + i * TPR) * 2 < wa) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + i * TPR * 16, 0, 0);
            }
#pragma unroll
            for (int d = 0; d < (GGA ? 4 : 1); d++) {
                const auto r = rsrc((GGA ? ao : aob) + d * cs + (size_t)g0 * lda + c0 * 16, nbb);
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
    // EVERY block of a slab runs the tile count of the largest rectangle (the smaller ones pad with repeats of their last tile):
    // blocks that issue the same MFMA stream per chunk walk through the slab in step, and the slab's rows then come out of
    // the XCD's L2 for all but the first reader.  With the exact counts (T = 26: 64 ... 81 tiles) the lighter blocks ran ahead and
    // FETCH_SIZE doubled (9.7 instead of 4.9 GB per launch at nao 412) for the same launch time
    const int ntmax = ((T + NR - 1) / NR) * ((T + NC - 1) / NC);
    int t0, nt;
    ws_deal(ntmax, wave, t0, nt);
    const WsTiles m{0, T, nc, r0, c0, t0, nr * nc};
    ws_consumer_n<MAXT, KCH, 4, WS2_GSA, WS2_GSB, WS2_XS, WS2_BUF>(nt, lds, vmat, ld, nchunk, m, LSA, LSB);
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

template <int MAXT, int NTL, bool TWO, int D = 2, int KG0 = 0, int NKG = 4, int GS = VWS_GS, int XS = VWS_XS>
__device__ __forceinline__ void wsu_chunk(const unsigned (&pi)[MAXT], const unsigned (&pj)[MAXT], v4d (&acc)[MAXT]) {
    // NTL <= MAXT: tiles actually looped over (waves that own one tile fewer skip the dummy MFMAs);
    // k-groups KG0 .. KG0 + NKG - 1 of the 16-point chunk (the fused kernel runs a chunk in two halves)
    constexpr int H = TWO ? 2 : 1, NS = NKG * NTL * H;
    double fa[D + 1], fb[D + 1];
    auto rd = [&](int s) {
        const int kk = KG0 + s / (NTL * H), t = (s % (NTL * H)) / H, h = s % H;
        // h = 0: A = Phi_i, B = Psi_j;   h = 1: A = Psi_i, B = Phi_j
        fa[s % (D + 1)] = *(lds_cdouble_t *)(pi[t] + (kk * GS + (h ? XS : 0)) * 8);
        fb[s % (D + 1)] = *(lds_cdouble_t *)(pj[t] + (kk * GS + (h ? 0 : XS)) * 8);
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

// the same with a (wave-uniform) run-time tile count in [0, MAXT]: one straight-line body per count, no dummy MFMAs
template <int MAXT, bool TWO, int N = MAXT>
__device__ __forceinline__ void wsu_chunk_n(int nt, const unsigned (&pi)[MAXT], const unsigned (&pj)[MAXT], v4d (&acc)[MAXT]) {
    if constexpr (N == 0) return;
    else if (nt == N) wsu_chunk<MAXT, N, TWO>(pi, pj, acc);
    else wsu_chunk_n<MAXT, TWO, N - 1>(nt, pi, pj, acc);
}

// the 4 producer waves of vxc_wsu_kernel / vxc_wsb_kernel (threads 512 .. 767): chunk c + 2 travels HBM -> registers while the
// consumers run the MFMAs of chunk c; chunk c + 1 is combined into (Phi, Psi) and written to LDS in the window between chunks
template <int NLP, bool GGA>
DQC_DEV void vwu_producer(double *lds, const double *__restrict__ ao, int ngrid, int ld, const double *__restrict__ w,
                          const double *__restrict__ vrho, const double *__restrict__ vgrad, int gs, int ge, int nchunk,
                          int lda, int LS) {
    // lda: row stride of the AO arrays in HBM; ld = 16 T: staged width; LS: LDS row stride (see vxc_kernel)
    constexpr int KCH = 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    (void)wave; (void)lane;
    const size_t cs = (size_t)ngrid * lda;
    // ------------------------------------------------------------------ producers (see vxc_ws_kernel)
    __builtin_amdgcn_s_setprio(3);
    constexpr int TPR = VWU_PROD / KCH;  // 16 threads per chunk row
    const int pt = tid - 512;
    const int prow = pt / TPR, pcol = pt % TPR;
    const unsigned voff0 = 8u * (unsigned)(prow * lda + pcol * 2);
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
        const size_t nb = (size_t)rows * lda * 8;
#pragma unroll
        for (int d = 0; d < (GGA ? 4 : 1); d++) {
            const auto r = rsrc(ao + d * cs + (size_t)g0 * lda, nb);
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
                                                           const double *__restrict__ vgrad, int slab, int lda, int LS) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int gs = blockIdx.x * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;

    if (wave >= VXC_WAVES) {
        vwu_producer<NLP, GGA>(lds, ao, ngrid, ld, w, vrho, vgrad, gs, ge, nchunk, lda, LS);
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
        wsu_chunk_n<MAXT, GGA>(nt, pi, pj, acc);
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
            for (int r = 0; r < 4; r++) acc_add(&vmat[(size_t)(ia + 4 * r) * ld + ib], sc * acc[t][r], g_vxc_det_scale);
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
constexpr int wsd_ls(int T) { return ((16 * T) & 31) == 16 ? 16 * T : 16 * T + 16; }

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
    constexpr int NT = NO + ND, LS = wsd_ls(T);  // rows of the output matrix: 16 T; LDS rows: == 16 (mod 32)
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
        for (int r = 0; r < 4; r++) acc_add(&vmat[(size_t)(ia + 4 * r) * (16 * T) + ib], acc[t][r], g_vxc_det_scale);
    }
}

template <int T, int NLP>
__global__ __launch_bounds__(VWU_NT, 3) void vxc_wsd_kernel(double *__restrict__ vmat, const double *__restrict__ ao, int ngrid,
                                                           const double *__restrict__ w, const double *__restrict__ vrho,
                                                           const double *__restrict__ vgrad, int slab, int lda) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int KCH = 16, ld = 16 * T;
    constexpr WsdDeal<T> DL{};
    static_assert(DL.max_tiles() <= 12, "more than 12 accumulator tiles per wave");
    const int wave = threadIdx.x >> 6;
    const int gs = blockIdx.x * slab, ge = min(gs + slab, ngrid);
    if (gs >= ngrid) return;
    const int nchunk = (ge - gs + KCH - 1) / KCH;
    if (wave >= VXC_WAVES) {
        vwu_producer<NLP, true>(lds, ao, ngrid, ld, w, vrho, vgrad, gs, ge, nchunk, lda, wsd_ls(T));
        return;
    }
#define DQC_WSD_CASE(W) case W: wsd_consumer<T, DL.no[W], DL.nd[W]>(lds, vmat, nchunk, DL.o0[W], DL.d0[W]); break;
    switch (wave) {  // wave-uniform; equal (NO, ND) pairs share one instantiation
        DQC_WSD_CASE(0) DQC_WSD_CASE(1) DQC_WSD_CASE(2) DQC_WSD_CASE(3)
        DQC_WSD_CASE(4) DQC_WSD_CASE(5) DQC_WSD_CASE(6) DQC_WSD_CASE(7)
    }
#undef DQC_WSD_CASE
}



// V = (M + M^T) / 2 on the (ld, ld) matrix (deterministic mode: M arrives as fixed-point integers).  Rows / columns nao .. ld - 1
// are ZEROED: the kernels stage 16 T columns per AO row, and where the row stride of the AO arrays is below that
// (dqc_ao_stride) the last tile's extra columns hold the first values of the next row -- finite numbers that only reach
// these padding rows / columns
__global__ void symmetrize_kernel(double *m, int ld, int nao) {
    const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
    const double sc = g_vxc_det_scale;
    if (i < ld && j < i) {
        const double v = i < nao ? 0.5 * (det_value(m[(size_t)i * ld + j], sc) + det_value(m[(size_t)j * ld + i], sc)) : 0.0;
        m[(size_t)i * ld + j] = v;
        m[(size_t)j * ld + i] = v;
    } else if (i < ld && j == i) {
        if (i >= nao) m[(size_t)i * ld + i] = 0.0;
        else if (sc != 0.0) m[(size_t)i * ld + i] = det_value(m[(size_t)i * ld + i], sc);
    }
}

// deterministic mode (dqc_set_deterministic): the device-side scale follows the host flag, per device
static int sync_vxc_det_scale() {
    static double current[64];  // what each device's g_vxc_det_scale holds (0: fp64 atomics)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    // |V_ij| <= max |v_xc| int |phi_i phi_j| stays far below 2^14 for normalised AOs: 2^47 leaves the sum 63 bits
    const double want = deterministic_mode() ? 140737488355328.0 : 0.0;
    if (current[dev] != want) {
        DQC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_vxc_det_scale), &want, sizeof(double)));
        current[dev] = want;
    }
    return 0;
}

template <int MAXT, int NL, int KCH, bool GGA>
static void launch_vxc_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                            const double *w, const double *vrho, const double *vgrad, int slab, int nsplit, int tps,
                            const double *aob, int lda, int LS) {
    (void)hipFuncSetAttribute((const void *)vxc_kernel<MAXT, NL, KCH, GGA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)shmem);
    hipLaunchKernelGGL((vxc_kernel<MAXT, NL, KCH, GGA>), grid, dim3(512), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab,
                       nsplit, tps, aob, lda, LS);
}

template <int MAXT, int NLP, int KCH, bool GGA>
static void launch_vxc_ws_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                               const double *w, const double *vrho, const double *vgrad, int slab, int nsplit, int tps,
                               const double *aob, int sym, int lda, int LS) {
    auto kern = vxc_ws_kernel<MAXT, NLP, KCH, GGA>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWS_NT), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob, sym, lda, LS);
}

template <bool GGA>
static int launch_vxc_ws(int maxt, int nlp, int kch, dim3 grid, size_t shmem, hipStream_t st, double *vmat,
                         const double *ao, int ngrid, int ld, const double *w, const double *vrho, const double *vgrad,
                         int slab, int nsplit, int tps, const double *aob, int sym, int lda, int LS) {
#define DQC_VWS_CASE(N, L)                                                                                          \
    if (maxt == N && nlp == L && kch == 16) {                                                                       \
        launch_vxc_ws_inst<N, L, 16, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob, sym, lda, LS); \
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
                                const double *w, const double *vrho, const double *vgrad, int slab, int lda, int LS) {
    auto kern = vxc_wsu_kernel<MAXT, NLP, GGA>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWU_NT), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, lda, LS);
}

template <int T>
static void launch_vxc_wsd(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, const double *w,
                           const double *vrho, const double *vgrad, int slab, int lda) {
    auto kern = vxc_wsd_kernel<T, 7>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWU_NT), shmem, st, vmat, ao, ngrid, w, vrho, vgrad, slab, lda);
}

template <bool GGA>
static int launch_vxc_wsu(int maxt, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                          const double *w, const double *vrho, const double *vgrad, int slab, int lda, int LS) {
    if (maxt <= 9) launch_vxc_wsu_inst<9, 7, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, lda, LS);
    else launch_vxc_wsu_inst<12, 7, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, lda, LS);
    return 0;
}

template <int MAXT, int NLA, int NLB, bool GGA>
static void launch_vxc_ws2_inst(dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao, int ngrid, int ld,
                                const double *w, const double *vrho, const double *vgrad, int slab, int NR, int NC, int LSA,
                                int LSB, const double *aob, int lda) {
    auto kern = vxc_ws2_kernel<MAXT, NLA, NLB, GGA>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, grid, dim3(VWS2_NT), shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, NR, NC, LSA, LSB, aob, lda);
}

template <bool GGA>
static int launch_vxc_ws2(int maxt, int nla, int nlb, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao,
                          int ngrid, int ld, const double *w, const double *vrho, const double *vgrad, int slab, int NR,
                          int NC, int LSA, int LSB, const double *aob, int lda) {
#define DQC_VW2_CASE(N, A, B)                                                                                        \
    if (maxt == N && nla == A && nlb == B) {                                                                          \
        launch_vxc_ws2_inst<N, A, B, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, NR, NC, LSA, LSB, aob, lda); \
        return 0;                                                                                                     \
    }
    DQC_VW2_CASE(8, 4, 4) DQC_VW2_CASE(8, 4, 6) DQC_VW2_CASE(11, 4, 4) DQC_VW2_CASE(11, 4, 6)
    DQC_VW2_CASE(8, 5, 4) DQC_VW2_CASE(8, 5, 6) DQC_VW2_CASE(11, 5, 4) DQC_VW2_CASE(11, 5, 6)  // 9-row rectangles (NLA = 5)
#undef DQC_VW2_CASE
    set_error("vxc_ws2: internal dispatch error");
    return DQC_EINVAL;
}

template <bool GGA>
static int launch_vxc(int maxt, int nl, int kch, dim3 grid, size_t shmem, hipStream_t st, double *vmat, const double *ao,
                      int ngrid, int ld, const double *w, const double *vrho, const double *vgrad, int slab, int nsplit,
                      int tps, const double *aob, int lda, int LS) {
#define DQC_VXC_CASE(N, L)                                                                                        \
    if (maxt == N && nl == L && kch == 16) {                                                                      \
        launch_vxc_inst<N, L, 16, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob, lda, LS);  \
        return 0;                                                                                                 \
    }                                                                                                             \
    if (maxt == N && nl == L && kch == 8) {                                                                       \
        launch_vxc_inst<N, L, 8, GGA>(grid, shmem, st, vmat, ao, ngrid, ld, w, vrho, vgrad, slab, nsplit, tps, aob, lda, LS);   \
        return 0;                                                                                                 \
    }
    DQC_VXC_CASE(2, 1) DQC_VXC_CASE(4, 1) DQC_VXC_CASE(8, 1) DQC_VXC_CASE(11, 1)
    DQC_VXC_CASE(2, 2) DQC_VXC_CASE(4, 2) DQC_VXC_CASE(8, 2) DQC_VXC_CASE(11, 2)
    DQC_VXC_CASE(2, 4) DQC_VXC_CASE(4, 4) DQC_VXC_CASE(8, 4) DQC_VXC_CASE(11, 4)
#undef DQC_VXC_CASE  // (this kernel only sees ld <= 256: wider bases take vxc_ws2_kernel)
    set_error("vxc: internal dispatch error");
    return DQC_EINVAL;
}

}  // namespace dqc

extern "C" {

static int grid_vxc_impl(double *d_vmat, const double *d_ao, const double *d_aob, int ncomp, int ngrid, int nao,
                         const double *d_w, const double *d_vrho, const double *d_vgrad, void *stream, bool raw = false) {
    // raw: the cross-block sums M are left as the kernels wrote them (V = (M + M^T) / 2 restricted to the first nao rows / columns is
    // formed by the consumer: dqc_fock_finish with vxc_raw) -- one launch less per build
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    const bool gga = d_vgrad != nullptr;
    if (gga && ncomp < 4) { set_error("dqc_grid_vxc: vgrad given but ao has < 4 components"); return DQC_EINVAL; }
    // ld = 16 T: rows / columns of the output matrix and the width the kernels stage; lda: row stride of the AO arrays;
    // LS: LDS row stride of a staged chunk, == 16 (mod 32) so that the ds_read_b64 fragment reads are conflict-free
    const int ld = dqc_padded_nao(nao), lda = dqc_ao_stride(nao), T = ld / 16, ttot = T * T;
    auto pad16 = [](int w_) { return (w_ & 31) == 16 ? w_ : w_ + 16; };
    const int LS = pad16(ld);
    if (sync_vxc_det_scale()) return DQC_EHIP;
    DQC_HIP(hipMemsetAsync(d_vmat, 0, sizeof(double) * (size_t)ld * ld, st));
    if (ngrid > 0) {
        static const char *impl_env = getenv("DQC_VXC_IMPL");  // "reg": the unspecialised vxc_kernel (A/B runs)
        // one-operand forms without a gradient term (LDA Vxc, the tau terms of a meta-GGA) are symmetric matrices: the
        // wave-specialised kernel then computes the upper-triangular tiles only
        const bool ws_shape = LS <= VWS_LSMAX && !(impl_env && impl_env[0] == 'r');
        const bool sym = ws_shape && !gga && d_aob == d_ao;
        const int ttot_w = sym ? T * (T + 1) / 2 : ttot;
        if (ttot_w > 2 * 11 * VXC_WAVES && !(impl_env && impl_env[0] == 'r')) {
            // larger bases: rectangular ownership (vxc_ws2_kernel), rectangles of at most 9 x 12 tiles
            // rectangle shape: a block stages nr Phi tile columns (one component) and 4 nc AO-component tile columns for its
            // nr x nc tiles -- (nr + 4 nc) / (nr nc) operand tile columns per MFMA: rows are cheap, columns dear.  The tallest
            // rectangle the layout allows (9 rows: LSA <= 144), then the widest that keeps <= 11 accumulator tiles per wave:
            // T = 26 (naphthalene / cc-pVTZ): 9 x 9 tiles, 9 blocks per slab (round 2: 7 x 9, 12 blocks); T = 17: 9 x 9 (6 x 9).
            // DQC_WS2_NR / DQC_WS2_NC override the block counts (A/B runs).
            int NR = (T + 8) / 9, NC = 1;
            {
                const int nrm = (T + NR - 1) / NR;
                while ((T + NC - 1) / NC > 12 || nrm * ((T + NC - 1) / NC) > 11 * VXC_WAVES) NC++;
                const char *e1 = getenv("DQC_WS2_NR"), *e2 = getenv("DQC_WS2_NC");
                if (e1 && atoi(e1) > 0) NR = atoi(e1);
                if (e2 && atoi(e2) > 0) NC = atoi(e2);
            }
            const int nrmax = (T + NR - 1) / NR, ncmax = (T + NC - 1) / NC;
            if (nrmax > 9 || ncmax > 12 || nrmax * ncmax > 11 * VXC_WAVES) { set_error("vxc_ws2: rectangle outside the kernel's limits"); return DQC_EINVAL; }
            const int need2 = (nrmax * ncmax + VXC_WAVES - 1) / VXC_WAVES;
            const int maxt2 = need2 <= 8 ? 8 : 11;
            const int LSA = pad16(nrmax * 16), LSB = pad16(ncmax * 16);
            const int nla = nrmax <= 8 ? 4 : 5, nlb = (ncmax * 8 + 15) / 16 <= 4 ? 4 : 6;  // b128 loads per producer thread and row
            const int nsplit2 = NR * NC;
            // two blocks per CU: the nine (NR x NC) blocks of a slab start together and stay in step (see the consumers), so the slab
            // is fetched from HBM about once; many small blocks (DQC_VXC_BLOCKS=3072: ~12 per CU) even out the tail but start
            // at different times -- same launch time, FETCH_SIZE x 2 (profiles/r04q_c4_blocks.txt)
            static const int target_blocks2 = [] { const char *e = getenv("DQC_VXC_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
            int nslab = std::max(8, std::min(target_blocks2 / nsplit2, std::max(ngrid / 1024, 512 / nsplit2)) / 8 * 8);
            int slab = (ngrid + nslab - 1) / nslab;
            slab = (slab + 15) / 16 * 16;
            nslab = ((ngrid + slab - 1) / slab + 7) / 8 * 8;
            const size_t shmem2 = sizeof(double) * 2 * WS2_BUF;  // fixed-stride chunk layout, two buffers
            if (LSA > 144 || LSB > 208) { set_error("vxc_ws2: internal layout error"); return DQC_EINVAL; }
            dim3 grid2(nslab * nsplit2);
            int rc = gga ? launch_vxc_ws2<true>(maxt2, nla, nlb, grid2, shmem2, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, NR, NC, LSA, LSB, d_aob, lda)
                         : launch_vxc_ws2<false>(maxt2, nla, nlb, grid2, shmem2, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, NR, NC, LSA, LSB, d_aob, lda);
            if (rc) return rc;
            DQC_CHECK_LAUNCH();
            if (!raw) hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld, nao);
            DQC_CHECK_LAUNCH();
            return DQC_OK;
        }
        // 145 <= nao <= 208 (10 <= T <= 13): one block per slab over the upper-triangular tiles (vxc_wsu_kernel) instead of two
        // blocks that each stage the whole slab.  One operand, symmetric result; DQC_VXC_IMPL=split keeps the two-block form.
        if (ws_shape && T >= 10 && T * (T + 1) / 2 <= 12 * VXC_WAVES && d_aob == d_ao && !(impl_env && impl_env[0] == 's')) {
            int ncu = stream_cus(st);  // one block per CU of the stream's partition (all 256 on an ordinary stream)
            if (vxc_cus_cap() > 0) ncu = std::max(8, std::min(ncu, vxc_cus_cap()));  // (leave CUs to other streams' kernels: host.hip)
            int nslab = ncu;
            int slab = (ngrid + nslab - 1) / nslab;
            slab = (slab + 15) / 16 * 16;
            nslab = (ngrid + slab - 1) / slab;
            const int need = (T * (T + 1) / 2 + VXC_WAVES - 1) / VXC_WAVES;
            const size_t shmem_u = sizeof(double) * 2 * VWS_BUF;
            int rc = 0;
            // GGA: one MFMA on the diagonal tiles (vxc_wsd_kernel); DQC_VXC_IMPL=upper keeps two on every tile (A/B runs)
            if (gga && !(impl_env && impl_env[0] == 'u')) {
                if (T == 13) launch_vxc_wsd<13>(dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, d_w, d_vrho, d_vgrad, slab, lda);
                else if (T == 12) launch_vxc_wsd<12>(dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, d_w, d_vrho, d_vgrad, slab, lda);
                else if (T == 11) launch_vxc_wsd<11>(dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, d_w, d_vrho, d_vgrad, slab, lda);
                else launch_vxc_wsd<10>(dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, d_w, d_vrho, d_vgrad, slab, lda);
            } else {
                rc = gga ? launch_vxc_wsu<true>(need, dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, lda, LS)
                         : launch_vxc_wsu<false>(need, dim3(nslab), shmem_u, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, lda, LS);
            }
            if (rc) return rc;
            DQC_CHECK_LAUNCH();
            if (!raw) hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld, nao);
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
        const int kch = (sizeof(double) * 2 * 2 * 16 * (size_t)LS <= 150 * 1024) ? 16 : 8;
        const int tpr = 512 / kch;  // threads per chunk row
        const int nlneed = (ld / 2 + tpr - 1) / tpr;  // double2 columns per thread
        const int nl = nlneed <= 1 ? 1 : (nlneed <= 2 ? 2 : (nlneed <= 4 ? 4 : 8));
        if (nlneed > 8) { set_error("dqc_grid_vxc: nao above 1008 is not supported by this build"); return DQC_EINVAL; }
        // one 8-wave block per CU; slabs in multiples of 8 so that the XCD-aware decode is exact
        int nslab = std::max(8, (stream_cus(st) / nsplit) / 8 * 8);
        int slab = (ngrid + nslab - 1) / nslab;
        slab = (slab + kch - 1) / kch * kch;
        nslab = ((ngrid + slab - 1) / slab + 7) / 8 * 8;
        const size_t shmem = sizeof(double) * 2 * 2 * kch * LS;
        dim3 grid(nslab * nsplit);
        // default: wave-specialised kernel (8 MFMA waves + 8 producer waves); the producers hold a whole chunk in
        // registers, which bounds ld; DQC_VXC_IMPL=reg selects the unspecialised kernel
        const int tprp = VWS_PROD / kch;
        const int nlpneed = (ld / 2 + tprp - 1) / tprp;
        const int nlp = nlpneed <= 1 ? 1 : (nlpneed <= 2 ? 2 : (nlpneed <= 4 ? 4 : (nlpneed <= 7 ? 7 : 8)));
        if (nlpneed <= 4 && kch == 16 && ws_shape) {
            const size_t shmem_ws = sizeof(double) * 2 * VWS_BUF;  // fixed-stride chunk layout, two buffers
            int rc = gga ? launch_vxc_ws<true>(maxt, nlp, kch, grid, shmem_ws, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob, 0, lda, LS)
                         : launch_vxc_ws<false>(maxt, nlp, kch, grid, shmem_ws, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob, sym ? 1 : 0, lda, LS);
            if (rc) return rc;
            DQC_CHECK_LAUNCH();
            if (!raw) hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld, nao);
            DQC_CHECK_LAUNCH();
            return DQC_OK;
        }
        int rc = gga ? launch_vxc<true>(maxt, nl, kch, grid, shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob, lda, LS)
                     : launch_vxc<false>(maxt, nl, kch, grid, shmem, st, d_vmat, d_ao, ngrid, ld, d_w, d_vrho, d_vgrad, slab, nsplit, tps, d_aob, lda, LS);
        if (rc) return rc;
        DQC_CHECK_LAUNCH();
        if (!raw) hipLaunchKernelGGL(symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, st, d_vmat, ld, nao);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}

int dqc_grid_vxc(double *d_vmat, const double *d_ao, int ncomp, int ngrid, int nao, const double *d_w,
                 const double *d_vrho, const double *d_vgrad, void *stream) {
    return grid_vxc_impl(d_vmat, d_ao, d_ao, ncomp, ngrid, nao, d_w, d_vrho, d_vgrad, stream);
}

int dqc_grid_vxc_raw(double *d_vmat, const double *d_ao, int ncomp, int ngrid, int nao, const double *d_w, const double *d_vrho,
                     const double *d_vgrad, double *h_scale, void *stream) {
    // dqc_grid_vxc without its closing symmetrisation launch: d_vmat <- the raw cross-block sums M (fixed-point integers in
    // deterministic mode; *h_scale <- their scale, 0: plain doubles), V = (M + M^T) / 2 on the first nao rows / columns
    if (h_scale) *h_scale = dqc::deterministic_mode() ? 140737488355328.0 : 0.0;
    return grid_vxc_impl(d_vmat, d_ao, d_ao, ncomp, ngrid, nao, d_w, d_vrho, d_vgrad, stream, true);
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


}  // extern "C"
