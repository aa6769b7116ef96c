// purify.hip -- occupied-space projector of a Fock matrix by trace-correcting purification (TC2), fused per iteration.
//
// The SCF step needs D = n P, P = projector onto the n_occ lowest eigenvectors of F (reference: `diagonalize` +
// `ao_orb2dm`, dqc/qccalc/hf.py:105-113, 227-247; dqc/hamilton/hcgto.py:272-281, done with xitorch.lsymeig).  On MI355X
// rocSOLVER's eigh of a 208 x 208 matrix (4.5 ms) costs more than twice the whole Fock build, so P is obtained from
// GEMMs only (dqc_amd/purify.py states the iteration).  One launch per iteration:
//     X2 = X X  (fp64 MFMA, one 4-wave block per 16 x 16 tile with the K range split over the waves, operands straight from L2 -- X is symmetric, so both fragments
//               are read as rows)
//     X' = done ? X : (tr X > n_occ ? X2 : 2 X - X2)
//     tr X' and max |X2 - X| accumulate into per-iteration slots of a small state array (atomics), which the NEXT
//     launch reads: no host decision anywhere, the whole sequence is hipGraph-capturable.
// `done` for iteration k = some earlier iteration reported max |X2 - X| < tol (the iterate is frozen from then on;
// continuing would let the trace test pick the error-doubling branch at round-off level).
#include <atomic>
#include "common.hpp"

namespace dqc {

typedef double pv4d __attribute__((ext_vector_type(4)));

// state layout: trace[k] = tr X_k, idem[k] = max |X_k^2 - X_k|  (k = 0 .. iters), stored as doubles
// One block of four waves per 16 x 16 tile: the K range is split over the waves (one batch of loads each for ld <= 208:
// a single L2 round trip per iteration instead of four), partial tiles are summed through LDS.  A frozen iterate only
// copies its tile (the launch is then a few microseconds).
// blockIdx.y = molecule of a batch (lockstep SCF of many molecules: dqc_amd/lockstep.py): matrices `xstride` doubles apart,
// state slots `sstride` doubles apart; every molecule freezes on its own.
__global__ __launch_bounds__(256) void purify_tc2_kernel(double *__restrict__ xout, const double *__restrict__ xin, int ld,
                                                         double nocc, double tol, int k, double *__restrict__ trace,
                                                         double *__restrict__ idem, size_t xstride, int sstride, double dsc) {
    // dsc != 0: deterministic mode -- the trace slots hold fixed-point integers (common.hpp: acc_add / det_value)
    __shared__ double red[3][4][64];
    xout += blockIdx.y * xstride;
    xin += blockIdx.y * xstride;
    trace += (size_t)blockIdx.y * sstride;
    idem += (size_t)blockIdx.y * sstride;
    const int T = ld >> 4;
    const int ti = blockIdx.x / T, tj = blockIdx.x % T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    bool done = false;
    for (int j = 0; j < k; j++) done = done || (idem[j] < tol);
    if (done) {  // wave-uniform, block-uniform
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const size_t e = (size_t)(ti * 16 + lk + 4 * r) * ld + tj * 16 + lr;
                xout[e] = xin[e];
            }
        }
        return;
    }
    const double tr = det_value(trace[k], dsc);
    pv4d acc = {0.0, 0.0, 0.0, 0.0};
    // A[i][kk] = X[kk][i] (symmetric): both operands are 4 rows x 128 bytes
    const double *pa = xin + (size_t)lk * ld + ti * 16 + lr;
    const double *pb = xin + (size_t)lk * ld + tj * 16 + lr;
    const int nk = ld >> 2, per = (nk + 3) >> 2;
    const int kbeg = wave * per, kend = min(kbeg + per, nk);
    for (int k0 = kbeg; k0 < kend; k0 += 13) {  // 26 loads in flight per batch
        double a[13], b[13];
#pragma unroll
        for (int q = 0; q < 13; q++) {
            const int kk = min(k0 + q, nk - 1);
            a[q] = pa[(size_t)kk * 4 * ld];
            b[q] = pb[(size_t)kk * 4 * ld];
        }
#pragma unroll
        for (int q = 0; q < 13; q++)
            if (k0 + q < kend) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    // C[row = lk + 4 r][col = lr]
    double tsum = 0.0, emax = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = ti * 16 + lk + 4 * r, col = tj * 16 + lr;
        const double x = xin[(size_t)row * ld + col], x2 = acc[r];
        const double out = tr > nocc ? x2 : 2.0 * x - x2;
        xout[(size_t)row * ld + col] = out;
        emax = fmax(emax, fabs(x2 - x));
        if (row == col) tsum += out;
    }
    for (int o = 32; o > 0; o >>= 1) {
        tsum += __shfl_xor(tsum, o);
        emax = fmax(emax, __shfl_xor(emax, o));
    }
    if (lane == 0) {
        if (ti == tj) acc_add(&trace[k + 1], tsum, dsc);
        // max of non-negative doubles == max of their bit patterns as unsigned integers
        atomicMax(reinterpret_cast<unsigned long long *>(&idem[k]), (unsigned long long)__double_as_longlong(emax));
    }
}

// ---------------------------------------------------------------------------------------------
// The whole TC2 sequence as ONE persistent launch on ONE XCD (round 4).  The per-iteration launches above cost 7-9 us each for
// 18 MFLOP (n = 208) -- 0.45-0.58 ms per SCF step, more than the Fock build of a benzene-size molecule and a fifth of a 20-atom
// one (tools/gpu_one_molecule_iteration.py).  A grid-wide barrier between iterations was tried in round 2 over the whole chip:
// every XCD has its own L2, so the iterate had to be written back / invalidated at each barrier and nothing was gained.  Here the
// WORKERS are the blocks 0, 8, 16, ... of the launch (the others return at once): the dispatcher deals workgroups to the XCDs
// round-robin, so all workers sit on XCD 0 and share ONE L2 -- the iterate lives there, is read with L1-bypassing loads and never
// needs a write-back.  The workers CHECK that (hardware register XCC_ID, OR-ed into a mask before the first barrier); if they do
// not share an XCD (another dispatch mode / partitioning) or a barrier times out, the kernel gives up, the projector's
// idempotency error stays large and the caller takes its fallback -- no wrong result can come out of a wrong assumption.
//   one wave per 16 x 16 tile (8 waves per worker, <= 32 workers: ld <= 256), the full K range per wave, all fragment loads of an
//   iteration issued up front; trace / idempotency through the same atomics and slots as the launch-per-iteration kernel.
// ctl: [0] barrier arrivals (monotone), [1] XCC mask, [2] status (0 ok, 1 workers on several XCDs, 2 barrier time-out)
// ---------------------------------------------------------------------------------------------
DQC_DEV double coh_load(const double *p) {  // agent-scope relaxed: bypasses the CU's L1 (the L2 is the point of coherence)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// agent-scope relaxed store (sc1, write-through).  Measured both ways on MI355X (tools/gpu_projector_stress.py, n = 208): these
// 0.60 ms per projector, PLAIN stores + s_waitcnt vmcnt(0) before the barrier (which keep the line in the L2) 0.68 ms -- the
// phases are bound by their chain of L2 round trips and by the one XCD's L2 bandwidth (9 MB of 8-byte fragment loads per
// phase), not by where the stored line lives
DQC_DEV void coh_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wall_clock64() ticks at 100 MHz on gfx950: 5 ms
constexpr unsigned long long PERSIST_TIMEOUT_TICKS = 500000ull;
DQC_DEV bool persist_barrier(unsigned *ctl, unsigned target) {
    // Every wave first waits for ITS OWN stores and atomics to be performed at the L2 (agent-scope release: s_waitcnt vmcnt(0) --
    // a workgroup-scope fence only orders LDS/L1 traffic and left tile stores in flight when the arrival was published), then the
    // block barrier, then one thread publishes the arrival (release) and polls (acquire) until every worker has arrived.  The
    // spin is bounded by wall clock (s_memrealtime, 100 MHz): a few milliseconds, then the kernel gives up and the caller
    // takes its eigh fallback.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __shared__ int ok_;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            if (__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            if (__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // somebody gave up
            if (wall_clock64() - t0 > PERSIST_TIMEOUT_TICKS) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(&ctl[2], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_ = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok_ != 0;
}

constexpr int PST_WAVES = 8;  // waves (tiles) per worker block
__global__ __launch_bounds__(64 * PST_WAVES) void purify_tc2_persist_kernel(double *__restrict__ x0, double *__restrict__ x1, int ld,
                                                                           double nocc, double tol, int iters,
                                                                           double *__restrict__ trace, double *__restrict__ idem,
                                                                           unsigned *__restrict__ ctl, int nworker, double dsc, int xsel) {
    if ((int)(blockIdx.x & 7) != xsel) return;  // workers = blocks xsel, xsel + 8, ...: one XCD under the round-robin dispatch
    const int w = blockIdx.x >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int T = ld >> 4, tile = w * PST_WAVES + wave;
    const bool has = tile < T * T;
    const int ti = has ? tile / T : 0, tj = has ? tile % T : 0;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(&ctl[1], 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned phase = 1;
    if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
    {
        const unsigned mask = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__popc(mask) != 1) {  // the workers do not share an L2: every worker sees the same mask and leaves
            if (w == 0 && threadIdx.x == 0) __hip_atomic_store(&ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    double *cur = x0, *nxt = x1;
    const int nk = ld >> 2;  // k-groups of 4
    for (int k = 0; k < iters; k++) {
        // frozen once an earlier iteration reported |X^2 - X| < tol (every worker reads the same slots: uniform decision)
        if (k > 0 && coh_load(&idem[k - 1]) < tol) break;
        const double tr = det_value(coh_load(&trace[k]), dsc);
        if (has) {
            pv4d acc = {0.0, 0.0, 0.0, 0.0};
            // A[i][kk] = X[kk][i] (symmetric iterate): both operands are 4 rows x 128 bytes
            const double *pa = cur + (size_t)lk * ld + ti * 16 + lr;
            const double *pb = cur + (size_t)lk * ld + tj * 16 + lr;
            for (int k0 = 0; k0 < nk; k0 += 16) {  // 32 loads in flight per batch
                double a[16], b[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int kk = min(k0 + q, nk - 1);
                    a[q] = coh_load(pa + (size_t)kk * 4 * ld);
                    b[q] = coh_load(pb + (size_t)kk * 4 * ld);
                }
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (k0 + q < nk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
            }
            double tsum = 0.0, emax = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = ti * 16 + lk + 4 * r, col = tj * 16 + lr;
                const double x = coh_load(&cur[(size_t)row * ld + col]), x2 = acc[r];
                const double out = tr > nocc ? x2 : 2.0 * x - x2;
                coh_store(&nxt[(size_t)row * ld + col], out);
                emax = fmax(emax, fabs(x2 - x));
                if (row == col) tsum += out;
            }
            for (int o = 32; o > 0; o >>= 1) {
                tsum += __shfl_xor(tsum, o);
                emax = fmax(emax, __shfl_xor(emax, o));
            }
            if (lane == 0) {
                if (ti == tj) acc_add(&trace[k + 1], tsum, dsc);
                atomicMax(reinterpret_cast<unsigned long long *>(&idem[k]), (unsigned long long)__double_as_longlong(emax));
            }
        }
        phase++;
        if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
        double *t_ = cur; cur = nxt; nxt = t_;
    }
    if (cur != x0 && has) {  // the result belongs in x0 (every wave copies the tile it owns)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t e = (size_t)(ti * 16 + lk + 4 * r) * ld + tj * 16 + lr;
            x0[e] = coh_load(&cur[e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// F -> P in ONE launch: the persistent kernel above with the steps around it folded in (the torch ops they were cost 5-6 us of
// launch latency each inside the SCF-iteration graph: ~30 nodes around 0.2 ms of purification):
//     Gershgorin bounds (row radii through atomics, then every wave reduces the n bounds itself),  X0 = (e_max I - F) / (e_max - e_min)
//     TC2 iterations until frozen
//     two McWeeny steps  X <- 3 X^2 - 2 X^3   (two tile GEMMs each, a barrier between them)
//     P = (X + X^T) / 2,   err = max |P^2 - P| + |tr P - n_occ|
// fock, pout: (n, n) contiguous; bufs: 3 ld^2 doubles (two iterates + the X^2 of the McWeeny steps); rad: ld doubles (zeroed by the
// host); fin: 4 doubles (zeroed): max |P^2 - P| bits, tr P, err, ran-to-the-end flag.  Same ctl / give-up protocol as purify_tc2_persist_kernel; on a
// give-up the flag stays 0 and projector_err_kernel reports a large error.
// ---------------------------------------------------------------------------------------------
DQC_DEV pv4d persist_tile_gemm(const double *__restrict__ a_rows, const double *__restrict__ b_rows, int ld, int ti, int tj, int lane,
                               int kpart = 0, int ksplit = 1) {
    // tile (ti, tj) of A B with A given through its TRANSPOSE rows (A symmetric: a_rows == A) -- both operands 4 rows x 128 bytes.
    // kpart / ksplit: this wave's share of the K range (split-K over the waves of a workgroup, summed by tile_reduce)
    const int lr = lane & 15, lk = lane >> 4, nk = ld >> 2;
    const int per = (nk + ksplit - 1) / ksplit, kbeg = kpart * per, kend = min(nk, kbeg + per);
    pv4d acc = {0.0, 0.0, 0.0, 0.0};
    const double *pa = a_rows + (size_t)lk * ld + ti * 16 + lr;
    const double *pb = b_rows + (size_t)lk * ld + tj * 16 + lr;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {  // 32 loads in flight per batch (64 per batch measured slower: 0.66 vs 0.60 ms at n = 208)
        double a[16], b[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int kk = min(k0 + q, nk - 1);
            a[q] = coh_load(pa + (size_t)kk * 4 * ld);
            b[q] = coh_load(pb + (size_t)kk * 4 * ld);
        }
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (k0 + q < kend) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    }
    return acc;
}

// sum of the KS partial tiles of a workgroup's waves (wave = KS * local tile + kpart) into the kpart == 0 wave, through LDS, in a
// fixed order (deterministic).  Called by EVERY wave of the block (two barriers).
template <int KS>
DQC_DEV pv4d tile_reduce(pv4d acc, double (*sred)[256], int wave, int lane) {
    if constexpr (KS == 1) return acc;
    const int kpart = wave % KS;
    if (kpart) {
#pragma unroll
        for (int r = 0; r < 4; r++) sred[wave][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (!kpart) {
#pragma unroll
        for (int p = 1; p < KS; p++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] += sred[wave + p][r * 64 + lane];
    }
    __syncthreads();
    return acc;
}

// KS: split-K factor.  A tile GEMM of one wave is a chain of dependent L2 round trips (ld / 4 fragment pairs, 32 in flight) and the
// kernel is a chain of ~35 such GEMMs with a grid barrier each: KS waves per tile shorten every link (n = 114: 32 -> 16 fragment
// pairs per wave, one batch) at the price of an LDS reduction -- pays for mid-sized matrices only (see dqc_projector_tc2).  gemm = has (this wave multiplies), own = the wave that owns the
// tile's epilogue (kpart == 0).
template <int KS>
__global__ __launch_bounds__(64 * PST_WAVES) void projector_persist_kernel(double *__restrict__ pout, const double *__restrict__ fock, int n,
                                                                          int ld, double nocc, double tol, int iters,
                                                                          double *__restrict__ bufs, double *__restrict__ rad,
                                                                          double *__restrict__ trace, double *__restrict__ idem,
                                                                          double *__restrict__ fin, unsigned *__restrict__ ctl,
                                                                          int nworker, double dsc, int xsel) {
    if ((int)(blockIdx.x & 7) != xsel) return;
    __shared__ double sred[KS == 1 ? 1 : PST_WAVES][256];
    const int w = blockIdx.x >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int kpart = wave % KS;
    const int T = ld >> 4, tile = w * (PST_WAVES / KS) + wave / KS;
    const bool gemm = tile < T * T;       // this wave multiplies its share of tile `tile`
    const int ti = gemm ? tile / T : 0, tj = gemm ? tile % T : 0;
    const bool has = gemm && kpart == 0;  // ... and owns the tile's element-wise work
    const size_t n2 = (size_t)ld * ld;
    double *cur = bufs, *nxt = bufs + n2, *yb = bufs + 2 * n2;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(&ctl[1], 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- Gershgorin radii: rad_i = sum_{j != i} |F_ij|
    if (has) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = ti * 16 + lk + 4 * r, col = tj * 16 + lr;
            double a = (row < n && col < n && row != col) ? fabs(fock[(size_t)row * n + col]) : 0.0;
            a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
            if (lr == 0 && row < n) acc_add(&rad[row], a, dsc);
        }
    }
    unsigned phase = 1;
    if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
    {
        const unsigned mask = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__popc(mask) != 1) {
            if (w == 0 && threadIdx.x == 0) __hip_atomic_store(&ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    // ---- bounds (every wave for itself: n values), X0 and its trace
    double emin = 1e300, emax = -1e300, trf = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double d = fock[(size_t)i * n + i], rr = det_value(coh_load(&rad[i]), dsc);
        emin = fmin(emin, d - rr);
        emax = fmax(emax, d + rr);
        trf += d;
    }
    for (int o = 32; o > 0; o >>= 1) {
        emin = fmin(emin, __shfl_xor(emin, o));
        emax = fmax(emax, __shfl_xor(emax, o));
        trf += __shfl_xor(trf, o);
    }
    const double isc = 1.0 / (emax - emin);
    if (has) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = ti * 16 + lk + 4 * r, col = tj * 16 + lr;
            const double v = (row < n && col < n) ? ((row == col ? emax : 0.0) - fock[(size_t)row * n + col]) * isc : 0.0;
            coh_store(&cur[(size_t)row * ld + col], v);
        }
    }
    if (w == 0 && threadIdx.x == 0) {
        const double tr0 = ((double)n * emax - trf) * isc;
        coh_store(&trace[0], dsc == 0.0 ? tr0 : __longlong_as_double(__double2ll_rn(tr0 * dsc)));
    }
    phase++;
    if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
    // ---- TC2
    for (int k = 0; k < iters; k++) {
        // the two scalars this iteration decides on and the tile's own elements are requested TOGETHER with the fragments (one L2
        // round trip instead of three dependent ones in front of the GEMM); a frozen iterate costs one wasted tile GEMM
        const double idprev = k > 0 ? coh_load(&idem[k - 1]) : 1e300;
        const double trraw = coh_load(&trace[k]);
        double xel[4] = {0.0, 0.0, 0.0, 0.0};
        pv4d acc = {0.0, 0.0, 0.0, 0.0};
        if (has) {
#pragma unroll
            for (int r = 0; r < 4; r++) xel[r] = coh_load(&cur[(size_t)(ti * 16 + lk + 4 * r) * ld + tj * 16 + lr]);
        }
        if (gemm) acc = persist_tile_gemm(cur, cur, ld, ti, tj, lane, kpart, KS);
        if (idprev < tol) break;
        acc = tile_reduce<KS>(acc, sred, wave, lane);
        const double tr = det_value(trraw, dsc);
        if (has) {
            double tsum = 0.0, em = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = ti * 16 + lk + 4 * r, col = tj * 16 + lr;
                const double x = xel[r], x2 = acc[r];
                const double out = tr > nocc ? x2 : 2.0 * x - x2;
                coh_store(&nxt[(size_t)row * ld + col], out);
                em = fmax(em, fabs(x2 - x));
                if (row == col) tsum += out;
            }
            for (int o = 32; o > 0; o >>= 1) {
                tsum += __shfl_xor(tsum, o);
                em = fmax(em, __shfl_xor(em, o));
            }
            if (lane == 0) {
                if (ti == tj) acc_add(&trace[k + 1], tsum, dsc);
                atomicMax(reinterpret_cast<unsigned long long *>(&idem[k]), (unsigned long long)__double_as_longlong(em));
            }
        }
        phase++;
        if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
        double *t_ = cur; cur = nxt; nxt = t_;
    }
    // ---- two McWeeny steps X <- 3 X^2 - 2 X^3 (contracting at both 0 and 1)
    for (int mw = 0; mw < 2; mw++) {
        {
            pv4d y = {0.0, 0.0, 0.0, 0.0};
            if (gemm) y = persist_tile_gemm(cur, cur, ld, ti, tj, lane, kpart, KS);
            y = tile_reduce<KS>(y, sred, wave, lane);
            if (has) {
#pragma unroll
                for (int r = 0; r < 4; r++) coh_store(&yb[(size_t)(ti * 16 + lk + 4 * r) * ld + tj * 16 + lr], y[r]);
            }
        }
        phase++;
        if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
        {
            pv4d z = {0.0, 0.0, 0.0, 0.0};
            // (X^2 symmetric: read through its rows)
            if (gemm) z = persist_tile_gemm(yb, cur, ld, ti, tj, lane, kpart, KS);
            z = tile_reduce<KS>(z, sred, wave, lane);
            if (has) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const size_t e = (size_t)(ti * 16 + lk + 4 * r) * ld + tj * 16 + lr;
                    coh_store(&nxt[e], 3.0 * coh_load(&yb[e]) - 2.0 * z[r]);
                }
            }
        }
        phase++;
        if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
        double *t_ = cur; cur = nxt; nxt = t_;
    }
    // ---- P = (X + X^T) / 2 (to the caller's (n, n) array and, padded, to `nxt` for the error), tr P
    if (has) {
        double tsum = 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = ti * 16 + lk + 4 * r, col = tj * 16 + lr;
            const double pv = 0.5 * (coh_load(&cur[(size_t)row * ld + col]) + coh_load(&cur[(size_t)col * ld + row]));
            coh_store(&nxt[(size_t)row * ld + col], pv);
            if (row < n && col < n) pout[(size_t)row * n + col] = pv;
            if (row == col) tsum += pv;
        }
        if (ti == tj) {
            for (int o = 32; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o);
            if (lane == 0) atomicAdd(&fin[1], tsum);
        }
    }
    phase++;
    if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
    pv4d q = {0.0, 0.0, 0.0, 0.0};
    if (gemm) q = persist_tile_gemm(nxt, nxt, ld, ti, tj, lane, kpart, KS);
    q = tile_reduce<KS>(q, sred, wave, lane);
    if (has) {
        double em = 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) em = fmax(em, fabs(q[r] - coh_load(&nxt[(size_t)(ti * 16 + lk + 4 * r) * ld + tj * 16 + lr])));
        for (int o = 32; o > 0; o >>= 1) em = fmax(em, __shfl_xor(em, o));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(&fin[0]), (unsigned long long)__double_as_longlong(em));
    }
    phase++;
    if (!persist_barrier(ctl, (unsigned)nworker * phase)) return;
    if (w == 0 && threadIdx.x == 0) {
        fin[2] = coh_load(&fin[0]) + fabs(coh_load(&fin[1]) - nocc);
        fin[3] = 1.0;  // ran to its end
    }
}

__global__ void projector_init_kernel(double *__restrict__ z, int nz) {
    for (int i = threadIdx.x; i < nz; i += blockDim.x) z[i] = 0.0;
}
__global__ void projector_init_u32_kernel(unsigned *__restrict__ z, int nz) {
    for (int i = threadIdx.x; i < nz; i += blockDim.x) z[i] = 0u;
}

// d_err <- the persistent kernel's error when it ran to its end (ctl[2] == 0 and the closing phase was reached), else a LARGE value
__global__ void projector_err_kernel(double *__restrict__ err, const double *__restrict__ fin, const unsigned *__restrict__ ctl) {
    if (threadIdx.x == 0) err[0] = (ctl[2] == 0u && fin[3] == 1.0) ? fin[2] : 1e300;
}

__global__ void purify_trace_kernel(const double *__restrict__ x, int ld, double *__restrict__ trace0, size_t xstride,
                                    int sstride, double dsc) {
    x += blockIdx.x * xstride;
    trace0 += (size_t)blockIdx.x * sstride;
    double s = 0.0;
    for (int i = threadIdx.x; i < ld; i += blockDim.x) s += x[(size_t)i * ld + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tr = part[0] + part[1] + part[2] + part[3];
        trace0[0] = dsc == 0.0 ? tr : __longlong_as_double(__double2ll_rn(tr * dsc));
    }
}

// Orthonormal basis of the range of a projector from GEMMs + one small launch: Y = P Omega (n x r, Omega a fixed random matrix,
// r = rank P) spans range(P); with G = Y^T Y = C C^T (Cholesky), Q = Y C^-T has orthonormal columns and Q Q^T = P.  The SCF
// step uses it to hand the purified density matrix to the Hamiltonian in FACTOR form (ao_orb2dm(Q, n): hcgto.py:272-281), so
// that the grid pass takes the rank-r density kernel.  Single-wave blocks (barriers cost nothing): every block factors G in
// its own LDS (left-looking, lane i owns row i) and forward-substitutes `rb` rows of Y, one per lane, also in LDS.
// A G that is not positive definite (purification failed) gives NaNs, which the caller's idempotency check turns into the
// eigh fallback.
__global__ __launch_bounds__(64) void orth_factor_kernel(double *__restrict__ q, const double *__restrict__ y,
                                                         const double *__restrict__ g, int n, int r, int rb) {
    extern __shared__ double sm[];
    q += (size_t)blockIdx.y * n * r;  // blockIdx.y = molecule of a batch
    y += (size_t)blockIdx.y * n * r;
    g += (size_t)blockIdx.y * r * r;
    const int ldc = r | 1, t = threadIdx.x;  // odd row stride: the rows of different lanes start in different banks
    double *c = sm, *rw = sm + (size_t)r * ldc;
    const int row0 = blockIdx.x * rb;
    for (int e = t; e < r * r; e += 64) c[(e / r) * ldc + e % r] = g[e];
    for (int e = t; e < rb * r; e += 64) {
        const int rr = e / r, k = e % r;
        rw[rr * ldc + k] = row0 + rr < n ? y[(size_t)(row0 + rr) * r + k] : 0.0;
    }
    __syncthreads();
    // dot(a[0:k], b[0:k]) with 16 LDS reads in flight
    auto dotk = [](const double *a, const double *b, int k) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int j = 0;
        for (; j + 7 < k; j += 8) {
            const double x0 = a[j], x1 = a[j + 1], x2 = a[j + 2], x3 = a[j + 3], x4 = a[j + 4], x5 = a[j + 5], x6 = a[j + 6], x7 = a[j + 7];
            const double y0 = b[j], y1 = b[j + 1], y2 = b[j + 2], y3 = b[j + 3], y4 = b[j + 4], y5 = b[j + 5], y6 = b[j + 6], y7 = b[j + 7];
            a0 += x0 * y0; a1 += x1 * y1; a2 += x2 * y2; a3 += x3 * y3;
            a0 += x4 * y4; a1 += x5 * y5; a2 += x6 * y6; a3 += x7 * y7;
        }
        for (; j < k; j++) a0 += a[j] * b[j];
        return (a0 + a1) + (a2 + a3);
    };
    // step k: column k of C (left-looking: finished columns only; lane i owns row i), then -- row k of C being final -- entry
    // k of every row of Q (Q[row] C^T = Y[row], one row per lane); the two dot products of a step overlap
    for (int k = 0; k < r; k++) {
        const double *ck = c + k * ldc;
        for (int i = k + t; i < r; i += 64) c[i * ldc + k] -= dotk(c + i * ldc, ck, k);
        __syncthreads();
        const double dinv = 1.0 / sqrt(ck[k]);
        __syncthreads();
        for (int i = k + t; i < r; i += 64) c[i * ldc + k] = i == k ? 1.0 / dinv : c[i * ldc + k] * dinv;
        for (int rr = t; rr < rb; rr += 64) {  // row k of C left of the diagonal is final since step k - 1
            double *w = rw + rr * ldc;
            w[k] = (w[k] - dotk(w, ck, k)) * dinv;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int e = t; e < rb * r; e += 64) {
        const int rr = e / r, k = e % r;
        if (row0 + rr < n) q[(size_t)(row0 + rr) * r + k] = rw[rr * ldc + k];
    }
}

// Pulay coefficients of a batch of DIIS problems, one 64-lane block per molecule.  gram (nmol, H, H): error-vector scalar
// products, the first m slots valid.  Solves  [[G/g, -1], [-1^T, 0]] (c, lambda) = (0, -1)  (g = largest diagonal of G: the
// coefficients do not depend on the scale, the conditioning does) in the least-squares / minimum-norm sense of
// numpy.linalg.lstsq(rcond=None), which is what the one-molecule driver does on the host (dqc_amd/qccalc.py): symmetric
// eigendecomposition by cyclic Jacobi rotations in LDS, eigenvalues below eps N |lambda|_max dropped.  Reference:
// the fixed-point solver of dqc/qccalc/scf_qccalc.py:109-113 (any convergent mixer has the same fixed point).
#define DQC_DIIS_NMAX 17
__global__ __launch_bounds__(64) void diis_solve_kernel(double *__restrict__ cout, const double *__restrict__ gram, int H, int m,
                                                        const long long *__restrict__ d_count) {
    __shared__ double A[DQC_DIIS_NMAX][DQC_DIIS_NMAX + 1], V[DQC_DIIS_NMAX][DQC_DIIS_NMAX + 1];
    // d_count: the number of valid slots is min(*d_count, H), read on the DEVICE (a step counter of an SCF loop that replays as
    // one hipGraph: the host never learns the iteration number); NULL: m as passed
    if (d_count != nullptr) m = (int)(*d_count < (long long)H ? (*d_count < 1 ? 1 : *d_count) : H);
    const int t = threadIdx.x, N = m + 1;
    gram += (size_t)blockIdx.x * H * H;
    cout += (size_t)blockIdx.x * H;
    double g = 0.0;
    for (int i = 0; i < m; i++) g = fmax(g, gram[i * H + i]);
    g = fmax(g, 1e-300);
    for (int e = t; e < N * N; e += 64) {
        const int i = e / N, j = e % N;
        double a;
        if (i < m && j < m) a = gram[i * H + j] / g;
        else if (i == m && j == m) a = 0.0;
        else a = -1.0;
        A[i][j] = a;
        V[i][j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    // Cyclic Jacobi in ROUND-ROBIN order (round 4): the N (N - 1) / 2 rotations of a sweep are done as Ne - 1 steps of Ne / 2 DISJOINT
    // pairs (the chess-tournament schedule on Ne = N rounded up to even; a pair that holds the padding index is skipped), each
    // step = angles by the first Ne / 2 lanes, then the row pairs, then the column pairs (+ eigenvectors) by all 64 lanes.  The
    // one-rotation-at-a-time form (a serial chain of ~550 rotations with five barriers each) took 0.33 ms for 12 stored vectors
    // -- invisible in the lockstep driver, a sixth of an iteration of the device-resident one-molecule loop (dqc_amd/devscf.py).
    __shared__ double cs[DQC_DIIS_NMAX / 2 + 1][2];
    __shared__ int pq[DQC_DIIS_NMAX / 2 + 1][2];
    const int Ne = N + (N & 1), npair = Ne / 2;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, dia = 0.0;  // every lane sums the whole matrix: uniform decision without a reduction
        for (int i = 0; i < N; i++) {
            dia += A[i][i] * A[i][i];
            for (int j = i + 1; j < N; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-34 * dia) break;  // quadratic convergence: the next sweep would not change a digit
        for (int step = 0; step < Ne - 1; step++) {
            if (t < npair) {
                // player Ne - 1 stays, the others rotate: pair 0 = (Ne - 1, step), pair k = (step + k, step - k) mod (Ne - 1)
                int p = t == 0 ? Ne - 1 : (step + t) % (Ne - 1), q = t == 0 ? step : (step - t + (Ne - 1)) % (Ne - 1);
                if (p > q) { const int x = p; p = q; q = x; }
                double c = 1.0, sn = 0.0;
                if (q < N) {
                    const double apq = A[p][q];
                    if (fabs(apq) > 1e-300) {
                        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        sn = tt * c;
                    }
                } else {
                    q = -1;  // the padding index: no rotation
                }
                cs[t][0] = c; cs[t][1] = sn;
                pq[t][0] = p; pq[t][1] = q;
            }
            __syncthreads();
            // A <- A J (columns p, q of every pair, all rows) and V <- V J; the pairs are disjoint, so no two items touch one element
            for (int e = t; e < npair * N; e += 64) {
                const int k = e / N, r = e - k * N, p = pq[k][0], q = pq[k][1];
                if (q < 0) continue;
                const double c = cs[k][0], sn = cs[k][1];
                const double akp = A[r][p], akq = A[r][q], vkp = V[r][p], vkq = V[r][q];
                A[r][p] = c * akp - sn * akq;
                A[r][q] = sn * akp + c * akq;
                V[r][p] = c * vkp - sn * vkq;
                V[r][q] = sn * vkp + c * vkq;
            }
            __syncthreads();
            // A <- J^T A (rows p, q of every pair, all columns)
            for (int e = t; e < npair * N; e += 64) {
                const int k = e / N, r = e - k * N, p = pq[k][0], q = pq[k][1];
                if (q < 0) continue;
                const double c = cs[k][0], sn = cs[k][1];
                const double akp = A[p][r], akq = A[q][r];
                A[p][r] = c * akp - sn * akq;
                A[q][r] = sn * akp + c * akq;
            }
            __syncthreads();
        }
    }
    double lmax = 0.0;
    for (int i = 0; i < N; i++) lmax = fmax(lmax, fabs(A[i][i]));
    const double cut = 2.220446049250313e-16 * N * lmax;
    if (t < H) {
        double x = 0.0;
        if (t < m)
            for (int i = 0; i < N; i++) {
                const double lam = A[i][i];
                if (fabs(lam) > cut) x += V[t][i] * (-V[m][i]) / lam;  // rhs = (0, ..., 0, -1)
            }
        cout[t] = x;
    }
}

}  // namespace dqc

// which XCD the next persistent projector runs on: round robin over the calls, so that the projectors of several molecules in flight
// (batch.run_concurrent, one hipGraph per molecule: the choice is baked into the graph at capture) spread over the eight XCDs instead
// of queueing on one.  Measured (tools/gpu_concurrent_projector_check.py, 16 C5 molecules, 8 / 16 in flight): 434 / 452 iterations/s
// against 431 / 441 with every projector on XCD 0 -- within noise, the drivers are not bound there; kept because it costs nothing and
// every one of the eight placements passes the kernel's own XCC check.  DQC_PROJECTOR_XCD = 0 ... 7 pins it (A/B runs).
static int persist_next_xcd() {
    static const int pin = [] { const char *e = getenv("DQC_PROJECTOR_XCD"); return e ? atoi(e) : -1; }();
    static std::atomic<unsigned> next{0};
    if (pin >= 0 && pin < 8) return pin;
    return (int)(next.fetch_add(1u, std::memory_order_relaxed) & 7u);
}

extern "C" int dqc_diis_solve(double *d_c, const double *d_gram, int nmol, int nhist, int m, void *stream) {
    // d_gram (nmol, nhist, nhist) Gram matrices of the stored error vectors, the first m slots valid; d_c (nmol, nhist) Pulay
    // coefficients (zero for the unused slots).  Enqueues only.
    using namespace dqc;
    if (nmol <= 0) return DQC_OK;
    if (m < 1 || m > nhist || nhist + 1 > DQC_DIIS_NMAX) { set_error("dqc_diis_solve: need 1 <= m <= nhist <= 16"); return DQC_EINVAL; }
    hipLaunchKernelGGL(diis_solve_kernel, dim3(nmol), dim3(64), 0, (hipStream_t)stream, d_c, d_gram, nhist, m, (const long long *)nullptr);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_diis_solve_dev(double *d_c, const double *d_gram, int nmol, int nhist, const long long *d_count, void *stream) {
    // the same with the number of valid slots min(*d_count, nhist) read from device memory (hipGraph-resident SCF loops)
    using namespace dqc;
    if (nmol <= 0) return DQC_OK;
    if (nhist < 1 || nhist + 1 > DQC_DIIS_NMAX || d_count == nullptr) { set_error("dqc_diis_solve_dev: need 1 <= nhist <= 16 and a device counter"); return DQC_EINVAL; }
    hipLaunchKernelGGL(diis_solve_kernel, dim3(nmol), dim3(64), 0, (hipStream_t)stream, d_c, d_gram, nhist, nhist, d_count);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_purify_tc2_batched(double *d_x, double *d_tmp, int ld, int nmol, double nocc, int iters, double tol,
                                      double *d_state, void *stream);

extern "C" int dqc_projector_tc2(double *d_p, double *d_err, const double *d_fock, int n, double nocc, int iters, double tol,
                                double *d_work, void *stream) {
    // P (n, n) <- projector onto the nocc lowest eigenvectors of the symmetric F (n, n), d_err[0] <- max |P^2 - P| + |tr P - nocc|
    // (large when the purification did not converge or the kernel gave up: the caller then falls back to an eigensolver).
    // n <= 256.  d_work: dqc_projector_work_doubles(n) doubles.  ONE persistent launch (projector_persist_kernel).  Enqueues only.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0 || n > 256) { set_error("dqc_projector_tc2: 1 <= n <= 256"); return DQC_EINVAL; }
    if (iters < 1) { set_error("dqc_projector_tc2: iters must be >= 1"); return DQC_EINVAL; }
    const int ld = (n + 15) / 16 * 16, T = ld / 16;
    // split-K (two waves per tile) for 80 < n <= 160, where a tile GEMM is a short chain of L2 round trips and more waves shorten it
    // (n = 114: 0.340 -> 0.284 ms); at n = 208 the XCD's L2 bandwidth is the bound and more waves only add contention (0.55 -> 0.63 ms;
    // four waves per tile: 1.3 ms), at n = 24 the LDS reduction costs more than the chain (0.13 -> 0.21 ms).
    // DQC_PROJECTOR_KSPLIT = 1 | 2 overrides (A/B runs)
    static const int ks_env = [] { const char *e = getenv("DQC_PROJECTOR_KSPLIT"); return e ? atoi(e) : 0; }();
    int ks = (T >= 6 && T <= 10) ? 2 : 1;
    if (ks_env == 1 || ks_env == 2) ks = ks_env;
    // (tried for n = 208 / 250, where a GEMM step takes 16 us: the block's eight waves sharing their tile row's A panel through LDS --
    // 44 % less L2 traffic, 0.555 -> 0.620 ms; so neither the L2 bandwidth nor the length of the load chain alone is the bound there)
    const int nworker = (T * T * ks + PST_WAVES - 1) / PST_WAVES;
    const int xsel = persist_next_xcd();
    const size_t n2 = (size_t)ld * ld;
    double *bufs = d_work, *rad = bufs + 3 * n2, *trace = rad + ld, *idem = trace + (iters + 2), *fin = idem + (iters + 2);
    unsigned *ctl = (unsigned *)(fin + 4);
    // rad, trace, idem, fin, ctl start at zero -- by a KERNEL, not hipMemsetAsync.  With a memset node here, the SECOND use of a
    // captured SCF iteration (dqc_amd/devscf.py: replays issued back to back, a rerun of the same object) reported err = 1e300
    // on perfectly good projectors in 6 of 6 reruns, and in none of 8 with this kernel; the same replays with a host sync between
    // them were fine either way.  tools/ubench/graph_memset_order.hip (memset node -> slow kernel -> check, 300 launches back to
    // back) does NOT reproduce a mis-ordered memset node, so the mechanism is not pinned down -- the control words of a kernel that
    // spins on them through L2 are simply not left to a copy-engine / blit path.
    hipLaunchKernelGGL(projector_init_kernel, dim3(1), dim3(256), 0, st, rad, ld + 2 * (iters + 2) + 4 + 2);
    DQC_CHECK_LAUNCH();
    const double dsc = deterministic_mode() ? 70368744177664.0 : 0.0;
    if (ks == 2)
        hipLaunchKernelGGL(projector_persist_kernel<2>, dim3(8 * nworker), dim3(64 * PST_WAVES), 0, st, d_p, d_fock, n, ld, nocc, tol, iters,
                           bufs, rad, trace, idem, fin, ctl, nworker, dsc, xsel);
    else
        hipLaunchKernelGGL(projector_persist_kernel<1>, dim3(8 * nworker), dim3(64 * PST_WAVES), 0, st, d_p, d_fock, n, ld, nocc, tol, iters,
                           bufs, rad, trace, idem, fin, ctl, nworker, dsc, xsel);
    DQC_CHECK_LAUNCH();
    hipLaunchKernelGGL(projector_err_kernel, dim3(1), dim3(64), 0, st, d_err, fin, ctl);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" size_t dqc_projector_work_doubles(int n, int iters) {
    const size_t ld = (size_t)(n + 15) / 16 * 16;
    return 3 * ld * ld + ld + 2 * (size_t)(iters + 2) + 4 + 2;
}

extern "C" int dqc_purify_tc2_persist(double *d_x, double *d_tmp, int ld, double nocc, int iters, double tol, double *d_state,
                                      unsigned *d_ctl, void *stream) {
    // dqc_purify_tc2 as ONE persistent launch on one XCD (purify_tc2_persist_kernel); d_ctl: 4 unsigned ints of scratch
    // (zeroed here).  ld <= 256.  When the kernel gives up (d_ctl[2] != 0: its workers did not share an XCD, or a barrier timed
    // out) d_x is NOT a projector -- the caller's idempotency check sees that, as for a purification that did not converge.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ld <= 0 || (ld & 15) || ld > 256) { set_error("dqc_purify_tc2_persist: ld must be a multiple of 16, <= 256"); return DQC_EINVAL; }
    if (iters < 1) { set_error("dqc_purify_tc2_persist: iters must be >= 1"); return DQC_EINVAL; }
    const int T = ld / 16, nworker = (T * T + PST_WAVES - 1) / PST_WAVES;
    double *trace = d_state, *idem = d_state + (iters + 2);
    // (zeroed by kernels, not memset nodes: see dqc_projector_tc2)
    hipLaunchKernelGGL(projector_init_kernel, dim3(1), dim3(256), 0, st, d_state, 2 * (iters + 2));
    hipLaunchKernelGGL(projector_init_u32_kernel, dim3(1), dim3(64), 0, st, d_ctl, 4);
    const double dsc = deterministic_mode() ? 70368744177664.0 : 0.0;
    hipLaunchKernelGGL(purify_trace_kernel, dim3(1), dim3(256), 0, st, d_x, ld, trace, (size_t)ld * ld, 2 * (iters + 2), dsc);
    DQC_CHECK_LAUNCH();
    const int xsel = persist_next_xcd();
    hipLaunchKernelGGL(purify_tc2_persist_kernel, dim3(8 * nworker), dim3(64 * PST_WAVES), 0, st, d_x, d_tmp, ld, nocc, tol, iters,
                       trace, idem, d_ctl, nworker, dsc, xsel);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_purify_tc2(double *d_x, double *d_tmp, int ld, double nocc, int iters, double tol, double *d_state,
                              void *stream) {
    // d_x (ld, ld): X0 on entry (spectrum in [0, 1], zero padded to ld = multiple of 16), the purified projector on
    // return; d_tmp (ld, ld) scratch; d_state: 2 * (iters + 2) doubles -- trace[0..iters], idem[0..iters]
    // (idem[k] = max |X_k^2 - X_k|, so the caller can tell convergence: min over k).  Enqueues only.
    return dqc_purify_tc2_batched(d_x, d_tmp, ld, 1, nocc, iters, tol, d_state, stream);
}

extern "C" int dqc_purify_tc2_batched(double *d_x, double *d_tmp, int ld, int nmol, double nocc, int iters, double tol,
                                      double *d_state, void *stream) {
    // the same for nmol matrices at once (one launch per iteration for the whole batch, blockIdx.y = molecule):
    // d_x, d_tmp (nmol, ld, ld); d_state (nmol, 2 * (iters + 2)).  Every matrix freezes on its own.  Enqueues only.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ld <= 0 || (ld & 15)) { set_error("dqc_purify_tc2: ld must be a positive multiple of 16"); return DQC_EINVAL; }
    if (iters < 1) { set_error("dqc_purify_tc2: iters must be >= 1"); return DQC_EINVAL; }
    if (nmol < 1 || nmol > 65535) { set_error("dqc_purify_tc2_batched: 1 <= nmol <= 65535"); return DQC_EINVAL; }
    const int sstride = 2 * (iters + 2);
    const size_t xstride = (size_t)ld * ld;
    double *trace = d_state, *idem = d_state + (iters + 2);
    DQC_HIP(hipMemsetAsync(d_state, 0, sizeof(double) * (size_t)sstride * nmol, st));
    // deterministic mode: traces (0 <= tr <= ld < 2^15) as fixed-point integers with 46 fractional bits
    const double dsc = deterministic_mode() ? 70368744177664.0 : 0.0;
    hipLaunchKernelGGL(purify_trace_kernel, dim3(nmol), dim3(256), 0, st, d_x, ld, trace, xstride, sstride, dsc);
    DQC_CHECK_LAUNCH();
    const int T = ld >> 4;
    double *cur = d_x, *nxt = d_tmp;
    for (int k = 0; k < iters; k++) {
        hipLaunchKernelGGL(purify_tc2_kernel, dim3(T * T, nmol), dim3(256), 0, st, nxt, cur, ld, nocc, tol, k, trace, idem, xstride,
                           sstride, dsc);
        DQC_CHECK_LAUNCH();
        std::swap(cur, nxt);
    }
    if (cur != d_x) DQC_HIP(hipMemcpyAsync(d_x, cur, sizeof(double) * xstride * nmol, hipMemcpyDeviceToDevice, st));
    return DQC_OK;
}

extern "C" int dqc_orth_factor_batched(double *d_q, const double *d_y, const double *d_g, int n, int r, int nmol, void *stream) {
    // d_y (nmol, n, r) row-major with full column rank, d_g (nmol, r, r) = Y^T Y  ->  d_q (nmol, n, r) = Y C^-T, G = C C^T.
    // Enqueues only.
    using namespace dqc;
    if (n <= 0 || r <= 0 || nmol <= 0) return DQC_OK;
    if (nmol > 65535) { set_error("dqc_orth_factor_batched: nmol <= 65535"); return DQC_EINVAL; }
    int rb = 64;  // rows of Y per block: as many as fit next to the r x r factor
    const int ldc = r | 1;
    while (rb > 8 && sizeof(double) * (size_t)(r + rb) * ldc > 150 * 1024) rb >>= 1;
    const size_t lds = sizeof(double) * (size_t)(r + rb) * ldc;
    if (lds > 150 * 1024) { set_error("dqc_orth_factor: r above 132 is not supported"); return DQC_EINVAL; }
    (void)hipFuncSetAttribute((const void *)orth_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(orth_factor_kernel, dim3((n + rb - 1) / rb, nmol), dim3(64), lds, (hipStream_t)stream, d_q, d_y, d_g, n, r, rb);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_orth_factor(double *d_q, const double *d_y, const double *d_g, int n, int r, void *stream) {
    // one matrix: d_y (n, r), d_g (r, r) -> d_q (n, r)
    return dqc_orth_factor_batched(d_q, d_y, d_g, n, r, 1, stream);
}
