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
__global__ __launch_bounds__(64) void diis_solve_kernel(double *__restrict__ cout, const double *__restrict__ gram, int H, int m) {
    __shared__ double A[DQC_DIIS_NMAX][DQC_DIIS_NMAX + 1], V[DQC_DIIS_NMAX][DQC_DIIS_NMAX + 1];
    __shared__ double rot[2];
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
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, dia = 0.0;  // every lane sums the whole matrix: uniform decision without a reduction
        for (int i = 0; i < N; i++) {
            dia += A[i][i] * A[i][i];
            for (int j = i + 1; j < N; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-34 * dia) break;  // quadratic convergence: the next sweep would not change a digit
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                if (t == 0) {
                    const double apq = A[p][q];
                    double c = 1.0, sn = 0.0;
                    if (fabs(apq) > 1e-300) {
                        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        sn = tt * c;
                    }
                    rot[0] = c;
                    rot[1] = sn;
                }
                __syncthreads();
                const double c = rot[0], sn = rot[1];
                // columns p, q of A and V (lane t = row), then rows p, q of A (lane t = column)
                double akp = 0, akq = 0;
                if (t < N) {
                    akp = A[t][p]; akq = A[t][q];
                    const double vkp = V[t][p], vkq = V[t][q];
                    V[t][p] = c * vkp - sn * vkq;
                    V[t][q] = sn * vkp + c * vkq;
                }
                __syncthreads();
                if (t < N) {
                    A[t][p] = c * akp - sn * akq;
                    A[t][q] = sn * akp + c * akq;
                }
                __syncthreads();
                if (t < N) {
                    akp = A[p][t]; akq = A[q][t];
                }
                __syncthreads();
                if (t < N) {
                    A[p][t] = c * akp - sn * akq;
                    A[q][t] = sn * akp + c * akq;
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

extern "C" int dqc_diis_solve(double *d_c, const double *d_gram, int nmol, int nhist, int m, void *stream) {
    // d_gram (nmol, nhist, nhist) Gram matrices of the stored error vectors, the first m slots valid; d_c (nmol, nhist) Pulay
    // coefficients (zero for the unused slots).  Enqueues only.
    using namespace dqc;
    if (nmol <= 0) return DQC_OK;
    if (m < 1 || m > nhist || nhist + 1 > DQC_DIIS_NMAX) { set_error("dqc_diis_solve: need 1 <= m <= nhist <= 16"); return DQC_EINVAL; }
    hipLaunchKernelGGL(diis_solve_kernel, dim3(nmol), dim3(64), 0, (hipStream_t)stream, d_c, d_gram, nhist, m);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

extern "C" int dqc_purify_tc2_batched(double *d_x, double *d_tmp, int ld, int nmol, double nocc, int iters, double tol,
                                      double *d_state, void *stream);

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
