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
__global__ __launch_bounds__(256) void purify_tc2_kernel(double *__restrict__ xout, const double *__restrict__ xin, int ld,
                                                         double nocc, double tol, int k, double *__restrict__ trace,
                                                         double *__restrict__ idem) {
    __shared__ double red[3][4][64];
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
    const double tr = trace[k];
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
        if (ti == tj) atomicAdd(&trace[k + 1], tsum);
        // max of non-negative doubles == max of their bit patterns as unsigned integers
        atomicMax(reinterpret_cast<unsigned long long *>(&idem[k]), (unsigned long long)__double_as_longlong(emax));
    }
}

__global__ void purify_trace_kernel(const double *__restrict__ x, int ld, double *__restrict__ trace0) {
    double s = 0.0;
    for (int i = threadIdx.x; i < ld; i += blockDim.x) s += x[(size_t)i * ld + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) trace0[0] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace dqc

extern "C" int dqc_purify_tc2(double *d_x, double *d_tmp, int ld, double nocc, int iters, double tol, double *d_state,
                              void *stream) {
    // d_x (ld, ld): X0 on entry (spectrum in [0, 1], zero padded to ld = multiple of 16), the purified projector on
    // return; d_tmp (ld, ld) scratch; d_state: 2 * (iters + 2) doubles -- trace[0..iters], idem[0..iters]
    // (idem[k] = max |X_k^2 - X_k|, so the caller can tell convergence: min over k).  Enqueues only.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (ld <= 0 || (ld & 15)) { set_error("dqc_purify_tc2: ld must be a positive multiple of 16"); return DQC_EINVAL; }
    if (iters < 1) { set_error("dqc_purify_tc2: iters must be >= 1"); return DQC_EINVAL; }
    double *trace = d_state, *idem = d_state + (iters + 2);
    DQC_HIP(hipMemsetAsync(d_state, 0, sizeof(double) * 2 * (iters + 2), st));
    hipLaunchKernelGGL(purify_trace_kernel, dim3(1), dim3(256), 0, st, d_x, ld, trace);
    DQC_CHECK_LAUNCH();
    const int T = ld >> 4;
    double *cur = d_x, *nxt = d_tmp;
    for (int k = 0; k < iters; k++) {
        hipLaunchKernelGGL(purify_tc2_kernel, dim3(T * T), dim3(256), 0, st, nxt, cur, ld, nocc, tol, k, trace, idem);
        DQC_CHECK_LAUNCH();
        std::swap(cur, nxt);
    }
    if (cur != d_x) DQC_HIP(hipMemcpyAsync(d_x, cur, sizeof(double) * (size_t)ld * ld, hipMemcpyDeviceToDevice, st));
    return DQC_OK;
}
