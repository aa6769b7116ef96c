// can a register-blocked GEMM inner loop built on v_mfma_f64_4x4x4 (operands from LDS) beat the 16x16x4 one?
//   variant A: 16x16x4, wave tile 16 x (16*NT): 1 + NT ds_read_b64 and NT MFMAs per k-step        (what grid.hip does)
//   variant B: 4x4x4 "A replicated": wave tile (4*TA) x (16*TB): TA broadcast + TB reads, TA*TB MFMAs per k-step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LS = 208;   // row stride (doubles), 16 mod 32
constexpr int KCH = 16;   // points per chunk
template <int NT>
__global__ __launch_bounds__(512, 2) void gemm16(double *out, int iters) {
    __shared__ double lds[4 * KCH * LS];
    for (int i = threadIdx.x; i < 4 * KCH * LS; i += 512) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    v4d acc[NT];
    for (int t = 0; t < NT; t++) acc[t] = v4d{0, 0, 0, 0};
    const double *A = lds + lk * LS + (wave % 13) * 16 + lr, *B = lds + KCH * LS + lk * LS + lr;
    for (int it = 0; it < iters; it++) {
        const int bo = (it & 1) * 2 * KCH * LS;
#pragma unroll
        for (int kk = 0; kk < KCH / 4; kk++) {
            const double a = A[bo + kk * 4 * LS];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[bo + kk * 4 * LS + t * 16], acc[t], 0, 0, 0);
        }
    }
    double s = 0;
    for (int t = 0; t < NT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NT>
__global__ __launch_bounds__(512, 2) void gemm16_2reads(double *out, int iters) {
    __shared__ double lds[4 * KCH * LS];
    for (int i = threadIdx.x; i < 4 * KCH * LS; i += 512) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    v4d acc[NT];
    unsigned off[NT];
    for (int t = 0; t < NT; t++) { acc[t] = v4d{0, 0, 0, 0}; const int tl = wave * NT + t; off[t] = (lk * LS + (tl / 13) * 16 + lr) | ((KCH * LS + lk * LS + (tl % 13) * 16 + lr) << 16); }
    for (int it = 0; it < iters; it++) {
        const double *base = lds + (it & 1) * 2 * KCH * LS;
#pragma unroll
        for (int kk = 0; kk < KCH / 4; kk++) {
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[kk * 4 * LS + (off[t] & 0xffff)], base[kk * 4 * LS + (off[t] >> 16)], acc[t], 0, 0, 0);
        }
    }
    double s = 0;
    for (int t = 0; t < NT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int TA, int TB>
__global__ __launch_bounds__(512, 2) void gemm4(double *out, int iters) {
    __shared__ double lds[4 * KCH * LS];
    for (int i = threadIdx.x; i < 4 * KCH * LS; i += 512) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    double acc[TA][TB];
    for (int a = 0; a < TA; a++) for (int b = 0; b < TB; b++) acc[a][b] = 0;
    const double *A = lds + lk * LS + wave * 4 * TA + (lr & 3), *B = lds + KCH * LS + lk * LS + lr;
    for (int it = 0; it < iters; it++) {
        const int bo = (it & 1) * 2 * KCH * LS;
#pragma unroll
        for (int kk = 0; kk < KCH / 4; kk++) {
            double bf[TB];
#pragma unroll
            for (int b = 0; b < TB; b++) bf[b] = B[bo + kk * 4 * LS + b * 16];
#pragma unroll
            for (int a = 0; a < TA; a++) {
                const double af = A[bo + kk * 4 * LS + 4 * a];
#pragma unroll
                for (int b = 0; b < TB; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bf[b], acc[a][b], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int a = 0; a < TA; a++) for (int b = 0; b < TB; b++) s += acc[a][b];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; hipMalloc(&out, sizeof(double) * 512 * 1024);
    const int iters = 3000;
    for (int blocks : {256, 512}) {
        float ms = timeit([&] { hipLaunchKernelGGL((gemm16<11>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("16x16x4  NT=11        blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 2048.0 * 11 * 4 * iters * blocks * 8 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((gemm16<13>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("16x16x4  NT=13        blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 2048.0 * 13 * 4 * iters * blocks * 8 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((gemm16_2reads<11>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("16x16x4  NT=11 2reads blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 2048.0 * 11 * 4 * iters * blocks * 8 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((gemm4<4, 13>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("4x4x4    TA=4 TB=13   blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 512.0 * 52 * 4 * iters * blocks * 8 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((gemm4<3, 13>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("4x4x4    TA=3 TB=13   blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 512.0 * 39 * 4 * iters * blocks * 8 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((gemm4<13, 4>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("4x4x4    TA=13 TB=4   blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 512.0 * 52 * 4 * iters * blocks * 8 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((gemm4<8, 6>), dim3(blocks), dim3(512), 0, 0, out, iters); });
        printf("4x4x4    TA=8 TB=6    blocks %d: %.2f ms  %.1f TF\n", blocks, ms, 512.0 * 48 * 4 * iters * blocks * 8 / ms / 1e9);
    }
    return 0;
}
