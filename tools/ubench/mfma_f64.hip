// microbenchmark: fp64 MFMA (16x16x4) and fp64 VALU FMA issue rates on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_k(double *out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void fma_k(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double *out; hipMalloc(&out, sizeof(double) * 256 * 4096);
    const int iters = 20000;
    for (int wpc : {4, 8, 16}) {  // waves per CU
        int blocks = 256 * wpc / 4;
        float ms = timeit([&] { hipLaunchKernelGGL(mfma_k<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * blocks * 4;
        printf("mfma_f64_16x16x4: %2d waves/CU, 8 acc: %.3f ms  %.1f TF  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpc, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (8.0 * iters * (wpc / 4.0)));
        ms = timeit([&] { hipLaunchKernelGGL(fma_k<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
        flops = 2.0 * 64 * 16.0 * iters * blocks * 4;
        printf("v_fma_f64       : %2d waves/CU, 16 acc: %.3f ms  %.1f TF\n", wpc, ms, flops / ms / 1e9);
    }
    return 0;
}
