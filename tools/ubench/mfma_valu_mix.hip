// microbenchmark: do fp64 MFMA waves and fp64 VALU-FMA waves overlap on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
// mode 0: all waves MFMA; 1: all waves VALU; 2: even waves MFMA, odd waves VALU
__global__ __launch_bounds__(512) void mix_k(double *out, int iters, double a0, double b0, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
    double a = a0 + threadIdx.x * 1e-9, b = b0, s = 0;
    if (do_mfma) {
        v4d acc[8];
        for (int i = 0; i < 8; i++) acc[i] = v4d{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
        for (int i = 0; i < 16; i++) acc[i] = i;
        for (int it = 0; it < iters * 16; it++)   // 16x more iterations: one MFMA = 2048 flop, one wave FMA = 128 flop
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = fma(a, acc[i], b);
        for (int i = 0; i < 16; i++) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out; (void)hipMalloc(&out, sizeof(double) * 512 * 512);
    const int iters = 4000;
    for (int mode = 0; mode < 3; mode++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(mix_k, dim3(256), dim3(512), 0, 0, out, 10, 1.0, 1e-9, mode);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mix_k, dim3(256), dim3(512), 0, 0, out, iters, 1.0000001, 1e-9, mode);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double waves = 256.0 * 8;
        double f_mfma = 2.0 * 16 * 16 * 4 * 8.0 * iters, f_valu = 2.0 * 64 * 16.0 * iters * 16;
        double flops = mode == 0 ? waves * f_mfma : mode == 1 ? waves * f_valu : waves / 2 * (f_mfma + f_valu);
        printf("mode %d (%s): %.3f ms  %.1f TF total\n", mode, mode == 0 ? "MFMA only" : mode == 1 ? "VALU only" : "half MFMA + half VALU", ms, flops / ms / 1e9);
    }
    return 0;
}
