// is the fp64 MFMA rate of this MI355X a per-CU pipeline limit or a chip-wide power/clock limit?
// Runs the same MFMA chain on 32/64/128/256 CUs' worth of blocks for ~0.3 s each and prints TF and TF per busy CU;
// tools/ubench/power_probe.sh samples rocm-smi power / sclk meanwhile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_k(double *out, int iters, double a0, double b0) {
    v4d acc[8];
    for (int i = 0; i < 8; i++) acc[i] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void clock_k(long long *out, int iters) {  // wall clock vs shader clock -> effective sclk
    long long t0 = wall_clock64(), c0 = clock64();
    double x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = fma(x, 1.0000001, 1e-9);
    long long t1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; out[2] = (long long)x; }
}
int main() {
    double *out; hipMalloc(&out, sizeof(double) * 256 * 8192);
    long long *clk; hipMalloc(&clk, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++)
    for (int blocks : {32, 64, 128, 256, 512, 1024}) {
        // fixed per-block work; ~60k iters * 8 MFMA * 64 clk = 30M clk = ~15 ms per launch; 20 launches
        const int iters = 60000, nl = 20;
        hipLaunchKernelGGL(mfma_k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int l = 0; l < nl; l++) hipLaunchKernelGGL(mfma_k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * (double)blocks * 4 * nl;
        int busy = blocks < 256 ? blocks : 256;
        printf("blocks %4d: %.1f ms  %.1f TF  %.3f TF per busy CU  (%.1f clk/MFMA/SIMD if 2.4 GHz)\n", blocks, ms,
               flops / ms / 1e9, flops / ms / 1e9 / busy, ms * 1e-3 * 2.4e9 / (8.0 * iters * nl * (blocks <= 256 ? 1 : blocks / 256)));
        fflush(stdout);
    }
    return 0;
}
