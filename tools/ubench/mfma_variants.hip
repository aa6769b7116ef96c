// fp64 MFMA issue-rate variants on gfx950: accumulator count, operand registers, waves per SIMD, 4x4x4 shape,
// and the effective shader clock (s_memtime vs s_memrealtime) while the matrix pipe is saturated.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, int VAR>
__global__ __launch_bounds__(256) void mfma_k(double *out, long long *clk, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = v4d{0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = b0 + i; }
    long long t0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (VAR == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i], 0, 0, 0);
            if (VAR == 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
            if (VAR == 2) { acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i], 0, 0, 0); asm volatile("s_nop 7"); }
        }
    }
    long long t1 = wall_clock64(), c1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma4_k(double *out, long long *clk, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = 0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long t0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = wall_clock64(), c1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}
template <typename F>
void run(const char *name, int blocks, int nmfma_per_iter, double flops_per_mfma, int iters, long long *clk, F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double waves_per_simd = blocks / 256.0;
    double tf = flops_per_mfma * nmfma_per_iter * (double)iters * blocks * 4 / ms / 1e9;
    printf("%-34s blocks %4d  %.2f ms  %6.1f TF  wallclk/MFMA/SIMD(100MHz ticks*24) %.1f  sclk-cycles/MFMA/SIMD %.1f  eff sclk %.0f MHz\n",
           name, blocks, ms, tf, h[0] * 24.0 / (nmfma_per_iter * (double)iters * (waves_per_simd < 1 ? 1 : waves_per_simd)),
           h[1] * 1.0 / (nmfma_per_iter * (double)iters * (waves_per_simd < 1 ? 1 : waves_per_simd)), h[1] * 100.0 / h[0]);
    fflush(stdout);
}
int main() {
    double *out; hipMalloc(&out, sizeof(double) * 256 * 8192);
    long long *clk; hipMalloc(&clk, 64);
    for (int iters : {2000, 40000}) {
        printf("---- iters %d\n", iters);
        for (int blocks : {256, 512, 1024}) {
            run("16x16x4 4acc same-operand", blocks, 4, 2048, iters, clk, [&] { hipLaunchKernelGGL((mfma_k<4, 0>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-9); });
            run("16x16x4 8acc same-operand", blocks, 8, 2048, iters, clk, [&] { hipLaunchKernelGGL((mfma_k<8, 0>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-9); });
            run("16x16x4 16acc same-operand", blocks, 16, 2048, iters, clk, [&] { hipLaunchKernelGGL((mfma_k<16, 0>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-9); });
            run("16x16x4 16acc distinct operands", blocks, 16, 2048, iters, clk, [&] { hipLaunchKernelGGL((mfma_k<16, 1>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-9); });
            run("16x16x4 8acc + s_nop 7", blocks, 8, 2048, iters, clk, [&] { hipLaunchKernelGGL((mfma_k<8, 2>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-9); });
            run("4x4x4(4 blocks) 8acc", blocks, 8, 512, iters, clk, [&] { hipLaunchKernelGGL((mfma4_k<8>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-9); });
        }
    }
    return 0;
}
