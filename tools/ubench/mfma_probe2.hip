// why did the 8-accumulator register-only probe stop at 47 TF while an LDS-fed GEMM loop reaches 75 TF?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, int NB, int THREADS, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k(double *out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = v4d{0, 0, 0, 0};
    double b[NB];
    for (int i = 0; i < NB; i++) b[i] = b0 * (i + 1) + threadIdx.x * 1e-7;
    const double a = a0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[i % NB], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
#define RUN(NACC, NB, THREADS, BLOCKS, MINW)                                                                              \
    {                                                                                                                \
        float ms = timeit([&] { hipLaunchKernelGGL((k<NACC, NB, THREADS, MINW>), dim3(BLOCKS), dim3(THREADS), 0, 0, out, iters, 1.0, 1e-3); }); \
        printf("acc %2d  distinct-b %2d  %4d thr x %4d blocks minw %d: %7.2f ms  %.1f TF\n", NACC, NB, THREADS, BLOCKS, MINW, ms,   \
               2048.0 * NACC * iters * (double)BLOCKS * (THREADS / 64) / ms / 1e9);                                   \
    }
int main() {
    double *out; hipMalloc(&out, sizeof(double) * 512 * 2048);
    const int iters = 20000;
    RUN(13, 13, 256, 512, 1) RUN(13, 13, 256, 512, 2) RUN(13, 13, 256, 1024, 2) RUN(13, 13, 256, 1024, 4)
    RUN(13, 13, 512, 256, 1) RUN(13, 13, 512, 256, 2) RUN(13, 13, 128, 1024, 2) RUN(13, 13, 1024, 128, 2) RUN(13, 13, 1024, 256, 4)
    RUN(13, 13, 64, 2048, 2) RUN(13, 13, 256, 256, 1)
    return 0;
}
