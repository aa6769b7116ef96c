// what does one s_barrier per 44-MFMA chunk cost the consumer waves of vxc_ws_kernel?
//   variants: NB = no barrier, B8 = barrier, 8 waves, B12 = barrier, 8 MFMA waves + 4 waves that only hit the barrier
//   each with compiler-scheduled fragment reads (S) or the hand-pipelined ws_chunk (P)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LS = 208, KCH = 16, NT = 11;
template <int D>
__device__ __forceinline__ void chunk_p(const double *base, const unsigned (&off)[NT], v4d (&acc)[NT]) {
    constexpr int NS = (KCH / 4) * NT;
    double fa[D + 1], fb[D + 1];
#pragma unroll
    for (int s = 0; s < D; s++) { const int ko = (s / NT) * 4 * LS, t = s % NT; fa[s % (D + 1)] = base[ko + (off[t] & 0xffffu)]; fb[s % (D + 1)] = base[ko + (off[t] >> 16)]; }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (s + D < NS) { const int s2 = s + D, ko = (s2 / NT) * 4 * LS, t = s2 % NT; fa[s2 % (D + 1)] = base[ko + (off[t] & 0xffffu)]; fb[s2 % (D + 1)] = base[ko + (off[t] >> 16)]; }
        __builtin_amdgcn_sched_barrier(0);
        acc[s % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s % (D + 1)], fb[s % (D + 1)], acc[s % NT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void chunk_s(const double *base, const unsigned (&off)[NT], v4d (&acc)[NT]) {
#pragma unroll
    for (int kk = 0; kk < KCH / 4; kk++)
#pragma unroll
        for (int t = 0; t < NT; t++)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[kk * 4 * LS + (off[t] & 0xffff)], base[kk * 4 * LS + (off[t] >> 16)], acc[t], 0, 0, 0);
}
template <int D>
__device__ __forceinline__ void chunk_r(const double *base, int ls, const unsigned (&off)[NT], v4d (&acc)[NT]) {
    constexpr int NS = (KCH / 4) * NT;
    double fa[D + 1], fb[D + 1];
#pragma unroll
    for (int s = 0; s < D; s++) { const int ko = (s / NT) * 4 * ls, t = s % NT; fa[s % (D + 1)] = base[ko + (off[t] & 0xffffu)]; fb[s % (D + 1)] = base[ko + (off[t] >> 16)]; }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (s + D < NS) { const int s2 = s + D, ko = (s2 / NT) * 4 * ls, t = s2 % NT; fa[s2 % (D + 1)] = base[ko + (off[t] & 0xffffu)]; fb[s2 % (D + 1)] = base[ko + (off[t] >> 16)]; }
        __builtin_amdgcn_sched_barrier(0);
        acc[s % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s % (D + 1)], fb[s % (D + 1)], acc[s % NT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__global__ __launch_bounds__(768, 3) void kr(double *out, int iters, int ls) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 4 * KCH * LS; i += 768) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    if (wave >= 8) { for (int it = 0; it < iters; it++) __syncthreads(); return; }
    v4d acc[NT];
    unsigned off[NT];
    for (int t = 0; t < NT; t++) { acc[t] = v4d{0, 0, 0, 0}; const int tl = wave * NT + t; off[t] = (lk * ls + (tl / 13) * 16 + lr) | ((KCH * ls + lk * ls + (tl % 13) * 16 + lr) << 16); }
    for (int it = 0; it < iters; it++) {
        chunk_r<4>(lds + (it & 1) * 2 * KCH * ls, ls, off, acc);
        __syncthreads();
    }
    double s = 0;
    for (int t = 0; t < NT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NTHR, bool BAR, bool PIPE>
__global__ __launch_bounds__(NTHR, NTHR == 768 ? 3 : 2) void k(double *out, int iters) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 4 * KCH * LS; i += NTHR) {
        if (iters & 1) {  // odd iteration count: full-entropy mantissas (splitmix64), values in [1, 2)
            unsigned long long z = (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            lds[i] = __longlong_as_double((long long)((z >> 12) | 0x3FF0000000000000ull)) - 1.5;
        } else lds[i] = 1e-3 * (i % 97);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    if (wave >= 8) {
        if (BAR) for (int it = 0; it < iters; it++) __syncthreads();
        return;
    }
    v4d acc[NT];
    unsigned off[NT];
    for (int t = 0; t < NT; t++) { acc[t] = v4d{0, 0, 0, 0}; const int tl = wave * NT + t; off[t] = (lk * LS + (tl / 13) * 16 + lr) | ((KCH * LS + lk * LS + (tl % 13) * 16 + lr) << 16); }
    for (int it = 0; it < iters; it++) {
        const double *base = lds + (it & 1) * 2 * KCH * LS;
        if (PIPE) chunk_p<4>(base, off, acc); else chunk_s(base, off, acc);
        if (BAR) __syncthreads();
    }
    double s = 0;
    for (int t = 0; t < NT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NTHR, bool BAR, bool PIPE>
void run(const char *name, double *d) {
    const int iters = 2000, nblk = 256;
    auto kern = k<NTHR, BAR, PIPE>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KCH * LS * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(NTHR), 4 * KCH * LS * 8, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(NTHR), 4 * KCH * LS * 8, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)nblk * 8 * iters * (KCH / 4) * NT * 2048.0;
    printf("%-28s %7.3f ms  %6.1f TF  (%s)\n", name, ms, fl / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}
int main() {
    double *d; hipMalloc(&d, 256 * 768 * 8);
    run<512, false, false>("NB  8 waves, sched", d);
    run<512, false, true>("NB  8 waves, pipelined", d);
    run<512, true, false>("B8  barrier, sched", d);
    run<512, true, true>("B8  barrier, pipelined", d);
    run<768, true, false>("B12 barrier+4 idle, sched", d);
    run<768, true, true>("B12 barrier+4 idle, pipe", d);
    {
        hipFuncSetAttribute((const void *)kr, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KCH * LS * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kr, dim3(256), dim3(768), 4 * KCH * LS * 8, 0, d, 10, LS);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kr, dim3(256), dim3(768), 4 * KCH * LS * 8, 0, d, 2000, LS);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("B12 pipelined, RUNTIME row stride: %.3f ms  %.1f TF\n", ms, 256.0 * 8 * 2000 * 4 * NT * 2048.0 / ms * 1e-9);
    }
    {   // the production shape: 174 chunks per launch, 20 launches back to back
        auto kern = k<768, true, true>;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int l = 0; l < 20; l++) hipLaunchKernelGGL(kern, dim3(256), dim3(768), 4 * KCH * LS * 8, 0, d, 174);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("174-chunk launches: %.4f ms each (%.1f TF)\n", ms / 20, 256.0 * 8 * 174 * 4 * NT * 2048.0 / (ms / 20) * 1e-9);
            hipEventRecord(e0);
            for (int l = 0; l < 20; l++) hipLaunchKernelGGL(kern, dim3(256), dim3(768), 4 * KCH * LS * 8, 0, d, 175);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("175-chunk launches, random mantissas: %.4f ms each (%.1f TF)\n", ms / 20, 256.0 * 8 * 175 * 4 * NT * 2048.0 / (ms / 20) * 1e-9);
        }
    }
    return 0;
}
