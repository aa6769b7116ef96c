// What is the best achievable streaming-READ bandwidth on this box, and with which access shape?  (J/K tile stream, jk.hip)
//   naive    : grid-stride, one 16-byte load per lane per iteration (the round-1 probe)
//   tile<U>  : one 256-thread block reads contiguous 32 KB "tiles", U tiles' worth of loads issued before the first use
//   ...nt    : the same with non-temporal loads
// plus the same kernels on a 96 MB buffer read repeatedly (Infinity-Cache resident) -- what a re-read costs when it hits MALL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void naive(const v2d *__restrict__ buf, size_t n2, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n2; i += stride) { v2d v = buf[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void tile(const v2d *__restrict__ buf, long long ntile, double *out) {
    // a tile = 2048 v2d; thread t loads 8 v2d of a tile: rows of 64 doubles, 4 x (2 v2d) like jk_tiles_kernel
    double s = 0;
    const int t = threadIdx.x;
    for (long long T = (long long)blockIdx.x * U; T < ntile; T += (long long)gridDim.x * U) {
        v2d g[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long TT = T + u < ntile ? T + u : T;
            const v2d *tp = buf + TT * 2048;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const v2d *p = tp + ((4 * (t >> 4) + r) * 32 + 2 * (t & 15));
                if (NT) { g[u][2 * r] = __builtin_nontemporal_load(p); g[u][2 * r + 1] = __builtin_nontemporal_load(p + 1); }
                else { g[u][2 * r] = p[0]; g[u][2 * r + 1] = p[1]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < 8; q++) s += g[u][q].x * 1.0000001 + g[u][q].y;
    }
    if (s == 1.2345) out[0] = s;
}

__global__ void fill_random(double *buf, size_t n, int kind) {
    // kind 1: uniform random mantissas in [-1, 1) (the round-1 probe's data);  kind 2: ERI-like (mostly tiny magnitudes)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = i * 6364136223846793005ull + 1442695040888963407ull;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33;
        double u = (double)(x >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
        if (kind == 2) { const double m = (double)((x >> 3) & 0xff) / 255.0; u *= exp(-30.0 * m * m); }
        buf[i] = u;
    }
}

template <typename F>
static double timeit(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 / reps;
}

int main() {
    double *out;
    hipMalloc(&out, 8);
    for (int kind = 0; kind < 3; kind++)
    for (size_t bytes : {(size_t)4 << 30, (size_t)96 << 20}) {
        if (kind > 0 && bytes < ((size_t)1 << 30)) continue;
        double *buf;
        hipMalloc(&buf, bytes);
        if (kind == 0) hipMemset(buf, 0, bytes);
        else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, kind);
        printf("data kind %d (0 zeros, 1 random, 2 mostly-tiny random)\n", kind);
        const size_t n2 = bytes / 16;
        const long long ntile = bytes / 32768;
        const int reps = bytes > ((size_t)1 << 30) ? 5 : 200;
        printf("buffer %.0f MB\n", bytes / 1048576.0);
        for (int grid : {2048, 4096, 8192}) {
            double t = timeit([&] { hipLaunchKernelGGL(naive, dim3(grid), dim3(256), 0, 0, (const v2d *)buf, n2, out); }, reps);
            printf("  naive      grid %5d : %7.1f GB/s\n", grid, bytes / t / 1e9);
        }
#define RUN(U, NT)                                                                                                        \
        for (int grid : {1024, 2048, 4096}) {                                                                             \
            double t = timeit([&] { hipLaunchKernelGGL((tile<U, NT>), dim3(grid), dim3(256), 0, 0, (const v2d *)buf, ntile, out); }, reps); \
            printf("  tile U=%d nt=%d grid %5d : %7.1f GB/s\n", U, (int)NT, grid, bytes / t / 1e9);                          \
        }
        RUN(1, false) RUN(1, true) RUN(2, false) RUN(2, true) RUN(4, false) RUN(4, true)
        hipFree(buf);
    }
    return 0;
}
