// Round 6, docs/LOG_r06.md section 4: does the SHAPE of the density pass's reads cost bandwidth?  The same 2.3 GB, the same bytes in
// flight per block, read (a) as contiguous 8 KB tiles, (b) as the density pass reads its phase-1 chunks: 64 rows x 128 bytes, rows
// 1664 bytes apart, 13 chunks per row block one after the other, (c) as its epilogue reads: per wave instruction 4 rows x 256 bytes.
//   hipcc --offload-arch=gfx950 -O3 -o read_shape read_shape.hip && ./read_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <dlfcn.h>
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int LD = 208;  // doubles per row (1664 bytes)

// (a) block b reads tiles b, b + grid, ...: 8 KB contiguous = 64 x 16 doubles; thread t loads 4 doubles
__global__ __launch_bounds__(256, 2) void contiguous(const double *__restrict__ buf, long long ntile, double *out) {
    double s = 0;
    const int t = threadIdx.x;
    for (long long T = blockIdx.x; T < ntile; T += 2 * gridDim.x) {
        const v2d *p0 = reinterpret_cast<const v2d *>(buf + T * 1024 + t * 4);
        const long long T2 = T + gridDim.x < ntile ? T + gridDim.x : T;
        const v2d *p1 = reinterpret_cast<const v2d *>(buf + T2 * 1024 + t * 4);
        const v2d a = p0[0], b = p0[1], c = p1[0], d = p1[1];
        s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    if (s == 1.2345) out[0] = s;
}

// (b) block b owns 64 consecutive rows; chunk kc = columns 16 kc .. 16 kc + 15 of those rows (128 bytes per row); thread t: row t / 4,
// 4 doubles at (t % 4) * 4 -- the staging pattern of density_lr_kernel's phase 1, two chunks in flight
__global__ __launch_bounds__(256, 2) void row_chunks(const double *__restrict__ buf, long long nrowblk, double *out) {
    double s = 0;
    const int t = threadIdx.x, row = t >> 2, seg = (t & 3) * 4;
    for (long long B = blockIdx.x; B < nrowblk; B += gridDim.x) {
        const double *base = buf + (B * 64 + row) * LD + seg;
#pragma unroll 1
        for (int kc = 0; kc < LD / 16; kc += 2) {
            const v2d *p0 = reinterpret_cast<const v2d *>(base + kc * 16);
            const int k1 = kc + 1 < LD / 16 ? kc + 1 : kc;
            const v2d *p1 = reinterpret_cast<const v2d *>(base + k1 * 16);
            const v2d a = p0[0], b = p0[1], c = p1[0], d = p1[1];
            s += a.x + a.y + b.x + b.y;
            if (kc + 1 < LD / 16) s += c.x + c.y + d.x + d.y;
        }
    }
    if (s == 1.2345) out[0] = s;
}

// (c) the epilogue's shape: a wave owns 16 rows; lane (lr = lane & 15, lk = lane >> 4) loads 16 bytes at column 2 lr of rows lk + 4 r,
// tile pairs m = 0 .. 5 (32 columns apart) + the odd tile: 7 loads per (row group r), 4 row groups
__global__ __launch_bounds__(256, 2) void accum_layout(const double *__restrict__ buf, long long nrowblk, double *out) {
    double s = 0;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, lk = lane >> 4;
    for (long long B = blockIdx.x; B < nrowblk; B += gridDim.x) {
        const double *base = buf + (B * 64 + wave * 16 + lk) * LD + 2 * lr;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v2d g[7];
#pragma unroll
            for (int m = 0; m < 6; m++) g[m] = *reinterpret_cast<const v2d *>(base + (4 * r) * LD + 32 * m);
            g[6] = *reinterpret_cast<const v2d *>(base + (4 * r) * LD + 192 - lr);  // (the odd last tile: 8-byte lanes; kept 16 bytes here)
#pragma unroll
            for (int m = 0; m < 7; m++) s += g[m].x + g[m].y;
        }
    }
    if (s == 1.2345) out[0] = s;
}

// (d) the library's ceiling probe (dqc_probe_stream_read, host.hip): 32 KB tiles, U tiles' loads in flight before the first use.
// GROUPED: block b takes tiles U b .. U b + U - 1, then + U grid (the round-5 probe); otherwise tile b + u grid.
template <int U, bool NT, bool GROUPED>
__global__ __launch_bounds__(256) void tile32(const v2d *__restrict__ buf, long long ntile, double *out) {
    double s = 0;
    const int t = threadIdx.x;
    for (long long T = GROUPED ? (long long)blockIdx.x * U : blockIdx.x; T < ntile; T += (long long)gridDim.x * U) {
        v2d g[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            long long TT = GROUPED ? T + u : T + (long long)u * gridDim.x;
            if (TT >= ntile) TT = T;
            const v2d *tp = buf + TT * 2048;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const v2d *p = tp + ((4 * (t >> 4) + r) * 32 + 2 * (t & 15));
                g[u][2 * r] = NT ? __builtin_nontemporal_load(p) : p[0];
                g[u][2 * r + 1] = NT ? __builtin_nontemporal_load(p + 1) : p[1];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < 8; q++) s += g[u][q].x + g[u][q].y;
    }
    if (s == 1.2345) out[0] = s;
}

// (e) the round-5 probe again with what the library's entry point adds: the wave-reduced atomicAdd of the sum, a memset before
template <bool ATOMIC>
__global__ __launch_bounds__(256) void tile32_sum(const v2d *__restrict__ buf, long long ntile, double *out) {
    constexpr int U = 4;
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
                g[u][2 * r] = __builtin_nontemporal_load(p);
                g[u][2 * r + 1] = __builtin_nontemporal_load(p + 1);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (T + u < ntile)
#pragma unroll
                for (int q = 0; q < 8; q++) s += g[u][q].x + g[u][q].y;
    }
    if (ATOMIC) {
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
    } else if (s == 1.2345) out[0] = s;
}

__global__ void fill(double *buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = 1e-3 * (double)(i % 977);
}

template <typename F>
static double ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float t; hipEventElapsedTime(&t, a, b);
    return t / 10;
}

// one launch at a time, bracketed by its own events and a synchronise (what bench.py's probe and tools/gpu_hbm_ceiling_bisect.py time)
template <typename F>
static double ms_isolated(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    double tot = 0;
    for (int i = 0; i < 10; i++) {
        hipEventRecord(a);
        f();
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        tot += t;
    }
    return tot / 10;
}

int main() {
    const long long nrow = 342656LL * 4;   // four AO components of a C5 molecule's live grid (a multiple of 64)
    const size_t n = (size_t)nrow * LD;
    double *buf, *out;
    hipMalloc(&buf, n * 8 + 4096); hipMalloc(&out, 8);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, buf, n + 512);
    const double gb = n * 8 / 1e9;
    for (int grid : {512, 2048, 5354}) {
        double ta = ms([&] { hipLaunchKernelGGL(contiguous, dim3(grid), dim3(256), 0, 0, buf, (long long)(n / 1024), out); });
        double tb = ms([&] { hipLaunchKernelGGL(row_chunks, dim3(grid), dim3(256), 0, 0, buf, nrow / 64, out); });
        double tc = ms([&] { hipLaunchKernelGGL(accum_layout, dim3(grid), dim3(256), 0, 0, buf, nrow / 64, out); });
        printf("grid %5d  %.2f GB:  contiguous 8 KB tiles %.3f ms = %.0f GB/s   phase-1 row chunks %.3f ms = %.0f GB/s   epilogue layout %.3f ms = %.0f GB/s\n",
               grid, gb, ta, gb / ta * 1e3, tb, gb / tb * 1e3, tc, gb / tc * 1e3);
    }
    {
        double ta = ms_isolated([&] { hipLaunchKernelGGL(contiguous, dim3(5354), dim3(256), 0, 0, buf, (long long)(n / 1024), out); });
        double tc = ms_isolated([&] { hipLaunchKernelGGL(accum_layout, dim3(5354), dim3(256), 0, 0, buf, nrow / 64, out); });
        printf("isolated launches (own events + synchronise each): contiguous %.3f ms = %.0f GB/s   epilogue layout %.3f ms = %.0f GB/s\n",
               ta, gb / ta * 1e3, tc, gb / tc * 1e3);
    }
    // (d) the probe's own geometry, on the probe's 2 GiB and on this buffer
    for (size_t bytes : {(size_t)2 << 30, n * 8}) {
        const long long nt = bytes / 32768;
        const double g2 = nt * 32768 / 1e9;
        printf("32 KB-tile probe variants on %.2f GB (isolated launches), GB/s by grid 256 / 512 / 1024 / 2048 / 4096:\n", g2);
#define ROW(U, NT, GR, label)                                                                                                  \
    {                                                                                                                          \
        printf("  %-46s", label);                                                                                              \
        for (int grid : {256, 512, 1024, 2048, 4096}) {                                                                        \
            double t_ = ms_isolated([&] { hipLaunchKernelGGL((tile32<U, NT, GR>), dim3(grid), dim3(256), 0, 0, (const v2d *)buf, nt, out); }); \
            printf(" %6.0f", g2 / t_ * 1e3);                                                                                   \
        }                                                                                                                      \
        printf("\n");                                                                                                          \
    }
        ROW(4, true, true, "U=4 non-temporal grouped (round-5 probe)")
        ROW(4, true, false, "U=4 non-temporal tile-strided")
        ROW(4, false, false, "U=4 cached tile-strided")
        ROW(2, true, false, "U=2 non-temporal tile-strided")
        ROW(2, false, false, "U=2 cached tile-strided")
        ROW(1, false, false, "U=1 cached tile-strided")
#undef ROW
    }
    {
        const long long nt = ((size_t)2 << 30) / 32768;
        const double g2 = nt * 32768 / 1e9;
        double t0 = ms_isolated([&] { hipLaunchKernelGGL((tile32_sum<false>), dim3(4096), dim3(256), 0, 0, (const v2d *)buf, nt, out); });
        double t1 = ms_isolated([&] { hipLaunchKernelGGL((tile32_sum<true>), dim3(4096), dim3(256), 0, 0, (const v2d *)buf, nt, out); });
        double t2 = ms_isolated([&] { hipMemsetAsync(out, 0, 8, 0); hipLaunchKernelGGL((tile32_sum<true>), dim3(4096), dim3(256), 0, 0, (const v2d *)buf, nt, out); });
        printf("round-5 probe body on 2.15 GB, grid 4096: no sum %.0f GB/s   + wave-reduced atomicAdd %.0f   + memset before %.0f\n",
               g2 / t0 * 1e3, g2 / t1 * 1e3, g2 / t2 * 1e3);
        // the library's own entry point from this bare process
        void *h = dlopen("../../dqc_amd/libdqc_amd.so", RTLD_NOW);
        if (h) {
            typedef int (*fn_t)(const double *, size_t, double *, void *);
            fn_t fn = (fn_t)dlsym(h, "dqc_probe_stream_read");
            double t3 = ms_isolated([&] { fn(buf, ((size_t)2 << 30) / 8, out, nullptr); });
            double t4 = ms([&] { fn(buf, ((size_t)2 << 30) / 8, out, nullptr); });
            printf("libdqc_amd.so dqc_probe_stream_read from this process: isolated %.0f GB/s   back to back %.0f GB/s\n", g2 / t3 * 1e3, g2 / t4 * 1e3);
        } else printf("dlopen failed: %s\n", dlerror());
    }
    return 0;
}
