// Round 6: is the density pass's epilogue bound by BYTES IN FLIGHT?  Its blocks hold 255 VGPRs (two blocks per CU) and read the three
// gradient arrays in batches of 7 x 16 bytes per lane: issue 7 loads, wait, consume, issue the next 7.  Registers for a second batch
// do not exist.  gfx950 can load global memory straight into LDS (global_load_lds_dwordx4: no VGPR holds the data in flight), so a ring
// of batches can be in flight at no register cost.  Same 2.28 GB, two 256-thread blocks per CU (80 KB of LDS each pins that):
//   (a) batch-synchronous register loads (the epilogue today)       (b) LDS ring, DEPTH batches ahead, consumed from LDS
//   hipcc --offload-arch=gfx950 -O3 -o lds_direct lds_direct.hip && ./lds_direct
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int NB = 7;  // 16-byte loads per lane and batch

// one batch = 256 lanes x 7 x 16 B = 28 KB contiguous; block b takes batches b, b + grid, ...
__global__ __launch_bounds__(256, 2) void reg_batches(const double *__restrict__ buf, long long nbatch, double *out) {
    extern __shared__ double lds[];
    double s = 0;
    const int t = threadIdx.x;
#pragma unroll 1
    for (long long B = blockIdx.x; B < nbatch; B += gridDim.x) {
        const v2d *p = reinterpret_cast<const v2d *>(buf + B * (256 * NB * 2)) + t;
        v2d g[NB];
#pragma unroll
        for (int m = 0; m < NB; m++) g[m] = p[m * 256];
#pragma unroll
        for (int m = 0; m < NB; m++) s += g[m].x * g[m].y;
    }
    if (s == 1.2345) { out[0] = s; lds[t] = s; }
}

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void lds_ring(const double *__restrict__ buf, long long nbatch, double *out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double s = 0;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    double *ring = lds + wave * ((DEPTH + 1) * NB * 128);  // per wave: (DEPTH + 1) slots x NB x 1 KB
    auto issue = [&](long long B, int slot) {
        const double *p = buf + B * (256 * NB * 2) + (wave * 64 + lane) * 2;
#pragma unroll
        for (int m = 0; m < NB; m++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + m * 512),
                                             (__attribute__((address_space(3))) void *)(ring + (slot * NB + m) * 128), 16, 0, 0);
    };
    long long B = blockIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const long long Bd = B + (long long)d * gridDim.x;
        issue(Bd < nbatch ? Bd : B, d);
    }
    int slot = 0;
#pragma unroll 1
    for (; B < nbatch; B += gridDim.x) {
        const long long Bn = B + (long long)DEPTH * gridDim.x;
        int ns = slot + DEPTH;
        if (ns > DEPTH) ns -= DEPTH + 1;
        issue(Bn < nbatch ? Bn : B, ns);
        // the oldest batch has landed when at most DEPTH x NB loads are outstanding
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        const v2d *r = reinterpret_cast<const v2d *>(ring + slot * NB * 128) + lane;
#pragma unroll
        for (int m = 0; m < NB; m++) {
            const v2d g = r[m * 64];
            s += g.x * g.y;
        }
        slot = slot == DEPTH ? 0 : slot + 1;
    }
    if (s == 1.2345) out[0] = s;
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

int main() {
    const long long nbatch = 79500;  // x 28 KB = 2.28 GB
    const size_t n = (size_t)nbatch * 256 * NB * 2;
    double *buf, *out;
    hipMalloc(&buf, n * 8); hipMalloc(&out, 8);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, buf, n);
    const double gb = n * 8 / 1e9;
    const int shm = 80 * 1024;  // two blocks per CU, as the density kernel
    hipFuncSetAttribute((const void *)reg_batches, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    hipFuncSetAttribute((const void *)lds_ring<1>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    const int shm2 = 84 * 1024;  // 2 ahead: 3 slots x 7 KB x 4 waves -- ONE block per CU then (the same bytes in flight per CU as 2 blocks x 1 ahead)
    hipFuncSetAttribute((const void *)lds_ring<2>, hipFuncAttributeMaxDynamicSharedMemorySize, shm2);
    for (int grid : {512, 5354}) {
        double ta = ms([&] { hipLaunchKernelGGL(reg_batches, dim3(grid), dim3(256), shm, 0, buf, nbatch, out); });
        double t1 = ms([&] { hipLaunchKernelGGL(lds_ring<1>, dim3(grid), dim3(256), shm, 0, buf, nbatch, out); });
        double t2 = ms([&] { hipLaunchKernelGGL(lds_ring<2>, dim3(grid), dim3(256), shm2, 0, buf, nbatch, out); });
        printf("grid %5d  %.2f GB, 2 blocks/CU:  register batches of 7 x 16 B  %.0f GB/s   LDS ring 1 ahead %.0f   2 ahead (1 block/CU) %.0f\n",
               grid, gb, gb / ta * 1e3, gb / t1 * 1e3, gb / t2 * 1e3);  // (3 ahead = 112 KB of ring: does not fit beside a second block)
    }
    return 0;
}
