// Is a memset NODE of a stream-captured hipGraph ordered after the PREVIOUS launch of the same graph on the same stream?
// graph = [memset(flag, 0, nbytes)] -> [slow kernel: waits ~`us` microseconds, then flag[0] = 1] -> [check kernel: bad += (flag[0] != 1)]
// launched N times back to back (no host sync between the launches).  If the memset of launch k + 1 is executed early -- while
// launch k's slow kernel is still waiting -- launch k's check kernel sees 0.  Build: hipcc --offload-arch=gfx950 -O2 -o
// /tmp/graph_memset_order tools/ubench/graph_memset_order.hip;  run: /tmp/graph_memset_order [N] [us] [nbytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void slow_kernel(unsigned long long *flag, long long ticks) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __hip_atomic_store(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void check_kernel(unsigned long long *flag, unsigned long long *bad) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1ull) bad[0] += 1;
    bad[1] += 1;
}
__global__ void zero_kernel(unsigned long long *flag, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) flag[i] = 0; }

static int trial(bool use_memset, int N, double us, size_t nbytes) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned long long *flag, *bad;
    CK(hipMalloc(&flag, nbytes < 8 ? 8 : nbytes));
    CK(hipMalloc(&bad, 16));
    CK(hipMemset(bad, 0, 16));
    const long long ticks = (long long)(us * 100.0);  // wall_clock64: 100 MHz
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    if (use_memset) CK(hipMemsetAsync(flag, 0, nbytes, st));
    else hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, st, flag, (int)(nbytes / 8));
    hipLaunchKernelGGL(slow_kernel, dim3(1), dim3(64), 0, st, flag, ticks);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(1), 0, st, flag, bad);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < N; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    unsigned long long h[2];
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("%-13s nbytes %6zu, slow kernel %6.0f us, %d launches back to back: check kernel saw a cleared flag %llu times (ran %llu)\n",
           use_memset ? "memset node" : "zero kernel", nbytes, us, N, h[0], h[1]);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(flag); hipFree(bad); hipStreamDestroy(st);
    return 0;
}

int main(int argc, char **argv) {
    int N = argc > 1 ? atoi(argv[1]) : 200;
    double us = argc > 2 ? atof(argv[2]) : 300.0;
    for (size_t nb : {(size_t)8, (size_t)16, (size_t)1184, (size_t)65536, (size_t)(4 << 20)})
        for (int m = 1; m >= 0; --m)
            if (trial(m == 1, N, us, nb)) return 1;
    return 0;
}
