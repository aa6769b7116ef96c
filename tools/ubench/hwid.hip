#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ void k(unsigned *out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the block alive a little so that all 512 are co-resident
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    unsigned *d; hipMalloc(&d, 8 * 1024);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d);
    unsigned h[1024]; hipMemcpy(h, d, 8 * 512, hipMemcpyDeviceToHost);
    std::map<unsigned, int> m, m2;
    for (int b = 0; b < 512; b++) {
        m[((h[2 * b + 1] & 0xf) << 8) | ((h[2 * b] >> 8) & 0xff)]++;
        m2[((h[2 * b + 1] & 0xf) << 16) | ((h[2 * b] >> 8) & 0xffff)]++;
    }
    printf("distinct slots (xcc, hw[15:8]): %zu ; with hw[23:8]: %zu\n", m.size(), m2.size());
    for (int b = 0; b < 24; b++) printf("block %2d hw %08x xcc %08x  cu_id %u sh %u se %u\n", b, h[2 * b], h[2 * b + 1], (h[2*b] >> 8) & 15, (h[2*b] >> 12) & 1, (h[2*b] >> 13) & 7);
    int hist[8] = {0};
    for (auto &kv : m) hist[kv.second < 8 ? kv.second : 7]++;
    for (int i = 0; i < 8; i++) printf("slots with %d blocks: %d\n", i, hist[i]);
    return 0;
}
