// decode the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 by one-hot probing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out) {
    const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
    double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[((size_t)la * 64 + lb) * 64 + lane] = d;
}
int main() {
    double *d; hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
    hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for each A lane: which B lanes pair with it, and where the product lands
    for (int la = 0; la < 64; la++) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; lb++)
            for (int l = 0; l < 64; l++)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    return 0;
}
