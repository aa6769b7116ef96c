// phase timeline of the factor-form density kernel from a register-neutral trace (see make_den_trace2.py); C5 shape, random AO data
//   python make_den_trace2.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o den_trace2 den_trace2.hip ../../dqc_amd/csrc/host.hip
#include "_den_trace2_kernel.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
__global__ void fillk(double *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = 1e-3 * (double)((i * 2654435761u) % 977) - 0.4;
}
int main(int argc, char **argv) {
    { int t = argc > 1 ? atoi(argv[1]) * 100 : 0; hipMemcpyToSymbol(HIP_SYMBOL(dqc::g_den_stagger), &t, sizeof(int)); printf("start offset of odd wave slots: %d us\n", t / 100); }
    const int nao = 208, ngrid = 342689, ld = dqc_padded_nao(nao), rp = 48;
    const size_t nd = dqc_ao_doubles(4, ngrid, nao);
    double *ao, *orb, *orbt, *rho, *grho;
    hipMalloc(&ao, sizeof(double) * nd);
    hipMalloc(&orb, sizeof(double) * ld * rp); hipMalloc(&orbt, sizeof(double) * ld * rp);
    hipMalloc(&rho, sizeof(double) * ngrid); hipMalloc(&grho, sizeof(double) * 3 * ngrid);
    hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, ao, nd);
    hipLaunchKernelGGL(fillk, dim3(64), dim3(256), 0, 0, orb, (size_t)ld * rp);
    hipLaunchKernelGGL(fillk, dim3(64), dim3(256), 0, 0, orbt, (size_t)ld * rp);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 50; it++) dqc_grid_density_lr(rho, grho, ao, 4, ngrid, nao, orb, orbt, rp, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 20; it++) dqc_grid_density_lr(rho, grho, ao, 4, ngrid, nao, orb, orbt, rp, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("traced kernel: %.3f ms per launch (product kernel: ~0.48)\n", ms / 20);
    const int nb = (ngrid + 63) / 64;
    std::vector<long long> h(5 * nb);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dqc::g_den_trace2), sizeof(long long) * 5 * nb);
    long long t0 = h[0], tend = 0;
    for (int b = 0; b < nb; b++) { t0 = std::min(t0, h[5 * b]); tend = std::max(tend, h[5 * b + 3]); }
    double s1 = 0, s2 = 0, s3 = 0;
    for (int b = 0; b < nb; b++) { s1 += h[5 * b + 1] - h[5 * b]; s2 += h[5 * b + 2] - h[5 * b + 1]; s3 += h[5 * b + 3] - h[5 * b + 2]; }
    printf("span %.1f us; per block: phase 1 %.2f us, phase 2 %.2f us, epilogue %.2f us; blocks %d\n", (tend - t0) / 100.0, s1 / nb / 100, s2 / nb / 100, s3 / nb / 100, nb);
#ifdef DEN_STEPS
    {
        std::vector<long long> st(8 * nb);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(dqc::g_den_steps), sizeof(long long) * 8 * nb);
        double m[8] = {0};
        for (int b = 512; b < nb - 512; b++) for (int q = 0; q < 8; q++) m[q] += st[8 * b + q];
        printf("direct phase 1, mean time since block start (us): first staging done %.2f;", m[0] / (nb - 1024) / 100);
        for (int q = 1; q < 8; q++) printf(" step %d %.2f", q - 1, m[q] / (nb - 1024) / 100);
        printf("\n");
    }
#endif
    // chip-wide phase occupancy over time: how many blocks are in each phase, sampled every 5 us over the middle of the launch
    printf("time(us): blocks in phase 1 / phase 2 / epilogue\n");
    for (double t = 200; t < 200 + 46 * 2; t += 4) {
        int c1 = 0, c2 = 0, c3 = 0;
        const long long tt = t0 + (long long)(t * 100);
        for (int b = 0; b < nb; b++) {
            if (tt >= h[5 * b] && tt < h[5 * b + 1]) c1++;
            else if (tt >= h[5 * b + 1] && tt < h[5 * b + 2]) c2++;
            else if (tt >= h[5 * b + 2] && tt < h[5 * b + 3]) c3++;
        }
        printf("  %6.1f: %4d %4d %4d\n", t, c1, c2, c3);
    }
    // one CU's timeline
    std::map<long long, std::vector<int>> cu;
    for (int b = 0; b < nb; b++) cu[h[5 * b + 4] >> 4 & 0xfffff0] .push_back(b);  // (drop the wave-slot bits)
    printf("CU ids seen: %zu\n", cu.size());
    auto &v = cu.begin()->second;
    std::sort(v.begin(), v.end(), [&](int a, int b) { return h[5 * a] < h[5 * b]; });
    for (size_t i = 0; i < v.size() && i < 14; i++) {
        const long long *o = &h[5 * v[i]];
        printf("  block %5d slot %2lld: start %7.2f  p1 end %7.2f  p2 end %7.2f  end %7.2f\n", v[i], o[4] & 15, (o[0] - t0) / 100.0, (o[1] - t0) / 100.0, (o[2] - t0) / 100.0, (o[3] - t0) / 100.0);
    }
    return 0;
}
