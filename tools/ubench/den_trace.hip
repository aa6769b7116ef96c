// timeline of the rank-r density kernel: which blocks share a CU and how their MFMA / epilogue segments overlap
#define DEN_TRACE 1
#include "../../dqc_amd/csrc/grid_density.hip"
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
int main() {
    const int nao = 208, ngrid = 353400, ld = dqc_padded_nao(nao), rp = 48;
    double *ao, *orb, *orbt, *rho, *grho;
    hipMalloc(&ao, sizeof(double) * 4 * (size_t)ngrid * ld);
    hipMalloc(&orb, sizeof(double) * ld * rp); hipMalloc(&orbt, sizeof(double) * ld * rp);
    hipMalloc(&rho, sizeof(double) * ngrid); hipMalloc(&grho, sizeof(double) * 3 * ngrid);
    hipMemset(ao, 0, sizeof(double) * 4 * (size_t)ngrid * ld); hipMemset(orb, 0, sizeof(double) * ld * rp); hipMemset(orbt, 0, sizeof(double) * ld * rp);
    for (int gga = 0; gga < 2; gga++) {
        for (int it = 0; it < 3; it++) dqc_grid_density_lr(rho, gga ? grho : nullptr, ao, 4, ngrid, nao, orb, orbt, rp, nullptr);
        hipDeviceSynchronize();
        const int nb = (ngrid + 63) / 64;
        std::vector<long long> h(4 * nb);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dqc::g_den_trace), sizeof(long long) * 4 * nb);
        long long t0 = h[0];
        for (int b = 0; b < nb; b++) t0 = std::min(t0, h[4 * b]);
        long long tend = 0;
        for (int b = 0; b < nb; b++) tend = std::max(tend, h[4 * b + 2]);
        double sm = 0, se = 0;
        for (int b = 0; b < nb; b++) { sm += h[4 * b + 1] - h[4 * b]; se += h[4 * b + 2] - h[4 * b + 1]; }
        printf("gga %d: kernel span %.1f us; mean MFMA segment %.2f us, mean epilogue %.2f us, blocks %d\n", gga, (tend - t0) / 100.0, sm / nb / 100.0, se / nb / 100.0, nb);
        // one CU's timeline
        std::map<long long, std::vector<int>> cu;
        for (int b = 0; b < nb; b++) cu[h[4 * b + 3]].push_back(b);
        printf("  CUs seen: %zu\n", cu.size());
        auto &v = cu.begin()->second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return h[4 * a] < h[4 * b]; });
        for (size_t i = 0; i < v.size() && i < 12; i++)
            printf("    block %5d: start %8.2f  mfma_end %8.2f  end %8.2f us\n", v[i], (h[4 * v[i]] - t0) / 100.0, (h[4 * v[i] + 1] - t0) / 100.0, (h[4 * v[i] + 2] - t0) / 100.0);
        // concurrency: average number of blocks resident per CU
        double res = 0;
        for (int b = 0; b < nb; b++) res += h[4 * b + 2] - h[4 * b];
        printf("  mean resident blocks per CU: %.2f\n", res / (double)(tend - t0) / cu.size());
    }
    return 0;
}
