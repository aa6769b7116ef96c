// timeline of the FULL-MATRIX density kernel (density_kernel<13, true>, nao 208): per block start / MFMA-end / end, which blocks share a
// CU, and how much of a CU's time has >= 1 block in its MFMA phase / in its epilogue / both.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -DDEN_TRACE -I dqc_amd/csrc -o /tmp/den_trace_dense tools/ubench/den_trace_dense.hip
#define DEN_TRACE 1
#include "../../dqc_amd/csrc/host.hip"
#include "../../dqc_amd/csrc/grid_density.hip"
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
int main() {
    const int nao = 208, ngrid = 353400, ld = dqc_padded_nao(nao);
    const size_t nd = dqc_ao_doubles(4, ngrid, nao);
    double *ao, *dm, *rho, *grho;
    hipMalloc(&ao, sizeof(double) * nd);
    hipMalloc(&dm, sizeof(double) * ld * ld);
    hipMalloc(&rho, sizeof(double) * ngrid); hipMalloc(&grho, sizeof(double) * 3 * ngrid);
    hipMemset(ao, 0, sizeof(double) * nd); hipMemset(dm, 0, sizeof(double) * ld * ld);
    for (int it = 0; it < 3; it++) dqc_grid_density(rho, grho, ao, 4, ngrid, nao, dm, nullptr);
    hipDeviceSynchronize();
    const int nb = (ngrid + 63) / 64;
    std::vector<long long> h(4 * nb);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dqc::g_den_trace), sizeof(long long) * 4 * nb);
    long long t0 = h[0], tend = 0;
    for (int b = 0; b < nb; b++) { t0 = std::min(t0, h[4 * b]); tend = std::max(tend, h[4 * b + 2]); }
    double sm = 0, se = 0;
    for (int b = 0; b < nb; b++) { sm += h[4 * b + 1] - h[4 * b]; se += h[4 * b + 2] - h[4 * b + 1]; }
    printf("kernel span %.1f us; mean MFMA segment %.2f us, mean epilogue %.2f us, blocks %d\n", (tend - t0) / 100.0, sm / nb / 100.0, se / nb / 100.0, nb);
    std::map<long long, std::vector<int>> cu;
    for (int b = 0; b < nb; b++) cu[h[4 * b + 3]].push_back(b);
    printf("CUs seen: %zu\n", cu.size());
    // per CU: time with >= 1 block in MFMA phase (m), in epilogue (e), both kinds at once (both), two in MFMA (mm), two in epilogue (ee)
    double T = 0, m1 = 0, e1 = 0, both = 0, mm = 0, ee = 0, idle = 0;
    for (auto &kv : cu) {
        std::vector<std::pair<long long, int>> ev;  // (time, +1/-1 mfma | +2/-2 epi)
        for (int b : kv.second) {
            ev.push_back({h[4 * b], 1}); ev.push_back({h[4 * b + 1], -1});
            ev.push_back({h[4 * b + 1], 2}); ev.push_back({h[4 * b + 2], -2});
        }
        std::sort(ev.begin(), ev.end());
        int nm = 0, ne = 0; long long prev = ev[0].first;
        for (auto &x : ev) {
            const double dt = (double)(x.first - prev);
            T += dt;
            if (nm && ne) both += dt; else if (nm >= 2) mm += dt; else if (ne >= 2) ee += dt; else if (nm) m1 += dt; else if (ne) e1 += dt; else idle += dt;
            prev = x.first;
            if (x.second == 1) nm++; else if (x.second == -1) nm--; else if (x.second == 2) ne++; else ne--;
        }
    }
    printf("share of CU time: one block, MFMA %.3f | one block, epilogue %.3f | MFMA + epilogue (the wanted overlap) %.3f | two in MFMA %.3f | two in epilogue %.3f | idle %.3f\n",
           m1 / T, e1 / T, both / T, mm / T, ee / T, idle / T);
    auto &v = cu.begin()->second;
    std::sort(v.begin(), v.end(), [&](int a, int b) { return h[4 * a] < h[4 * b]; });
    for (size_t i = 0; i < v.size() && i < 14; i++)
        printf("    block %5d: start %8.2f  mfma_end %8.2f  end %8.2f us\n", v[i], (h[4 * v[i]] - t0) / 100.0, (h[4 * v[i] + 1] - t0) / 100.0, (h[4 * v[i] + 2] - t0) / 100.0);
    return 0;
}
