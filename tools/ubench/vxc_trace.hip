// per-chunk timeline of vxc_ws_kernel on the C5 shape: how long is a chunk period, who waits for whom?
//   slot 0 = loop entry, slot c + 1 = wave finished its work of period c (before the barrier); role 0 = consumer wave 0,
//   role 1 = producer wave 8.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics [-DABL_...] -o /tmp/vt tools/ubench/vxc_trace.hip
#define VXC_TRACE 1
#include "../../dqc_amd/csrc/grid.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
int main() {
    const int nao = 208, ngrid = 353400, ld = dqc_padded_nao(nao);
    double *ao, *w, *vr, *vg, *vm;
    hipMalloc(&ao, sizeof(double) * 4 * (size_t)ngrid * ld); hipMalloc(&w, 8 * ngrid); hipMalloc(&vr, 8 * ngrid);
    hipMalloc(&vg, 8 * 3 * ngrid); hipMalloc(&vm, 8 * ld * ld);
    hipMemset(ao, 0, sizeof(double) * 4 * (size_t)ngrid * ld); hipMemset(w, 0, 8 * ngrid); hipMemset(vr, 0, 8 * ngrid); hipMemset(vg, 0, 8 * 3 * ngrid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; it++) dqc_grid_vxc(vm, ao, 4, ngrid, nao, w, vr, vg, nullptr);
    hipEventRecord(e0);
    for (int it = 0; it < 10; it++) dqc_grid_vxc(vm, ao, 4, ngrid, nao, w, vr, vg, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dqc_grid_vxc: %.4f ms per call (%s)\n", ms / 10, dqc_last_error());
    const int W = dqc::VXC_TRACE_MAXC + 2;
    std::vector<long long> h(256 * 2 * W);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dqc::g_vxc_trace), sizeof(long long) * h.size());
    int nch = 0;
    while (nch + 1 < W && h[nch + 1] > h[0]) nch++;
    long long t0 = h[0], t1 = 0;
    for (int b = 0; b < 256; b++) { t0 = std::min(t0, std::min(h[(2 * b) * W], h[(2 * b + 1) * W])); t1 = std::max(t1, h[(2 * b) * W + nch]); }
    printf("chunks %d; span first loop entry -> last consumer done: %.2f us\n", nch, (t1 - t0) / 100.0);
    double ssum = 0; long long smax = 0;
    for (int b = 0; b < 256; b++) { long long d = h[2 * b * W] - t0; ssum += d; smax = std::max(smax, d); }
    printf("consumer loop entry after first block: mean %.2f us, max %.2f us\n", ssum / 256 / 100.0, smax / 100.0);
    // period statistics over all blocks: consumer done -> next consumer done
    std::vector<double> per, lag;
    for (int b = 0; b < 256; b++)
        for (int c = 1; c < nch; c++) {
            per.push_back((h[2 * b * W + c + 1] - h[2 * b * W + c]) / 100.0);
            lag.push_back((h[(2 * b + 1) * W + c + 1] - h[2 * b * W + c + 1]) / 100.0);  // > 0: the producer finished its period later than the consumer
        }
    std::sort(per.begin(), per.end()); std::sort(lag.begin(), lag.end());
    auto q = [](std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    printf("chunk period (us): p05 %.2f p25 %.2f p50 %.2f p75 %.2f p95 %.2f p99 %.2f  mean %.3f\n", q(per, .05), q(per, .25), q(per, .5), q(per, .75), q(per, .95), q(per, .99),
           [&] { double s = 0; for (double x : per) s += x; return s / per.size(); }());
    printf("producer-done minus consumer-done (us): p05 %.2f p25 %.2f p50 %.2f p75 %.2f p95 %.2f\n", q(lag, .05), q(lag, .25), q(lag, .5), q(lag, .75), q(lag, .95));
    printf("block 0, periods 20..39 (consumer done, producer done, relative to period start):\n");
    for (int c = 20; c < 40 && c < nch; c++)
        printf("  c %3d: period %.2f  consumer %.2f  producer %.2f\n", c, (h[c + 1] - h[c]) / 100.0, (h[c + 1] - h[c]) / 100.0, (h[W + c + 1] - h[c]) / 100.0);
    return 0;
}
