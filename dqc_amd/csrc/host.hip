// host.hip -- error handling, table parsing and small utilities of libdqc_amd.so
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include <mutex>

namespace dqc {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// F_0 .. F_8 at X_i = i / 8 (series for F_8 in long double, downward recursion): the Boys table of eri_core.hpp
const std::vector<double> &boys_table_host() {
    static std::mutex mu;
    static std::vector<double> tab;
    std::lock_guard<std::mutex> lk(mu);
    if (tab.empty()) {
        constexpr int W = 9, ROWS = 321;
        tab.resize(W * ROWS);
        for (int i = 0; i < ROWS; i++) {
            const long double X = i * 0.125L, ex = expl(-X);
            const int mtop = W - 1;
            long double term = 1.0L / (2 * mtop + 1), sum = term;  // F_m(X) = e^-X sum_j (2X)^j / ((2m+1)(2m+3)...(2m+2j+1))
            for (int j = 1; j < 400; j++) {
                term *= 2.0L * X / (2 * mtop + 2 * j + 1);
                sum += term;
                if (term < 1e-22L * sum) break;
            }
            long double f = ex * sum;
            tab[i * W + mtop] = (double)f;
            for (int m = mtop; m >= 1; m--) {
                f = (2.0L * X * f + ex) / (2 * m - 1);
                tab[i * W + m - 1] = (double)f;
            }
        }
    }
    return tab;
}

// process-wide switches, read by every entry point from whatever thread calls it: atomics
static std::atomic<bool> g_deterministic{false};
bool deterministic_mode() { return g_deterministic.load(std::memory_order_relaxed); }
static std::atomic<int> g_generic_eri{-1};  // -1: not set (environment DQC_ERI_GENERIC decides)
bool generic_eri_forced() {
    int v = g_generic_eri.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = std::getenv("DQC_ERI_GENERIC");
        v = (e && e[0] == '1') ? 1 : 0;
        int expect = -1;
        g_generic_eri.compare_exchange_strong(expect, v);  // (a concurrent dqc_set_generic_eri wins)
        v = g_generic_eri.load(std::memory_order_relaxed);
    }
    return v == 1;
}

// ---- compute-unit partitions (round 6): streams created with a CU mask and the CU count the grid kernels size their launches for ----
// One block per CU is how vxc_ws* / vxc_wsd fill the chip; on a stream that owns only part of the CUs (the Coulomb stream of another
// molecule runs on the rest, dqc_stream_create_partition) the slab count follows the partition.  Registry: stream -> CUs.
static std::mutex g_cu_mu;
static std::vector<std::pair<hipStream_t, int>> g_stream_cus;
static int device_cus() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
// Cap on the CUs the one-block-per-CU Vxc kernels occupy (0: none).  Those blocks hold 408 of 512 VGPRs per SIMD and 128 of 160 KB of LDS for the whole launch, so
// while a Vxc kernel runs no block of the other hot kernels is resident anywhere; with a cap of e.g. 208 the launch leaves 48 CUs to the kernels other
// streams have queued (the HBM-bound Coulomb / density passes of other molecules of a batch).  dqc_set_vxc_cus / DQC_VXC_CUS.
static std::atomic<int> g_vxc_cus{-1};
int vxc_cus_cap() {
    int v = g_vxc_cus.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = std::getenv("DQC_VXC_CUS");
        v = e ? std::max(0, atoi(e)) : 0;
        int expect = -1;
        g_vxc_cus.compare_exchange_strong(expect, v);
        v = g_vxc_cus.load(std::memory_order_relaxed);
    }
    return v;
}
int stream_cus(hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(g_cu_mu);
        for (auto &e : g_stream_cus)
            if (e.first == st) return e.second;
    }
    return device_cus();
}

// ---- pinned staging blocks of the stream-ordered DevPool ----
static std::mutex g_stg_mu;
static std::vector<Staging *> g_stg;

Staging *staging_acquire(size_t bytes) {
    std::lock_guard<std::mutex> lk(g_stg_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (Staging *s : g_stg) {
        if (s->dev != dev || s->cap < bytes || s->held) continue;
        if (s->busy && hipEventQuery(s->ev) != hipSuccess) continue;  // its copy is still in flight
        s->busy = false;
        s->held = true;
        return s;
    }
    Staging *s = new Staging;
    size_t cap = 1 << 16;
    while (cap < bytes) cap <<= 1;
    if (hipHostMalloc(&s->host, cap, hipHostMallocDefault) != hipSuccess) { delete s; return nullptr; }
    if (hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(s->host); delete s; return nullptr; }
    s->cap = cap;
    s->dev = dev;
    s->held = true;
    g_stg.push_back(s);
    return s;
}

void staging_release(Staging *s, hipStream_t st) {
    // stays `busy` until the event recorded behind the copy has completed (checked at the next acquire)
    std::lock_guard<std::mutex> lk(g_stg_mu);
    s->busy = hipEventRecord(s->ev, st) == hipSuccess;
    s->held = false;
}

int parse_basis(Basis &b, const int *atm, int natm, const int *bas, int nbas, const double *env,
                int nenv, const double *zs) {
    // layout per dqc/hamilton/intor/lcintwrap.py:57, 81-83
    b.shells.clear();
    b.exps.clear();
    b.coefs.clear();
    b.natm = natm;
    b.atom_xyz.resize((size_t)natm * 3);
    b.atom_z.resize(natm);
    for (int ia = 0; ia < natm; ia++) {
        int p = atm[ia * 6 + 1];
        if (p < 0 || p + 3 > nenv) { set_error("atm: coordinate pointer outside env"); return DQC_EINVAL; }
        for (int d = 0; d < 3; d++) b.atom_xyz[ia * 3 + d] = env[p + d];
        b.atom_z[ia] = zs ? zs[ia] : (double)atm[ia * 6 + 0];
    }
    int ao = 0;
    for (int i = 0; i < nbas; i++) {
        const int *s = bas + i * 8;
        HostShell h;
        h.atom = s[0]; h.l = s[1]; h.nprim = s[2];
        if (s[3] != 1) { set_error("bas: nctr must be 1 (general contractions are split upstream)"); return DQC_EINVAL; }
        if (h.l < 0 || h.l > DQC_LMAX) { set_error("bas: angular momentum above g is not supported"); return DQC_EINVAL; }
        if (h.atom < 0 || h.atom >= natm) { set_error("bas: atom index out of range"); return DQC_EINVAL; }
        // the integral kernels split a primitive-quartet index through a float reciprocal that is exact below 2^22 (eri_core.hpp):
        // (nprim_a nprim_b)(nprim_c nprim_d) < 2^22 holds for nprim <= 45 on every shell
        if (h.nprim < 1 || h.nprim > 45) { set_error("bas: a shell must have 1 ... 45 primitives"); return DQC_EINVAL; }
        if (s[5] < 0 || s[5] + h.nprim > nenv || s[6] < 0 || s[6] + h.nprim > nenv) {
            set_error("bas: exponent/coefficient pointer outside env"); return DQC_EINVAL;
        }
        h.ao_off = ao;
        h.prim_off = (int)b.exps.size();
        for (int d = 0; d < 3; d++) h.r[d] = b.atom_xyz[h.atom * 3 + d];
        for (int p = 0; p < h.nprim; p++) {
            b.exps.push_back(env[s[5] + p]);
            b.coefs.push_back(env[s[6] + p]);
        }
        ao += 2 * h.l + 1;
        b.shells.push_back(h);
    }
    b.nao = ao;
    return 0;
}

void group_s_shells(const Basis &b, Basis &g, bool merge) {
    g = Basis();
    g.nao = b.nao;
    g.natm = b.natm;
    g.atom_xyz = b.atom_xyz;
    g.atom_z = b.atom_z;
    for (size_t ish = 0; ish < b.shells.size(); ish++) {
        const HostShell &h = b.shells[ish];
        int hit = -1;
        if (merge && h.l == 0)
            for (size_t k = 0; k < g.shells.size() && hit < 0; k++) {
                const HostShell &o = g.shells[k];
                if (o.l != 0 || o.atom != h.atom || o.nprim != h.nprim || g.ao_off1[k] >= 0) continue;
                bool same = true;
                for (int p = 0; p < h.nprim && same; p++) same = g.exps[o.prim_off + p] == b.exps[h.prim_off + p];
                if (same) hit = (int)k;
            }
        if (hit >= 0) {
            g.sh_id1[hit] = (int)ish;
            g.ao_off1[hit] = h.ao_off;
            for (int p = 0; p < h.nprim; p++) g.coefs1[g.shells[hit].prim_off + p] = b.coefs[h.prim_off + p];
            continue;
        }
        HostShell n = h;
        n.prim_off = (int)g.exps.size();
        for (int p = 0; p < h.nprim; p++) {
            g.exps.push_back(b.exps[h.prim_off + p]);
            g.coefs.push_back(b.coefs[h.prim_off + p]);
            g.coefs1.push_back(0.0);
        }
        g.shells.push_back(n);
        g.ao_off1.push_back(-1);
        g.sh_id0.push_back((int)ish);
        g.sh_id1.push_back(-1);
    }
}

int upload_shells(DevShells &d, const Basis &b, DevPool &pool, hipStream_t st) {
    int n = (int)b.shells.size();
    std::vector<int> l(n), np(n), ao(n), po(n);
    std::vector<double> xyz((size_t)n * 3);
    for (int i = 0; i < n; i++) {
        l[i] = b.shells[i].l; np[i] = b.shells[i].nprim; ao[i] = b.shells[i].ao_off; po[i] = b.shells[i].prim_off;
        for (int k = 0; k < 3; k++) xyz[i * 3 + k] = b.shells[i].r[k];
    }
    int rc;
    if ((rc = pool.upload(&d.l, l, st))) return rc;
    if ((rc = pool.upload(&d.nprim, np, st))) return rc;
    if ((rc = pool.upload(&d.ao_off, ao, st))) return rc;
    if ((rc = pool.upload(&d.prim_off, po, st))) return rc;
    if ((rc = pool.upload(&d.xyz, xyz, st))) return rc;
    if ((rc = pool.upload(&d.exps, b.exps, st))) return rc;
    if ((rc = pool.upload(&d.coefs, b.coefs, st))) return rc;
    if (b.grouped() && (rc = pool.upload(&d.ao_off1, b.ao_off1, st))) return rc;
    d.nsh = n;
    return 0;
}

}  // namespace dqc

extern "C" {

const char *dqc_last_error(void) { return dqc::g_err.c_str(); }

int dqc_set_deterministic(int on) {
    // process-wide: the cross-block accumulations of the Fock build (J / K accumulators, split-K Vxc partial sums, the trace of
    // the purification iterate) switch from fp64 atomics to fixed-point integer atomics (common.hpp: acc_add), which makes
    // every result bit-reproducible from run to run.  Returns the previous setting.
    return dqc::g_deterministic.exchange(on != 0) ? 1 : 0;
}
int dqc_get_deterministic(void) { return dqc::deterministic_mode() ? 1 : 0; }

int dqc_device_cu_count(void) { return dqc::device_cus(); }

int dqc_stream_create_partition(void **stream_out, int cu_begin, int cu_end, int priority) {
    // A stream whose kernels run on the compute units [cu_begin, cu_end) of EVERY XCD only (hipExtStreamCreateWithCUMask; the
    // mask's bit i is CU i / nxcd of XCD i % nxcd on the multi-die parts, so a contiguous per-XCD range is a strided bit set).
    // The grid kernels size their launches for the partition (stream_cus).  priority: 0 normal, < 0 higher.
    using namespace dqc;
    if (!stream_out) { set_error("dqc_stream_create_partition: null output"); return DQC_EINVAL; }
    const int ncu = device_cus();
    const int nxcd = ncu >= 64 && ncu % 8 == 0 ? 8 : 1, per = ncu / nxcd;
    if (cu_begin < 0 || cu_end > per || cu_begin >= cu_end) { set_error("dqc_stream_create_partition: CU range outside [0, CUs per XCD]"); return DQC_EINVAL; }
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c = cu_begin; c < cu_end; c++)
        for (int x = 0; x < nxcd; x++) {
            const int bit = c * nxcd + x;
            mask[bit >> 5] |= 1u << (bit & 31);
        }
    hipStream_t st = nullptr;
    (void)priority;  // (hipExtStreamCreateWithCUMask has no priority argument; kept in the ABI for a later driver)
    DQC_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    {
        std::lock_guard<std::mutex> lk(g_cu_mu);
        g_stream_cus.emplace_back(st, (cu_end - cu_begin) * nxcd);
    }
    *stream_out = (void *)st;
    return DQC_OK;
}

int dqc_stream_destroy(void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    {
        std::lock_guard<std::mutex> lk(g_cu_mu);
        for (size_t i = 0; i < g_stream_cus.size(); i++)
            if (g_stream_cus[i].first == st) { g_stream_cus.erase(g_stream_cus.begin() + i); break; }
    }
    DQC_HIP(hipStreamDestroy(st));
    return DQC_OK;
}

int dqc_stream_cus(void *stream) { return dqc::stream_cus((hipStream_t)stream); }

int dqc_set_vxc_cus(int ncu) {
    // process-wide; returns the previous setting (0: no cap)
    const int prev = dqc::vxc_cus_cap();
    dqc::g_vxc_cus.store(ncu > 0 ? ncu : 0);
    return prev;
}



int dqc_set_generic_eri(int on) {
    // process-wide: every shell-quartet class through the runtime-angular-momentum kernel (eri_generic.hpp) instead of only the
    // classes with a g shell -- the cross-check of the two implementations.  Returns the previous setting.
    const int prev = dqc::generic_eri_forced() ? 1 : 0;
    dqc::g_generic_eri.store(on != 0 ? 1 : 0);
    return prev;
}
int dqc_version(void) { return 100; }

int dqc_nao(const int *bas, int nbas) {
    int n = 0;
    for (int i = 0; i < nbas; i++) n += 2 * bas[i * 8 + 1] + 1;
    return n;
}

int dqc_padded_nao(int nao) {
    // rows / columns of the zero-padded AO-indexed SQUARE matrices (D, V, L): whole 16 x 16 MFMA tiles.  (Until round 4 this was
    // also the row stride of the AO-on-grid arrays and had to be == 16 (mod 32) for the LDS fragment reads: benzene / cc-pVDZ,
    // nao 114, lived in 144 columns -- 1.26 x the traffic and 9 tile columns of MFMA work instead of 8.  The kernels now keep
    // their own LDS strides and the arrays their own row stride, dqc_ao_stride.)
    return (nao + 15) / 16 * 16;
}

int dqc_ao_stride(int nao) {
    // row stride (doubles) of the AO-on-grid arrays: nao rounded up to 8 doubles -- 64-byte aligned rows.  Measured on the
    // benzene / naphthalene grids (profiles/r04a_grid_ab.txt): rows at the kernels' minimum alignment (2 doubles = their 16-byte
    // loads) cost the factor-form density kernel 3-5 % although they are 5 % fewer bytes than 8-aligned rows (a 128-byte row
    // segment then straddles cache lines); 8- and 16-aligned rows time the same, and 8 keeps benzene / cc-pVDZ (nao 114) at
    // 1.05 x the algorithmic bytes instead of 1.12 x.  DQC_AO_ALIGN = 2 | 4 | 8 | 16 overrides (A/B runs).
    static const int align = [] {
        const char *e = getenv("DQC_AO_ALIGN");
        const int a = e ? atoi(e) : 8;
        return (a == 2 || a == 4 || a == 8 || a == 16) ? a : 8;
    }();
    // (measured and not it: rows an odd number of 128-byte lines apart -- 432 instead of 416 doubles at nao 412 -- leave the L2
    // traffic of the naphthalene / cc-pVTZ Vxc pass where it is, profiles/r04r_c4_stride.txt)
    return (nao + align - 1) / align * align;
}

size_t dqc_ao_doubles(int ncomp, int ngrid, int nao) {
    // what an AO-on-grid array of ncomp components must hold: the kernels read whole 16-column tiles, i.e. up to
    // dqc_padded_nao - dqc_ao_stride doubles past the end of the last row
    const int over = dqc_padded_nao(nao) - dqc_ao_stride(nao);
    return (size_t)ncomp * (size_t)ngrid * (size_t)dqc_ao_stride(nao) + (size_t)(over > 0 ? over : 0);
}

size_t dqc_eri_store_doubles(int nao) {
    // doubles of the packed tile store (common.hpp): what dqc_eri_fill_tiles writes
    return nao <= 0 ? 0 : (size_t)dqc::eri_store_data_doubles(nao);
}

size_t dqc_eri_tile_count(int nao) {
    size_t nb = (size_t)(nao + DQC_TILE_B - 1) / DQC_TILE_B;
    size_t np = nb * (nb + 1) / 2;
    return np * (np + 1) / 2;
}

// streaming-read probe used by bench.py as the MEASURED HBM read ceiling.  Round 5: the access shape of tools/ubench/read_bw.hip
// that reads fastest on this chip -- contiguous 32 KB tiles per 256-thread block, 16-byte NON-TEMPORAL loads, four tiles' worth of
// loads in flight before the first use, two blocks per CU (6.3-6.4 TB/s on 2 GB, 6.5 on large buffers)
typedef double probe_v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void probe_read_kernel(const probe_v2d *__restrict__ buf, long long ntile, size_t n2, double *out) {
    constexpr int U = 4;
    double s = 0;
    const int t = threadIdx.x;
    for (long long T = (long long)blockIdx.x * U; T < ntile; T += (long long)gridDim.x * U) {
        probe_v2d g[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long TT = T + u < ntile ? T + u : T;
            const probe_v2d *tp = buf + TT * 2048;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const probe_v2d *p = tp + ((4 * (t >> 4) + r) * 32 + 2 * (t & 15));
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
    // (the tail that does not fill a tile)
    for (size_t i = (size_t)ntile * 2048 + (size_t)blockIdx.x * blockDim.x + t; i < n2; i += (size_t)gridDim.x * blockDim.x) s += buf[i].x + buf[i].y;
    // ONE atomic per block (round 6): the 16384 per-wave fp64 atomicAdds of rounds 1-5 to the one address serialise at ~4.6 ns each --
    // 75-80 us per launch, which is what made this probe read 5.2 TB/s on 2 GB and 6.1 on 8 GB and was mistaken for a launch ramp
    // of the chip (tools/ubench/read_shape.hip: the same loop without the atomics reads 6.2-6.4 TB/s on 2 GB)
    __shared__ double part[4];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// fp64 MFMA issue-rate probe: 8 independent 16x16x4 accumulators per wave, `iters` rounds.
// __launch_bounds__(256, 2) matters: with the default bounds the compiler keeps the accumulators in AGPRs and the
// same loop runs at 47.7 TF instead of 77.8 TF (tools/ubench/mfma_probe2.hip); the product kernels are all built for
// >= 2 waves per SIMD, i.e. the VGPR form.
typedef double probe_v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void probe_mfma_kernel(double *out, int iters, double a0, double b0) {
    probe_v4d acc[8];
    for (int i = 0; i < 8; i++) acc[i] = probe_v4d{0, 0, 0, 0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int dqc_probe_mfma_f64(double *d_out, int iters, void *stream) {
    // 256 CUs x 8 waves; d_out needs 512*256 doubles; flops = 2*16*16*4 * 8 * iters * 2048 waves
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, d_out, iters, 1.0, 1e-9);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_probe_stream_read(const double *d_buf, size_t n, double *d_out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    DQC_HIP(hipMemsetAsync(d_out, 0, sizeof(double), st));
    hipLaunchKernelGGL(probe_read_kernel, dim3(2 * dqc::stream_cus(st)), dim3(256), 0, st, (const probe_v2d *)d_buf, (long long)(n / 4096), n / 2, d_out);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
