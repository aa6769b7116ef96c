// common.hpp -- shared host/device helpers of libdqc_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dqc_amd.h"

#define DQC_LMAX 4          // highest shell angular momentum accepted (s,p,d,f,g)
#define DQC_TILE_B 8        // AO block edge of the ERI tile storage
#define DQC_TILE_SZ 4096    // B^4 doubles per tile

namespace dqc {

// ---------------------------------------------------------------------------------------
// Packed 8-fold-unique ERI tile store.  AOs in blocks of 8; block pairs P = (A >= B), index P = A (A + 1) / 2 + B; tile
// (IJ >= KL) holds the sub-tensor g[(i,j)][(k,l)] as R(IJ) rows x C(KL) columns.  A DIAGONAL block pair (A == B) keeps only its
// a >= b elements (36 rows / columns, index a (a + 1) / 2 + b) -- the a < b ones are the same integrals -- any other pair all 64
// (index 8 a + b).  Tiles follow each other in the order (IJ, KL <= IJ); tile (IJ, KL) starts at
//     tile_row_off(I, J) + R(IJ) * (64 KL - 28 K),
//     tile_row_off(I, J) = sum_{P < IJ} R(P) * (64 (P + 1) - 28 (A_P + [A_P == B_P])) = 8 I (64 I^3 + 16 I^2 + 129 I - 47) + 256 J (8 I^2 + I + 8 J + 8)
// (a closed form: a table lookup in front of every tile's loads cost the J + K kernels 6-9 %; the last block row: tile_base below).  For nao = 208 the store is
// 1.895 GB instead of the 2.024 GB of full 8^4 tiles (ideal nao^4 / 8 doubles: 1.872 GB).
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int tile_dim(bool diag) { return diag ? 36 : 64; }
__host__ __device__ inline int tile_pidx(bool diag, int a, int b) {  // local pair (a, b) -> packed row / column (diag: any order)
    if (!diag) return a * 8 + b;
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    return hi * (hi + 1) / 2 + lo;
}
__host__ __device__ inline long long tile_row_off(long long I, long long J) {
    return 8 * I * (64 * I * I * I + 16 * I * I + 129 * I - 47) + 256 * J * (8 * I * I + I + 8 * J + 8);
}
// Round 4: the LAST AO block is stored at its true width.  nao = 8 last + wl (1 <= wl <= 8): a block pair (last, J) has only
// 8 wl (J < last) or wl (wl + 1) / 2 (J == last) rows that belong to AOs -- a PREFIX of its 64 / 36 rows in the pair's own
// element order (8 a + b resp. a (a + 1) / 2 + b with a < wl) -- and only those are stored (benzene / cc-pVDZ, nao 114 = 14 x 8 + 2:
// 0.213 -> 0.172 GB, 1.26 -> 1.02 x nao^4 bytes; naphthalene / cc-pVTZ 30.1 -> 29.1 GB).  The pairs (last, .) are the last ones in
// the order of the store, so every tile in front of them keeps its closed-form offset; COLUMNS are left alone (a padded column
// needs K == last, which only the last block row itself has: 0.1 % of the store).
struct TileLay {
    int last, wl;
    __host__ __device__ TileLay(int nao) : last((nao - 1) >> 3), wl(nao - 8 * ((nao - 1) >> 3)) {}
};
__host__ __device__ inline int tile_rows(int I, int J, const TileLay &ly) {  // stored rows of the tiles of block pair (I, J)
    if (I == ly.last && ly.wl < 8) return I == J ? ly.wl * (ly.wl + 1) / 2 : 8 * ly.wl;
    return tile_dim(I == J);
}
__host__ __device__ inline long long tile_base(int I, int J, int K, int KL, const TileLay &ly) {  // first double of tile ((I, J), KL = (K, .))
    const long long colpre = 64LL * KL - 28LL * K;  // columns of the pairs in front of KL
    if (I < ly.last || ly.wl == 8) return tile_row_off(I, J) + (long long)tile_dim(I == J) * colpre;
    // row (last, J): everything up to (last, 0), then the J truncated pairs (last, J' < J) of 64 (P0 + J' + 1) - 28 last columns each
    const long long L = ly.last, P0 = L * (L + 1) / 2;
    const long long pre = tile_row_off(L, 0) + 8LL * ly.wl * ((64 * (P0 + 1) - 28 * L) * J + 32LL * J * (J - 1));
    return pre + (long long)tile_rows(I, J, ly) * colpre;
}
inline long long eri_store_data_doubles(int nao) {
    if (nao <= 0) return 0;
    const TileLay ly(nao);
    const long long L = ly.last, PL = L * (L + 1) / 2 + L;
    return tile_base(ly.last, ly.last, 0, 0, ly) + (long long)tile_rows(ly.last, ly.last, ly) * (64 * (PL + 1) - 28 * (L + 1));
}

void set_error(const std::string &msg);
bool deterministic_mode();  // dqc_set_deterministic (host.hip)
bool generic_eri_forced();  // dqc_set_generic_eri (host.hip)
int stream_cus(hipStream_t st);  // CUs the stream's kernels may run on (dqc_stream_create_partition; else the device's) (host.hip)

#define DQC_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            dqc::set_error(std::string(#call) + ": " + hipGetErrorString(e_));             \
            return DQC_EHIP;                                                               \
        }                                                                                  \
    } while (0)

#define DQC_CHECK_LAUNCH() DQC_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------
// Host-side view of the libcint tables (reference: dqc/hamilton/intor/lcintwrap.py:37-86)
// ---------------------------------------------------------------------------------------
struct HostShell {
    int l, nprim, atom, ao_off, prim_off;  // prim_off: offset into the flat exps/coefs arrays
    double r[3];
};

struct Basis {
    std::vector<HostShell> shells;
    std::vector<double> exps, coefs;  // flat per-primitive arrays (coefs already radially normalised)
    int nao = 0, natm = 0;
    std::vector<double> atom_xyz, atom_z;
    // grouped view (group_s_shells; eri_core.hpp "General contractions"): `shells` are then GROUPS -- an s group may hold a second
    // member shell over the same exponents: its AO offset (-1: none) and its coefficients (same indexing as coefs; 0: none)
    std::vector<int> ao_off1;
    std::vector<double> coefs1;
    std::vector<int> sh_id0, sh_id1;  // the members' indices in the original shell table (sh_id1 = -1: none)
    bool grouped() const { return !ao_off1.empty(); }
};
// grouped view of a basis: the s shells of one atom that share their exponent list are merged, two per group (host.hip);
// merge = false: every shell its own group (the tables then only have the grouped LAYOUT)
void group_s_shells(const Basis &b, Basis &g, bool merge = true);

// returns 0 or DQC_EINVAL
int parse_basis(Basis &b, const int *atm, int natm, const int *bas, int nbas, const double *env,
                int nenv, const double *zs);

// device mirror of the shell table (SoA, uploaded once per call)
struct DevShells {
    int *l = nullptr, *nprim = nullptr, *ao_off = nullptr, *prim_off = nullptr;
    int *ao_off1 = nullptr;  // grouped view only: AO offset of a group's second member (-1: none)
    double *xyz = nullptr;  // (nsh,3)
    double *exps = nullptr, *coefs = nullptr;
    int nsh = 0;
};

// pinned host staging blocks, recycled once the copy that read them has completed (host.hip)
struct Staging {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int dev = 0;
    bool held = false;  // handed to an open DevPool: its copy has not even been enqueued yet
    bool busy = false;  // an event was recorded behind its copy: free again once that event has completed
};
Staging *staging_acquire(size_t bytes);
void staging_release(Staging *s, hipStream_t st);

// Device scratch of a setup call (shell / pair tables).  Two modes:
//   DevPool pool;       synchronous: hipMalloc / pageable copies / hipFree -- the caller synchronises the stream before return;
//   DevPool pool(st);   stream-ordered: hipMallocAsync, copies staged through recycled pinned blocks, hipFreeAsync -- the call
//                       only ENQUEUES (no device-wide synchronisation by hipFree, no host wait), so the setup of the next
//                       molecule overlaps this one's kernels.
struct DevPool {
    std::vector<void *> ptrs;
    std::vector<Staging *> stg;
    hipStream_t ast = nullptr;
    bool async = false;
    DevPool() {}
    explicit DevPool(hipStream_t st) : ast(st), async(true) {}
    DevPool(const DevPool &) = delete;
    DevPool &operator=(const DevPool &) = delete;
    int dmalloc(void **p, size_t bytes) {
        hipError_t e = async ? hipMallocAsync(p, bytes ? bytes : 8, ast) : hipMalloc(p, bytes ? bytes : 8);
        if (e != hipSuccess) return DQC_ENOMEM;
        ptrs.push_back(*p);
        return 0;
    }
    template <typename T>
    int upload(T **dst, const std::vector<T> &src, hipStream_t st) {
        size_t bytes = src.size() * sizeof(T);
        void *p = nullptr;
        if (dmalloc(&p, bytes)) return DQC_ENOMEM;
        if (bytes) {
            const void *from = src.data();
            if (async) {
                Staging *s = staging_acquire(bytes);
                if (!s) return DQC_ENOMEM;
                stg.push_back(s);
                std::memcpy(s->host, src.data(), bytes);
                from = s->host;
            }
            if (hipMemcpyAsync(p, from, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return DQC_EHIP;
        }
        *dst = (T *)p;
        return 0;
    }
    // host vector -> caller-owned device memory (staged through a pinned block in the stream-ordered mode)
    template <typename T>
    int copy_to(T *d_dst, const std::vector<T> &src, hipStream_t st) {
        const size_t bytes = src.size() * sizeof(T);
        if (!bytes) return 0;
        const void *from = src.data();
        if (async) {
            Staging *s = staging_acquire(bytes);
            if (!s) return DQC_ENOMEM;
            stg.push_back(s);
            std::memcpy(s->host, src.data(), bytes);
            from = s->host;
        }
        return hipMemcpyAsync(d_dst, from, bytes, hipMemcpyHostToDevice, st) == hipSuccess ? 0 : DQC_EHIP;
    }
    template <typename T>
    int alloc(T **dst, size_t n) {
        void *p = nullptr;
        if (dmalloc(&p, n * sizeof(T) + 8)) return DQC_ENOMEM;
        *dst = (T *)p;
        return 0;
    }
    void release() {
        for (void *p : ptrs) (void)(async ? hipFreeAsync(p, ast) : hipFree(p));
        ptrs.clear();
        for (Staging *s : stg) staging_release(s, ast);
        stg.clear();
    }
    ~DevPool() { release(); }
};

int upload_shells(DevShells &d, const Basis &b, DevPool &pool, hipStream_t st);

inline int ncart(int l) { return (l + 1) * (l + 2) / 2; }

// Side streams for the class launches of one ERI pass (fill, direct J / K, gradient).  A pass is ~55 launches of 0.2-3 ms, each
// latency-bound with a long tail of deep quartets: dealt round-robin to a few streams, the next class fills the chip while the
// last waves of the previous one drain (naphthalene / cc-pVTZ direct Coulomb pass 67.7 -> 63.7 ms with three streams).
// fork(st): the side streams wait for what `st` holds so far; join(st): `st` waits for all of them.  One set per device and
// process (DQC_SIDE_STREAMS = 0 / 1: none, default 3, at most 4); calls that share it are ordered by the events alone.
struct SideStreams {
    static constexpr int MAX = 4;
    hipStream_t s[MAX] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX] = {};
    int n = 0, next = 0;
    hipStream_t take() { return s[next++ % n]; }
    int fork(hipStream_t st) {
        if (hipEventRecord(ev_fork, st) != hipSuccess) return DQC_EHIP;
        for (int i = 0; i < n; i++)
            if (hipStreamWaitEvent(s[i], ev_fork, 0) != hipSuccess) return DQC_EHIP;
        return DQC_OK;
    }
    int join(hipStream_t st) {
        for (int i = 0; i < n; i++)
            if (hipEventRecord(ev_join[i], s[i]) != hipSuccess || hipStreamWaitEvent(st, ev_join[i], 0) != hipSuccess) return DQC_EHIP;
        return DQC_OK;
    }
};
// joins the side streams back into `st` on EVERY exit after a successful fork: an early error return would otherwise hand the
// stream-ordered scratch of the call (DevPool) back while launches queued on the side streams may still read it, and leave the
// caller's stream unordered behind them.  done(): the normal join, with its return code.  (One set of streams per device and process,
// round-robin counter unsynchronised: the entry points that use them are meant for one host thread per device.)
struct SideJoin {
    SideStreams *s = nullptr;
    hipStream_t st = nullptr;
    void arm(SideStreams *s_, hipStream_t st_) { s = s_; st = st_; }
    int done() {
        SideStreams *t = s;
        s = nullptr;
        return t ? t->join(st) : DQC_OK;
    }
    ~SideJoin() { if (s) (void)s->join(st); }
};
inline SideStreams *side_streams() {  // nullptr: switched off, or the streams could not be created
    constexpr int MAXDEV = 16;
    static SideStreams pool[MAXDEV];
    static int state[MAXDEV] = {};  // 0: not tried, 1: there, -1: none
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev] == 0) {
        const char *e = getenv("DQC_SIDE_STREAMS");
        int want = e ? atoi(e) : 3;
        want = want < 2 ? 0 : (want > SideStreams::MAX ? SideStreams::MAX : want);
        SideStreams &p = pool[dev];
        bool ok = want > 0 && hipEventCreateWithFlags(&p.ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < want; i++)
            ok = hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&p.ev_join[i], hipEventDisableTiming) == hipSuccess;
        p.n = ok ? want : 0;
        state[dev] = ok ? 1 : -1;
    }
    return state[dev] == 1 ? &pool[dev] : nullptr;
}

}  // namespace dqc

// ---------------------------------------------------------------------------------------
// device-side tables
// ---------------------------------------------------------------------------------------
#ifdef __HIPCC__
#define DQC_DEV __device__ __forceinline__

// Accumulation into a shared fp64 slot from many blocks.  scale == 0: fp64 atomic add (order of the additions, hence the last
// bits, vary from run to run).  scale = 2^k: DETERMINISTIC mode -- the contribution is rounded to a multiple of 2^-k and added
// as a 64-bit integer; integer addition is associative, so the sum is bit-identical whatever the order (and, two's
// complement being modular, only the FINAL sum has to fit 63 bits).  The reader converts with det_value().
__device__ __forceinline__ void acc_add(double *p, double v, double scale) {
    if (scale == 0.0) atomicAdd(p, v);
    else atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double2ll_rn(v * scale));
}
__device__ __forceinline__ double det_value(double stored, double scale) {
    return scale == 0.0 ? stored : (double)__double_as_longlong(stored) / scale;
}

// real solid harmonics (generated, libcint convention)
#define C2S_QUAL __device__ const
#include "cart2sph.inc"
#undef C2S_QUAL

#define RYS_QUAL __device__ const
#include "rys_tables.inc"
#undef RYS_QUAL

namespace dqc {

template <int L>
struct Cart {
    static constexpr int n = (L + 1) * (L + 2) / 2;
};

// powers (lx,ly,lz) of Cartesian component c of shell l (libcint order), computed arithmetically
DQC_DEV void cart_pow(int l, int c, int &lx, int &ly, int &lz) {
    // rows: lx = l, l-1, ... ; row with lx has (l-lx+1) entries
    int row = 0, acc = 0;
    while (acc + row + 1 <= c) { acc += row + 1; row++; }
    lx = l - row;
    int k = c - acc;
    ly = row - k;
    lz = k;
}

// Rys roots u[r] (= t^2) and weights w[r] for given X = rho*|PQ|^2
template <int N>
DQC_DEV void rys_roots(double X, double *u, double *w) {
    constexpr double XMAX = 35.0 + 5.0 * N;
    if (X >= XMAX) {
        double ix = 1.0 / X, isx = sqrt(ix);
#pragma unroll
        for (int r = 0; r < N; r++) {
            u[r] = RYS_HERM_X2[N][r] * ix;
            w[r] = RYS_HERM_W[N][r] * isx;
        }
        return;
    }
    int it = (int)(X * (1.0 / 2.5));
    double x = (X - (it * 2.5 + 1.25)) * (1.0 / 1.25);
    const double *tab = RYS_TAB + RYS_OFF[N - 1] + (size_t)it * (2 * N) * (RYS_DEG + 1);
    double x2 = 2.0 * x;
#pragma unroll
    for (int q = 0; q < 2 * N; q++) {
        const double *c = tab + q * (RYS_DEG + 1);
        // Clenshaw
        double b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int k = RYS_DEG; k >= 1; k--) {
            double t = x2 * b1 - b2 + c[k];
            b2 = b1;
            b1 = t;
        }
        double v = x * b1 - b2 + c[0];
        if (q < N) u[q] = v; else w[q - N] = v;
    }
}

// single root/weight r of the N-point rule from an LDS copy of the N table in the layout of rys_stage_lds: per (interval,
// root) one row of (u-coefficient, w-coefficient) pairs at an odd row stride (29 doubles: rows of different intervals fall
// into different banks).  Different lanes look up different intervals -- a gather; as ds_read_b128 pairs from LDS instead of
// 28 uncoalesced global loads per call.
constexpr int RYS_LDS_ROW = 2 * (RYS_DEG + 1) + 1;
template <int N>
constexpr int rys_lds_doubles() { return (14 + 2 * N) * N * RYS_LDS_ROW; }

template <int N>
DQC_DEV void rys_stage_lds(__attribute__((address_space(3))) double *lt, int tid, int nthreads) {
    const double *src = RYS_TAB + RYS_OFF[N - 1];
    constexpr int NI = 14 + 2 * N;
    for (int e = tid; e < NI * N * (RYS_DEG + 1); e += nthreads) {
        const int k = e % (RYS_DEG + 1), r = (e / (RYS_DEG + 1)) % N, it = e / ((RYS_DEG + 1) * N);
        const double *row = src + (size_t)it * (2 * N) * (RYS_DEG + 1);
        lt[(it * N + r) * RYS_LDS_ROW + 2 * k] = row[r * (RYS_DEG + 1) + k];
        lt[(it * N + r) * RYS_LDS_ROW + 2 * k + 1] = row[(N + r) * (RYS_DEG + 1) + k];
    }
}

template <int N>
DQC_DEV void rys_root1_lds(const __attribute__((address_space(3))) double *lt, double X, int r, double &u, double &w) {
    constexpr double XMAX = 35.0 + 5.0 * N;
    if (X >= XMAX) {
        double ix = 1.0 / X;
        u = RYS_HERM_X2[N][r] * ix;
        w = RYS_HERM_W[N][r] * sqrt(ix);
        return;
    }
    int it = (int)(X * (1.0 / 2.5));
    double x = (X - (it * 2.5 + 1.25)) * (1.0 / 1.25);
    const __attribute__((address_space(3))) double *c = lt + (it * N + r) * RYS_LDS_ROW;
    double x2 = 2.0 * x, a1 = 0, a2 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int k = RYS_DEG; k >= 1; k--) {
        double t = x2 * a1 - a2 + c[2 * k]; a2 = a1; a1 = t;
        double s = x2 * b1 - b2 + c[2 * k + 1]; b2 = b1; b1 = s;
    }
    u = x * a1 - a2 + c[0];
    w = x * b1 - b2 + c[1];
}

// single root/weight r (runtime) -- used when different lanes need different roots
template <int N>
DQC_DEV void rys_root1(double X, int r, double &u, double &w) {
    constexpr double XMAX = 35.0 + 5.0 * N;
    if (X >= XMAX) {
        double ix = 1.0 / X;
        u = RYS_HERM_X2[N][r] * ix;
        w = RYS_HERM_W[N][r] * sqrt(ix);
        return;
    }
    int it = (int)(X * (1.0 / 2.5));
    double x = (X - (it * 2.5 + 1.25)) * (1.0 / 1.25);
    const double *cu = RYS_TAB + RYS_OFF[N - 1] + ((size_t)it * (2 * N) + r) * (RYS_DEG + 1);
    const double *cw = cu + N * (RYS_DEG + 1);
    double x2 = 2.0 * x, a1 = 0, a2 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int k = RYS_DEG; k >= 1; k--) {
        double t = x2 * a1 - a2 + cu[k]; a2 = a1; a1 = t;
        double s = x2 * b1 - b2 + cw[k]; b2 = b1; b1 = s;
    }
    u = x * a1 - a2 + cu[0];
    w = x * b1 - b2 + cw[0];
}

}  // namespace dqc
#endif
