// eri.hip -- two-electron repulsion integrals (ij|kl) over contracted real-spherical Gaussians.
// Replaces  GTOnr2e_fill_drv(int2e_sph, GTOnr2e_fill_s4, prescreen=NULL)  + fills4
// (reference call sites dqc/hamilton/intor/molintor.py:667-688, symmetry.py:40-69).
//
// MI355X design
//   * shell pairs are sorted by angular-momentum class (la>=lb) and, inside a class, by the number of
//     surviving primitive pairs, so every launch is one template instantiation <LA,LB,LC,LD> with
//     uniform loop shapes and coalesced reads of the pair tables;
//   * Rys quadrature (roots from Chebyshev tables, rys_tables.inc): per primitive quartet the 2D
//     integrals Ix,Iy,Iz (one (direction, root) item per lane) are staged in LDS, then every lane of
//     the quartet's lane group accumulates its share of the Cartesian outputs in registers;
//   * TPQ lanes cooperate on one shell quartet (1 for (ss|ss)-like classes ... 256 for (ff|ff)), so the
//     register footprint per lane stays ~20-40 accumulators for every class;
//   * after the primitive loops the Cartesian block is transformed to real solid harmonics in LDS and
//     scattered straight into the 8-fold-unique TILE storage the J/K kernels stream (no nao^4 tensor).
//   * the NR roots of a primitive quartet depend on X only: lane s of the quartet's lane group evaluates root s % NR once and the
//     3 NR (direction, root) items fetch theirs by shuffle (round 1 evaluated the Clenshaw series per item: three times the
//     work in groups of 1 or 4 lanes).  20-atom cc-pVDZ fill 24.7 -> 20.1 ms, benzene 12.4 -> 9.6, CH4 / cc-pVTZ 15.9 -> 12.2;
//   * the lanes of a quartet synchronise at wave level (eri_group_sync: a lane group of <= 64 lanes lies inside one wave) and a
//     wave runs until ITS longest quartet is done -- no block barrier, no block-uniform primitive-quartet count; the
//     recurrence coefficients use reciprocals formed once per primitive quartet instead of five fp64 divisions per item:
//     20.2 -> 18.6 ms;
//     (tried: incremental (bra pair, ket pair) counters with the bra pair's data held in registers instead of the per-iteration
//     integer division and re-read -- 17.85 -> 21.5 ms of kernel time: the extra live registers cost more than the division);
//   * classes with one or two Rys roots keep their root table in LDS (rys_stage_lds: (u, w) coefficient pairs per row, odd row
//     stride): lanes of a wave work on different primitive quartets, so a root lookup is a gather -- 28 uncoalesced global
//     loads per (direction, root) item, one VMEM read per 8.6 VALU instructions in (ps|ss).  20-atom cc-pVDZ fill 27.8 ->
//     24.7 ms, naphthalene / cc-pVTZ 157 -> 153 ms; a generic (flat) pointer to the copy made it 42.7 ms, the 14 KB table of
//     the three-root classes costs more occupancy than it saves.  PMC (profiles/r02u_*): the kernels are VALU-dependency
//     bound (48 % of the wave cycles in issue stalls: Clenshaw recurrences, fp64 divisions), ~35 % VALU utilisation.
//   The stored fill screens no shell quartet (the reference passes prescreen = NULL); primitive pairs whose prefactor cannot
//   reach 1e-20 are dropped when the pair tables are built (eri_core.hpp: prim_pair_negligible).  The DIRECT path (nothing
//   stored, every build re-evaluates) does screen: Cauchy-Schwarz bounds, density-weighted, see the dqc_direct_* context below.
#include <algorithm>
#include <memory>

#include "eri_generic.hpp"

namespace dqc {

// ---------------------------------------------------------------------------------------------
// expansion of the tile storage to the reference's dense tensor (tests / tiny systems)
// ---------------------------------------------------------------------------------------------
__global__ void tiles_to_dense_kernel(double *__restrict__ dense, const double *__restrict__ tiles, int nao) {
    const size_t n = nao, total = n * n * n * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int l = e % n, k = (e / n) % n, j = (e / (n * n)) % n, i = e / (n * n * n);
        int a = i, b = j, c = k, d = l;
        if ((a >> 3) < (b >> 3)) { int t = a; a = b; b = t; }
        if ((c >> 3) < (d >> 3)) { int t = c; c = d; d = t; }
        int IJ = (a >> 3) * ((a >> 3) + 1) / 2 + (b >> 3), KL = (c >> 3) * ((c >> 3) + 1) / 2 + (d >> 3);
        if (IJ < KL) { int t = a; a = c; c = t; t = b; b = d; d = t; t = IJ; IJ = KL; KL = t; }
        const int A = a >> 3, B = b >> 3, Cb = c >> 3, Db = d >> 3;
        dense[e] = tiles[tile_base(A, B, Cb, KL, TileLay(nao)) + (long long)tile_pidx(A == B, a & 7, b & 7) * tile_dim(Cb == Db) + tile_pidx(Cb == Db, c & 7, d & 7)];
    }
}

// ---------------------------------------------------------------------------------------------
// class dispatch
// ---------------------------------------------------------------------------------------------
// the pool the per-launch wave tables of a fill are uploaded through
struct WaveMaps {
    DevPool *pool = nullptr;      // stream-ordered pool of the wave tables (nullptr: flat task maps)
    SideStreams *side = nullptr;  // class launches dealt round-robin to the side streams (common.hpp)
};

// One class launch of the fill.  Bra and ket pairs come from their own tables: the same one (a whole-store fill: diagonal classes
// keep bra pair >= ket pair) or, for a slice of a store spread over several GPUs (dqc_eri_fill_tiles_part), the pairs whose block
// pairs lie inside / below the slice's rows -- then every (bra, ket) combination of the two lists is wanted (pairing FILL_CROSS)
// and the diagonal classes of the second cross launch are left out (FILL_CROSS_OFFDIAG: the first one covered them)
enum { FILL_SAME_TABLE = 0, FILL_CROSS = 1, FILL_CROSS_OFFDIAG = 2 };
struct FillTables {
    const DevPairs *dpb, *dpk;
    const HostPairs *hpb, *hpk;
    int pairing;
};

template <int LA, int LB, int LC, int LD>
static int launch_class(double *tiles, const DevShells &ds, const FillTables &ft, hipStream_t st, const EriOut &og,
                        const WaveMaps *wm = nullptr) {
    using Cfg = EriCfg<LA, LB, LC, LD>;
    const int cb = LA * (LA + 1) / 2 + LB, ck = LC * (LC + 1) / 2 + LD;
    const HostPairs &hpb = *ft.hpb, &hpk = *ft.hpk;
    const DevPairs &dpb = *ft.dpb, &dpk = *ft.dpk;
    const int nb = hpb.cls_count[cb], nk = hpk.cls_count[ck];
    if (nb == 0 || nk == 0 || hl_forced()) return 0;
    if (ft.pairing == FILL_CROSS_OFFDIAG && cb == ck) return 0;
    const int same = (cb == ck && ft.pairing == FILL_SAME_TABLE) ? 1 : 0;
    const long long ntask = same ? (long long)nb * (nb + 1) / 2 : (long long)nb * nk;
    const long long nblk = eri_num_blocks<Cfg>(nb, nk, ntask);
    // grouped tables (general contractions merged: eri_core.hpp): the instantiation with the coefficient slots of the two pair classes
    constexpr int NPB = PairSlots<LA, LB>::N, NPK = PairSlots<LC, LD>::N;
    if (dpb.stride != PP_STRIDE_G || dpk.stride != PP_STRIDE_G) { set_error("eri fill: pair tables without the grouped layout"); return DQC_EINVAL; }
    auto kern = eri_kernel<LA, LB, LC, LD, ERI_OUT_TILES, NPB, NPK>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    // lane groups of <= 16 lanes: depth-binned wave map (eri_core.hpp: eri_split_lanes, eri_wave_table) -- where the class pair has
    // contractions deep enough to be split at all (pairs are sorted by depth: the first pair of a class is its deepest); the
    // single-primitive classes of a cc-pVTZ fill are 1-2 % faster under the flat map
    const int dmax = (hpb.pp_off[hpb.cls_start[cb] + 1] - hpb.pp_off[hpb.cls_start[cb]]) * (hpk.pp_off[hpk.cls_start[ck] + 1] - hpk.pp_off[hpk.cls_start[ck]]);
    if (wm != nullptr && wm->side != nullptr) st = wm->side->take();  // (the wave table's upload and the launch on the same stream)
    if (Cfg::TPQ <= 16 && wm != nullptr && wm->pool != nullptr && dmax > 16) {
        EriOut o2 = og;
        long long nwave;
        if (same) {  // diagonal class (bra pair >= ket pair): one table entry per wave
            std::vector<int2> wtab;
            eri_wave_table(wtab, o2.wbin, hpb, hpb.cls_start[cb], nb, hpk, hpk.cls_start[ck], nk, true, Cfg::TPQ);
            nwave = (long long)wtab.size();
            if (nwave == 0) return 0;
            int2 *d_wtab = nullptr;
            if (wm->pool->upload(&d_wtab, wtab, st)) { set_error("dqc_eri_fill_tiles: device upload failed"); return DQC_ENOMEM; }
            o2.wtab = d_wtab;
        } else {     // runs of ket pairs of equal depth
            std::vector<WaveRun> runs;
            nwave = eri_wave_runs(runs, o2.wbin, hpb, hpb.cls_start[cb], nb, hpk, hpk.cls_start[ck], nk, Cfg::TPQ);
            if (nwave == 0) return 0;
            WaveRun *d_runs = nullptr;
            if (wm->pool->upload(&d_runs, runs, st)) { set_error("dqc_eri_fill_tiles: device upload failed"); return DQC_ENOMEM; }
            o2.wruns = d_runs;
            o2.nruns = (int)runs.size();
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)((nwave + 3) / 4)), dim3(256), Cfg::LDS_BYTES, st, tiles, ds, dpb, dpk, hpb.cls_start[cb], nb,
                           hpk.cls_start[ck], nk, same, nwave, o2);
        DQC_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES, st, tiles, ds, dpb, dpk, hpb.cls_start[cb], nb,
                       hpk.cls_start[ck], nk, same, ntask, og);
    DQC_CHECK_LAUNCH();
    return 0;
}

// all classes with (LA>=LB), (LC>=LD), class(bra) >= class(ket)
template <int CB, int CK>
struct ClassLoop {
    static int run(double *tiles, const DevShells &ds, const FillTables &ft, hipStream_t st, const EriOut &og, const WaveMaps *wm = nullptr) {
        constexpr int LA = CB < 1 ? 0 : (CB < 3 ? 1 : (CB < 6 ? 2 : 3)), LB = CB - LA * (LA + 1) / 2;
        constexpr int LC = CK < 1 ? 0 : (CK < 3 ? 1 : (CK < 6 ? 2 : 3)), LD = CK - LC * (LC + 1) / 2;
        int rc = launch_class<LA, LB, LC, LD>(tiles, ds, ft, st, og, wm);
        if (rc) return rc;
        if constexpr (CK > 0) return ClassLoop<CB, CK - 1>::run(tiles, ds, ft, st, og, wm);
        else if constexpr (CB > 0) return ClassLoop<CB - 1, CB - 1>::run(tiles, ds, ft, st, og, wm);
        else return 0;
    }
};

// ---------------------------------------------------------------------------------------------
// direct SCF: J / K straight from the shell quartets (nothing stored)
// ---------------------------------------------------------------------------------------------
// screened task maps of one J / K pass (direct-SCF context below): per class pair (cb >= ck) the slice of the device array of
// prefix offsets (EriOut::toff) and the number of surviving tasks (waves, for the one-lane-per-quartet classes)
constexpr int NCLS_ALL = (DQC_LMAX + 1) * (DQC_LMAX + 2) / 2;
struct DirectCtx;
struct ScreenPlan {
    const long long *d_toff = nullptr;
    const int *d_bins = nullptr;  // (NCLS_ALL, SCREEN_NBIN + 1) bin starts of every pair class, relative to the class start
    long long start[NCLS_ALL][NCLS_ALL], total[NCLS_ALL][NCLS_ALL];
    // the entries of a class pair are computed and uploaded right before its launch (plan_screen_pair): the device works on the first
    // class pairs while the host plans the later ones -- planning everything first left it idle for ~1 ms per pass
    DirectCtx *ctx = nullptr;
    const double *dl = nullptr;  // 5 x 5 density maxima by angular momentum (or nullptr: tc_all)
    double tc_all = 0.0, tau = 0.0;
    bool with_k = false;
    hipStream_t st = nullptr;
    bool upload = false;  // upload every planned slice on `st` (which may be the null stream)
    // class launches dealt round-robin to the side streams (common.hpp: SideStreams; forked from / joined to the caller's stream)
    SideStreams *side = nullptr;
};
static int plan_screen_pair(ScreenPlan &sp, int cb, int ck);

template <int LA, int LB, int LC, int LD>
static int launch_class_jk(const DevShells &ds, const DevPairs &dp, const HostPairs &hp, const EriOut &og, hipStream_t st,
                           ScreenPlan *sp = nullptr) {
    using Cfg = EriCfg<LA, LB, LC, LD>;
    const int cb = LA * (LA + 1) / 2 + LB, ck = LC * (LC + 1) / 2 + LD;
    const int nb = hp.cls_count[cb], nk = hp.cls_count[ck];
    if (nb == 0 || nk == 0 || hl_forced()) return 0;
    const int same = cb == ck;
    long long ntask = same ? (long long)nb * (nb + 1) / 2 : (long long)nb * nk;
    long long nblk = eri_num_blocks<Cfg>(nb, nk, ntask);
    EriOut o2 = og;
    if (sp) {
        if (sp->side) st = sp->st = sp->side->take();  // (slice upload and launch on the same stream)
        if (int rc = plan_screen_pair(*sp, cb, ck)) return rc;
        ntask = sp->total[cb][ck];
        if (ntask == 0) return 0;
        o2.toff = sp->d_toff + sp->start[cb][ck];
        // (one-lane classes: entries per ket pair, tasks = waves -- unless the class takes the bra-uniform entries, eri_core.hpp FLIP1)
        const bool ket_entries = Cfg::TPQ == 1 && !(og.flip1 && Cfg::SA * Cfg::SB > Cfg::SC * Cfg::SD);
        o2.pbin = sp->d_bins + (ket_entries ? cb : ck) * (SCREEN_NBIN + 1);  // bins of the PARTNER list the prefixes run over
        nblk = ket_entries ? (ntask + 3) / 4 : (ntask + Cfg::QPB - 1) / Cfg::QPB;
    }
    nblk = (nblk + o2.nparts - 1) / o2.nparts;  // this rank's share of the blocks (dqc_direct_jk_part)
    // Coulomb only (Kohn-Sham): the mode without the exchange accumulators in LDS (eri_core.hpp: ERI_OUT_J)
    const bool jonly = o2.kacc == nullptr;
    if (dp.stride != PP_STRIDE_G) { set_error("direct J / K: pair tables without the grouped layout"); return DQC_EINVAL; }
    constexpr int NPB = PairSlots<LA, LB>::N, NPK = PairSlots<LC, LD>::N;
    auto kern = jonly ? eri_kernel<LA, LB, LC, LD, ERI_OUT_J, NPB, NPK> : eri_kernel<LA, LB, LC, LD, ERI_OUT_JK, NPB, NPK>;
    const size_t lds_bytes = Cfg::LDS_BYTES;  // (both direct modes digest the block in place: the fill's LDS footprint)
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds_bytes, st, (double *)nullptr, ds, dp, dp, hp.cls_start[cb],
                       nb, hp.cls_start[ck], nk, same, ntask, o2);
    DQC_CHECK_LAUNCH();
    return 0;
}

template <int CB, int CK>
struct ClassLoopJK {
    static int run(const DevShells &ds, const DevPairs &dp, const HostPairs &hp, const EriOut &og, hipStream_t st,
                   ScreenPlan *sp = nullptr) {
        constexpr int LA = CB < 1 ? 0 : (CB < 3 ? 1 : (CB < 6 ? 2 : 3)), LB = CB - LA * (LA + 1) / 2;
        constexpr int LC = CK < 1 ? 0 : (CK < 3 ? 1 : (CK < 6 ? 2 : 3)), LD = CK - LC * (LC + 1) / 2;
        int rc = launch_class_jk<LA, LB, LC, LD>(ds, dp, hp, og, st, sp);
        if (rc) return rc;
        if constexpr (CK > 0) return ClassLoopJK<CB, CK - 1>::run(ds, dp, hp, og, st, sp);
        else if constexpr (CB > 0) return ClassLoopJK<CB - 1, CB - 1>::run(ds, dp, hp, og, st, sp);
        else return 0;
    }
};

// Schwarz bounds: the diagonal quartets (ab|ab) of every pair class (task map `same` = 2), max |.| into d_q[pair]
template <int CB>
struct ClassLoopSchwarz {
    static int run(double *d_q, const DevShells &ds, const DevPairs &dp, const HostPairs &hp, hipStream_t st) {
        constexpr int LA = CB < 1 ? 0 : (CB < 3 ? 1 : (CB < 6 ? 2 : 3)), LB = CB - LA * (LA + 1) / 2;
        using Cfg = EriCfg<LA, LB, LA, LB>;
        const int nb = hp.cls_count[CB];
        if (nb > 0 && !hl_forced()) {
            constexpr int NP = PairSlots<LA, LB>::N;
            auto kern = eri_kernel<LA, LB, LA, LB, ERI_OUT_SCHWARZ, NP, NP>;  // d_q: FOUR slots per pair (member pairs of a grouped pair)
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
            const long long nblk = ((long long)nb + Cfg::QPB - 1) / Cfg::QPB;
            hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES, st, d_q, ds, dp, dp, hp.cls_start[CB], nb,
                               hp.cls_start[CB], nb, 2, (long long)nb, EriOut{0, 0, 0, 0});
            DQC_CHECK_LAUNCH();
        }
        if constexpr (CB > 0) return ClassLoopSchwarz<CB - 1>::run(d_q, ds, dp, hp, st);
        else return 0;
    }
};

// the classes the compile-time kernels leave out -- any pair class with a g shell -- through the runtime kernel
// (eri_generic.hpp); DQC_ERI_GENERIC=1: all classes
template <int MODE>
static int run_generic_classes(double *tiles, const DevShells &ds, const DevPairs &dp, const HostPairs &hp, const EriOut &og,
                               hipStream_t st, ScreenPlan *sp = nullptr) {
    for (int la = 0; la <= DQC_LMAX; la++)
        for (int lb = 0; lb <= la; lb++)
            for (int lc = 0; lc <= la; lc++)
                for (int ld = 0; ld <= lc; ld++) {
                    const int cb = la * (la + 1) / 2 + lb, ck = lc * (lc + 1) / 2 + ld;
                    if (ck > cb) continue;
                    if (!hl_forced() && la <= ERI_LMAX && lc <= ERI_LMAX) continue;
                    if (MODE == ERI_OUT_SCHWARZ && ck != cb) continue;  // diagonal quartets only
                    EriOut o2 = og;
                    long long nscr = -1;
                    if (sp) {
                        sp->st = st;
                        if (int rc = plan_screen_pair(*sp, cb, ck)) return rc;
                        nscr = sp->total[cb][ck];
                        o2.toff = sp->d_toff + sp->start[cb][ck];
                        o2.pbin = sp->d_bins + ck * (SCREEN_NBIN + 1);
                    }
                    int rc = launch_hl<MODE>(tiles, ds, dp, dp, hp.cls_start[cb], hp.cls_count[cb], hp.cls_start[ck],
                                             hp.cls_count[ck], MODE == ERI_OUT_SCHWARZ ? 2 : (cb == ck), o2, la, lb, lc, ld, st, nscr);
                    if (rc) return rc;
                }
    return 0;
}

__global__ void jk_direct_prep_kernel(double *__restrict__ dsym, double *__restrict__ a, double *__restrict__ b,
                                      const double *__restrict__ dm, int nao) {
    const size_t n2 = (size_t)nao * nao;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (size_t)gridDim.x * blockDim.x) {
        const int i = e / nao, j = e % nao;
        dsym[e] = 0.5 * (dm[e] + dm[(size_t)j * nao + i]);
        a[e] = 0.0;
        if (b) b[e] = 0.0;
    }
}

__global__ void jk_direct_finish_kernel(double *__restrict__ J, double *__restrict__ K, const double *__restrict__ a,
                                        const double *__restrict__ b, int nao) {
    const size_t n2 = (size_t)nao * nao;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (size_t)gridDim.x * blockDim.x) {
        const int i = e / nao, j = e % nao;
        const size_t et = (size_t)j * nao + i;
        J[e] = 0.5 * (a[e] + a[et]);
        if (K) K[e] = b[e] + b[et];
    }
}


// ---------------------------------------------------------------------------------------------
// direct-SCF context: pair tables and Schwarz bounds resident on the device, screened J / K passes
// ---------------------------------------------------------------------------------------------
// Round-3 direct builds re-parsed the basis and re-built / re-uploaded the pair tables in every call and evaluated every unique
// shell quartet.  The context keeps the tables on the device, holds Q_ab = sqrt(max |(ab|ab)|) for every shell pair (one pass
// over the diagonal quartets at creation) and sorts the pairs of every class by Q, descending.  A J / K pass of a density D
// then launches, per class pair, only the tasks with Q_ab Q_cd 4 max|D| >= tau -- for a bra pair these are a PREFIX of the ket
// list, so the surviving tasks are described by one prefix-offset array (two-pointer sweep on the host, O(pairs) per class
// pair) and the kernel finds its bra pair by bisection; inside the kernel the same test runs per quartet with the maxima of
// the shell-pair density blocks that quartet actually touches.  tau = 0: no screening (every quartet, as dqc_jk_direct).
// J and K are linear in D, so the SCF driver hands over density DIFFERENCES (hamilton.py): as the SCF converges max|dD| falls
// and the screened share of the quartets with it.
struct DirectCtx {
    Basis b;                // the caller's shells
    Basis bg;               // grouped view the kernels work on (general contractions merged: eri_core.hpp; trivial groups with g shells)
    HostPairs hp;           // GROUP pairs, Schwarz-sorted inside every class
    std::vector<double> q;  // bound of every group pair (max over its member shell pairs), table order
    std::vector<double> qm; // (npair, 4): bound of every member shell pair (slot 2 xa + xb, or xb for (l s| pairs; 0: absent)
    std::vector<long long> wpre;  // prefix sums of the pairs' weights (= member shell pairs per group pair), table order
    DevPool pool;           // synchronous pool: owns the context's device memory until destroy
    DevShells ds;
    DevPairs dp{nullptr, nullptr, nullptr};
    double *d_q = nullptr, *d_dsh = nullptr, *d_sym = nullptr, *d_a = nullptr, *d_b = nullptr, *d_dmax = nullptr;
    long long *d_toff = nullptr;
    int *d_bins = nullptr;
    int split_per = 16;     // (plan_screen / EriOut::split_per; DQC_ERI_SPLIT_PER)
    int flip1 = 1;          // (plan_screen / EriOut::flip1; DQC_ERI_FLIP1=0: A/B runs)
    std::vector<int> bins;  // (NCLS_ALL, SCREEN_NBIN + 1): starts of the contraction-depth bins of every class (relative)
    // offsets of the screened task maps, PINNED: every class pair's slice is uploaded on its own in front of its launch (plan_screen_pair),
    // asynchronously
    long long *h_toff = nullptr;
    size_t n_toff = 0;
    std::vector<long long> cnt_scratch;
    ~DirectCtx() { if (h_toff) (void)hipHostFree(h_toff); }
    long long stat_total = 0, stat_launched = 0;  // unique quartets / quartets launched, last pass
    double stat_dmax = 0.0;
};

static void class_l(int c, int &la, int &lb) {
    la = 0;
    while ((la + 1) * (la + 2) / 2 <= c) la++;
    lb = c - la * (la + 1) / 2;
}

// pairs of every class re-ordered by contraction-depth bin (deepest first), then by their Schwarz bound, descending;
// bins: (NCLS_ALL, SCREEN_NBIN + 1) bin starts relative to the class start
static void sort_pairs_by_bound(HostPairs &hp, std::vector<double> &q, std::vector<int> &bins, std::vector<double> *qm = nullptr) {
    const size_t np = q.size();
    std::vector<int> perm(np);
    for (size_t i = 0; i < np; i++) perm[i] = (int)i;
    auto binof = [&](int x) { return screen_bin(hp.pp_off[x + 1] - hp.pp_off[x]); };
    bins.assign((size_t)NCLS_ALL * (SCREEN_NBIN + 1), 0);
    for (int c = 0; c < NCLS_ALL; c++) {
        if (hp.cls_count[c] == 0) continue;
        std::stable_sort(perm.begin() + hp.cls_start[c], perm.begin() + hp.cls_start[c] + hp.cls_count[c], [&](int x, int y) {
            const int bx = binof(x), by = binof(y);
            if (bx != by) return bx < by;
            return q[x] > q[y];
        });
        int *bs = bins.data() + (size_t)c * (SCREEN_NBIN + 1);
        for (int i = 0; i < hp.cls_count[c]; i++) bs[binof(perm[hp.cls_start[c] + i]) + 1]++;
        for (int k = 0; k < SCREEN_NBIN; k++) bs[k + 1] += bs[k];
    }
    HostPairs n;
    n.stride = hp.stride;
    for (int c = 0; c < 48; c++) { n.cls_start[c] = hp.cls_start[c]; n.cls_count[c] = hp.cls_count[c]; }
    std::vector<double> nq(np);
    n.pp_off.push_back(0);
    n.sh.reserve(hp.sh.size());
    n.pp.reserve(hp.pp.size());
    for (size_t i = 0; i < np; i++) {
        const int o = perm[i];
        nq[i] = q[o];
        n.sh.push_back(hp.sh[2 * o]);
        n.sh.push_back(hp.sh[2 * o + 1]);
        n.pp.insert(n.pp.end(), hp.pp.begin() + (size_t)hp.pp_off[o] * hp.stride, hp.pp.begin() + (size_t)hp.pp_off[o + 1] * hp.stride);
        n.pp_off.push_back((int)n.pp.size() / hp.stride);
    }
    if (qm) {  // (four member-pair bounds per pair ride along)
        std::vector<double> nm(qm->size());
        for (size_t i = 0; i < np; i++)
            for (int e = 0; e < 4; e++) nm[4 * i + e] = (*qm)[4 * (size_t)perm[i] + e];
        *qm = std::move(nm);
    }
    hp = std::move(n);
    q = std::move(nq);
}

// member shell pairs of a group pair: (slot, original shell of the first member, of the second); a pair inside one group keeps
// xa >= xb.  Slots as in build_pairs: 2 xa + xb for (s s| pairs, xb for (l s| pairs, 0 otherwise
static int group_pair_members(const Basis &bg, int ga, int gb, int slot[4], int sa[4], int sb[4]) {
    int n = 0;
    const bool ss = bg.shells[ga].l == 0 && bg.shells[gb].l == 0;
    for (int xa = 0; xa < 2; xa++)
        for (int xb = 0; xb < 2; xb++) {
            const int ia = xa ? bg.sh_id1[ga] : bg.sh_id0[ga], ib = xb ? bg.sh_id1[gb] : bg.sh_id0[gb];
            if (ia < 0 || ib < 0) continue;
            if (ga == gb && xa < xb) continue;
            slot[n] = ss ? 2 * xa + xb : xb;
            sa[n] = ia;
            sb[n] = ib;
            n++;
        }
    return n;
}

// prefix offsets of the surviving tasks of every class pair for the coarse test Q_b Q_k >= tc: SCREEN_NBIN entries per pair,
// one per contraction-depth bin of the partner class (inside a bin the partners are sorted by their bound, descending, so the
// survivors are a prefix and a two-pointer sweep per (bin, bin) block finds them all)
// dl: nullptr (one threshold `tc` for every class pair) or the 5 x 5 table of density maxima by angular momentum -- then the
// threshold of a class pair is tau / max(4 dl[la][lb], 4 dl[lc][ld], and with K: dl[la][lc], dl[la][ld], dl[lb][lc], dl[lb][ld])
static void plan_screen_begin(DirectCtx &c, ScreenPlan &sp, double tc_all, const double *dl = nullptr, double tau = 0.0, bool with_k = false,
                              bool upload = false, hipStream_t upload_on = nullptr) {
    // slices of the offset array: fixed per class pair (the larger of the two lists owns the entries at most)
    size_t n = 0;
    for (int cb = 0; cb < NCLS_ALL; cb++)
        for (int ck = 0; ck <= cb; ck++) {
            sp.start[cb][ck] = (long long)n;
            sp.total[cb][ck] = 0;
            if (c.hp.cls_count[cb] && c.hp.cls_count[ck]) n += (size_t)std::max(c.hp.cls_count[cb], c.hp.cls_count[ck]) * SCREEN_NBIN + 1;
        }
    (void)n;  // (== n_toff: dqc_direct_create sized the host and device arrays with the same formula)
    c.stat_total = c.stat_launched = 0;
    sp.ctx = &c;
    sp.dl = dl;
    sp.tc_all = tc_all;
    sp.tau = tau;
    sp.with_k = with_k;
    sp.st = upload_on;
    sp.upload = upload;
}

static int plan_screen_pair(ScreenPlan &sp, int cb, int ck) {
    DirectCtx &c = *sp.ctx;
    const double *dl = sp.dl;
    const double tau = sp.tau;
    const bool with_k = sp.with_k;
    std::vector<long long> &cnt = c.cnt_scratch;
    long long *toff = c.h_toff + sp.start[cb][ck];
    {
        {
            sp.total[cb][ck] = 0;
            const int nb = c.hp.cls_count[cb], nk = c.hp.cls_count[ck];
            if (nb == 0 || nk == 0) return 0;
            int la, lb, lc, ld;
            class_l(cb, la, lb);
            class_l(ck, lc, ld);
            const bool same = cb == ck;
            // statistics in SHELL quartets: a group pair stands for w member shell pairs (wpre: prefix sums of w in table order)
            const long long *wb_ = c.wpre.data() + c.hp.cls_start[cb], *wk_ = c.wpre.data() + c.hp.cls_start[ck];
            auto Wb = [&](int lo_, int hi_) { return wb_[hi_] - wb_[lo_]; };
            auto Wk = [&](int lo_, int hi_) { return wk_[hi_] - wk_[lo_]; };
            const long long Sb = Wb(0, nb), Sk = Wk(0, nk);
            c.stat_total += same ? Sb * (Sb + 1) / 2 : Sb * Sk;
            double tc = sp.tc_all;
            if (dl) {
                auto d2 = [&](int x, int y) { return std::max(dl[5 * x + y], dl[5 * y + x]); };
                double m = 4.0 * std::max(d2(la, lb), d2(lc, ld));
                if (with_k) m = std::max(std::max(m, std::max(d2(la, lc), d2(la, ld))), std::max(d2(lb, lc), d2(lb, ld)));
                tc = m > 0.0 ? tau / m : INFINITY;
            }
            // (one-lane classes with a larger bra than ket block may take the bra-uniform entries: eri_core.hpp, FLIP1)
            const bool flip = c.flip1 && (2 * la + 1) * (2 * lb + 1) > (2 * lc + 1) * (2 * ld + 1);
            const bool tpq1 = !hl_forced() && la <= ERI_LMAX && lc <= ERI_LMAX && eri_tpq(ncart(la) * ncart(lb) * ncart(lc) * ncart(ld)) == 1 && !flip;
            const double *qb = c.q.data() + c.hp.cls_start[cb], *qk = c.q.data() + c.hp.cls_start[ck];
            const int *bb = c.bins.data() + (size_t)cb * (SCREEN_NBIN + 1), *bk = c.bins.data() + (size_t)ck * (SCREEN_NBIN + 1);
            const int nown = tpq1 ? nk : nb;  // the list the entries belong to: ket pairs (one lane per quartet) or bra pairs
            cnt.assign((size_t)nown * SCREEN_NBIN, 0);
            for (int ab = 0; ab < SCREEN_NBIN; ab++)      // bra bin
                for (int kb = 0; kb < SCREEN_NBIN; kb++) {  // ket bin
                    const int b0 = bb[ab], b1 = bb[ab + 1], k0 = bk[kb], k1 = bk[kb + 1];
                    if (b0 == b1 || k0 == k1) continue;
                    if (tpq1) {  // per ket pair: 64-bra-pair chunks of bra bin `ab` (the triangle keeps bra >= ket)
                        int ptr = b1 - b0;
                        for (int ik = k0; ik < k1; ik++) {
                            while (ptr > 0 && !(qb[b0 + ptr - 1] * qk[ik] >= tc)) ptr--;
                            // (chunks of 64 / PS bra pairs: the kernel spreads a quartet's primitive quartets over PS lanes)
                            const int per = 64 / eri_split_lanes(ab, c.hp.pp_off[c.hp.cls_start[ck] + ik + 1] - c.hp.pp_off[c.hp.cls_start[ck] + ik], 1);
                            const int c0 = (same && ik > b0) ? ((ik - b0) / per) : 0;
                            int chunks = (ptr + per - 1) / per - c0;
                            if (chunks < 0) chunks = 0;
                            cnt[(size_t)ik * SCREEN_NBIN + ab] = chunks;
                            const int lo = same ? std::max(ik - b0, 0) : 0;
                            if (ptr > lo) {
                                const long long wk1 = Wk(ik, ik + 1);
                                c.stat_launched += wk1 * Wb(b0 + lo, b0 + ptr);
                                if (same && ik >= b0 + lo && ik < b0 + ptr) c.stat_launched -= wk1 * (wk1 - 1) / 2;  // the quartet of the pair with itself
                            }
                        }
                    } else {  // per bra pair: a prefix of ket bin `kb`
                        int ptr = k1 - k0;
                        for (int ib = b0; ib < b1; ib++) {
                            while (ptr > 0 && !(qb[ib] * qk[k0 + ptr - 1] >= tc)) ptr--;
                            int n = ptr;
                            if (same) n = std::max(0, std::min(n, ib + 1 - k0));
                            cnt[(size_t)ib * SCREEN_NBIN + kb] = n;
                            if (n > 0) {
                                const long long wb1 = Wb(ib, ib + 1);
                                c.stat_launched += wb1 * Wk(k0, k0 + n);
                                if (same && ib >= k0 && ib < k0 + n) c.stat_launched -= wb1 * (wb1 - 1) / 2;
                            }
                        }
                    }
                }
            // multi-lane classes with lane groups of <= 16 lanes: PS slots per task (eri_split_lanes of the ket bin's depth bound and
            // the bra pair's primitive count), every entry starting at a multiple of its PS (eri_core.hpp, screened map)
            const int tpq_ = (!hl_forced() && la <= ERI_LMAX && lc <= ERI_LMAX) ? eri_tpq(ncart(la) * ncart(lb) * ncart(lc) * ncart(ld)) : 0;
            const bool split = tpq_ >= 1 && tpq_ <= 16 && !tpq1;
            long long run = 0;
            size_t w = 0;
            for (size_t e = 0; e < cnt.size(); e++) {
                int ps = 1;
                if (split) {
                    const int ib = c.hp.cls_start[cb] + (int)(e / SCREEN_NBIN);
                    ps = eri_split_lanes((int)(e % SCREEN_NBIN), c.hp.pp_off[ib + 1] - c.hp.pp_off[ib], tpq_, c.split_per);
                    run = (run + ps - 1) / ps * ps;
                }
                toff[w++] = run;
                run += cnt[e] * ps;
            }
            toff[w++] = run;
            sp.total[cb][ck] = run;
            if (sp.upload && run > 0)
                DQC_HIP(hipMemcpyAsync(c.d_toff + sp.start[cb][ck], toff, sizeof(long long) * w, hipMemcpyHostToDevice, sp.st));
        }
    }
    return 0;
}

// all class pairs at once, nothing uploaded (statistics of an unscreened pass)
static void plan_screen(DirectCtx &c, double tc_all, ScreenPlan &sp) {
    plan_screen_begin(c, sp, tc_all);
    for (int cb = 0; cb < NCLS_ALL; cb++)
        for (int ck = 0; ck <= cb; ck++) (void)plan_screen_pair(sp, cb, ck);
}

// max |D| over the AO block of every shell pair, and over the whole matrix (non-negative doubles order like their bit patterns)
__global__ void shell_dmax_kernel(double *__restrict__ dsh, double *__restrict__ dmax, const double *__restrict__ dsym,
                                  const int *__restrict__ ao_off, const int *__restrict__ ao_off1, const int *__restrict__ shl, int nsh,
                                  int nao) {
    // (nsh GROUPS: a group's block is the union of its members' blocks -- ao_off1: second member or -1)
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double m = 0.0;
    if (e < (long long)nsh * nsh) {
        const int i = (int)(e / nsh), j = (int)(e % nsh);
        const int ni = 2 * shl[i] + 1, nj = 2 * shl[j] + 1;
        for (int xi = 0; xi < 2; xi++)
            for (int xj = 0; xj < 2; xj++) {
                const int oi = xi ? (ao_off1 ? ao_off1[i] : -1) : ao_off[i], oj = xj ? (ao_off1 ? ao_off1[j] : -1) : ao_off[j];
                if (oi < 0 || oj < 0) continue;
                const double *p = dsym + (size_t)oi * nao + oj;
                for (int a = 0; a < ni; a++)
                    for (int b_ = 0; b_ < nj; b_++) m = fmax(m, fabs(p[(size_t)a * nao + b_]));
            }
        dsh[e] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0)
        atomicMax(reinterpret_cast<unsigned long long *>(dmax), (unsigned long long)__double_as_longlong(m));
}

// the same maxima per PAIR OF ANGULAR MOMENTA: dl[1 + 5 li + lj] = max |D| over the blocks of all shell pairs (l = li, l = lj).
// The launch maps of a class pair are cut with the maxima of the blocks its quartets can touch -- the d and f blocks of a
// density matrix are one to two orders below its s and p blocks, and the classes they belong to are the expensive ones
__global__ void shell_dmax_by_l_kernel(double *__restrict__ dl, const double *__restrict__ dsh, const int *__restrict__ shl, int nsh) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)nsh * nsh) return;
    const double m = dsh[e];
    const int li = shl[e / nsh], lj = shl[e % nsh];
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(dl) + 1 + 5 * li + lj;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    if (m > 0.0 && bits > *slot) atomicMax(slot, bits);  // (the plain read only saves atomics: a stale value is smaller)
}
}  // namespace dqc

extern "C" {

int dqc_jk_direct(double *d_J, double *d_K, const double *d_dm, const int *atm, int natm, const int *bas, int nbas,
                  const double *env, int nenv, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if (nbas == 0 || b.nao == 0) return DQC_OK;
    if ((rc = boys_table_ensure())) return rc;
    // grouped view (general contractions merged: eri_core.hpp; trivial groups with g shells, whose runtime kernel reads one
    // coefficient per primitive pair)
    bool merge = !hl_forced();
    for (const HostShell &h : b.shells) merge = merge && h.l <= ERI_LMAX;
    Basis bg;
    group_s_shells(b, bg, merge);
    HostPairs hp;
    build_pairs(bg, hp);
    DevPool pool(st);
    DevShells ds;
    if ((rc = upload_shells(ds, bg, pool, st))) { set_error("dqc_jk_direct: device upload failed"); return rc; }
    int *d_sh = nullptr, *d_off = nullptr;
    double *d_pp = nullptr, *d_sym = nullptr, *d_a = nullptr, *d_b = nullptr;
    const size_t n2 = (size_t)b.nao * b.nao;
    if ((rc = pool.upload(&d_sh, hp.sh, st)) || (rc = pool.upload(&d_off, hp.pp_off, st)) || (rc = pool.upload(&d_pp, hp.pp, st)) ||
        (rc = pool.alloc(&d_sym, n2)) || (rc = pool.alloc(&d_a, n2)) || (d_K && (rc = pool.alloc(&d_b, n2)))) {
        set_error("dqc_jk_direct: device allocation failed");
        return rc;
    }
    hipLaunchKernelGGL(jk_direct_prep_kernel, dim3(256), dim3(256), 0, st, d_sym, d_a, d_b, d_dm, b.nao);
    DQC_CHECK_LAUNCH();
    DevPairs dp{d_sh, d_off, d_pp, hp.stride};
    EriOut og{b.nao, 0, 0, 0};
    og.dmat = d_sym;
    og.jacc = d_a;
    og.kacc = d_b;
    constexpr int NCLS = (ERI_LMAX + 1) * (ERI_LMAX + 2) / 2;
    rc = ClassLoopJK<NCLS - 1, NCLS - 1>::run(ds, dp, hp, og, st);
    if (rc) return rc;
    if ((rc = run_generic_classes<ERI_OUT_JK>(nullptr, ds, dp, hp, og, st))) return rc;
    hipLaunchKernelGGL(jk_direct_finish_kernel, dim3(256), dim3(256), 0, st, d_J, d_K, d_a, d_b, b.nao);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_direct_create(void **ctx_out, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv,
                      void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx_out) { set_error("dqc_direct_create: null handle"); return DQC_EINVAL; }
    *ctx_out = nullptr;
    std::unique_ptr<DirectCtx> c(new DirectCtx());
    int rc = parse_basis(c->b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if (nbas == 0 || c->b.nao == 0) { set_error("dqc_direct_create: empty basis"); return DQC_EINVAL; }
    if ((rc = boys_table_ensure())) return rc;
    // the kernels work on the grouped view: general contractions merged (eri_core.hpp) unless the basis has g shells (the runtime
    // kernel of those classes reads one coefficient per primitive pair) or DQC_ERI_GROUP=0
    static const bool group_env = [] { const char *e = getenv("DQC_ERI_GROUP"); return !(e && e[0] == '0'); }();
    bool merge = group_env && !hl_forced();
    for (const HostShell &h : c->b.shells) merge = merge && h.l <= ERI_LMAX;
    group_s_shells(c->b, c->bg, merge);
    build_pairs(c->bg, c->hp);
    const size_t np = c->hp.sh.size() / 2, n2 = (size_t)c->b.nao * c->b.nao, nsh = c->bg.shells.size();
    if ((rc = upload_shells(c->ds, c->bg, c->pool, st))) { set_error("dqc_direct_create: device upload failed"); return rc; }
    c->q.assign(np, 0.0);
    c->qm.assign(np * 4, 0.0);
    {   // Schwarz bounds from the unsorted tables (scratch of this block only): four slots per pair, one per member shell pair
        DevPool tmp;
        int *d_sh = nullptr, *d_off = nullptr;
        double *d_pp = nullptr, *d_q0 = nullptr;
        if ((rc = tmp.upload(&d_sh, c->hp.sh, st)) || (rc = tmp.upload(&d_off, c->hp.pp_off, st)) || (rc = tmp.upload(&d_pp, c->hp.pp, st)) ||
            (rc = tmp.alloc(&d_q0, np * 4))) {
            set_error("dqc_direct_create: device allocation failed");
            return rc;
        }
        DQC_HIP(hipMemsetAsync(d_q0, 0, sizeof(double) * np * 4, st));
        DevPairs dp0{d_sh, d_off, d_pp, c->hp.stride};
        constexpr int NCLS = (ERI_LMAX + 1) * (ERI_LMAX + 2) / 2;
        if ((rc = ClassLoopSchwarz<NCLS - 1>::run(d_q0, c->ds, dp0, c->hp, st))) return rc;
        if ((rc = run_generic_classes<ERI_OUT_SCHWARZ>(d_q0, c->ds, dp0, c->hp, EriOut{0, 0, 0, 0}, st))) return rc;
        DQC_HIP(hipMemcpyAsync(c->qm.data(), d_q0, sizeof(double) * np * 4, hipMemcpyDeviceToHost, st));
        DQC_HIP(hipStreamSynchronize(st));
    }
    for (double &v : c->qm) v = std::sqrt(v);
    for (size_t i = 0; i < np; i++) c->q[i] = std::max(std::max(c->qm[4 * i], c->qm[4 * i + 1]), std::max(c->qm[4 * i + 2], c->qm[4 * i + 3]));
    sort_pairs_by_bound(c->hp, c->q, c->bins, &c->qm);
    c->wpre.assign(np + 1, 0);
    for (size_t i = 0; i < np; i++) {
        int slot[4], sa[4], sb[4];
        c->wpre[i + 1] = c->wpre[i] + group_pair_members(c->bg, c->hp.sh[2 * i], c->hp.sh[2 * i + 1], slot, sa, sb);
    }
    int *d_sh = nullptr, *d_off = nullptr;
    double *d_pp = nullptr;
    size_t ntoff = 0;
    for (int cb = 0; cb < NCLS_ALL; cb++)
        for (int ck = 0; ck <= cb; ck++)
            if (c->hp.cls_count[cb] && c->hp.cls_count[ck])
                ntoff += (size_t)std::max(c->hp.cls_count[cb], c->hp.cls_count[ck]) * SCREEN_NBIN + 1;
    if ((rc = c->pool.upload(&d_sh, c->hp.sh, st)) || (rc = c->pool.upload(&d_off, c->hp.pp_off, st)) ||
        (rc = c->pool.upload(&d_pp, c->hp.pp, st)) || (rc = c->pool.upload(&c->d_q, c->q, st)) || (rc = c->pool.upload(&c->d_bins, c->bins, st)) || (rc = c->pool.alloc(&c->d_dsh, nsh * nsh)) ||
        (rc = c->pool.alloc(&c->d_sym, n2)) || (rc = c->pool.alloc(&c->d_a, n2)) || (rc = c->pool.alloc(&c->d_b, n2)) ||
        (rc = c->pool.alloc(&c->d_dmax, 26)) || (rc = c->pool.alloc(&c->d_toff, ntoff))) {
        set_error("dqc_direct_create: device allocation failed");
        return rc;
    }
    DQC_HIP(hipStreamSynchronize(st));  // the uploads read host vectors that may move
    c->dp = DevPairs{d_sh, d_off, d_pp, c->hp.stride};
    c->n_toff = ntoff;
    if (hipHostMalloc((void **)&c->h_toff, sizeof(long long) * std::max<size_t>(ntoff, 1), hipHostMallocDefault) != hipSuccess) {
        c->h_toff = nullptr;
        set_error("dqc_direct_create: pinned host allocation failed");
        return DQC_ENOMEM;
    }
    *ctx_out = c.release();
    return DQC_OK;
}

int dqc_direct_destroy(void *ctx) {
    delete static_cast<dqc::DirectCtx *>(ctx);  // the pool frees the device memory
    return DQC_OK;
}

int dqc_direct_npairs(void *ctx) { return ctx ? (int)static_cast<dqc::DirectCtx *>(ctx)->wpre.back() : 0; }

int dqc_direct_bounds(void *ctx, double *h_q, int *h_shells) { return dqc_direct_bounds_groups(ctx, h_q, h_shells, nullptr); }

int dqc_direct_bounds_groups(void *ctx, double *h_q, int *h_shells, int *h_group) {
    // HOST arrays: the Schwarz bound of every SHELL pair (npairs), its two shells (npairs, 2) -- indices of the caller's table -- and
    // (h_group, optional) the index of the group pair it belongs to: the screening works on group pairs (general contractions
    // merged, eri_core.hpp) with the largest bound of their members
    using namespace dqc;
    if (!ctx) { set_error("dqc_direct_bounds: null context"); return DQC_EINVAL; }
    const DirectCtx &c = *static_cast<DirectCtx *>(ctx);
    size_t o = 0;
    for (size_t i = 0; i < c.q.size(); i++) {
        int slot[4], sa[4], sb[4];
        const int n = group_pair_members(c.bg, c.hp.sh[2 * i], c.hp.sh[2 * i + 1], slot, sa, sb);
        for (int e = 0; e < n; e++, o++) {
            if (h_q) h_q[o] = c.qm[4 * i + slot[e]];
            if (h_shells) { h_shells[2 * o] = sa[e]; h_shells[2 * o + 1] = sb[e]; }
            if (h_group) h_group[o] = (int)i;
        }
    }
    return DQC_OK;
}

int dqc_direct_stats(void *ctx, long long *quartets_total, long long *quartets_launched, double *dmax) {
    if (!ctx) { dqc::set_error("dqc_direct_stats: null context"); return DQC_EINVAL; }
    const dqc::DirectCtx &c = *static_cast<dqc::DirectCtx *>(ctx);
    if (quartets_total) *quartets_total = c.stat_total;
    if (quartets_launched) *quartets_launched = c.stat_launched;
    if (dmax) *dmax = c.stat_dmax;
    return DQC_OK;
}

int dqc_direct_jk(void *ctx, double *d_J, double *d_K, const double *d_dm, double tau, void *stream) {
    return dqc_direct_jk_part(ctx, d_J, d_K, d_dm, tau, 0, 1, stream);
}

int dqc_direct_jk_part(void *ctx, double *d_J, double *d_K, const double *d_dm, double tau, int part, int nparts, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx) { set_error("dqc_direct_jk: null context"); return DQC_EINVAL; }
    if (!(tau >= 0.0)) { set_error("dqc_direct_jk: tau must be >= 0"); return DQC_EINVAL; }
    if (nparts < 1 || part < 0 || part >= nparts) { set_error("dqc_direct_jk_part: need 0 <= part < nparts"); return DQC_EINVAL; }
    DirectCtx &c = *static_cast<DirectCtx *>(ctx);
    const int nao = c.b.nao, nsh = (int)c.bg.shells.size();  // (groups: the kernels' "shells")
    hipLaunchKernelGGL(jk_direct_prep_kernel, dim3(256), dim3(256), 0, st, c.d_sym, c.d_a, d_K ? c.d_b : nullptr, d_dm, nao);
    DQC_CHECK_LAUNCH();
    SideJoin sj;  // joins the side streams on every exit after the fork below
    EriOut og{nao, 0, 0, 0};
    og.dmat = c.d_sym;
    og.jacc = c.d_a;
    og.kacc = d_K ? c.d_b : nullptr;
    og.part = part;
    og.nparts = nparts;
    if (const char *e = getenv("DQC_ERI_DBG")) og.dbg = atoi(e);  // timing experiments (eri_core.hpp)
    if (const char *e = getenv("DQC_ERI_SPLIT_PER")) c.split_per = atoi(e) > 0 ? atoi(e) : 1 << 30;
    og.split_per = c.split_per;
    if (const char *e = getenv("DQC_ERI_FLIP1")) c.flip1 = atoi(e) != 0;
    og.flip1 = c.flip1;
    ScreenPlan sp;
    ScreenPlan *spp = nullptr;
    double dmx[26];  // [0]: max |D|, [1 + 5 li + lj]: by angular momentum of the block (read by the planner until the last launch)
    int rc;
    if (tau > 0.0) {
        DQC_HIP(hipMemsetAsync(c.d_dmax, 0, sizeof(double) * 26, st));
        const long long npr = (long long)nsh * nsh;
        hipLaunchKernelGGL(shell_dmax_kernel, dim3((unsigned)((npr + 255) / 256)), dim3(256), 0, st, c.d_dsh, c.d_dmax, c.d_sym, c.ds.ao_off,
                           c.ds.ao_off1, c.ds.l, nsh, nao);
        DQC_CHECK_LAUNCH();
        hipLaunchKernelGGL(shell_dmax_by_l_kernel, dim3((unsigned)((npr + 255) / 256)), dim3(256), 0, st, c.d_dmax, c.d_dsh, c.ds.l, nsh);
        DQC_CHECK_LAUNCH();
        DQC_HIP(hipMemcpyAsync(dmx, c.d_dmax, sizeof(dmx), hipMemcpyDeviceToHost, st));
        DQC_HIP(hipStreamSynchronize(st));  // the launch sizes of this pass depend on the density maxima
        c.stat_dmax = dmx[0];
        plan_screen_begin(c, sp, 0.0, dmx + 1, tau, d_K != nullptr, true, st);  // (the class pairs are planned one by one in front of their launches)
        if ((sp.side = side_streams()) != nullptr && (rc = sp.side->fork(st))) { set_error("dqc_direct_jk: stream fork failed"); return rc; }
        sj.arm(sp.side, st);
        sp.d_toff = c.d_toff;
        sp.d_bins = c.d_bins;
        spp = &sp;
        og.pq = c.d_q;
        og.dsh = c.d_dsh;
        og.tau = tau;
        og.nsh = nsh;
    } else {
        plan_screen(c, 0.0, sp);  // statistics only: every quartet is launched through the dense maps
        c.stat_launched = c.stat_total;
    }
    constexpr int NCLS = (ERI_LMAX + 1) * (ERI_LMAX + 2) / 2;
    if ((rc = ClassLoopJK<NCLS - 1, NCLS - 1>::run(c.ds, c.dp, c.hp, og, st, spp))) return rc;
    if ((rc = run_generic_classes<ERI_OUT_JK>(nullptr, c.ds, c.dp, c.hp, og, st, spp))) return rc;
    if ((rc = sj.done())) { set_error("dqc_direct_jk: stream join failed"); return rc; }
    hipLaunchKernelGGL(jk_direct_finish_kernel, dim3(256), dim3(256), 0, st, d_J, d_K, c.d_a, c.d_b, nao);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_eri_fill_tiles(double *d_tiles, const int *atm, int natm, const int *bas, int nbas, const double *env,
                       int nenv, void *stream) {
    return dqc_eri_fill_tiles_part(d_tiles, atm, natm, bas, nbas, env, nenv, 0, -1, stream);
}

int dqc_eri_fill_tiles_part(double *d_tiles_part, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv,
                            long long tile_begin, long long tile_end, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if (nbas == 0 || b.nao == 0) return DQC_OK;
    const long long nt_all = (long long)dqc_eri_tile_count(b.nao);
    if (tile_end < 0) tile_end = nt_all;
    if (tile_begin < 0 || tile_end > nt_all || tile_begin > tile_end) { set_error("dqc_eri_fill_tiles_part: tile range outside the store"); return DQC_EINVAL; }
    // a slice [tile_begin, tile_end) of the store (one rank's share when the store is spread over several GPUs): every shell
    // quartet is still evaluated -- the integrals of a quartet scatter over several tiles -- but only the slice is written
    const long long lo = dqc_eri_tile_offset(b.nao, tile_begin), hi = dqc_eri_tile_offset(b.nao, tile_end);
    if (hi == lo) return DQC_OK;
    double *d_tiles = d_tiles_part - lo;  // virtual origin: the kernels address by offsets in the whole store
    DQC_HIP(hipMemsetAsync(d_tiles_part, 0, sizeof(double) * (size_t)(hi - lo), st));  // packed store (common.hpp)
    if ((rc = boys_table_ensure())) return rc;
    // general contractions: s shells of one atom over the same exponents are evaluated together (eri_core.hpp).  The runtime
    // kernel of the g-shell classes reads ungrouped tables: a basis with g shells (or DQC_ERI_GENERIC) is not grouped
    // (DQC_ERI_GROUP=0: A/B runs)
    static const bool group_env = [] { const char *e = getenv("DQC_ERI_GROUP"); return !(e && e[0] == '0'); }();
    bool group = group_env && !hl_forced();
    for (const HostShell &h : b.shells) group = group && h.l <= ERI_LMAX;
    Basis bg;
    group_s_shells(b, bg, group);  // (merge = false: one shell per group, the tables still have the grouped layout the kernels read)
    const Basis &bu = bg;
    HostPairs hp;
    build_pairs(bu, hp);
    DevPool pool(st);  // stream-ordered scratch: this call only enqueues
    SideJoin sj;       // (after the pool: destroyed -- i.e. the side streams joined -- before the scratch is released)
    DevShells ds;
    if ((rc = upload_shells(ds, bu, pool, st))) { set_error("dqc_eri_fill_tiles: device upload failed"); return rc; }
    int *d_sh = nullptr, *d_off = nullptr;
    double *d_pp = nullptr;
    if ((rc = pool.upload(&d_sh, hp.sh, st)) || (rc = pool.upload(&d_off, hp.pp_off, st)) ||
        (rc = pool.upload(&d_pp, hp.pp, st))) {
        set_error("dqc_eri_fill_tiles: device upload failed");
        return rc;
    }
    DevPairs dp{d_sh, d_off, d_pp, hp.stride};
    constexpr int NCLS = (ERI_LMAX + 1) * (ERI_LMAX + 2) / 2;
    EriOut og{0, 0, 0, 0};
    og.st_lo = lo;
    og.st_hi = hi;
    og.st_nao = b.nao;
    if (const char *e = getenv("DQC_ERI_DBG")) og.dbg = atoi(e);
    // depth-binned wave maps of the one-lane classes (DQC_ERI_WMAP=0: the plain wave-transposed map, A/B runs)
    static const bool wmap_env = [] { const char *e = getenv("DQC_ERI_WMAP"); return !(e && e[0] == '0'); }();
    WaveMaps wm;
    wm.pool = wmap_env ? &pool : nullptr;
    wm.side = side_streams();
    const WaveMaps *wmp = &wm;
    if (tile_begin == 0 && tile_end == nt_all) {
        const FillTables ft{&dp, &dp, &hp, &hp, FILL_SAME_TABLE};
        if (wm.side && (rc = wm.side->fork(st))) { set_error("dqc_eri_fill_tiles: stream fork failed"); return rc; }
        sj.arm(wm.side, st);
        rc = ClassLoop<NCLS - 1, NCLS - 1>::run(d_tiles, ds, ft, st, og, wmp);
        if (rc) return rc;
    } else {
        // A slice [tile_begin, tile_end) of a store spread over several GPUs.  Tiles follow each other in the order (IJ, KL <= IJ)
        // of their block pairs, so the slice is the block-pair rows IJ in [r_lo, r_hi] (its first and last row partly).  An integral
        // (ab|cd) lands in the row max(P_ab, P_cd) of its pairs' block pairs: only quartets with one pair IN the rows of the slice
        // and the other in or below them can write into it.  Round 4 evaluated every quartet on every rank (109 ms x N for
        // naphthalene / cc-pVTZ); now the pair table is split into the pairs inside (h1) and below (h0) the rows -- a shell that
        // straddles two AO blocks counts for every block pair it touches -- and the rank evaluates h1 x h1, h1 x h0 and h0 x h1.
        auto row_of = [](long long t) {
            long long r = (long long)((std::sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while (r * (r + 1) / 2 > t) r--;
            while ((r + 1) * (r + 2) / 2 <= t) r++;
            return r;
        };
        const long long r_lo = row_of(tile_begin), r_hi = row_of(tile_end - 1);
        const size_t np = hp.sh.size() / 2;
        HostPairs h1, h0;
        for (HostPairs *h : {&h1, &h0}) {
            h->stride = hp.stride;
            h->pp_off.push_back(0);
            for (int c = 0; c < 48; c++) { h->cls_start[c] = 0; h->cls_count[c] = 0; }
        }
        auto blocks_of = [&](int g, int out[4]) {  // AO blocks the members of group / shell g touch -> count
            int n = 0;
            const int w = 2 * bu.shells[g].l;
            for (int m = 0; m < 2; m++) {
                const int a0 = m == 0 ? bu.shells[g].ao_off : (bu.grouped() ? bu.ao_off1[g] : -1);
                if (a0 < 0) continue;
                out[n++] = a0 >> 3;
                if (((a0 + w) >> 3) != (a0 >> 3)) out[n++] = (a0 + w) >> 3;
            }
            return n;
        };
        for (int c = 0; c < NCLS; c++)
            for (int i = 0; i < hp.cls_count[c]; i++) {
                const size_t x = (size_t)hp.cls_start[c] + i;
                int ba[4], bb[4];
                const int na = blocks_of(hp.sh[2 * x], ba), nb_ = blocks_of(hp.sh[2 * x + 1], bb);
                long long pmin = 0x7fffffffffffffffLL, pmax = -1;
                for (int u = 0; u < na; u++)
                    for (int v = 0; v < nb_; v++) {
                        const long long I = std::max(ba[u], bb[v]), J = std::min(ba[u], bb[v]), P = I * (I + 1) / 2 + J;
                        pmin = std::min(pmin, P);
                        pmax = std::max(pmax, P);
                    }
                if (pmin > r_hi) continue;  // above the rows of the slice: every image of its quartets lies above it
                HostPairs &h = pmax >= r_lo ? h1 : h0;
                if (h.cls_count[c] == 0) h.cls_start[c] = (int)(h.sh.size() / 2);
                h.cls_count[c]++;
                h.sh.push_back(hp.sh[2 * x]);
                h.sh.push_back(hp.sh[2 * x + 1]);
                h.pp.insert(h.pp.end(), hp.pp.begin() + (size_t)hp.pp_off[x] * hp.stride, hp.pp.begin() + (size_t)hp.pp_off[x + 1] * hp.stride);
                h.pp_off.push_back((int)(h.pp.size() / hp.stride));
            }
        (void)np;
        DevPairs d1{nullptr, nullptr, nullptr, hp.stride}, d0{nullptr, nullptr, nullptr, hp.stride};
        auto up = [&](HostPairs &h, DevPairs &d) {
            int *q_sh = nullptr, *q_off = nullptr;
            double *q_pp = nullptr;
            int r;
            if ((r = pool.upload(&q_sh, h.sh, st)) || (r = pool.upload(&q_off, h.pp_off, st)) || (r = pool.upload(&q_pp, h.pp, st))) return r;
            d = DevPairs{q_sh, q_off, q_pp, h.stride};
            return 0;
        };
        if ((rc = up(h1, d1)) || (rc = up(h0, d0))) { set_error("dqc_eri_fill_tiles_part: device upload failed"); return rc; }
        const FillTables f11{&d1, &d1, &h1, &h1, FILL_SAME_TABLE}, f10{&d1, &d0, &h1, &h0, FILL_CROSS}, f01{&d0, &d1, &h0, &h1, FILL_CROSS_OFFDIAG};
        if (wm.side && (rc = wm.side->fork(st))) { set_error("dqc_eri_fill_tiles_part: stream fork failed"); return rc; }
        sj.arm(wm.side, st);
        if ((rc = ClassLoop<NCLS - 1, NCLS - 1>::run(d_tiles, ds, f11, st, og, wmp)) || (rc = ClassLoop<NCLS - 1, NCLS - 1>::run(d_tiles, ds, f10, st, og, wmp)) ||
            (rc = ClassLoop<NCLS - 1, NCLS - 1>::run(d_tiles, ds, f01, st, og, wmp)))
            return rc;
    }
    if ((rc = run_generic_classes<ERI_OUT_TILES>(d_tiles, ds, dp, hp, og, st))) return rc;
    if ((rc = sj.done())) { set_error("dqc_eri_fill_tiles: stream join failed"); return rc; }
    return DQC_OK;
}

int dqc_eri_pair_stats(const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, int merge, long long *h_out) {
    // HOST only (no device call): the pair tables the fill / direct kernels would be launched with -- h_out[0] groups (= shells when
    // merge == 0), [1] pairs, [2] primitive pairs kept after the negligible-pair cut, [3] primitive quartets over all class pairs
    // (bra class >= ket class, bra pair >= ket pair inside a class).  What merging the general contractions buys, and a check of
    // the host logic that runs without a GPU
    using namespace dqc;
    if (!h_out) { set_error("dqc_eri_pair_stats: null output"); return DQC_EINVAL; }
    Basis b, bg;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    group_s_shells(b, bg, merge != 0);
    HostPairs hp;
    build_pairs(bg, hp);
    h_out[0] = (long long)bg.shells.size();
    h_out[1] = (long long)hp.sh.size() / 2;
    h_out[2] = hp.pp_off.empty() ? 0 : hp.pp_off.back();
    long long tot = 0;
    for (int cb = 0; cb < NCLS_ALL; cb++)
        for (int ck = 0; ck <= cb; ck++) {
            long long sb = 0, sk = 0, sq = 0;
            for (int i = 0; i < hp.cls_count[cb]; i++) {
                const long long n = hp.pp_off[hp.cls_start[cb] + i + 1] - hp.pp_off[hp.cls_start[cb] + i];
                sb += n;
                sq += n * n;
            }
            for (int i = 0; i < hp.cls_count[ck]; i++) sk += hp.pp_off[hp.cls_start[ck] + i + 1] - hp.pp_off[hp.cls_start[ck] + i];
            tot += cb == ck ? (sb * sb + sq) / 2 : sb * sk;
        }
    h_out[3] = tot;
    return DQC_OK;
}

int dqc_eri_tiles_to_dense(double *d_dense, const double *d_tiles, int nao, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    hipLaunchKernelGGL(tiles_to_dense_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, d_dense, d_tiles, nao);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
