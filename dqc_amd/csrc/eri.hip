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
//   No integral screening (the reference passes prescreen = NULL); primitive pairs whose Gaussian
//   product prefactor underflows (exp(-100)) are dropped when the pair tables are built.
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
        dense[e] = tiles[tile_base(A, B, Cb, KL) + (long long)tile_pidx(A == B, a & 7, b & 7) * tile_dim(Cb == Db) + tile_pidx(Cb == Db, c & 7, d & 7)];
    }
}

// ---------------------------------------------------------------------------------------------
// class dispatch
// ---------------------------------------------------------------------------------------------
template <int LA, int LB, int LC, int LD>
static int launch_class(double *tiles, const DevShells &ds, const DevPairs &dp, const HostPairs &hp, hipStream_t st) {
    using Cfg = EriCfg<LA, LB, LC, LD>;
    const int cb = LA * (LA + 1) / 2 + LB, ck = LC * (LC + 1) / 2 + LD;
    const int nb = hp.cls_count[cb], nk = hp.cls_count[ck];
    if (nb == 0 || nk == 0 || hl_forced()) return 0;
    const int same = cb == ck;
    const long long ntask = same ? (long long)nb * (nb + 1) / 2 : (long long)nb * nk;
    const long long nblk = eri_num_blocks<Cfg>(nb, nk, ntask);
    (void)hipFuncSetAttribute((const void *)eri_kernel<LA, LB, LC, LD, ERI_OUT_TILES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)Cfg::LDS_BYTES);
    EriOut og{0, 0, 0, 0};
    hipLaunchKernelGGL((eri_kernel<LA, LB, LC, LD, ERI_OUT_TILES>), dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES, st, tiles, ds, dp,
                       dp, hp.cls_start[cb], nb, hp.cls_start[ck], nk, same, ntask, og);
    DQC_CHECK_LAUNCH();
    return 0;
}

// all classes with (LA>=LB), (LC>=LD), class(bra) >= class(ket)
template <int CB, int CK>
struct ClassLoop {
    static int run(double *tiles, const DevShells &ds, const DevPairs &dp, const HostPairs &hp, hipStream_t st) {
        constexpr int LA = CB < 1 ? 0 : (CB < 3 ? 1 : (CB < 6 ? 2 : 3)), LB = CB - LA * (LA + 1) / 2;
        constexpr int LC = CK < 1 ? 0 : (CK < 3 ? 1 : (CK < 6 ? 2 : 3)), LD = CK - LC * (LC + 1) / 2;
        int rc = launch_class<LA, LB, LC, LD>(tiles, ds, dp, hp, st);
        if (rc) return rc;
        if constexpr (CK > 0) return ClassLoop<CB, CK - 1>::run(tiles, ds, dp, hp, st);
        else if constexpr (CB > 0) return ClassLoop<CB - 1, CB - 1>::run(tiles, ds, dp, hp, st);
        else return 0;
    }
};

// ---------------------------------------------------------------------------------------------
// direct SCF: J / K straight from the shell quartets (nothing stored)
// ---------------------------------------------------------------------------------------------
template <int LA, int LB, int LC, int LD>
static int launch_class_jk(const DevShells &ds, const DevPairs &dp, const HostPairs &hp, const EriOut &og, hipStream_t st) {
    using Cfg = EriCfg<LA, LB, LC, LD>;
    const int cb = LA * (LA + 1) / 2 + LB, ck = LC * (LC + 1) / 2 + LD;
    const int nb = hp.cls_count[cb], nk = hp.cls_count[ck];
    if (nb == 0 || nk == 0 || hl_forced()) return 0;
    const int same = cb == ck;
    const long long ntask = same ? (long long)nb * (nb + 1) / 2 : (long long)nb * nk;
    const long long nblk = eri_num_blocks<Cfg>(nb, nk, ntask);
    auto kern = eri_kernel<LA, LB, LC, LD, ERI_OUT_JK>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES_JK);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES_JK, st, (double *)nullptr, ds, dp, dp, hp.cls_start[cb],
                       nb, hp.cls_start[ck], nk, same, ntask, og);
    DQC_CHECK_LAUNCH();
    return 0;
}

template <int CB, int CK>
struct ClassLoopJK {
    static int run(const DevShells &ds, const DevPairs &dp, const HostPairs &hp, const EriOut &og, hipStream_t st) {
        constexpr int LA = CB < 1 ? 0 : (CB < 3 ? 1 : (CB < 6 ? 2 : 3)), LB = CB - LA * (LA + 1) / 2;
        constexpr int LC = CK < 1 ? 0 : (CK < 3 ? 1 : (CK < 6 ? 2 : 3)), LD = CK - LC * (LC + 1) / 2;
        int rc = launch_class_jk<LA, LB, LC, LD>(ds, dp, hp, og, st);
        if (rc) return rc;
        if constexpr (CK > 0) return ClassLoopJK<CB, CK - 1>::run(ds, dp, hp, og, st);
        else if constexpr (CB > 0) return ClassLoopJK<CB - 1, CB - 1>::run(ds, dp, hp, og, st);
        else return 0;
    }
};

// the classes the compile-time kernels leave out -- any pair class with a g shell -- through the runtime kernel
// (eri_generic.hpp); DQC_ERI_GENERIC=1: all classes
template <int MODE>
static int run_generic_classes(double *tiles, const DevShells &ds, const DevPairs &dp, const HostPairs &hp, const EriOut &og,
                               hipStream_t st) {
    for (int la = 0; la <= DQC_LMAX; la++)
        for (int lb = 0; lb <= la; lb++)
            for (int lc = 0; lc <= la; lc++)
                for (int ld = 0; ld <= lc; ld++) {
                    const int cb = la * (la + 1) / 2 + lb, ck = lc * (lc + 1) / 2 + ld;
                    if (ck > cb) continue;
                    if (!hl_forced() && la <= ERI_LMAX && lc <= ERI_LMAX) continue;
                    int rc = launch_hl<MODE>(tiles, ds, dp, dp, hp.cls_start[cb], hp.cls_count[cb], hp.cls_start[ck],
                                             hp.cls_count[ck], cb == ck, og, la, lb, lc, ld, st);
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
    HostPairs hp;
    build_pairs(b, hp);
    DevPool pool(st);
    DevShells ds;
    if ((rc = upload_shells(ds, b, pool, st))) { set_error("dqc_jk_direct: device upload failed"); return rc; }
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
    DevPairs dp{d_sh, d_off, d_pp};
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

int dqc_eri_fill_tiles(double *d_tiles, const int *atm, int natm, const int *bas, int nbas, const double *env,
                       int nenv, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if (nbas == 0 || b.nao == 0) return DQC_OK;
    DQC_HIP(hipMemsetAsync(d_tiles, 0, sizeof(double) * (size_t)eri_store_data_doubles(b.nao), st));  // packed store (common.hpp)
    if ((rc = boys_table_ensure())) return rc;
    HostPairs hp;
    build_pairs(b, hp);
    DevPool pool(st);  // stream-ordered scratch: this call only enqueues
    DevShells ds;
    if ((rc = upload_shells(ds, b, pool, st))) { set_error("dqc_eri_fill_tiles: device upload failed"); return rc; }
    int *d_sh = nullptr, *d_off = nullptr;
    double *d_pp = nullptr;
    if ((rc = pool.upload(&d_sh, hp.sh, st)) || (rc = pool.upload(&d_off, hp.pp_off, st)) ||
        (rc = pool.upload(&d_pp, hp.pp, st))) {
        set_error("dqc_eri_fill_tiles: device upload failed");
        return rc;
    }
    DevPairs dp{d_sh, d_off, d_pp};
    constexpr int NCLS = (ERI_LMAX + 1) * (ERI_LMAX + 2) / 2;
    rc = ClassLoop<NCLS - 1, NCLS - 1>::run(d_tiles, ds, dp, hp, st);
    if (rc) return rc;
    if ((rc = run_generic_classes<ERI_OUT_TILES>(d_tiles, ds, dp, hp, EriOut{0, 0, 0, 0}, st))) return rc;
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
