// eri_core.hpp -- the Rys-quadrature shell-quartet kernel shared by the 4-centre ERI fill (eri.hip) and the
// density-fitting integrals (df.hip).  See eri.hip for the design notes.
#pragma once
#include <atomic>
#include <algorithm>

#include "common.hpp"

namespace dqc {

constexpr int ERI_LMAX = 3;

// Boys functions F_0, F_1 for the one-root classes with closed-form integrals ((ss|ss), (ps|ss): eri_core.hpp).  Table of
// F_0 .. F_8 on a grid of spacing 1/8 up to X = 40 (host-generated once per device: boys_table_ensure, host.hip), staged in
// LDS by the block; a lookup is 9 LDS reads + a 7-term Taylor step (|delta| <= 1/16: truncation 3e-16) instead of the two
// 14-coefficient Clenshaw evaluations of the Rys table; beyond X = 40 the asymptotic forms are exact to round-off.
constexpr int BOYS_W = 9, BOYS_ROWS = 321, BOYS_DOUBLES = BOYS_W * BOYS_ROWS;
static __device__ double g_boys_tab[BOYS_DOUBLES];  // one copy per translation unit (no relocatable device code)
const std::vector<double> &boys_table_host();       // host.hip

// upload of this translation unit's copy, once per device
static int boys_table_ensure() {
    static std::atomic<bool> done[64];  // (two threads racing here both upload the same table: harmless)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DQC_EHIP;
    if (done[dev].load(std::memory_order_acquire)) return 0;
    const std::vector<double> &tab = boys_table_host();
    DQC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_boys_tab), tab.data(), sizeof(double) * BOYS_DOUBLES));
    done[dev].store(true, std::memory_order_release);
    return 0;
}

DQC_DEV void boys_stage_lds(__attribute__((address_space(3))) double *lt, int tid, int nthreads) {
    for (int e = tid; e < BOYS_DOUBLES; e += nthreads) lt[e] = g_boys_tab[e];
}

DQC_DEV void boys01_lds(const __attribute__((address_space(3))) double *lt, double X, double &f0, double &f1) {
    if (X >= 40.0) {
        const double ix = 1.0 / X;
        f0 = 0.88622692545275801 * sqrt(ix);  // sqrt(pi) / 2
        f1 = 0.5 * f0 * ix;
        return;
    }
    const int i = (int)(X * 8.0 + 0.5);
    const double d = i * 0.125 - X;  // -delta
    const __attribute__((address_space(3))) double *c = lt + i * BOYS_W;
    double a = c[7] * (1.0 / 5040.0), b = c[8] * (1.0 / 5040.0);
    a = a * d + c[6] * (1.0 / 720.0); b = b * d + c[7] * (1.0 / 720.0);
    a = a * d + c[5] * (1.0 / 120.0); b = b * d + c[6] * (1.0 / 120.0);
    a = a * d + c[4] * (1.0 / 24.0);  b = b * d + c[5] * (1.0 / 24.0);
    a = a * d + c[3] * (1.0 / 6.0);   b = b * d + c[4] * (1.0 / 6.0);
    a = a * d + c[2] * 0.5;           b = b * d + c[3] * 0.5;
    a = a * d + c[1];                 b = b * d + c[2];
    f0 = a * d + c[0];
    f1 = b * d + c[1];
}


struct DevPairs {
    const int *sh;       // (npair, 2): first shell has the higher (or equal) l
    const int *pp_off;   // (npair+1)
    const double *pp;    // (npp, stride): p, Px, Py, Pz, ca*cb*Kab / p  [grouped tables: four coefficient slots, see below]
    int stride = 5;
};

// ---------------------------------------------------------------------------------------------------------------------------
// General contractions (round 5).  The reference splits a generally contracted shell into one shell per contraction
// (dqc/api/loadbasis.py:72-82): a cc-pVDZ carbon arrives with TWO s shells over the same eight exponents (and a third, single
// primitive one).  Every primitive integral over those exponents was evaluated once per contraction -- up to 16 times for an
// (ss|ss) quartet of four such shells, and the deep s contractions are where a cc-pVDZ / cc-pVTZ fill spends its time.  The
// fill therefore works on GROUPS: s shells of one atom with identical exponent lists are merged (at most two per group: a third
// one starts a new group), a group pair carries up to FOUR coefficient products per primitive pair (slot 2 xa + xb for members
// xa of the first and xb of the second group; c_a c_b K_ab / p, zero for an absent member), the primitive integral -- formed
// WITHOUT coefficients -- is accumulated into NPB x NPK accumulator sets, and the output phase runs once per member
// combination (each is an ordinary shell quartet).  Only l = 0 groups are merged (the accumulators of a class scale with the
// number of combinations, and s shells are the ones the basis sets of the configs contract generally); a pair class (la, 0)
// has two slots, (0, 0) four, every other one.  PP_STRIDE_G doubles per primitive pair: p, P, four coefficient slots.
constexpr int PP_STRIDE_G = 8;
template <int LA, int LB>
struct PairSlots {  // coefficient slots of a grouped pair class
    static constexpr int N = LB == 0 ? (LA == 0 ? 4 : 2) : 1;
};

__host__ __device__ constexpr int c_ncart(int l) { return (l + 1) * (l + 2) / 2; }

// all images of the integral (ij|kl) = v into the packed tile store: the eight permutational images collapse to
// ONE canonical location -- block pair (I >= J; inside a diagonal block pair the a >= b element), the same for the ket, bra
// block pair >= ket block pair -- plus its transpose when both block pairs coincide ((ij|kl) and (kl|ij) live in the same tile).
// Eight guarded stores per element (index arithmetic + the closed-form tile offset each time) were most of the instructions of
// the shallow (cc-pVTZ-type, one primitive quartet) classes.
// [lo, hi): the double offsets of the slice of the store this launch fills (a rank's share of a store sharded over several
// GPUs: `tiles` is then the slice's virtual origin, slice - lo); whole store: 0 ... LLONG_MAX
DQC_DEV void tile_put_all(double *__restrict__ tiles, int i, int j, int k, int l, double v, long long lo, long long hi, int nao) {
    int I = i >> 3, J = j >> 3, K = k >> 3, L = l >> 3;
    int il = i & 7, jl = j & 7, kl = k & 7, ll = l & 7;
    if (I < J || (I == J && il < jl)) { int t = I; I = J; J = t; t = il; il = jl; jl = t; }
    if (K < L || (K == L && kl < ll)) { int t = K; K = L; L = t; t = kl; kl = ll; ll = t; }
    int IJ = I * (I + 1) / 2 + J, KL = K * (K + 1) / 2 + L;
    if (IJ < KL) {
        int t = I; I = K; K = t; t = J; J = L; L = t; t = il; il = kl; kl = t; t = jl; jl = ll; ll = t; t = IJ; IJ = KL; KL = t;
    }
    const int C = tile_dim(K == L);
    const int r = tile_pidx(I == J, il, jl), c = tile_pidx(K == L, kl, ll);
    const long long base = tile_base(I, J, K, KL, TileLay(nao));  // (rows of valid AOs are a prefix of every pair: r < tile_rows)
    if (base < lo || base >= hi) return;
    double *tb = tiles + base;
    tb[(long long)r * C + c] = v;
    if (IJ == KL) tb[(long long)c * C + r] = v;
}

template <int LA, int LB, int LC, int LD>
struct EriCfg {
    static constexpr int NR = (LA + LB + LC + LD) / 2 + 1;
    static constexpr int NCA = c_ncart(LA), NCB = c_ncart(LB), NCC = c_ncart(LC), NCD = c_ncart(LD);
    static constexpr int SA = 2 * LA + 1, SB = 2 * LB + 1, SC = 2 * LC + 1, SD = 2 * LD + 1;
    static constexpr int NOUT = NCA * NCB * NCC * NCD;
    static constexpr int G1 = (LA + 1) * (LB + 1) * (LC + 1) * (LD + 1);
    static constexpr int TPQ = NOUT <= 9 ? 1 : (NOUT <= 81 ? 4 : (NOUT <= 324 ? 16 : (NOUT <= 1296 ? 64 : 256)));  // = eri_tpq(NOUT)
    static constexpr int QPB = 256 / TPQ;
    static constexpr int NPT = (NOUT + TPQ - 1) / TPQ;
    static constexpr int GSZ = 3 * NR * G1;
    static constexpr int BUF1 = SA * NCB * NCC * NCD;
    static constexpr int REG0 = GSZ > NOUT + BUF1 ? GSZ : NOUT + BUF1;
    static constexpr int REGION = REG0 | 1;  // odd stride: conflict-free when every lane owns a region
    // Rys table of this class's root count staged in LDS behind the regions when it is small (NR <= 2: 3.7 and 8.4 KB; with the 14 KB table of NR = 3 cc-pVTZ fills got 7 % slower).  Lanes of
    // a wave work on different primitive quartets, so a root lookup is a gather: 28 uncoalesced global loads per (direction,
    // root) item -- PMC: one VMEM read per 8.6 VALU instructions and 48 % of the wave cycles in issue stalls for (ps|ss)
    // (ss|ss) and (ps|ss): the integral is a closed form in F_0, F_1 -- no 2D-integral staging, Boys table instead of the Rys table
    static constexpr bool BOYS01 = (NR == 1 && LB == 0 && LC == 0 && LD == 0);
#ifdef ERI_NO_LDS_TAB  // A/B builds
    static constexpr int TAB_DOUBLES = 0;
#else
    static constexpr int TAB_DOUBLES = BOYS01 ? BOYS_DOUBLES : (NR <= 2 ? rys_lds_doubles<NR>() : 0);
#endif
    static constexpr size_t REG_DOUBLES = (size_t)REGION * QPB;
    static constexpr size_t LDS_BYTES = sizeof(double) * (REG_DOUBLES + TAB_DOUBLES);
    // GRAD mode contracts straight from the accumulators: only the 2D-integral staging area is needed
    static constexpr int REGION_G = GSZ | 1;
    static constexpr size_t REG_DOUBLES_G = (size_t)(REGION_G * QPB > 16 ? REGION_G * QPB : 16);
    static constexpr size_t LDS_BYTES_G = sizeof(double) * (REG_DOUBLES_G + TAB_DOUBLES);
};

// positions of the Cartesian output n = ((ca NCB + cb) NCC + cc) NCD + cd in the three staged 2D-integral arrays, packed
// ixx | iyy << 10 | izz << 20 -- a compile-time table per class (computing it in the kernel -- four cart_pow loops and three
// divisions per output -- was most of the 18 ms a naphthalene / cc-pVTZ fill spent outside the primitive loops and the output phase)
constexpr void c_cart_pow(int l, int c, int &lx, int &ly, int &lz) {
    int row = 0, acc = 0;
    while (acc + row + 1 <= c) { acc += row + 1; row++; }
    lx = l - row;
    const int k = c - acc;
    ly = row - k;
    lz = k;
}
template <int LA, int LB, int LC, int LD>
struct OidxTab {
    static constexpr int NCA = c_ncart(LA), NCB = c_ncart(LB), NCC = c_ncart(LC), NCD = c_ncart(LD), NOUT = NCA * NCB * NCC * NCD;
    int v[NOUT];
    constexpr OidxTab() : v{} {
        for (int n = 0; n < NOUT; n++) {
            const int cd = n % NCD, cc = (n / NCD) % NCC, cb = (n / (NCD * NCC)) % NCB, ca = n / (NCD * NCC * NCB);
            int ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0, cx = 0, cy = 0, cz = 0, dx = 0, dy = 0, dz = 0;
            c_cart_pow(LA, ca, ax, ay, az);
            c_cart_pow(LB, cb, bx, by, bz);
            c_cart_pow(LC, cc, cx, cy, cz);
            c_cart_pow(LD, cd, dx, dy, dz);
            const int ixx = ((ax * (LB + 1) + bx) * (LC + 1) + cx) * (LD + 1) + dx;
            const int iyy = ((ay * (LB + 1) + by) * (LC + 1) + cy) * (LD + 1) + dy;
            const int izz = ((az * (LB + 1) + bz) * (LC + 1) + cz) * (LD + 1) + dz;
            v[n] = ixx | (iyy << 10) | (izz << 20);
        }
    }
};

// output modes of the kernel
//   ERI_OUT_TILES : (ij|kl) scattered into the blocked-s8 tile storage (all 8 images)
//   ERI_OUT_3C    : (ij|k)  with the ket pair = (auxiliary shell, unit function) -> out[(i, j), k] and [(j, i), k]
//   ERI_OUT_2C    : (k|l)   bra and ket pairs = (auxiliary shell, unit function) -> out[k, l]
//   ERI_OUT_GRAD  : nuclear-gradient contraction (grad.hip).  The first shell of the bra pair is the "up" (l+1,
//                   coefficients 2 alpha c) or "down" (l-1) companion of an orbital shell a, so the Cartesian block IS
//                   the derivative d/dA of (a b|c d); it is contracted on the fly with Cartesian density matrices,
//                   sum [jfac D_ab D_cd - k (D_ac D_bd + D_ad D_bc)], and added to the gradient of a's atom
//   ERI_OUT_JK    : direct SCF -- nothing is stored; the spherical block of every unique shell quartet is contracted with the
//                   density on the fly (J_ab += (ab|cd) D_cd, J_cd += (ab|cd) D_ab, four exchange products): passes over the
//                   quartet's block in LDS, a lane per result element, one global atomic per (shell-pair) element -- or per
//                   WAVE for the Coulomb block the wave's quartets share
//   ERI_OUT_SCHWARZ : the diagonal quartets (ab|ab) only (task map `same` = 2): max over the spherical block of |(ab|ab)| per
//                   shell pair -> tiles[pair] (as the bit pattern of a non-negative double, atomicMax) -- the Schwarz bounds
//                   Q_ab = sqrt(max |(ab|ab)|), |(ab|cd)| <= Q_ab Q_cd, of the screened direct SCF (dqc_direct_*)
//   ERI_OUT_J     : ERI_OUT_JK without the exchange blocks (Kohn-Sham builds).  The Coulomb products need no LDS accumulators (two
//                   matrix-vector passes over the block, see the kernel), so this mode keeps the fill's LDS footprint -- the
//                   six accumulator blocks of ERI_OUT_JK (294 doubles per quartet for (ff|ff)) halve the occupancy of the
//                   high-angular-momentum classes
enum { ERI_OUT_TILES = 0, ERI_OUT_3C = 1, ERI_OUT_2C = 2, ERI_OUT_GRAD = 3, ERI_OUT_JK = 4, ERI_OUT_SCHWARZ = 5, ERI_OUT_J = 6 };

// lane-group size of a compile-time class by its Cartesian block size (EriCfg::TPQ; the host's screened task maps need it too)
__host__ __device__ constexpr int eri_tpq(int nout) { return nout <= 9 ? 1 : (nout <= 81 ? 4 : (nout <= 324 ? 16 : (nout <= 1296 ? 64 : 256))); }

// 256-thread blocks of one class launch: QPB consecutive tasks per block, or -- one-lane-per-quartet classes, wave-transposed
// task map (see the kernel) -- one (64-bra-pair chunk, ket pair) per wave
template <class Cfg>
inline long long eri_num_blocks(int nb, int nk, long long ntask) {
    if (Cfg::TPQ == 1) return (((long long)(nb + 63) / 64) * nk + 3) / 4;
    return (ntask + Cfg::QPB - 1) / Cfg::QPB;
}

// a run of consecutive ket pairs with the same primitive-pair count under the depth-binned wave map of an off-diagonal class
// launch: every ket pair of the run owns W waves, cum[bin] of them in front of the bra depth bin `bin` (eri_wave_runs)
struct WaveRun {
    long long off;  // first wave of the run
    int k0, W;      // first ket pair (relative to the class start), waves per ket pair
    int cum[9];     // prefix of the waves per bra depth bin inside one ket pair's W
    int pad_;
};

struct EriOut {
    int nao;      // orbital AOs (3C: leading dimensions)
    int naux;     // auxiliary AOs (3C / 2C: fastest dimension)
    int ao0;      // AO offset of the first orbital shell
    int aux0;     // AO offset of the first auxiliary shell
    // ---- GRAD mode
    const double *dcart = nullptr;  // (ncart, ncart) Cartesian-basis density matrix
    int ncart = 0;
    const int *cao = nullptr;       // Cartesian AO offset of every ORIGINAL shell
    const int *sh_atom = nullptr;   // atom of every original shell
    double *gpart = nullptr;        // (nslot, natm, 3) partial gradients (spread to keep atomics apart)
    int nslot = 1, natm = 0, norig = 0;
    int dirn = 0;                   // +1: first bra shell is an "up" companion, -1: "down"
    double jscale = 1.0;            // weight of the Coulomb-type product
    double kscale = 0.0;            // weight of the exchange-type products (1: HF, 0: pure J)
    // density-fitting gradient (gmode 1: (d_A a b|k) D_ab c_k -> +2 to a's atom, -2 to k's atom;
    //                           gmode 2: (d_A k|l) c_k c_l    -> -1 to k's atom); ccart: fit coefficients, Cartesian
    int gmode = 0;
    const double *ccart = nullptr;
    // ---- JK mode: symmetric AO density (nao, nao), accumulators A (J = (A + A^T) / 2) and B (K = B + B^T; NULL: J only)
    const double *dmat = nullptr;
    double *jacc = nullptr, *kacc = nullptr;
    // ---- screened task map (JK mode of the direct-SCF context, eri.hip): pairs sorted by their Schwarz bound inside a class
    //   toff  : prefix offsets of the surviving tasks.  Lane groups of > 1 lane: per BRA pair ib the kets [0, toff[ib+1] -
    //           toff[ib]) survive (a prefix: the ket list is sorted too), task -> ib by bisection.  One lane per quartet
    //           (wave-transposed map): per KET pair the number of 64-bra-pair chunks, wave -> ket by bisection.
    //   pq    : Schwarz bound of every pair (index = position in the pair table)
    //   dsh   : (nsh, nsh) max |D| over the AO block of every shell pair -- the per-quartet test
    //           Q_ab Q_cd max(4 |D_ab|, 4 |D_cd|, |D_ac|, |D_ad|, |D_bc|, |D_bd|) < tau skips the quartet (exchange blocks only with K)
    //   pbin  : the pairs of a class are ordered by contraction-depth bin (SCREEN_NBIN bins of the primitive-pair count, deepest
    //           first), then by bound: neighbouring lanes keep equally deep primitive loops (sorted by the bound alone a C5-size
    //           pass was 37 % slower), and the surviving partners of a pair are one prefix PER BIN -- toff has SCREEN_NBIN
    //           entries per pair, pbin the SCREEN_NBIN + 1 bin starts of the partner class (relative to the class start)
    const long long *toff = nullptr;
    const double *pq = nullptr, *dsh = nullptr;
    const int *pbin = nullptr;
    double tau = 0.0;
    int nsh = 0;
    // ---- depth-binned wave map of the one-lane-per-quartet classes (fill; see eri_split_lanes): wtab = per wave (8 ket pair +
    //      bra depth bin, first bra pair of the wave), wbin = the bra class's bin starts
    //      Off-diagonal class launches describe the same map by RUNS of ket pairs with equal primitive-pair count (wruns, nruns:
    //      a few dozen entries instead of one per wave -- the gradient's ordered pair lists made per-wave tables of 40 MB)
    const int2 *wtab = nullptr;
    const WaveRun *wruns = nullptr;
    int nruns = 0;
    int wbin[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int flip1 = 0;      // screened maps of the one-lane classes with a larger bra than ket block: bra-uniform entries (see FLIP1)
    int split_per = 16; // screened maps of the multi-lane classes: primitive quartets per lane group below which a quartet is not spread further
    int dbg = 0;  // timing experiments (DQC_ERI_DBG): 1 = skip the primitive loops, 2 = skip the output phase, 4 = skip the tile stores only, 8 = skip the Coulomb atomics of the direct modes
    // ---- one molecule sharded over GPUs (dqc_direct_jk_part): this launch is part `part` of `nparts` interleaved block sets
    int part = 0, nparts = 1;
    // ---- TILES mode: slice [st_lo, st_hi) (double offsets) of the store this launch fills (dqc_eri_fill_tiles_part)
    long long st_lo = 0, st_hi = 0x7fffffffffffffffLL;
    int st_nao = 0;  // basis size (the last block row of the store is kept at its true width: common.hpp TileLay)
};
constexpr int SCREEN_NBIN = 8;
// contraction-depth bin of a pair with npp surviving primitive pairs: 0 = deepest (> 64) ... 7 = one primitive pair (or none)
__host__ __device__ constexpr int screen_bin(int npp) {
    return npp > 64 ? 0 : (npp > 32 ? 1 : (npp > 16 ? 2 : (npp > 8 ? 3 : (npp > 4 ? 4 : (npp > 2 ? 5 : (npp > 1 ? 6 : 7))))));
}

// Lane GROUPS per shell quartet under the depth-binned wave map (round 5; classes whose lane group has <= 16 lanes).  A deep contraction -- two
// 8-primitive s shells on each side: up to 4096 primitive quartets -- walked by ONE lane is a serial chain of dependent loads,
// LDS lookups and fp64 divisions/rsqrt: the deepest waves of the (ss|ss) launch ran ~1 ms on their own while most of the
// chip idled (the fill was latency-bound, not throughput-bound: merging the general contractions cut the primitive quartets of
// that class 5x and its time not at all).  The PRIMITIVE quartets of a quartet are therefore spread over PS = 2^k lane groups
// (of TPQ lanes each, every group with its own LDS region), the partial sums combined by a butterfly of shuffles, and the member
// combinations of the output phase dealt to the same groups.  PS is uniform per wave: a wave takes ONE ket pair and
// 64 / (PS TPQ) bra pairs of ONE depth bin (pairs are sorted by depth inside a class), so it follows from the bin's bound and
// the ket pair's primitive count: ~8 primitive quartets per lane group.  (Measured the other way round -- bra pair uniform, ket
// pairs over the lanes, so that a wave writes one row range of the store: C5 fill 8.4 -> 8.9 ms, gradient 0.16 -> 0.22 s.
// The gradient's class launches keep the flat task map: under this map they run 0.16 s against 0.11 s.)
__host__ __device__ constexpr int screen_bin_bound(int bin) {  // largest primitive-pair count of the bin (bin 0: open, 128 stands in)
    return bin == 0 ? 128 : (128 >> bin);
}
__host__ __device__ inline int eri_split_lanes(int bin, int nkp, int tpq, int per = 8) {  // tpq: lanes of one lane group (EriCfg::TPQ <= 16)
    const int depth = screen_bin_bound(bin) * nkp;
    int ps = 1;
    while (ps * tpq < 64 && ps * per < depth) ps <<= 1;
    return ps;
}

// largest i in [0, n) with off[i] <= t (off has n + 1 non-decreasing entries, off[0] = 0 <= t < off[n])
DQC_DEV int screen_find(const long long *__restrict__ off, int n, long long t) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// the per-quartet density-weighted Schwarz test of the screened direct SCF: true = the quartet contributes less than tau
DQC_DEV bool screen_skip(const EriOut &og, int ib, int ik, int ish, int jsh, int ksh, int lsh) {
    const double *d = og.dsh;
    const size_t n = og.nsh;
    double m = 4.0 * fmax(d[ish * n + jsh], d[ksh * n + lsh]);
    if (og.kacc) m = fmax(fmax(m, fmax(d[ish * n + ksh], d[ish * n + lsh])), fmax(d[jsh * n + ksh], d[jsh * n + lsh]));
    return og.pq[ib] * og.pq[ik] * m < og.tau;
}

// 1 / x from v_rcp_f64 and two Newton steps (x > 0, normal range: Gaussian exponent sums)
DQC_DEV double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// index of the Cartesian component (lx, ly, lz) of shell l (inverse of cart_pow)
DQC_DEV int cart_index(int l, int lx, int lz) {
    const int row = l - lx;
    return row * (row + 1) / 2 + lz;
}

// synchronisation of the TPQ lanes that share a shell quartet: a lane group of at most 64 lanes lies inside one wave, whose LDS
// instructions execute in program order -- a compiler-level wave barrier is enough and the block's other waves never wait
// for this one (round 1 used __syncthreads and a block-uniform primitive-quartet count: every wave waited for the slowest
// quartet of the block twice per primitive quartet); 256-lane groups are the block
template <int TPQ>
DQC_DEV void eri_group_sync() {
    if constexpr (TPQ > 64) {
        __syncthreads();
    } else if constexpr (TPQ > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// compile-time copy of the solid-harmonic tables: the fibre transform below unrolls over it and keeps the non-zero terms only
namespace c2s_ce {
#define C2S_QUAL constexpr
#include "cart2sph.inc"
#undef C2S_QUAL
}  // namespace c2s_ce

// one Cartesian -> solid-harmonic transform of the output phase: src [NOUTER][NC][NINNER] -> dst [NOUTER][NS][NINNER], the fibres
// (outer, inner) dealt to the TPQ lanes of the quartet's lane group
template <int L, int NOUTER, int NINNER, int TPQ>
DQC_DEV void c2s_fibres(const double *__restrict__ src, double *__restrict__ dst, int s) {
    constexpr int NC = (L + 1) * (L + 2) / 2, NS = 2 * L + 1;
    for (int f = s; f < NOUTER * NINNER; f += TPQ) {
        const int o = f / NINNER, in = f - o * NINNER;
        double x[NC], y[NS];
#pragma unroll
        for (int c = 0; c < NC; c++) x[c] = src[(o * NC + c) * NINNER + in];
#pragma unroll
        for (int m = 0; m < NS; m++) {
            y[m] = 0.0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                constexpr int off = c2s_ce::C2S_OFF[L];
                const double cf = c2s_ce::C2S[off + m * NC + c];
                if (cf != 0.0) y[m] += cf * x[c];
            }
        }
#pragma unroll
        for (int m = 0; m < NS; m++) dst[(o * NS + m) * NINNER + in] = y[m];
    }
}

template <int LA, int LB, int LC, int LD, int MODE, int NPB = 1, int NPK = 1>
__global__ __launch_bounds__(256) void eri_kernel(double *__restrict__ tiles, DevShells sh, DevPairs prs, DevPairs prk,
                                                  int b0, int nb, int k0, int nk, int same, long long ntask, EriOut og) {
    using Cfg = EriCfg<LA, LB, LC, LD>;
    // member combinations of a grouped quartet (1: ordinary shell quartets, the tables may have any stride)
    constexpr int NE = NPB * NPK;
    static_assert(NE == 1 || MODE == ERI_OUT_TILES || MODE == ERI_OUT_JK || MODE == ERI_OUT_J || MODE == ERI_OUT_SCHWARZ, "grouped tables: fill / direct modes only");
    constexpr bool DIRECT = MODE == ERI_OUT_JK || MODE == ERI_OUT_J;  // digestion with the density, nothing stored
    static_assert((NPB == 1 || NPB == PairSlots<LA, LB>::N) && (NPK == 1 || NPK == PairSlots<LC, LD>::N), "slot count of the pair class");
    constexpr int NR = Cfg::NR, TPQ = Cfg::TPQ, QPB = Cfg::QPB, NPT = Cfg::NPT, G1 = Cfg::G1, NOUT = Cfg::NOUT;
    constexpr int NMAX = LA + LB, MMAX = LC + LD;
    // one-lane classes whose bra block is larger than the ket block ((ps|ss), (pp|ss), (ds|ss)): under the wave-transposed map every
    // lane adds its own Coulomb bra block (9 atomics per quartet in (pp|ss): that launch ran at the atomic rate, 1.26 ms against
    // 0.55 ms in the fill); the screened direct maps may take the bra-uniform entries of the multi-lane classes instead (og.flip1)
    constexpr bool FLIP1 = DIRECT && TPQ == 1 && Cfg::SA * Cfg::SB > Cfg::SC * Cfg::SD;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int s_maxq;

    const int tid = threadIdx.x;
    const int q = tid / TPQ, s = tid % TPQ;  // quartet slot in the block, lane inside the quartet group
    constexpr int REGION = MODE == ERI_OUT_GRAD ? Cfg::REGION_G : Cfg::REGION;  // (the direct modes digest the block in place: the fill's footprint)
    double *reg = lds + (size_t)q * REGION;
    constexpr bool TAB_LDS = Cfg::TAB_DOUBLES > 0;
    typedef __attribute__((address_space(3))) double lds_double_t;
    lds_double_t *ltab = (lds_double_t *)lds + (MODE == ERI_OUT_GRAD ? Cfg::REG_DOUBLES_G : Cfg::REG_DOUBLES);
    if constexpr (TAB_LDS) {
        if constexpr (Cfg::BOYS01) boys_stage_lds(ltab, tid, 256);
        else rys_stage_lds<NR>(ltab, tid, 256);
        __syncthreads();
    }

    // (one molecule over several GPUs: rank `part` of `nparts` takes the blocks part, part + nparts, ... of the launch)
    const long long bidx = (long long)blockIdx.x * og.nparts + og.part;
    long long task = bidx * QPB + q;
    bool active = task < ntask;
    if (!active) task = ntask - 1;
    int ib, ik;
    int psl = 1, psj = 0;  // lane groups that share this lane's quartet (wave map) and this lane's group among them
    if (same == 2) {  // diagonal quartets (ab|ab) only (Schwarz bounds): ntask = nb
        ib = ik = (int)task;
    } else if (og.toff != nullptr) {
        // screened map (direct SCF): the surviving tasks of every bra pair -- or, one lane per quartet, the surviving 64-bra-pair
        // chunks of every ket pair -- are a prefix of the Schwarz-sorted partner list; ntask counts tasks (waves)
        if (TPQ == 1 && !(FLIP1 && og.flip1)) {
            const long long wv = bidx * 4 + (tid >> 6);
            const bool inr = wv < ntask;
            const long long w2 = inr ? wv : ntask - 1;
            const int e = screen_find(og.toff, nk * SCREEN_NBIN, w2);
            const int ikl = e / SCREEN_NBIN, bin = e % SCREEN_NBIN;
            const int bs = og.pbin[bin];                             // the bra bin [bs, be) of this wave's chunk
            // the primitive quartets of a quartet are spread over PS lanes (eri_split_lanes: from the bra bin's depth bound and
            // the ket pair's primitive count, uniform per wave): a chunk is 64 / PS bra pairs
            const int nkp_ = prk.pp_off[k0 + ikl + 1] - prk.pp_off[k0 + ikl];
            psl = eri_split_lanes(bin, nkp_, 1);
            const int per = 64 / psl;
            const int c0 = (same && ikl > bs) ? ((ikl - bs) / per) : 0;  // (triangle: chunks wholly below the ket pair are not launched)
            const int ibl = bs + (int)(w2 - og.toff[e] + c0) * per + (tid & 63) / psl;
            psj = (tid & 63) % psl;
            active = inr && ibl < og.pbin[bin + 1] && (!same || ibl >= ikl);
            ib = ibl < nb ? ibl : nb - 1;
            ik = ikl;
        } else {
            const int e = screen_find(og.toff, nb * SCREEN_NBIN, task);
            ib = e / SCREEN_NBIN;
            const int bin = e % SCREEN_NBIN;
            if constexpr (TPQ <= 16) {
                // the primitive quartets of a quartet spread over PS lane groups here too (from the ket bin's depth bound and the
                // bra pair's primitive count): an entry holds PS slots per task and starts at a multiple of its PS (plan_screen), so
                // the PS groups of a quartet are an aligned run of lane groups of one wave.  The slots that pad an entry's tail map
                // to partners behind its prefix: past the bin, above the diagonal, or under the threshold (screen_skip)
                const int nbp_ = prs.pp_off[b0 + ib + 1] - prs.pp_off[b0 + ib];
                psl = eri_split_lanes(bin, nbp_, TPQ, og.split_per);
                const int rel = (int)(task - og.toff[e]);
                ik = og.pbin[bin] + rel / psl;
                psj = rel % psl;
                active = active && ik < og.pbin[bin + 1] && (!same || ik <= ib);
                ik = ik < nk ? ik : nk - 1;
            } else {
                ik = og.pbin[bin] + (int)(task - og.toff[e]);
            }
        }
    } else
    if (TPQ <= 16 && og.wruns != nullptr) {
        // depth-binned wave map by runs (off-diagonal classes): wave -> run of ket pairs (scan), ket pair, bra depth bin (scan), chunk
        const long long wv = bidx * 4 + (tid >> 6);
        const bool inr = wv < ntask;
        const long long w2 = inr ? wv : ntask - 1;
        int r = 0;
        for (int t = 1; t < og.nruns; t++) r = og.wruns[t].off <= w2 ? t : r;
        const WaveRun &wr = og.wruns[r];
        const int rel = (int)(w2 - wr.off), ikl = wr.k0 + rel / wr.W, rem = rel % wr.W;
        int bin = 0;
#pragma unroll
        for (int t = 1; t < SCREEN_NBIN; t++) bin = wr.cum[t] <= rem ? t : bin;
        const int nkp_ = prk.pp_off[k0 + ikl + 1] - prk.pp_off[k0 + ikl];
        psl = eri_split_lanes(bin, nkp_, TPQ);
        const int grp = (tid & 63) / TPQ;
        const int ibl = og.wbin[bin] + (rem - wr.cum[bin]) * (64 / (psl * TPQ)) + grp / psl;
        psj = grp % psl;
        active = inr && ibl < og.wbin[bin + 1];
        ib = ibl < nb ? ibl : nb - 1;
        ik = ikl;
    } else
    if (TPQ <= 16 && og.wtab != nullptr) {
        // depth-binned wave map, one table entry per wave (diagonal classes: bra pair >= ket pair): (8 ket pair + bra depth bin,
        // first bra pair of the wave) -- PS lane groups per quartet, 64 / (PS TPQ) bra pairs per wave
        const long long wv = bidx * 4 + (tid >> 6);
        const bool inr = wv < ntask;
        const int2 we = og.wtab[inr ? wv : ntask - 1];
        const int ikl = we.x >> 3, bin = we.x & 7;
        const int nkp_ = prk.pp_off[k0 + ikl + 1] - prk.pp_off[k0 + ikl];
        psl = eri_split_lanes(bin, nkp_, TPQ);
        const int grp = (tid & 63) / TPQ;  // lane group inside the wave
        const int ibl = we.y + grp / psl;
        psj = grp % psl;
        active = inr && ibl < og.wbin[bin + 1];
        ib = ibl < nb ? ibl : nb - 1;
        ik = ikl;
    } else
    if constexpr (TPQ == 1) {
        // one lane per shell quartet: WAVE-TRANSPOSED task map -- the 64 lanes of a wave take 64 consecutive BRA pairs and ONE
        // ket pair.  The primitive loops then read the ket pair's data wave-uniformly (one cache line per load instead of 64:
        // with consecutive KET pairs per lane every primitive quartet was five uncoalesced loads served from L2) and the
        // ket primitive count is the same for all lanes; the pairs of a class are sorted by primitive count, so neighbouring
        // bra pairs are equally deep.  `same` classes keep ib >= ik (waves entirely below the diagonal retire at once).
        const long long wv = bidx * 4 + (tid >> 6);
        const int nchunk = (nb + 63) >> 6;
        int ikl = (int)(wv / nchunk);
        int ibl = (int)(wv - (long long)ikl * nchunk) * 64 + (tid & 63);
        active = ikl < nk && ibl < nb && (!same || ibl >= ikl);
        ib = ibl < nb ? ibl : nb - 1;
        ik = ikl < nk ? ikl : nk - 1;
    } else if (same) {
        long long r = (long long)((sqrt(8.0 * (double)task + 1.0) - 1.0) * 0.5);
        while (r * (r + 1) / 2 > task) r--;
        while ((r + 1) * (r + 2) / 2 <= task) r++;
        ib = (int)r;
        ik = (int)(task - r * (r + 1) / 2);
    } else {
        ib = (int)(task / nk);
        ik = (int)(task % nk);
    }
    ib += b0;
    ik += k0;
    const int ish = prs.sh[2 * ib], jsh = prs.sh[2 * ib + 1], ksh = prk.sh[2 * ik], lsh = prk.sh[2 * ik + 1];
    if constexpr (DIRECT)
        if (og.pq != nullptr && active && screen_skip(og, ib, ik, ish, jsh, ksh, lsh)) active = false;
    double A[3], Cc[3], AB[3], CD[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        A[d] = sh.xyz[ish * 3 + d];
        AB[d] = A[d] - sh.xyz[jsh * 3 + d];
        Cc[d] = sh.xyz[ksh * 3 + d];
        CD[d] = Cc[d] - sh.xyz[lsh * 3 + d];
    }
    const int pb0 = prs.pp_off[ib], nbp = prs.pp_off[ib + 1] - pb0;
    const int pk0 = prk.pp_off[ik], nkp = prk.pp_off[ik + 1] - pk0;
    const bool loops_on = active && !(og.dbg & 1);
    const int nq = loops_on ? nbp * nkp : 0;

    int maxq = nq;  // groups inside a wave: the wave simply runs until its longest quartet is done (`on` masks the others)
    if (TPQ > 64) {  // the group is the block: block-uniform trip count so that the barriers below are legal
        if (tid == 0) s_maxq = 0;
        __syncthreads();
        atomicMax(&s_maxq, nq);
        __syncthreads();
        maxq = s_maxq;
    }

    // output ownership: lane s owns Cartesian outputs n = s + TPQ*m
    static constexpr OidxTab<LA, LB, LC, LD> c_oidx{};
    int oidx[NPT];
    double acc[NE][NPT];
#pragma unroll
    for (int m = 0; m < NPT; m++) {
#pragma unroll
        for (int e = 0; e < NE; e++) acc[e][m] = 0.0;
        int n = s + TPQ * m;
        if (n >= NOUT) n = NOUT - 1;
        oidx[m] = c_oidx.v[n];
    }

    if constexpr (Cfg::BOYS01 && TAB_LDS) {
        // ---------------- (ss|ss), (ps|ss): [s s|s s] = pref F_0(X),  [p_d s|s s] = pref (F_0 (P - A)_d - F_1 q/(p+q) (P - Q)_d) ----
        // (the general path below stages three 2D integrals per primitive quartet in LDS and reads them back; these deep s
        // contractions -- up to 4096 primitive quartets per shell quartet in cc-pVDZ -- were 27 % of a 20-atom fill)
        static_assert(TPQ == 1, "closed-form classes run one lane per shell quartet");
        // the psl lanes of the quartet tile its primitive pairs: psk lanes along the ket list (wave-uniform), the rest along the bra list
        int psk = 1;
        while (psk < psl && psk < nkp) psk <<= 1;
        const int psb = psl / psk, jb = psj / psk, jk = psj % psk;
        for (int ipb = jb; ipb < (loops_on ? nbp : 0); ipb += psb) {
            const double *pb = prs.pp + (size_t)(pb0 + ipb) * prs.stride;
            // NE == 1: the coefficient rides in the prefactor; grouped: the primitive integral is formed without coefficients,
            // summed over the ket primitives per ket slot (tk) and spread over the bra slots once per bra primitive pair
            const double p = pb[0], P0 = pb[1], P1 = pb[2], P2 = pb[3], kb = (NE == 1 ? pb[4] : 1.0) * 34.986836655249725;  // 2 pi^(5/2) K_ab / p
            double tk[NPK][NPT];
            if constexpr (NE > 1) {
#pragma unroll
                for (int k = 0; k < NPK; k++)
#pragma unroll
                    for (int m = 0; m < NPT; m++) tk[k][m] = 0.0;
            }
            for (int ipk = jk; ipk < nkp; ipk += psk) {
                const double *pk = prk.pp + (size_t)(pk0 + ipk) * prk.stride;
                const double qq = pk[0];
                const double d0 = P0 - pk[1], d1 = P1 - pk[2], d2 = P2 - pk[3];
                const double rs_ = rsqrt(p + qq), ipq = rs_ * rs_;  // no division in this loop: the pair table carries K / p
                const double X = p * qq * ipq * (d0 * d0 + d1 * d1 + d2 * d2);
                const double pref = kb * (NE == 1 ? pk[4] : 1.0) * rs_;  // 2 pi^(5/2) K_ab K_cd / (p q sqrt(p + q))
                double f0, f1;
                boys01_lds(ltab, X, f0, f1);
                double val[NPT];
                if constexpr (LA == 0) {
                    val[0] = pref * f0;
                } else {
                    const double w0 = pref * f0, w1 = pref * f1 * qq * ipq;
                    const double v[3] = {w0 * (P0 - A[0]) - w1 * d0, w0 * (P1 - A[1]) - w1 * d1, w0 * (P2 - A[2]) - w1 * d2};
#pragma unroll
                    for (int m = 0; m < NPT; m++) {
                        int ax, ay, az;
                        cart_pow(1, m, ax, ay, az);
                        val[m] = ax ? v[0] : (ay ? v[1] : v[2]);
                    }
                }
                if constexpr (NE == 1) {
#pragma unroll
                    for (int m = 0; m < NPT; m++) acc[0][m] += val[m];
                } else {
#pragma unroll
                    for (int k = 0; k < NPK; k++) {
                        const double ck = pk[4 + k];
#pragma unroll
                        for (int m = 0; m < NPT; m++) tk[k][m] += ck * val[m];
                    }
                }
            }
            if constexpr (NE > 1) {
#pragma unroll
                for (int b = 0; b < NPB; b++) {
                    const double cb = pb[4 + b];
#pragma unroll
                    for (int k = 0; k < NPK; k++)
#pragma unroll
                        for (int m = 0; m < NPT; m++) acc[b * NPK + k][m] += cb * tk[k][m];
                }
            }
        }
    } else {
    const float inv_nkp = 1.0f / (float)(nkp > 0 ? nkp : 1);
    for (int iq = psj; iq < maxq; iq += psl) {  // (psl = 1, psj = 0 except under the wave map)
        const bool on = iq < nq;
        // ---------------- phase A: 2D integrals for every (direction, root) ----------------
        // (iq / nkp through a float reciprocal: exact for iq < 2^22 -- (iq + 1/2) / nkp stays 1 / (2 nkp) away from the integers --
        // where the integer division costs ~25 instructions per primitive quartet)
        const int ipb = on ? (int)(((float)iq + 0.5f) * inv_nkp) : 0, ipk = on ? iq - ipb * nkp : 0;
        const double *pb = prs.pp + (size_t)(pb0 + ipb) * prs.stride, *pk = prk.pp + (size_t)(pk0 + ipk) * prk.stride;
        if (on) {
            const double p = pb[0], qq = pk[0];
            const double P[3] = {pb[1], pb[2], pb[3]}, Q[3] = {pk[1], pk[2], pk[3]};
            // reciprocals once per primitive quartet (the recurrence coefficients below were five fp64 divisions per item) -- and
            // none of them a full IEEE division: 1 / (p + q) and its square root from ONE rsqrt, 1 / p and 1 / q from the hardware
            // reciprocal refined by two Newton steps (error ~1 ulp; a division expands to ~13 instructions, four of them plus a
            // square root per primitive quartet were a sixth of the loop of the small classes)
            const double pq = p + qq;
            const double rs_ = rsqrt(pq), ipq = rs_ * rs_, rho = p * qq * ipq;
            const double PQ[3] = {P[0] - Q[0], P[1] - Q[1], P[2] - Q[2]};
            const double X = rho * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]);
            const double ip = fast_rcp(p), iqq = fast_rcp(qq);
            // 2 pi^(5/2) K_ab K_cd / (p q sqrt(p + q)): the table holds K / p (grouped: coefficients applied in phase B)
            const double pref = (NE == 1 ? pb[4] * pk[4] : 1.0) * 34.986836655249725 * rs_;
            // the NR roots depend on X only, not on the direction: lane s of the quartet's group evaluates root s % NR ONCE and
            // the (direction, root) items fetch theirs by shuffle (a one-lane group loops over all roots) -- per item this
            // was a Clenshaw evaluation of its own, i.e. three times the work in groups of 1 or 4 lanes
            double ru[TPQ == 1 ? NR : 1], rw[TPQ == 1 ? NR : 1];
            if constexpr (TPQ == 1) {
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if constexpr (TAB_LDS) rys_root1_lds<NR>(ltab, X, r, ru[r], rw[r]);
                    else rys_root1<NR>(X, r, ru[r], rw[r]);
                }
            } else {
                static_assert(TPQ >= NR, "a quartet's lane group holds fewer lanes than the class has roots");
                const int rmine = s < NR ? s : 0;
                ru[0] = rw[0] = 0.0;
                if (TPQ <= 64 || s < 64) {  // (a 256-lane group: its items all sit in the first wave)
                    if constexpr (TAB_LDS) rys_root1_lds<NR>(ltab, X, rmine, ru[0], rw[0]);
                    else rys_root1<NR>(X, rmine, ru[0], rw[0]);
                }
            }
            constexpr int NROUND = (3 * NR + TPQ - 1) / TPQ;
#pragma unroll
            for (int round = 0; round < NROUND; round++) {
                const int item_ = s + round * TPQ;
                const int item = item_ < 3 * NR ? item_ : 3 * NR - 1;  // every lane of the wave takes part in the shuffles
                const int d = item / NR, r = item - d * NR;
                double u, w;
                if constexpr (TPQ == 1) {
                    u = ru[0]; w = rw[0];
#pragma unroll
                    for (int rr = 1; rr < NR; rr++)
                        if (r == rr) { u = ru[rr]; w = rw[rr]; }
                } else {  // lane r of this group holds root r (the items of a 256-lane group all sit in its first wave)
                    const int src = ((tid & 63) - (s & 63)) + r;
                    u = __shfl(ru[0], src, 64);
                    w = __shfl(rw[0], src, 64);
                }
                if (item_ >= 3 * NR) continue;
                const double uq = u * qq * ipq, up = u * p * ipq;
                const double b00 = 0.5 * u * ipq;
                const double b10 = 0.5 * (1.0 - uq) * ip;
                const double b01 = 0.5 * (1.0 - up) * iqq;
                const double c00 = (P[d] - A[d]) - uq * PQ[d];
                const double c0p = (Q[d] - Cc[d]) + up * PQ[d];
                // vertical recurrence g[n][m], n <= NMAX, m <= MMAX
                double g[NMAX + 1][MMAX + 1];
                g[0][0] = (d == 2) ? w * pref : 1.0;
#pragma unroll
                for (int n = 0; n < NMAX; n++) g[n + 1][0] = c00 * g[n][0] + (n ? n * b10 * g[n - 1][0] : 0.0);
#pragma unroll
                for (int m = 0; m < MMAX; m++)
#pragma unroll
                    for (int n = 0; n <= NMAX; n++)
                        g[n][m + 1] = c0p * g[n][m] + (m ? m * b01 * g[n][m - 1] : 0.0) + (n ? n * b00 * g[n - 1][m] : 0.0);
                // horizontal recurrence on the bra, in place: h[i][j][m]
                double h[NMAX + 1][LB + 1][MMAX + 1];
#pragma unroll
                for (int i = 0; i <= NMAX; i++)
#pragma unroll
                    for (int m = 0; m <= MMAX; m++) h[i][0][m] = g[i][m];
#pragma unroll
                for (int j = 1; j <= LB; j++)
#pragma unroll
                    for (int i = 0; i <= NMAX - j; i++)
#pragma unroll
                        for (int m = 0; m <= MMAX; m++) h[i][j][m] = h[i + 1][j - 1][m] + AB[d] * h[i][j - 1][m];
                // horizontal recurrence on the ket and store
                double *G = reg + (size_t)(d * NR + r) * G1;
#pragma unroll
                for (int i = 0; i <= LA; i++)
#pragma unroll
                    for (int j = 0; j <= LB; j++) {
                        double kk[MMAX + 1][LD + 1];
#pragma unroll
                        for (int m = 0; m <= MMAX; m++) kk[m][0] = h[i][j][m];
#pragma unroll
                        for (int l = 1; l <= LD; l++)
#pragma unroll
                            for (int m = 0; m <= MMAX - l; m++) kk[m][l] = kk[m + 1][l - 1] + CD[d] * kk[m][l - 1];
#pragma unroll
                        for (int k = 0; k <= LC; k++)
#pragma unroll
                            for (int l = 0; l <= LD; l++) G[((i * (LB + 1) + j) * (LC + 1) + k) * (LD + 1) + l] = kk[k][l];
                    }
            }
        }
        eri_group_sync<TPQ>();
        // ---------------- phase B: accumulate the Cartesian outputs ----------------
        if (on) {
            double cbk[NE];
            if constexpr (NE > 1) {
#pragma unroll
                for (int b = 0; b < NPB; b++)
#pragma unroll
                    for (int k = 0; k < NPK; k++) cbk[b * NPK + k] = pb[4 + b] * pk[4 + k];
            }
#pragma unroll
            for (int m = 0; m < NPT; m++) {
                const int ixx = oidx[m] & 1023, iyy = (oidx[m] >> 10) & 1023, izz = oidx[m] >> 20;
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < NR; r++)
                    v += reg[r * G1 + ixx] * reg[(NR + r) * G1 + iyy] * reg[(2 * NR + r) * G1 + izz];
                if constexpr (NE == 1) acc[0][m] += v;
                else {
#pragma unroll
                    for (int e = 0; e < NE; e++) acc[e][m] += cbk[e] * v;
                }
            }
        }
        eri_group_sync<TPQ>();
    }

    }  // (general path)
    if constexpr (TPQ <= 16) {
        // the lane groups that shared the quartet's primitive quartets combine their partial sums (every group ends with the total)
        // (psl is wave-uniform under the fill's wave maps; the screened maps of the multi-lane classes mix entries in a wave: the
        // shuffles run to the wave's largest count, every lane adds only inside its own run of groups)
        int pslw = psl;
        if constexpr (DIRECT)
            for (int o = TPQ; o < 64; o <<= 1) pslw = max(pslw, __shfl_xor(pslw, o));
        for (int o = 1; o < pslw; o <<= 1)
#pragma unroll
            for (int e = 0; e < NE; e++)
#pragma unroll
                for (int m = 0; m < NPT; m++) {
                    const double t_ = __shfl_xor(acc[e][m], o * TPQ);
                    if (o < psl) acc[e][m] += t_;
                }
    }
    if constexpr (MODE == ERI_OUT_GRAD) {
        // ---------------- gradient contraction straight from the Cartesian accumulators ----------------
        const int a = ish % og.norig;             // original shell behind the up / down companion
        const int la = LA - og.dirn;
        const int ca0 = og.cao[a], cb0 = og.cao[jsh], cc0 = og.cao[ksh], cd0 = og.cao[lsh];
        const bool same_cd = ksh == lsh;
        const double jfac = (same_cd ? 2.0 : 4.0) * og.jscale;
        const double *D = og.dcart;
        const size_t nc = og.ncart;
        double g[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int m = 0; m < NPT; m++) {
            const int n = s + TPQ * m;
            // (lane groups that shared the quartet's primitive quartets hold the same totals: they split the outputs)
            if (n < NOUT && active && (m & (psl - 1)) == psj) {
                const int cd = n % Cfg::NCD, cc = (n / Cfg::NCD) % Cfg::NCC, cb = (n / (Cfg::NCD * Cfg::NCC)) % Cfg::NCB,
                          cu = n / (Cfg::NCD * Cfg::NCC * Cfg::NCB);
                int u[3];
                cart_pow(LA, cu, u[0], u[1], u[2]);
                const size_t ib = cb0 + cb, ic = cc0 + cc, id = cd0 + cd;
                double dcd = 0.0, dbd = 0.0, dbc = 0.0;
                const bool with_k = og.kscale != 0.0;  // (Kohn-Sham: Coulomb-type product only -- four density loads per output fewer)
                if (og.gmode == 0) {
                    dcd = D[ic * nc + id];
                    if (with_k) { dbd = D[ib * nc + id]; dbc = D[ib * nc + ic]; }
                }
#pragma unroll
                for (int dir = 0; dir < 3; dir++) {
                    int o[3] = {u[0], u[1], u[2]};
                    double coef;
                    if (og.dirn > 0) {
                        if (o[dir] == 0) continue;
                        o[dir]--;
                        coef = 1.0;
                    } else {
                        coef = -(o[dir] + 1.0);
                        o[dir]++;
                    }
                    const size_t ia = ca0 + cart_index(la, o[0], o[2]);
                    double f;
                    if (og.gmode == 0) {
                        f = jfac * D[ia * nc + ib] * dcd;
                        if (with_k) f -= og.kscale * (D[ia * nc + ic] * dbd + (same_cd ? 0.0 : D[ia * nc + id] * dbc));
                    }
                    else if (og.gmode == 1)
                        f = 2.0 * D[ia * nc + ib] * og.ccart[ic];
                    else
                        f = -og.ccart[ia] * og.ccart[ic];
                    g[dir] += coef * acc[0][m] * f;
                }
            }
        }
        // reduce over the TPQ lanes of the quartet, one atomic per direction
        constexpr int WRED = TPQ < 64 ? TPQ : 64;
#pragma unroll
        for (int dir = 0; dir < 3; dir++)
#pragma unroll
            for (int o = WRED / 2; o > 0; o >>= 1) g[dir] += __shfl_xor(g[dir], o);
        double *gp = og.gpart + ((size_t)(blockIdx.x % og.nslot) * og.natm + og.sh_atom[a]) * 3;
        double *gk = og.gpart + ((size_t)(blockIdx.x % og.nslot) * og.natm + og.sh_atom[ksh]) * 3;  // gmode 1 only
        if (TPQ <= 64) {
            if (s == 0 && active)
                for (int dir = 0; dir < 3; dir++) {
                    atomicAdd(&gp[dir], g[dir]);
                    if (og.gmode == 1) atomicAdd(&gk[dir], -g[dir]);
                }
        } else {  // the whole block is one quartet: combine the four waves through LDS
            __syncthreads();
            if ((tid & 63) == 0)
                for (int dir = 0; dir < 3; dir++) lds[(tid >> 6) * 3 + dir] = g[dir];
            __syncthreads();
            if (tid == 0 && active)
                for (int dir = 0; dir < 3; dir++) {
                    const double v = lds[dir] + lds[3 + dir] + lds[6 + dir] + lds[9 + dir];
                    atomicAdd(&gp[dir], v);
                    if (og.gmode == 1) atomicAdd(&gk[dir], -v);
                }
        }
        return;
    }
    if (og.dbg & 2) return;
    // ---------------- phase C: Cartesian -> solid harmonics (LDS), scatter to tiles ----------------
    // once per member combination of a grouped quartet (NE == 1: the shell quartet itself).  Members: bra slot eb = 2 xa + xb
    // ((s s| pairs; xb alone for (l s|), ket alike; a combination is an ordinary shell quartet with its own AO offsets.  Of the
    // combinations that are the same integrals -- (xa, xb) and (xb, xa) inside one group, (eb, ek) and (ek, eb) when bra and ket
    // are the same group pair -- one is kept, so every shell quartet is visited exactly once
    double *buf0 = reg, *buf1 = reg + NOUT;
#pragma unroll
    for (int e = 0; e < NE; e++) {
    const int eb = e / NPK, ek = e % NPK;
    const int xa = NPB == 4 ? eb >> 1 : 0, xb = NPB == 4 ? eb & 1 : eb, xc = NPK == 4 ? ek >> 1 : 0, xd = NPK == 4 ? ek & 1 : ek;
    int ai = sh.ao_off[ish], aj = sh.ao_off[jsh], ak = sh.ao_off[ksh], al = sh.ao_off[lsh];
    bool act = active && (e & (psl - 1)) == psj;  // (the combinations dealt to the lane groups that shared the quartet)
    if constexpr (NE > 1) {
        if (xa) ai = sh.ao_off1[ish];
        if (xb) aj = sh.ao_off1[jsh];
        if (xc) ak = sh.ao_off1[ksh];
        if (xd) al = sh.ao_off1[lsh];
        act = act && ai >= 0 && aj >= 0 && ak >= 0 && al >= 0 && !(ish == jsh && xa < xb) && !(ksh == lsh && xc < xd) &&
              !(ib == ik && prs.sh == prk.sh && eb < ek);  // (bra and ket pairs may come from different tables: a sliced fill)
        if constexpr (TPQ > 64) {  // (the group is the block: the barriers below stay block-uniform)
            if (!act) continue;
        } else if (!__any(act)) continue;
        eri_group_sync<TPQ>();  // the previous combination's last stage has read buf1
    }
#pragma unroll
    for (int m = 0; m < NPT; m++) {
        const int n = s + TPQ * m;
        if (n < NOUT) buf0[n] = acc[e][m];
    }
    eri_group_sync<TPQ>();
    // four fibre transforms (c2s_fibres: a lane takes whole Cartesian fibres -- NC LDS reads, NS writes, compile-time sparse
    // coefficients -- where every output element used to read its NC inputs and global coefficients), ping-pong between the buffers;
    // an s index has nothing to transform: its constant factor is applied with the last pass
    double *cur = buf0, *oth = buf1;
    if constexpr (LA > 0) {  // [ca][rest] -> [ma][rest]
        c2s_fibres<LA, 1, Cfg::NCB * Cfg::NCC * Cfg::NCD, TPQ>(cur, oth, s);
        double *t_ = cur; cur = oth; oth = t_;
        eri_group_sync<TPQ>();
    }
    if constexpr (LB > 0) {  // [ma][cb][rest] -> [ma][mb][rest]
        c2s_fibres<LB, Cfg::SA, Cfg::NCC * Cfg::NCD, TPQ>(cur, oth, s);
        double *t_ = cur; cur = oth; oth = t_;
        eri_group_sync<TPQ>();
    }
    if constexpr (LC > 0) {  // [mab][cc][cd] -> [mab][mc][cd]
        c2s_fibres<LC, Cfg::SA * Cfg::SB, Cfg::NCD, TPQ>(cur, oth, s);
        double *t_ = cur; cur = oth; oth = t_;
        eri_group_sync<TPQ>();
    }
    if constexpr (LD > 0) {  // [mabc][cd] -> [mabc][md]
        c2s_fibres<LD, Cfg::SA * Cfg::SB * Cfg::SC, 1, TPQ>(cur, oth, s);
        double *t_ = cur; cur = oth; oth = t_;
        eri_group_sync<TPQ>();
    }
    if (MODE != ERI_OUT_J && MODE != ERI_OUT_JK && act) {  // (the direct modes digest the block in the passes below: no per-element work)
        // scatter: value (ma, mb, mc, md) -> all block-canonical images (md fastest over the lanes: runs of consecutive addresses)
        constexpr double S0 = 0.28209479177387864;  // the l = 0 solid harmonic
        constexpr double SCALE = (LA == 0 ? S0 : 1.0) * (LB == 0 ? S0 : 1.0) * (LC == 0 ? S0 : 1.0) * (LD == 0 ? S0 : 1.0);
        for (int e_ = s; e_ < Cfg::SA * Cfg::SB * Cfg::SC * Cfg::SD; e_ += TPQ) {
            const int md = e_ % Cfg::SD, mabc = e_ / Cfg::SD;
            const double v = cur[e_] * SCALE;
            const int mc = mabc % Cfg::SC, mb = (mabc / Cfg::SC) % Cfg::SB, ma = mabc / (Cfg::SC * Cfg::SB);
            const int i = ai + ma, j = aj + mb, k = ak + mc, l = al + md;
            if constexpr (MODE == ERI_OUT_SCHWARZ) {
                // (four slots per pair: the bound of every member pair of a grouped pair on its own -- slot = the bra combination;
                // the mixed combinations eb != ek are no diagonal quartets of a shell pair)
                if (ma == mc && mb == md && eb == ek)  // non-negative doubles order like their bit patterns
                    atomicMax(reinterpret_cast<unsigned long long *>(tiles) + (size_t)ib * 4 + eb, (unsigned long long)__double_as_longlong(fabs(v)));
            } else
            if (MODE == ERI_OUT_TILES) {
                // A quartet with a repeated shell -- (a a|c d), (a b|c c), (a b|a b) -- holds every value twice ((m, m') and (m', m),
                // equal by symmetry but summed in another order: the last bit may differ) and the image of one lands on the store
                // position of the other: whichever lane wrote last stayed, two fills differed in the last bit of ~0.3 % of the
                // elements.  ONE of the two is written (its images cover both positions): the store is bitwise reproducible
                const bool twin = (ai == aj && mb > ma) || (ak == al && md > mc) ||
                                  (ai == ak && aj == al && mc * Cfg::SD + md > ma * Cfg::SB + mb);
                if (!twin && (!(og.dbg & 4) || v == 12345.678)) tile_put_all(tiles, i, j, k, l, v, og.st_lo, og.st_hi, og.st_nao);
            } else if (MODE == ERI_OUT_3C) {
                const size_t io = i - og.ao0, jo = j - og.ao0, kx = k - og.aux0;
                tiles[(io * og.nao + jo) * og.naux + kx] = v;
                tiles[(jo * og.nao + io) * og.naux + kx] = v;
            } else {
                tiles[(size_t)(i - og.aux0) * og.naux + (k - og.aux0)] = v;
            }
        }
    }
    if constexpr (DIRECT) {
        // Coulomb products of the block V[ab][cd] (in `cur`): J_ab += sum_cd V D_cd (a lane per row, the density element uniform
        // over the lanes) and J_cd += sum_ab V D_ab (a lane per column, stride-1 LDS reads), one global atomic per result.  Round 4
        // added v D to LDS accumulators element by element: two ds_add_f64 per integral, up to 49 lanes on one address -- the
        // output phase of an unscreened naphthalene / cc-pVTZ pass took 66 of 121 ms.
        // The lane groups of a wave mostly share one of their two pairs -- the bra pair under the flat / screened maps of the
        // multi-lane classes (consecutive tasks: one bra pair, consecutive ket pairs), the ket pair under the wave-transposed map
        // of the one-lane classes -- so that block's results are summed across the groups (shuffle butterfly) and added ONCE
        // per wave: the ~1.8e9 global fp64 atomics of an unscreened naphthalene / cc-pVTZ pass were 24 ms of its 105.
        {
            constexpr double S0_ = 0.28209479177387864;
            constexpr double SCALE_ = (LA == 0 ? S0_ : 1.0) * (LB == 0 ? S0_ : 1.0) * (LC == 0 ? S0_ : 1.0) * (LD == 0 ? S0_ : 1.0);
            constexpr int NAB = Cfg::SA * Cfg::SB, NCDS = Cfg::SC * Cfg::SD;
            const double degj = act ? 4.0 * SCALE_ * (ai == aj ? 0.5 : 1.0) * (ak == al ? 0.5 : 1.0) * ((ai == ak && aj == al) ? 0.5 : 1.0) : 0.0;
            const double *D = og.dmat;
            const size_t n = og.nao;
            // wave-uniform shared pair?
            constexpr bool CAN_SHARE = TPQ <= 32;  // (grouped quartets: per member combination -- this code runs inside the combination loop)
            bool share_ab = false, share_cd = false;
            int ai0 = ai, aj0 = aj, ak0 = ak, al0 = al;
            if constexpr (CAN_SHARE) {
                const unsigned long long am = __ballot(act);
                if (am != 0ull) {
                    const int first = __ffsll((long long)am) - 1;
                    ai0 = __shfl(ai, first); aj0 = __shfl(aj, first); ak0 = __shfl(ak, first); al0 = __shfl(al, first);
                    if (TPQ > 1 || (FLIP1 && og.flip1 && og.toff != nullptr)) share_ab = __all(!act || (ai == ai0 && aj == aj0));
                    else share_cd = __all(!act || (ak == ak0 && al == al0));
                }
            }
            const int grp0 = (tid & 63) < TPQ;  // the lanes of the wave's first lane group issue the shared block's atomics
            if (share_ab || act)
            for (int ab = s; ab < NAB; ab += TPQ) {
                double a_ = 0.0;
                if (act) {
                    const double *row = cur + ab * NCDS;
#pragma unroll
                    for (int mc = 0; mc < Cfg::SC; mc++)
#pragma unroll
                        for (int md = 0; md < Cfg::SD; md++) a_ += row[mc * Cfg::SD + md] * D[(size_t)(ak + mc) * n + al + md];
                    a_ *= degj;
                }
                if (share_ab) {
                    for (int o = TPQ; o < 64; o <<= 1) a_ += __shfl_xor(a_, o);
                    if (grp0 && a_ != 0.0 && (!(og.dbg & 8) || a_ == 12345.678)) atomicAdd(og.jacc + (size_t)(ai0 + ab / Cfg::SB) * n + aj0 + ab % Cfg::SB, a_);
                } else if (act && (!(og.dbg & 8) || a_ == 12345.678)) {
                    atomicAdd(og.jacc + (size_t)(ai + ab / Cfg::SB) * n + aj + ab % Cfg::SB, a_);
                }
            }
            if (share_cd || act)
            for (int cd = s; cd < NCDS; cd += TPQ) {
                double a_ = 0.0;
                if (act) {
#pragma unroll
                    for (int ma = 0; ma < Cfg::SA; ma++)
#pragma unroll
                        for (int mb = 0; mb < Cfg::SB; mb++) a_ += cur[(ma * Cfg::SB + mb) * NCDS + cd] * D[(size_t)(ai + ma) * n + aj + mb];
                    a_ *= degj;
                }
                if (share_cd) {
                    for (int o = 1; o < 64; o <<= 1) a_ += __shfl_xor(a_, o);
                    if ((tid & 63) == 0 && a_ != 0.0 && (!(og.dbg & 8) || a_ == 12345.678)) atomicAdd(og.jacc + (size_t)(ak0 + cd / Cfg::SD) * n + al0 + cd % Cfg::SD, a_);
                } else if (act && (!(og.dbg & 8) || a_ == 12345.678)) {
                    atomicAdd(og.jacc + (size_t)(ak + cd / Cfg::SD) * n + al + cd % Cfg::SD, a_);
                }
            }
        }
    }
    if constexpr (MODE == ERI_OUT_JK) {
        // The four exchange blocks, each a pass over the block V[a b c d] in `cur` with a lane per result element and the two summed
        // indices in the inner loops: B_ac += sum_bd V D_bd, B_ad += sum_bc V D_bc, B_bc += sum_ad V D_ad, B_bd += sum_ac V D_ac (K = B + B^T
        // with 1/2 per coincidence a == b, c == d, (ab) == (cd): every unique quartet is visited once), one global atomic per element.
        // Round 4 added v D to LDS accumulators element by element (four ds_add_f64 per integral) -- and the accumulator blocks
        // (294 doubles per (ff|ff) quartet) halved the occupancy of the high classes
        if (act && og.kacc) {
            constexpr double S0_ = 0.28209479177387864;
            constexpr double SCALE_ = (LA == 0 ? S0_ : 1.0) * (LB == 0 ? S0_ : 1.0) * (LC == 0 ? S0_ : 1.0) * (LD == 0 ? S0_ : 1.0);
            constexpr int SA = Cfg::SA, SB = Cfg::SB, SC = Cfg::SC, SD = Cfg::SD;
            constexpr int O2 = SA * SC, O3 = O2 + SA * SD, O4 = O3 + SB * SC, NK4 = O4 + SB * SD;
            const double deg = SCALE_ * (ai == aj ? 0.5 : 1.0) * (ak == al ? 0.5 : 1.0) * ((ai == ak && aj == al) ? 0.5 : 1.0);
            const double *D = og.dmat;
            const size_t n = og.nao;
            for (int x = s; x < NK4; x += TPQ) {
                double a_ = 0.0;
                double *dst;
                if (x < O2) {  // (a, c): sum over b, d
                    const int ma = x / SC, mc = x % SC;
#pragma unroll
                    for (int mb = 0; mb < SB; mb++)
#pragma unroll
                        for (int md = 0; md < SD; md++) a_ += cur[((ma * SB + mb) * SC + mc) * SD + md] * D[(size_t)(aj + mb) * n + al + md];
                    dst = og.kacc + (size_t)(ai + ma) * n + ak + mc;
                } else if (x < O3) {  // (a, d): sum over b, c
                    const int y = x - O2, ma = y / SD, md = y % SD;
#pragma unroll
                    for (int mb = 0; mb < SB; mb++)
#pragma unroll
                        for (int mc = 0; mc < SC; mc++) a_ += cur[((ma * SB + mb) * SC + mc) * SD + md] * D[(size_t)(aj + mb) * n + ak + mc];
                    dst = og.kacc + (size_t)(ai + ma) * n + al + md;
                } else if (x < O4) {  // (b, c): sum over a, d
                    const int y = x - O3, mb = y / SC, mc = y % SC;
#pragma unroll
                    for (int ma = 0; ma < SA; ma++)
#pragma unroll
                        for (int md = 0; md < SD; md++) a_ += cur[((ma * SB + mb) * SC + mc) * SD + md] * D[(size_t)(ai + ma) * n + al + md];
                    dst = og.kacc + (size_t)(aj + mb) * n + ak + mc;
                } else {  // (b, d): sum over a, c
                    const int y = x - O4, mb = y / SD, md = y % SD;
#pragma unroll
                    for (int ma = 0; ma < SA; ma++)
#pragma unroll
                        for (int mc = 0; mc < SC; mc++) a_ += cur[((ma * SB + mb) * SC + mc) * SD + md] * D[(size_t)(ai + ma) * n + ak + mc];
                    dst = og.kacc + (size_t)(aj + mb) * n + al + md;
                }
                atomicAdd(dst, deg * a_);
            }
        }
    }
    }  // member combinations
}

// ---------------------------------------------------------------------------------------------
// host: pair tables
// ---------------------------------------------------------------------------------------------
struct HostPairs {
    std::vector<int> sh, pp_off;
    std::vector<double> pp;
    int stride = 5;  // doubles per primitive pair (PP_STRIDE_G for a grouped basis)
    int cls_start[48], cls_count[48];  // class c(la,lb) = la(la+1)/2+lb  (grad.hip: la*8+lb... see there)
};

// A primitive pair enters every integral through c_a c_b K_ab / p times factors of order one (Boys functions <= 1, root
// weights, |P - A|^l <= R^l): below PRIM_PAIR_EPS times (1 + R)^(la + lb) it cannot move any integral by more than ~1e-18,
// six orders under the 1e-12 parity bar -- but the exp(-100) cut alone keeps, e.g., two 21-bohr^-2 s primitives on neighbouring
// carbons (K = 6e-29).  On a 20-atom cc-pVDZ molecule 32 % of the primitive quartets go (naphthalene / cc-pVTZ: 26 %).
constexpr double PRIM_PAIR_EPS = 1e-20;
inline bool prim_pair_negligible(double ck, double ab2, int lsum) {
    double f = std::fabs(ck);
    const double r1 = 1.0 + std::sqrt(ab2);
    for (int i = 0; i < lsum; i++) f *= r1;
    return f < PRIM_PAIR_EPS;
}

// shell pairs (i >= j) of the shells [s0, s1); unit >= 0: instead the "pairs" (i, unit shell) used by the 2- and
// 3-centre integrals (the unit shell has exponent 0, so P = A and K = 1)
static void build_pairs(const Basis &b, HostPairs &hp, int s0 = 0, int s1 = -1, int unit = -1) {
    if (s1 < 0) s1 = (int)b.shells.size();
    struct P { int a, b, cls, npp; std::vector<double> pp; };
    std::vector<P> all;
    all.reserve((size_t)(s1 - s0) * (s1 - s0 + 1) / 2);
    // grouped view (Basis::grouped): PP_STRIDE_G doubles per primitive pair -- four coefficient slots 2 xa + xb over the members
    // of the two groups (only s groups have a second member); otherwise five doubles, one coefficient
    const bool grp = b.grouped();
    const int stride = grp ? PP_STRIDE_G : 5;
    for (int i = s0; i < s1; i++)
        for (int j = (unit >= 0 ? unit : s0); j <= (unit >= 0 ? unit : i); j++) {
            int a = i, c = j;
            if (b.shells[c].l > b.shells[a].l) std::swap(a, c);
            const HostShell &A = b.shells[a], &B = b.shells[c];
            P pr;
            pr.a = a; pr.b = c; pr.cls = A.l * (A.l + 1) / 2 + B.l;
            double ab2 = 0;
            for (int d = 0; d < 3; d++) ab2 += (A.r[d] - B.r[d]) * (A.r[d] - B.r[d]);
            for (int ip = 0; ip < A.nprim; ip++)
                for (int jp = 0; jp < B.nprim; jp++) {
                    const double ea = b.exps[A.prim_off + ip], eb = b.exps[B.prim_off + jp], p = ea + eb;
                    const double arg = ea * eb / p * ab2;
                    if (arg > 100.0) continue;  // exp(-100) ~ 4e-44: numerically zero contribution
                    const double K = std::exp(-arg);
                    const double ca0 = b.coefs[A.prim_off + ip], cb0 = b.coefs[B.prim_off + jp];
                    double cf[4] = {ca0 * cb0 * K / p, 0.0, 0.0, 0.0};  // c_a c_b K_ab / p
                    if (grp) {
                        const double ca1 = b.coefs1[A.prim_off + ip], cb1 = b.coefs1[B.prim_off + jp];
                        cf[1] = ca0 * cb1 * K / p;
                        cf[2] = ca1 * cb0 * K / p;
                        cf[3] = ca1 * cb1 * K / p;
                    }
                    const double cmax = std::max(std::max(std::fabs(cf[0]), std::fabs(cf[1])), std::max(std::fabs(cf[2]), std::fabs(cf[3])));
                    if (prim_pair_negligible(cmax, ab2, A.l + B.l)) continue;
                    pr.pp.push_back(p);
                    for (int d = 0; d < 3; d++) pr.pp.push_back((ea * A.r[d] + eb * B.r[d]) / p);
                    pr.pp.push_back(cf[0]);
                    if (grp) {
                        // slots of an (l > 0, s) pair: xb alone (the s group is the SECOND one: slot 2 xa + xb with xa = 0)
                        pr.pp.push_back(cf[1]);
                        pr.pp.push_back(cf[2]);
                        pr.pp.push_back(cf[3]);
                    }
                }
            pr.npp = (int)pr.pp.size() / stride;
            all.push_back(std::move(pr));
        }
    std::stable_sort(all.begin(), all.end(), [](const P &x, const P &y) {
        if (x.cls != y.cls) return x.cls < y.cls;
        return x.npp > y.npp;
    });
    for (int c = 0; c < 48; c++) { hp.cls_start[c] = 0; hp.cls_count[c] = 0; }
    hp.pp_off.push_back(0);
    for (size_t n = 0; n < all.size(); n++) {
        const P &pr = all[n];
        if (hp.cls_count[pr.cls] == 0) hp.cls_start[pr.cls] = (int)n;
        hp.cls_count[pr.cls]++;
        hp.sh.push_back(pr.a);
        hp.sh.push_back(pr.b);
        hp.pp.insert(hp.pp.end(), pr.pp.begin(), pr.pp.end());
        hp.pp_off.push_back((int)hp.pp.size() / stride);
    }
    hp.stride = stride;
}

// depth-binned wave table of one DIAGONAL class launch (eri_split_lanes; kernel branch og.wtab): per (ket pair, bra depth bin) the
// waves that cover the bin's bra pairs (>= the ket pair) with PS lane groups per quartet.  The pairs of a class are sorted by
// primitive-pair count, descending, so the bins are ranges: wbin = the starts of the BRA class's bins relative to the class start
static_assert(SCREEN_NBIN == 8, "wave table packs the bin into three bits");
static void eri_wave_bins(int wbin[SCREEN_NBIN + 1], const HostPairs &hb, int b0, int nb) {
    for (int k = 0; k <= SCREEN_NBIN; k++) wbin[k] = 0;
    for (int i = 0; i < nb; i++) wbin[screen_bin(hb.pp_off[b0 + i + 1] - hb.pp_off[b0 + i]) + 1]++;
    for (int k = 0; k < SCREEN_NBIN; k++) wbin[k + 1] += wbin[k];
}
static void eri_wave_table(std::vector<int2> &wtab, int wbin[SCREEN_NBIN + 1], const HostPairs &hb, int b0, int nb, const HostPairs &hk,
                           int k0, int nk, bool same, int tpq) {
    eri_wave_bins(wbin, hb, b0, nb);
    wtab.clear();
    for (int ik = 0; ik < nk; ik++) {
        const int nkp = hk.pp_off[k0 + ik + 1] - hk.pp_off[k0 + ik];
        for (int bin = 0; bin < SCREEN_NBIN; bin++) {
            const int bs = (same && ik > wbin[bin]) ? ik : wbin[bin];
            const int per = 64 / (eri_split_lanes(bin, nkp, tpq) * tpq);
            for (int b1 = bs; b1 < wbin[bin + 1]; b1 += per) wtab.push_back(make_int2(ik * 8 + bin, b1));
        }
    }
}

// the same map for an OFF-DIAGONAL class launch as runs of ket pairs with equal primitive-pair count (the pairs of a class are
// sorted by that count): returns the number of waves
static long long eri_wave_runs(std::vector<WaveRun> &runs, int wbin[SCREEN_NBIN + 1], const HostPairs &hb, int b0, int nb,
                               const HostPairs &hk, int k0, int nk, int tpq) {
    eri_wave_bins(wbin, hb, b0, nb);
    runs.clear();
    long long nwave = 0;
    int prev = -1;
    for (int ik = 0; ik < nk; ik++) {
        const int nkp = hk.pp_off[k0 + ik + 1] - hk.pp_off[k0 + ik];
        if (nkp != prev) {
            WaveRun r;
            r.off = nwave; r.k0 = ik; r.pad_ = 0;
            r.cum[0] = 0;
            for (int bin = 0; bin < SCREEN_NBIN; bin++) {
                const int per = 64 / (eri_split_lanes(bin, nkp, tpq) * tpq), cnt = wbin[bin + 1] - wbin[bin];
                r.cum[bin + 1] = r.cum[bin] + (cnt + per - 1) / per;
            }
            r.W = r.cum[SCREEN_NBIN];
            if (r.W == 0) r.W = 1;  // (no bra pairs: never launched)
            runs.push_back(r);
            prev = nkp;
        }
        nwave += runs.back().W;
    }
    return nb > 0 ? nwave : 0;
}

}  // namespace dqc
