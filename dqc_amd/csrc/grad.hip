// grad.hip -- nuclear gradients of the SCF energy: the integral-derivative contractions.
//
// The reference obtains dE/dR by autograd through its integral wrappers (the "ip" derivative integrals,
// dqc/hamilton/intor/molintor.py:463-500) and the implicit-function backward of the SCF fixed point
// (dqc/qccalc/scf_qccalc.py:63-67, 109-113).  At a converged SCF point that derivative is the classic
// Hellmann-Feynman + Pulay expression, which needs no response equations:
//     dE/dR_A = sum D dh/dR_A - sum W dS/dR_A + 1/2 sum D D d(ab|cd)/dR_A [J - K/2 weights] + dE_xc/dR_A + dE_nn/dR_A
// This file provides the two integral-derivative terms, contracted on the fly (no derivative tensor is stored):
//   dqc_eri_grad   : g_A = sum_{a in A} sum_bcd (d_A a b|c d) [2 j D_ab D_cd - k D_ac D_bd]
//   dqc_int1e_grad : g_A = 2 sum_{a in A} sum_b [D_ab (d_A a|T + V|b) - W_ab (d_A a|b)],  g_C -= 2 sum_ab D_ab (d_A a|v_C|b)
// d/dA of a contracted Cartesian Gaussian of angular momentum l is an (l+1)-shell with coefficients 2 alpha c minus
// an (l-1)-shell; the 2e term therefore reuses the Rys shell-quartet kernel unchanged, with those companion shells
// first in the bra pair, in its GRAD output mode (eri_core.hpp).  Densities enter in the Cartesian AO basis
// (D_cart = T^T D T, T = dqc_cart2sph_matrix), so no solid-harmonic transform is needed on the device.
// Supported: shells up to g (companions up to h: the classes with a g shell or an h companion run through the runtime
// kernel of eri_generic.hpp).
#include "eri_generic.hpp"

namespace dqc {

// host copy of the solid-harmonic tables (the ones in common.hpp are device symbols)
namespace hostc2s {
#undef C2S_LMAX
#undef C2S_LEN
#define C2S_QUAL static const
#include "cart2sph.inc"
#undef C2S_QUAL
}  // namespace hostc2s

constexpr int GRAD_LMAX = 3;  // orbital shells up to f in the compile-time classes (companions up to g); above: eri_generic.hpp

static void cart_offsets(const Basis &b, int nsh, std::vector<int> &cao, int &ncart) {
    cao.resize(nsh);
    ncart = 0;
    for (int i = 0; i < nsh; i++) {
        cao[i] = ncart;
        ncart += (b.shells[i].l + 1) * (b.shells[i].l + 2) / 2;
    }
}

// ordered pairs (first in [f0, f1), second in [0, nsecond)), class index = la * 8 + lb, NO swap
static void build_pairs_ordered(const Basis &b, HostPairs &hp, int f0, int f1, int s0, int s1) {
    struct P { int a, b, cls, npp; std::vector<double> pp; };
    std::vector<P> all;
    for (int i = f0; i < f1; i++) {
        const HostShell &A = b.shells[i];
        if (A.l < 0) continue;  // placeholder (no down companion of an s shell)
        for (int j = s0; j < s1; j++) {
            const HostShell &B = b.shells[j];
            P pr;
            pr.a = i; pr.b = j; pr.cls = A.l * 8 + B.l;
            double ab2 = 0;
            for (int d = 0; d < 3; d++) ab2 += (A.r[d] - B.r[d]) * (A.r[d] - B.r[d]);
            for (int ip = 0; ip < A.nprim; ip++)
                for (int jp = 0; jp < B.nprim; jp++) {
                    const double ea = b.exps[A.prim_off + ip], eb = b.exps[B.prim_off + jp], p = ea + eb;
                    const double arg = ea * eb / p * ab2;
                    if (arg > 100.0) continue;
                    if (prim_pair_negligible(b.coefs[A.prim_off + ip] * b.coefs[B.prim_off + jp] * std::exp(-arg) / p, ab2, A.l + B.l)) continue;
                    pr.pp.push_back(p);
                    for (int d = 0; d < 3; d++) pr.pp.push_back((ea * A.r[d] + eb * B.r[d]) / p);
                    pr.pp.push_back(b.coefs[A.prim_off + ip] * b.coefs[B.prim_off + jp] * std::exp(-arg) / p);  // c_a c_b K_ab / p
                }
            pr.npp = (int)pr.pp.size() / 5;
            all.push_back(std::move(pr));
        }
    }
    std::stable_sort(all.begin(), all.end(), [](const P &x, const P &y) {
        if (x.cls != y.cls) return x.cls < y.cls;
        return x.npp > y.npp;
    });
    for (int c = 0; c < 48; c++) { hp.cls_start[c] = 0; hp.cls_count[c] = 0; }
    hp.sh.clear(); hp.pp.clear(); hp.pp_off.clear();
    hp.pp_off.push_back(0);
    for (size_t n = 0; n < all.size(); n++) {
        const P &pr = all[n];
        if (hp.cls_count[pr.cls] == 0) hp.cls_start[pr.cls] = (int)n;
        hp.cls_count[pr.cls]++;
        hp.sh.push_back(pr.a);
        hp.sh.push_back(pr.b);
        hp.pp.insert(hp.pp.end(), pr.pp.begin(), pr.pp.end());
        hp.pp_off.push_back((int)hp.pp.size() / 5);
    }
}

struct GradCtx {
    DevShells ds;
    DevPairs dbra, dket;
    const HostPairs *hbra, *hket;
    EriOut og;
    DevPool *pool = nullptr;  // stream-ordered pool for the wave tables (nullptr: flat task maps)
    SideStreams *side = nullptr;  // class launches dealt round-robin to the side streams (common.hpp)
    int wmap_depth = 1 << 30;  // class pairs whose deepest contraction has at least this many primitive quartets take the wave map
};

template <int LA, int LB, int LC, int LD>
static int launch_grad_class(const GradCtx &c, hipStream_t st) {
    using Cfg = EriCfg<LA, LB, LC, LD>;
    const int cb = LA * 8 + LB, ck = LC * (LC + 1) / 2 + LD;
    const int nb = c.hbra->cls_count[cb], nk = c.hket->cls_count[ck];
    if (nb == 0 || nk == 0 || hl_forced()) return 0;
    const long long ntask = (long long)nb * nk;
    const long long nblk = eri_num_blocks<Cfg>(nb, nk, ntask);
    auto kern = eri_kernel<LA, LB, LC, LD, ERI_OUT_GRAD>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES_G);
    if (c.side != nullptr) st = c.side->take();
    // (only where the class pair's deepest contraction -- first pair of each class: the lists are sorted by depth -- is worth splitting)
    const int dmax = (c.hbra->pp_off[c.hbra->cls_start[cb] + 1] - c.hbra->pp_off[c.hbra->cls_start[cb]]) *
                     (c.hket->pp_off[c.hket->cls_start[ck] + 1] - c.hket->pp_off[c.hket->cls_start[ck]]);
    if (Cfg::TPQ <= 16 && c.pool != nullptr && dmax >= c.wmap_depth) {
        // depth-binned wave map (eri_core.hpp: eri_split_lanes, eri_wave_table): the deep (companion, s) x (s, s) contractions were
        // serial chains in single lane groups
        std::vector<WaveRun> runs;
        EriOut o2 = c.og;
        const long long nwave = eri_wave_runs(runs, o2.wbin, *c.hbra, c.hbra->cls_start[cb], nb, *c.hket, c.hket->cls_start[ck], nk, Cfg::TPQ);
        if (nwave == 0) return 0;
        WaveRun *d_runs = nullptr;
        if (c.pool->upload(&d_runs, runs, st)) { set_error("dqc_eri_grad: device upload failed"); return DQC_ENOMEM; }
        o2.wruns = d_runs;
        o2.nruns = (int)runs.size();
        hipLaunchKernelGGL(kern, dim3((unsigned)((nwave + 3) / 4)), dim3(256), Cfg::LDS_BYTES_G, st, (double *)nullptr, c.ds, c.dbra, c.dket,
                           c.hbra->cls_start[cb], nb, c.hket->cls_start[ck], nk, 0, nwave, o2);
        DQC_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES_G, st, (double *)nullptr, c.ds, c.dbra, c.dket,
                       c.hbra->cls_start[cb], nb, c.hket->cls_start[ck], nk, 0, ntask, c.og);
    DQC_CHECK_LAUNCH();
    return 0;
}

template <int LA, int LB>
static int launch_grad_bra(const GradCtx &c, hipStream_t st) {
    int rc;
#define DQC_GK(LC, LD) \
    if ((rc = launch_grad_class<LA, LB, LC, LD>(c, st))) return rc;
    DQC_GK(0, 0) DQC_GK(1, 0) DQC_GK(1, 1) DQC_GK(2, 0) DQC_GK(2, 1) DQC_GK(2, 2) DQC_GK(3, 0) DQC_GK(3, 1) DQC_GK(3, 2) DQC_GK(3, 3)
#undef DQC_GK
    return 0;
}

// density-fitting gradients: ket pairs are (auxiliary shell, unit), classes (LC, 0), LC up to f
template <int LA, int LB>
static int launch_grad_bra_df(const GradCtx &c, hipStream_t st) {
    int rc;
#define DQC_GK(LC) \
    if ((rc = launch_grad_class<LA, LB, LC, 0>(c, st))) return rc;
    DQC_GK(0) DQC_GK(1) DQC_GK(2) DQC_GK(3)
#undef DQC_GK
    return 0;
}

static int launch_grad_generic(const GradCtx &c, bool df, int lbmax, hipStream_t st);

static int launch_grad_df3c(const GradCtx &c, hipStream_t st) {
    int rc;
    if ((rc = launch_grad_generic(c, true, DQC_LMAX, st))) return rc;
#define DQC_GB(LA) \
    if ((rc = launch_grad_bra_df<LA, 0>(c, st)) || (rc = launch_grad_bra_df<LA, 1>(c, st)) || (rc = launch_grad_bra_df<LA, 2>(c, st)) || (rc = launch_grad_bra_df<LA, 3>(c, st))) return rc;
    DQC_GB(0) DQC_GB(1) DQC_GB(2) DQC_GB(3) DQC_GB(4)
#undef DQC_GB
    return 0;
}

static int launch_grad_df2c(const GradCtx &c, hipStream_t st) {
    int rc;
    if ((rc = launch_grad_generic(c, true, 0, st))) return rc;
    if ((rc = launch_grad_bra_df<0, 0>(c, st)) || (rc = launch_grad_bra_df<1, 0>(c, st)) || (rc = launch_grad_bra_df<2, 0>(c, st)) ||
        (rc = launch_grad_bra_df<3, 0>(c, st)) || (rc = launch_grad_bra_df<4, 0>(c, st)))
        return rc;
    return 0;
}

// the classes outside the compile-time set (h companions, g shells) through the runtime kernel; ket pairs (lc >= ld), or
// (auxiliary shell, unit) when `df`
static int launch_grad_generic(const GradCtx &c, bool df, int lbmax, hipStream_t st) {
    for (int la = 0; la <= DQC_LMAX + 1; la++)
        for (int lb = 0; lb <= lbmax; lb++)
            for (int lc = 0; lc <= DQC_LMAX; lc++)
                for (int ld = 0; ld <= (df ? 0 : lc); ld++) {
                    if (!hl_forced() && la <= GRAD_LMAX + 1 && lb <= GRAD_LMAX && lc <= GRAD_LMAX) continue;
                    const int cb = la * 8 + lb, ck = lc * (lc + 1) / 2 + ld;
                    int rc = launch_hl<ERI_OUT_GRAD>(nullptr, c.ds, c.dbra, c.dket, c.hbra->cls_start[cb], c.hbra->cls_count[cb],
                                                     c.hket->cls_start[ck], c.hket->cls_count[ck], 0, c.og, la, lb, lc, ld, st);
                    if (rc) return rc;
                }
    return 0;
}

static int launch_grad_all(const GradCtx &c, hipStream_t st) {
    int rc;
    if ((rc = launch_grad_generic(c, false, DQC_LMAX, st))) return rc;
#define DQC_GB(LA) \
    if ((rc = launch_grad_bra<LA, 0>(c, st)) || (rc = launch_grad_bra<LA, 1>(c, st)) || (rc = launch_grad_bra<LA, 2>(c, st)) || (rc = launch_grad_bra<LA, 3>(c, st))) return rc;
    DQC_GB(0) DQC_GB(1) DQC_GB(2) DQC_GB(3) DQC_GB(4)
#undef DQC_GB
    return 0;
}

// ---------------------------------------------------------------------------------------------
// one-electron part: one thread per ORDERED shell pair (a, b), derivative on a's centre
// ---------------------------------------------------------------------------------------------
constexpr int G1 = DQC_LMAX + 2, G3 = DQC_LMAX + 3;  // bra index up to l + 1 (the derivative), ket index up to l + 2 (kinetic)

DQC_DEV void overlap_1d_g(double s[G1][G3], int la, int lb, double PA, double PB, double hp) {
    s[0][0] = 1.0;
    for (int i = 0; i < la; i++) s[i + 1][0] = PA * s[i][0] + (i ? i * hp * s[i - 1][0] : 0.0);
    for (int j = 0; j < lb; j++)
        for (int i = 0; i <= la; i++)
            s[i][j + 1] = PB * s[i][j] + (j ? j * hp * s[i][j - 1] : 0.0) + (i ? i * hp * s[i - 1][j] : 0.0);
}

template <int N>
DQC_DEV void nuc_grad_accumulate(double g3[3], int la, int lb, int nb, double a, double p, const double *P, const double *A,
                                 const double *AB, const double *C, double pref, const double *dblk, size_t nc) {
    // sum_{ca,cb} D[ca][cb] * d/dA (ca | 1/|r - C| | cb), added to g3 (pref carries -Z, coefficients, K, 2 pi / p)
    const double X = p * ((P[0] - C[0]) * (P[0] - C[0]) + (P[1] - C[1]) * (P[1] - C[1]) + (P[2] - C[2]) * (P[2] - C[2]));
    double u[N], w[N];
    rys_roots<N>(X, u, w);
    for (int r = 0; r < N; r++) {
        double g[3][2 * DQC_LMAX + 2][G1];
        const double b10 = 0.5 * (1.0 - u[r]) / p;
        for (int d = 0; d < 3; d++) {
            const double c00 = (P[d] - A[d]) - u[r] * (P[d] - C[d]);
            g[d][0][0] = 1.0;
            g[d][1][0] = c00;
            for (int n = 1; n < la + 1 + lb; n++) g[d][n + 1][0] = c00 * g[d][n][0] + n * b10 * g[d][n - 1][0];
            for (int j = 1; j <= lb; j++)
                for (int i = 0; i <= la + 1 + lb - j; i++) g[d][i][j] = g[d][i + 1][j - 1] + AB[d] * g[d][i][j - 1];
        }
        const double wr = pref * w[r];
        int ca = 0;
        for (int ax = la; ax >= 0; ax--)
            for (int ay = la - ax; ay >= 0; ay--, ca++) {
                const int az = la - ax - ay;
                int cb = 0;
                for (int bx = lb; bx >= 0; bx--)
                    for (int by = lb - bx; by >= 0; by--, cb++) {
                        const int bz = lb - bx - by;
                        const double dd = wr * dblk[ca * nc + cb];
                        const double gx = g[0][ax][bx], gy = g[1][ay][by], gz = g[2][az][bz];
                        const double dgx = 2.0 * a * g[0][ax + 1][bx] - (ax ? ax * g[0][ax - 1][bx] : 0.0);
                        const double dgy = 2.0 * a * g[1][ay + 1][by] - (ay ? ay * g[1][ay - 1][by] : 0.0);
                        const double dgz = 2.0 * a * g[2][az + 1][bz] - (az ? az * g[2][az - 1][bz] : 0.0);
                        g3[0] += dd * dgx * gy * gz;
                        g3[1] += dd * gx * dgy * gz;
                        g3[2] += dd * gx * gy * dgz;
                    }
            }
    }
}

__global__ __launch_bounds__(64) void int1e_grad_kernel(double *__restrict__ grad, DevShells sh, const int *__restrict__ cao,
                                  const int *__restrict__ sh_atom, const double *__restrict__ dcart,
                                  const double *__restrict__ wcart, int ncart, int natm,
                                  const double *__restrict__ atom_xyz, const double *__restrict__ atom_z) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= sh.nsh * sh.nsh) return;
    const int ish = pair / sh.nsh, jsh = pair % sh.nsh;
    const int la = sh.l[ish], lb = sh.l[jsh];
    const int nb = (lb + 1) * (lb + 2) / 2;
    const size_t nc = ncart;
    const double A[3] = {sh.xyz[ish * 3], sh.xyz[ish * 3 + 1], sh.xyz[ish * 3 + 2]};
    const double B[3] = {sh.xyz[jsh * 3], sh.xyz[jsh * 3 + 1], sh.xyz[jsh * 3 + 2]};
    const double AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
    const double ab2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
    const double *dblk = dcart + (size_t)cao[ish] * nc + cao[jsh];
    const double *wblk = wcart + (size_t)cao[ish] * nc + cao[jsh];
    double ga[3] = {0.0, 0.0, 0.0};  // basis-centre terms, go to a's atom
    const bool one_nuc = (int)gridDim.y >= natm;
    double gn[3] = {0.0, 0.0, 0.0};  // operator-derivative term of this thread's nucleus (one_nuc)
    for (int ip = 0; ip < sh.nprim[ish]; ip++)
        for (int jp = 0; jp < sh.nprim[jsh]; jp++) {
            const double a = sh.exps[sh.prim_off[ish] + ip], b = sh.exps[sh.prim_off[jsh] + jp];
            const double cc = sh.coefs[sh.prim_off[ish] + ip] * sh.coefs[sh.prim_off[jsh] + jp];
            const double p = a + b, hp = 0.5 / p;
            const double K = exp(-a * b / p * ab2);
            double P[3];
            for (int d = 0; d < 3; d++) P[d] = (a * A[d] + b * B[d]) / p;
            // ---- overlap and kinetic
            double s[3][G1][G3];
            for (int d = 0; d < 3; d++) overlap_1d_g(s[d], la + 1, lb + 2, P[d] - A[d], P[d] - B[d], hp);
            const double pref = cc * K * pow(M_PI / p, 1.5);
            auto tt = [&](int d, int i, int j) {  // -1/2 d2/dx2 on the ket index, 1D
                return -2.0 * b * b * s[d][i][j + 2] + b * (2 * j + 1) * s[d][i][j] - (j >= 2 ? 0.5 * j * (j - 1) * s[d][i][j - 2] : 0.0);
            };
            int ca = 0;
            // (the launch is split over the nuclei along grid.y -- one thread per shell pair walked all primitive pairs and all
            // nuclei alone: 9 ms for a 20-atom molecule on 144 waves; the overlap / kinetic part belongs to split 0)
            for (int ax = (blockIdx.y == 0 ? la : -1); ax >= 0; ax--)
                for (int ay = la - ax; ay >= 0; ay--, ca++) {
                    const int az = la - ax - ay;
                    const int av[3] = {ax, ay, az};
                    int cb = 0;
                    for (int bx = lb; bx >= 0; bx--)
                        for (int by = lb - bx; by >= 0; by--, cb++) {
                            const int bz = lb - bx - by;
                            const int bv[3] = {bx, by, bz};
                            double sv[3], tv[3], dsv[3], dtv[3];
                            for (int d = 0; d < 3; d++) {
                                sv[d] = s[d][av[d]][bv[d]];
                                tv[d] = tt(d, av[d], bv[d]);
                                dsv[d] = 2.0 * a * s[d][av[d] + 1][bv[d]] - (av[d] ? av[d] * s[d][av[d] - 1][bv[d]] : 0.0);
                                dtv[d] = 2.0 * a * tt(d, av[d] + 1, bv[d]) - (av[d] ? av[d] * tt(d, av[d] - 1, bv[d]) : 0.0);
                            }
                            const double dd = pref * dblk[ca * nc + cb], ww = pref * wblk[ca * nc + cb];
                            for (int d = 0; d < 3; d++) {
                                const int e = (d + 1) % 3, f = (d + 2) % 3;
                                const double dS = dsv[d] * sv[e] * sv[f];
                                const double dT = dtv[d] * sv[e] * sv[f] + dsv[d] * tv[e] * sv[f] + dsv[d] * sv[e] * tv[f];
                                ga[d] += dd * dT - ww * dS;
                            }
                        }
                }
            // ---- nuclear attraction, nucleus by nucleus (operator derivative by translational invariance)
            const int nroots = (la + 1 + lb) / 2 + 1;
            for (int ic = blockIdx.y; ic < natm; ic += gridDim.y) {
                const double prn = -atom_z[ic] * cc * K * 2.0 * M_PI / p;
                const double *C = atom_xyz + ic * 3;
                double g3[3] = {0.0, 0.0, 0.0};
                switch (nroots) {
                case 1: nuc_grad_accumulate<1>(g3, la, lb, nb, a, p, P, A, AB, C, prn, dblk, nc); break;
                case 2: nuc_grad_accumulate<2>(g3, la, lb, nb, a, p, P, A, AB, C, prn, dblk, nc); break;
                case 3: nuc_grad_accumulate<3>(g3, la, lb, nb, a, p, P, A, AB, C, prn, dblk, nc); break;
                case 4: nuc_grad_accumulate<4>(g3, la, lb, nb, a, p, P, A, AB, C, prn, dblk, nc); break;
                default: nuc_grad_accumulate<5>(g3, la, lb, nb, a, p, P, A, AB, C, prn, dblk, nc); break;
                }
                for (int d = 0; d < 3; d++) {
                    ga[d] += g3[d];
                    // (one nucleus per thread when the launch has a y-slice per nucleus: its operator-derivative term is summed over
                    // the primitive pairs and added once -- 35 M atomics on 60 addresses were most of this kernel's time)
                    if (one_nuc) gn[d] += g3[d];
                    else atomicAdd(&grad[ic * 3 + d], -2.0 * g3[d]);
                }
            }
        }
    for (int d = 0; d < 3; d++) atomicAdd(&grad[sh_atom[ish] * 3 + d], 2.0 * ga[d]);
    if (one_nuc)
        for (int d = 0; d < 3; d++) atomicAdd(&grad[blockIdx.y * 3 + d], -2.0 * gn[d]);
}

}  // namespace dqc

extern "C" {

int dqc_ncart(const int *bas, int nbas) {
    int n = 0;
    for (int i = 0; i < nbas; i++) n += (bas[i * 8 + 1] + 1) * (bas[i * 8 + 1] + 2) / 2;
    return n;
}

int dqc_cart2sph_matrix(double *h_out, const int *bas, int nbas) {
    // HOST array (nao, ncart), block diagonal: chi_m = sum_c T[m][c] g_c for every shell
    using namespace dqc;
    int nao = 0, ncart = dqc_ncart(bas, nbas);
    for (int i = 0; i < nbas; i++) nao += 2 * bas[i * 8 + 1] + 1;
    for (size_t i = 0; i < (size_t)nao * ncart; i++) h_out[i] = 0.0;
    int ao = 0, co = 0;
    for (int i = 0; i < nbas; i++) {
        const int l = bas[i * 8 + 1];
        if (l > C2S_LMAX) { set_error("dqc_cart2sph_matrix: angular momentum above g"); return DQC_EINVAL; }
        const int ns = 2 * l + 1, ncl = (l + 1) * (l + 2) / 2;
        for (int m = 0; m < ns; m++)
            for (int c = 0; c < ncl; c++) h_out[(size_t)(ao + m) * ncart + co + c] = hostc2s::C2S[hostc2s::C2S_OFF[l] + m * ncl + c];
        ao += ns;
        co += ncl;
    }
    return DQC_OK;
}

int dqc_eri_grad(double *d_grad, const double *d_dcart, double jscale, double kscale, const int *atm, int natm, const int *bas, int nbas,
                 const double *env, int nenv, void *stream) {
    // d_grad (natm, 3) += sum_{a in A} sum_bcd (d_A a b|c d) [2 jscale D_ab D_cd - kscale D_ac D_bd]; d_dcart (ncart, ncart)
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if (nbas == 0) return DQC_OK;
    const int N = nbas;
    std::vector<int> cao, sh_atom(N);
    int ncart;
    cart_offsets(b, N, cao, ncart);
    for (int i = 0; i < N; i++) sh_atom[i] = b.shells[i].atom;
    // companion shells: [N, 2N) up (l+1, coefficients 2 alpha c), [2N, 3N) down (l-1; l = -1 marks "none")
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < N; i++) {
            HostShell h = b.shells[i];
            h.l = pass == 0 ? h.l + 1 : h.l - 1;
            const int po = (int)b.exps.size();
            for (int p = 0; p < h.nprim; p++) {
                const double e = b.exps[b.shells[i].prim_off + p], c = b.coefs[b.shells[i].prim_off + p];
                b.exps.push_back(e);
                b.coefs.push_back(pass == 0 ? 2.0 * e * c : c);
            }
            h.prim_off = po;
            b.shells.push_back(h);
        }
    HostPairs hup, hdown, hket;
    build_pairs_ordered(b, hup, N, 2 * N, 0, N);
    build_pairs_ordered(b, hdown, 2 * N, 3 * N, 0, N);
    build_pairs(b, hket, 0, N);
    // upload_shells needs l >= 0 everywhere: placeholders become s shells (never referenced by a pair)
    for (HostShell &h : b.shells)
        if (h.l < 0) h.l = 0;
    if ((rc = boys_table_ensure())) return rc;
    DevPool pool;
    GradCtx c;
    // the depth-binned wave map of the fill (eri_core.hpp) is OFF here: measured on a 20-atom cc-pVDZ gradient it is slower for every
    // depth threshold -- 0.160 s (classes deeper than 64 primitive quartets), 0.149 (256), 0.140 (1024) against 0.107 s for the flat
    // task maps (DQC_GRAD_WMAP = the depth threshold for A/B runs, 0 / unset = never)
    DevPool wpool(st);
    static const int wmap_env = [] { const char *e = getenv("DQC_GRAD_WMAP"); return e ? atoi(e) : 0; }();
    if (wmap_env > 0) { c.pool = &wpool; c.wmap_depth = wmap_env; }
    if ((rc = upload_shells(c.ds, b, pool, st))) { set_error("dqc_eri_grad: device upload failed"); return rc; }
    auto up = [&](HostPairs &hp, DevPairs &dp) {
        int *d_sh = nullptr, *d_off = nullptr;
        double *d_pp = nullptr;
        int r;
        if ((r = pool.upload(&d_sh, hp.sh, st)) || (r = pool.upload(&d_off, hp.pp_off, st)) || (r = pool.upload(&d_pp, hp.pp, st)))
            return r;
        dp = DevPairs{d_sh, d_off, d_pp};
        return 0;
    };
    DevPairs dup, ddown, dket;
    int *d_cao = nullptr, *d_atom = nullptr;
    if ((rc = up(hup, dup)) || (rc = up(hdown, ddown)) || (rc = up(hket, dket)) || (rc = pool.upload(&d_cao, cao, st)) ||
        (rc = pool.upload(&d_atom, sh_atom, st))) {
        set_error("dqc_eri_grad: device upload failed");
        return rc;
    }
    const int nslot = 64;
    double *d_part = nullptr;
    if (hipMalloc((void **)&d_part, sizeof(double) * nslot * natm * 3) != hipSuccess) { set_error("dqc_eri_grad: out of memory"); return DQC_ENOMEM; }
    pool.ptrs.push_back(d_part);
    DQC_HIP(hipMemsetAsync(d_part, 0, sizeof(double) * nslot * natm * 3, st));
    c.dket = dket;
    c.hket = &hket;
    c.og = EriOut{0, 0, 0, 0};
    c.og.dcart = d_dcart; c.og.ncart = ncart; c.og.cao = d_cao; c.og.sh_atom = d_atom; c.og.gpart = d_part;
    c.og.nslot = nslot; c.og.natm = natm; c.og.norig = N; c.og.jscale = jscale; c.og.kscale = kscale;
    SideJoin sj;  // (declared after the scratch pools of this call: destroyed, i.e. joined, before they are released)
    if ((c.side = side_streams()) != nullptr && (rc = c.side->fork(st))) { set_error("dqc_eri_grad: stream fork failed"); return rc; }
    sj.arm(c.side, st);
    c.dbra = dup; c.hbra = &hup; c.og.dirn = +1;
    if ((rc = launch_grad_all(c, st))) return rc;
    c.dbra = ddown; c.hbra = &hdown; c.og.dirn = -1;
    if ((rc = launch_grad_all(c, st))) return rc;
    if ((rc = sj.done())) { set_error("dqc_eri_grad: stream join failed"); return rc; }
    // fold the slots into d_grad on the host side of the stream: tiny
    std::vector<double> part((size_t)nslot * natm * 3), g((size_t)natm * 3);
    DQC_HIP(hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
    DQC_HIP(hipMemcpyAsync(g.data(), d_grad, sizeof(double) * g.size(), hipMemcpyDeviceToHost, st));
    DQC_HIP(hipStreamSynchronize(st));
    for (int sidx = 0; sidx < nslot; sidx++)
        for (size_t i = 0; i < g.size(); i++) g[i] += part[(size_t)sidx * natm * 3 + i];
    DQC_HIP(hipMemcpyAsync(d_grad, g.data(), sizeof(double) * g.size(), hipMemcpyHostToDevice, st));
    DQC_HIP(hipStreamSynchronize(st));
    return DQC_OK;
}

int dqc_int1e_grad(double *d_grad, const double *d_dcart, const double *d_wcart, const int *atm, int natm, const int *bas,
                   int nbas, const double *env, int nenv, const double *zs, void *stream) {
    // d_grad (natm, 3) += 2 sum_{a in A} sum_b [D_ab (d_A a|T + V|b) - W_ab (d_A a|b)]  and the Hellmann-Feynman
    // term of every nucleus; d_dcart / d_wcart: Cartesian-basis density and energy-weighted density (ncart, ncart)
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, zs);
    if (rc) return rc;
    if (nbas == 0) return DQC_OK;
    std::vector<int> cao, sh_atom(nbas);
    int ncart;
    cart_offsets(b, nbas, cao, ncart);
    for (int i = 0; i < nbas; i++) sh_atom[i] = b.shells[i].atom;
    DevPool pool;
    DevShells ds;
    if ((rc = upload_shells(ds, b, pool, st))) { set_error("dqc_int1e_grad: device upload failed"); return rc; }
    double *d_xyz = nullptr, *d_z = nullptr;
    int *d_cao = nullptr, *d_atom = nullptr;
    if ((rc = pool.upload(&d_xyz, b.atom_xyz, st)) || (rc = pool.upload(&d_z, b.atom_z, st)) ||
        (rc = pool.upload(&d_cao, cao, st)) || (rc = pool.upload(&d_atom, sh_atom, st))) {
        set_error("dqc_int1e_grad: device upload failed");
        return rc;
    }
    const int npair = nbas * nbas;
    hipLaunchKernelGGL(int1e_grad_kernel, dim3((npair + 63) / 64, natm < 32 ? natm : 32), dim3(64), 0, st, d_grad, ds, d_cao, d_atom, d_dcart, d_wcart,
                       ncart, natm, d_xyz, d_z);
    DQC_CHECK_LAUNCH();
    DQC_HIP(hipStreamSynchronize(st));
    return DQC_OK;
}

int dqc_df_grad(double *d_grad, const double *d_dcart, const double *d_ccart, const int *atm, int natm, const int *bas,
                int nbas, const double *env, int nenv, int sh0, int sh1, int k0, int k1, void *stream) {
    // gradient of the density-fitted Coulomb energy E_J = 1/2 t^T M^-1 t (t_k = sum D_ij (ij|k), M = (k|l), c = M^-1 t):
    //   d_grad (natm, 3) += sum D_ij c_k d(ij|k) - 1/2 sum c_k c_l d(k|l)
    // over the CONCATENATED tables (orbital shells [sh0, sh1), auxiliary shells [k0, k1)); d_dcart (ncart, ncart) and
    // d_ccart (ncart): density matrix and fit coefficients in the Cartesian basis of ALL shells of the table
    // (zero outside the orbital block / auxiliary segment).
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if (sh0 < 0 || sh1 > nbas || sh0 > sh1 || k0 < 0 || k1 > nbas || k0 > k1) { set_error("dqc_df_grad: shell ranges outside the table"); return DQC_EINVAL; }
    if (sh0 == sh1 || k0 == k1) return DQC_OK;
    const int N = nbas;
    std::vector<int> cao, sh_atom(3 * N + 1, 0);
    int ncart;
    cart_offsets(b, N, cao, ncart);
    cao.resize(3 * N + 1, 0);
    for (int i = 0; i < N; i++) sh_atom[i] = b.shells[i].atom;
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < N; i++) {
            HostShell h = b.shells[i];
            h.l = pass == 0 ? h.l + 1 : h.l - 1;
            const int po = (int)b.exps.size();
            for (int p = 0; p < h.nprim; p++) {
                const double e = b.exps[b.shells[i].prim_off + p], c = b.coefs[b.shells[i].prim_off + p];
                b.exps.push_back(e);
                b.coefs.push_back(pass == 0 ? 2.0 * e * c : c);
            }
            h.prim_off = po;
            b.shells.push_back(h);
        }
    HostShell u;
    u.atom = 0; u.l = 0; u.nprim = 1; u.ao_off = b.nao; u.prim_off = (int)b.exps.size();
    u.r[0] = u.r[1] = u.r[2] = 0.0;
    b.exps.push_back(0.0);
    b.coefs.push_back(1.0);  // GRAD mode contracts CARTESIAN blocks: the unit function is the Cartesian s function 1
                             // (df.hip uses sqrt(4 pi) because there the l = 0 solid-harmonic factor is applied)
    b.shells.push_back(u);
    const int unit = 3 * N;
    HostPairs h3up, h3down, h2up, h2down, hket;
    build_pairs_ordered(b, h3up, N + sh0, N + sh1, sh0, sh1);
    build_pairs_ordered(b, h3down, 2 * N + sh0, 2 * N + sh1, sh0, sh1);
    build_pairs_ordered(b, h2up, N + k0, N + k1, unit, unit + 1);
    build_pairs_ordered(b, h2down, 2 * N + k0, 2 * N + k1, unit, unit + 1);
    build_pairs(b, hket, k0, k1, unit);
    for (HostShell &h : b.shells)
        if (h.l < 0) h.l = 0;
    if ((rc = boys_table_ensure())) return rc;
    DevPool pool;
    GradCtx c;
    if ((rc = upload_shells(c.ds, b, pool, st))) { set_error("dqc_df_grad: device upload failed"); return rc; }
    auto up = [&](HostPairs &hp, DevPairs &dp) {
        int *d_sh = nullptr, *d_off = nullptr;
        double *d_pp = nullptr;
        int r;
        if ((r = pool.upload(&d_sh, hp.sh, st)) || (r = pool.upload(&d_off, hp.pp_off, st)) || (r = pool.upload(&d_pp, hp.pp, st)))
            return r;
        dp = DevPairs{d_sh, d_off, d_pp};
        return 0;
    };
    DevPairs d3up, d3down, d2up, d2down, dket;
    int *d_cao = nullptr, *d_atom = nullptr;
    if ((rc = up(h3up, d3up)) || (rc = up(h3down, d3down)) || (rc = up(h2up, d2up)) || (rc = up(h2down, d2down)) ||
        (rc = up(hket, dket)) || (rc = pool.upload(&d_cao, cao, st)) || (rc = pool.upload(&d_atom, sh_atom, st))) {
        set_error("dqc_df_grad: device upload failed");
        return rc;
    }
    const int nslot = 64;
    double *d_part = nullptr;
    if (hipMalloc((void **)&d_part, sizeof(double) * nslot * natm * 3) != hipSuccess) { set_error("dqc_df_grad: out of memory"); return DQC_ENOMEM; }
    pool.ptrs.push_back(d_part);
    DQC_HIP(hipMemsetAsync(d_part, 0, sizeof(double) * nslot * natm * 3, st));
    c.dket = dket;
    c.hket = &hket;
    c.og = EriOut{0, 0, 0, 0};
    c.og.dcart = d_dcart; c.og.ccart = d_ccart; c.og.ncart = ncart; c.og.cao = d_cao; c.og.sh_atom = d_atom; c.og.gpart = d_part;
    c.og.nslot = nslot; c.og.natm = natm; c.og.norig = N;
    c.og.gmode = 1;
    c.dbra = d3up; c.hbra = &h3up; c.og.dirn = +1;
    if ((rc = launch_grad_df3c(c, st))) return rc;
    c.dbra = d3down; c.hbra = &h3down; c.og.dirn = -1;
    if ((rc = launch_grad_df3c(c, st))) return rc;
    c.og.gmode = 2;
    c.dbra = d2up; c.hbra = &h2up; c.og.dirn = +1;
    if ((rc = launch_grad_df2c(c, st))) return rc;
    c.dbra = d2down; c.hbra = &h2down; c.og.dirn = -1;
    if ((rc = launch_grad_df2c(c, st))) return rc;
    std::vector<double> part((size_t)nslot * natm * 3), g((size_t)natm * 3);
    DQC_HIP(hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
    DQC_HIP(hipMemcpyAsync(g.data(), d_grad, sizeof(double) * g.size(), hipMemcpyDeviceToHost, st));
    DQC_HIP(hipStreamSynchronize(st));
    for (int sidx = 0; sidx < nslot; sidx++)
        for (size_t i = 0; i < g.size(); i++) g[i] += part[(size_t)sidx * natm * 3 + i];
    DQC_HIP(hipMemcpyAsync(d_grad, g.data(), sizeof(double) * g.size(), hipMemcpyHostToDevice, st));
    DQC_HIP(hipStreamSynchronize(st));
    return DQC_OK;
}

}  // extern "C"
