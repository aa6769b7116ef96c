// eri_generic.hpp -- the shell-quartet kernel with RUNTIME angular momenta: every class that holds a g shell (or, in the
// gradient path, an h companion), i.e. what the compile-time classes of eri_core.hpp (s ... f) leave out.  Same algorithm
// (Rys quadrature, 2D integrals per (direction, root), Cartesian products, solid harmonics, the five output modes), arranged
// so that nothing is sized by the class at compile time:
//   * (gg|gg) is 50 625 Cartesian integrals per shell quartet -- 198 accumulators per lane in a 256-lane group.  The outputs
//     are therefore produced in SLICES: one Cartesian component (ax, ay, az) of shell a at a time (<= 15^3 = 3375 outputs,
//     14 per lane).  The primitive-quartet loop runs once per slice; what is recomputed is the vertical recurrence only
//     (27 lanes, ~90 terms), the expensive part -- the products of the three 2D integrals -- is not.  Small classes whose
//     2D integrals of every a-component fit the LDS budget run as ONE slice (`full`).
//   * a slice needs the 2D integrals at i = a_d only, so the LDS table is 3 n_roots (lb+1)(lc+1)(ld+1) doubles (27 KB for
//     (gg|gg)) instead of 135 KB.
//   * the vertical recurrence runs in registers, two rolling columns of <= 10 terms; both horizontal recurrences are done by the
//     whole lane group as binomial transfers out of LDS (coefficients C(j, a) AB^(j-a) tabulated once per quartet).
//   * the slice is transformed to solid harmonics on b, c, d in LDS; the a-transform accumulates into the quartet's spherical
//     block, which stays in LDS (<= 9^4 doubles) until the scatter.  The gradient mode contracts the Cartesian slice directly.
// Lane groups of 16, 64 or 256 lanes per shell quartet, chosen per class on the host (hl_plan).
#pragma once
#include "eri_core.hpp"

namespace dqc {

constexpr int HL_NPT = 14;   // Cartesian outputs per lane and slice
constexpr int HL_LDS_LIMIT = 150 * 1024;

struct HlClass {
    int la, lb, lc, ld;
    int nr;      // Rys roots
    int full;    // 1: one slice holding every Cartesian component of shell a; 0: one component per slice
    int region;  // doubles per shell quartet in LDS (odd)
    int off_h, off_G, off_c, off_jk, off_s;  // areas of a region: g at 0, h, G, transfer coefficients, J/K sums, spherical block
    int tab_doubles;                         // block-level element descriptors of the two transfer steps, in front of the regions
};

// root r of the n-point rule (n at run time; tables and asymptotics as rys_root1)
DQC_DEV void rys_root1_rt(int N, double X, int r, double &u, double &w) {
    if (X >= 35.0 + 5.0 * N) {
        const double ix = 1.0 / X;
        u = RYS_HERM_X2[N][r] * ix;
        w = RYS_HERM_W[N][r] * sqrt(ix);
        return;
    }
    const int it = (int)(X * (1.0 / 2.5));
    const double x = (X - (it * 2.5 + 1.25)) * (1.0 / 1.25);
    const double *cu = RYS_TAB + RYS_OFF[N - 1] + ((size_t)it * (2 * N) + r) * (RYS_DEG + 1);
    const double *cw = cu + N * (RYS_DEG + 1);
    const double x2 = 2.0 * x;
    double a1 = 0, a2 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int k = RYS_DEG; k >= 1; k--) {
        const double t = x2 * a1 - a2 + cu[k]; a2 = a1; a1 = t;
        const double s = x2 * b1 - b2 + cw[k]; b2 = b1; b1 = s;
    }
    u = x * a1 - a2 + cu[0];
    w = x * b1 - b2 + cw[0];
}

template <int TPQ, int MODE>
__global__ __launch_bounds__(256) void eri_hl_kernel(double *__restrict__ tiles, DevShells sh, DevPairs prs, DevPairs prk, int b0,
                                                     int nb, int k0, int nk, int same, long long ntask, EriOut og, HlClass hc) {
    constexpr int QPB = 256 / TPQ;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int s_maxq;
    const int tid = threadIdx.x;
    const int q = tid / TPQ, s = tid % TPQ;
    double *reg = lds + hc.tab_doubles + (size_t)q * hc.region;
    const int la = hc.la, lb = hc.lb, lc = hc.lc, ld = hc.ld, nr = hc.nr;
    const int nmax = la + lb, mmax = lc + ld, M1 = mmax + 1;
    const int nca = c_ncart(la), ncb = c_ncart(lb), ncc = c_ncart(lc), ncd = c_ncart(ld);
    const int nitem = 3 * nr;
    const int gsz = (nmax + 1) * M1;                  // vertical-recurrence table of one item
    const int ni = hc.full ? la + 1 : 1;              // bra-a indices the 2D tables hold
    const int hsz = ni * (lb + 1) * M1;
    const int g1 = (lb + 1) * (lc + 1) * (ld + 1), Gsz = ni * g1;
    double *gl = reg, *hl = reg + hc.off_h, *Gl = reg + hc.off_G, *ctab = reg + hc.off_c;
    // element descriptors of the transfer steps A2 / A3, the same for every quartet, slice and primitive quartet of the launch:
    // decoded once per block (five integer divisions per element -- inside the primitive loop they cost more than the sums)
    const int nA2 = nitem * hsz, nA3 = nitem * Gsz;
    int *tabA2 = reinterpret_cast<int *>(lds), *tabA3 = tabA2 + nA2;
    {
        const int jm = (lb + 1) * M1, kl = (lc + 1) * (ld + 1);
        for (int e = tid; e < nA2; e += 256) {
            const int item = e / hsz, rem = e - item * hsz;
            const int isel = rem / jm, j = (rem / M1) % (lb + 1), m = rem % M1;
            tabA2[e] = (item * gsz + isel * M1 + m) | ((item / nr) << 12) | (j << 14);
        }
        for (int e = tid; e < nA3; e += 256) {
            const int item = e / Gsz, rem = e - item * Gsz;
            const int ij = rem / kl, r2 = rem - ij * kl, k = r2 / (ld + 1), l = r2 - k * (ld + 1);
            tabA3[e] = (item * hsz + ij * M1 + k) | ((item / nr) << 13) | (l << 15);
        }
        __syncthreads();
    }

    long long task = ((long long)blockIdx.x * og.nparts + og.part) * QPB + q;
    bool active = task < ntask;
    if (!active) task = ntask - 1;
    int ib, ik;
    if (same == 2) {  // diagonal quartets only (Schwarz bounds)
        ib = ik = (int)task;
    } else if (og.toff != nullptr) {  // screened map of the direct-SCF context (eri_core.hpp)
        const int e = screen_find(og.toff, nb * SCREEN_NBIN, task);
        ib = e / SCREEN_NBIN;
        ik = og.pbin[e % SCREEN_NBIN] + (int)(task - og.toff[e]);
    } else
    if (same) {
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
    if constexpr (MODE == ERI_OUT_JK)
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
    const int nq = active ? nbp * nkp : 0;
    int maxq = nq;
    if (TPQ > 64) {  // the group is the block: block-uniform trip count (barriers)
        if (tid == 0) s_maxq = 0;
        __syncthreads();
        atomicMax(&s_maxq, nq);
        __syncthreads();
        maxq = s_maxq;
    }

    // transfer coefficients of the horizontal recurrences: ctab[d][j][a] = C(j, a) AB_d^(j-a), ctab[75 + ...] the same with CD
    for (int e = s; e < 150; e += TPQ) {
        const int r = e % 75, d = r / 25, j = (r / 5) % 5, a = r % 5;
        const double x = e < 75 ? (d == 0 ? AB[0] : (d == 1 ? AB[1] : AB[2])) : (d == 0 ? CD[0] : (d == 1 ? CD[1] : CD[2]));
        double v = 0.0;
        if (a <= j) {
            v = 1.0;
            for (int t = 0; t < a; t++) v = v * (j - t) / (t + 1);  // binomial
            for (int t = 0; t < j - a; t++) v *= x;
        }
        ctab[e] = v;
    }
    if constexpr (MODE == ERI_OUT_JK) {
        const int njk = (2 * la + 1) * (2 * lb + 1) + (2 * lc + 1) * (2 * ld + 1) + (2 * la + 1) * (2 * lc + 1) +
                        (2 * la + 1) * (2 * ld + 1) + (2 * lb + 1) * (2 * lc + 1) + (2 * lb + 1) * (2 * ld + 1);
        for (int e = s; e < njk; e += TPQ) reg[hc.off_jk + e] = 0.0;
    }

    const int sa = 2 * la + 1, sb = 2 * lb + 1, sc = 2 * lc + 1, sd = 2 * ld + 1;
    const int nsph = sa * sb * sc * sd;
    double *sphl = reg + hc.off_s;
    if constexpr (MODE != ERI_OUT_GRAD)
        for (int e = s; e < nsph; e += TPQ) sphl[e] = 0.0;
    double gsum[3] = {0.0, 0.0, 0.0};  // GRAD mode

    const int nslice = hc.full ? 1 : nca, ncas = hc.full ? nca : 1;
    const int rcd = ncc * ncd, rbcd = ncb * rcd, nouts = ncas * rbcd;

    // 2D-table indices of the outputs a lane owns.  They do not depend on the slice: a slice's tables hold i = a_d only
    // (index 0), the one-slice form holds every i
    int oidx[HL_NPT];
#pragma unroll
    for (int m = 0; m < HL_NPT; m++) {
        int n = s + TPQ * m;
        if (n >= nouts) n = nouts - 1;
        const int cd = n % ncd, cc = (n / ncd) % ncc, cb = (n / rcd) % ncb, cal = n / rbcd;
        int ax = 0, ay = 0, az = 0, bx, by, bz, cx, cy, cz, dx, dy, dz;
        if (hc.full) cart_pow(la, cal, ax, ay, az);
        cart_pow(lb, cb, bx, by, bz);
        cart_pow(lc, cc, cx, cy, cz);
        cart_pow(ld, cd, dx, dy, dz);
        const int ixx = ((ax * (lb + 1) + bx) * (lc + 1) + cx) * (ld + 1) + dx;
        const int iyy = ((ay * (lb + 1) + by) * (lc + 1) + cy) * (ld + 1) + dy;
        const int izz = ((az * (lb + 1) + bz) * (lc + 1) + cz) * (ld + 1) + dz;
        oidx[m] = ixx | (iyy << 10) | (izz << 20);
    }

    for (int slice = 0; slice < nslice; slice++) {
        int i0[3] = {0, 0, 0};  // first bra-a index of the 2D tables, per direction
        if (!hc.full) cart_pow(la, slice, i0[0], i0[1], i0[2]);
        double acc[HL_NPT];
#pragma unroll
        for (int m = 0; m < HL_NPT; m++) acc[m] = 0.0;
        eri_group_sync<TPQ>();  // the previous slice's transform buffers are free (they alias the tables below)

        for (int iq = 0; iq < maxq; iq++) {
            const bool on = iq < nq;
            // ---------------- A1: vertical recurrence of every (direction, root) item, rolling columns in registers ----------------
            if (on) {
                const int ipb = iq / nkp, ipk = iq - ipb * nkp;
                const double *pb = prs.pp + (size_t)(pb0 + ipb) * prs.stride, *pk = prk.pp + (size_t)(pk0 + ipk) * prk.stride;
                const double p = pb[0], qq = pk[0];
                const double P[3] = {pb[1], pb[2], pb[3]}, Q[3] = {pk[1], pk[2], pk[3]};
                const double pq = p + qq, rho = p * qq / pq;
                const double PQ[3] = {P[0] - Q[0], P[1] - Q[1], P[2] - Q[2]};
                const double X = rho * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]);
                const double ipq = 1.0 / pq, ip = 1.0 / p, iqq = 1.0 / qq;
                const double pref = pb[4] * pk[4] * 34.986836655249725 * sqrt(ipq);  // the pair table holds K / p
                for (int item = s; item < nitem; item += TPQ) {
                    const int d = item / nr, r = item - d * nr;
                    double u, w;
                    rys_root1_rt(nr, X, r, u, w);
                    const double Pd = d == 0 ? P[0] : (d == 1 ? P[1] : P[2]), Qd = d == 0 ? Q[0] : (d == 1 ? Q[1] : Q[2]);
                    const double Ad = d == 0 ? A[0] : (d == 1 ? A[1] : A[2]), Cd = d == 0 ? Cc[0] : (d == 1 ? Cc[1] : Cc[2]);
                    const double PQd = Pd - Qd;
                    const double uq = u * qq * ipq, up = u * p * ipq;
                    const double b00 = 0.5 * u * ipq, b10 = 0.5 * (1.0 - uq) * ip, b01 = 0.5 * (1.0 - up) * iqq;
                    const double c00 = (Pd - Ad) - uq * PQd, c0p = (Qd - Cd) + up * PQd;
                    const int nd = (hc.full ? la : (d == 0 ? i0[0] : (d == 1 ? i0[1] : i0[2]))) + lb;  // highest n needed
                    double *gi = gl + (size_t)item * gsz;
                    double c0[10], c1[10];
                    c1[0] = (d == 2) ? w * pref : 1.0;
                    c0[0] = 0.0;
#pragma unroll
                    for (int n = 0; n < 9; n++) {
                        c0[n + 1] = 0.0;
                        c1[n + 1] = n < nd ? c00 * c1[n] + (n ? n * b10 * c1[n - 1] : 0.0) : 0.0;
                    }
#pragma unroll
                    for (int n = 0; n < 10; n++)
                        if (n <= nd) gi[n * M1] = c1[n];
                    for (int m = 0; m < mmax; m++) {
                        double cn[10];
#pragma unroll
                        for (int n = 0; n < 10; n++)
                            cn[n] = c0p * c1[n] + m * b01 * c0[n] + (n ? n * b00 * c1[n - 1] : 0.0);
#pragma unroll
                        for (int n = 0; n < 10; n++) {
                            c0[n] = c1[n];
                            c1[n] = cn[n];
                            if (n <= nd) gi[n * M1 + m + 1] = cn[n];
                        }
                    }
                }
            }
            eri_group_sync<TPQ>();
            // ---------------- A2: bra transfer h[item][i][j][m] = sum_a C(j, a) AB^(j-a) g[item][i + a][m] ----------------
            if (on) {
                for (int e = s; e < nA2; e += TPQ) {
                    const int ds_ = tabA2[e], d = (ds_ >> 12) & 3, j = ds_ >> 14;
                    const double *gi = gl + (ds_ & 4095) + (d == 0 ? i0[0] : (d == 1 ? i0[1] : i0[2])) * M1;
                    const double *cf = ctab + d * 25 + j * 5;
                    double v = 0.0;
                    for (int a = 0; a <= j; a++) v += cf[a] * gi[a * M1];
                    hl[e] = v;
                }
            }
            eri_group_sync<TPQ>();
            // ---------------- A3: ket transfer G[item][i][j][k][l] = sum_b C(l, b) CD^(l-b) h[item][i][j][k + b] ----------------
            if (on) {
                for (int e = s; e < nA3; e += TPQ) {
                    const int ds_ = tabA3[e], d = (ds_ >> 13) & 3, l = ds_ >> 15;
                    const double *hi = hl + (ds_ & 8191);
                    const double *cf = ctab + 75 + d * 25 + l * 5;
                    double v = 0.0;
                    for (int b = 0; b <= l; b++) v += cf[b] * hi[b];
                    Gl[e] = v;
                }
            }
            eri_group_sync<TPQ>();
            // ---------------- B: Cartesian outputs of the slice ----------------
            if (on) {
#pragma unroll
                for (int m = 0; m < HL_NPT; m++) {
                    if (s + TPQ * m < nouts) {
                        const double *gx = Gl + (oidx[m] & 1023), *gy = Gl + (size_t)nr * Gsz + ((oidx[m] >> 10) & 1023),
                                     *gz = Gl + (size_t)2 * nr * Gsz + (oidx[m] >> 20);
                        double v = 0.0;
                        for (int r = 0; r < nr; r++) v += gx[r * Gsz] * gy[r * Gsz] * gz[r * Gsz];
                        acc[m] += v;
                    }
                }
            }
            eri_group_sync<TPQ>();
        }

        if constexpr (MODE == ERI_OUT_GRAD) {
            // ---------------- gradient contraction of the slice (see eri_core.hpp) ----------------
            const int a = ish % og.norig;
            const int lorig = la - og.dirn;
            const int ca0 = og.cao[a], cb0 = og.cao[jsh], cc0 = og.cao[ksh], cd0 = og.cao[lsh];
            const bool same_cd = ksh == lsh;
            const double jfac = (same_cd ? 2.0 : 4.0) * og.jscale;
            const double *D = og.dcart;
            const size_t nc = og.ncart;
#pragma unroll
            for (int m = 0; m < HL_NPT; m++) {
                const int n = s + TPQ * m;
                if (n < nouts && active) {
                    const int cd = n % ncd, cc = (n / ncd) % ncc, cb = (n / rcd) % ncb, cal = n / rbcd;
                    int u[3];
                    cart_pow(la, hc.full ? cal : slice, u[0], u[1], u[2]);
                    const size_t ibb = cb0 + cb, ic = cc0 + cc, id = cd0 + cd;
                    double dcd = 0.0, dbd = 0.0, dbc = 0.0;
                    if (og.gmode == 0) { dcd = D[ic * nc + id]; dbd = D[ibb * nc + id]; dbc = D[ibb * nc + ic]; }
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
                        const size_t ia = ca0 + cart_index(lorig, o[0], o[2]);
                        double f;
                        if (og.gmode == 0)
                            f = jfac * D[ia * nc + ibb] * dcd -
                                og.kscale * (D[ia * nc + ic] * dbd + (same_cd ? 0.0 : D[ia * nc + id] * dbc));
                        else if (og.gmode == 1)
                            f = 2.0 * D[ia * nc + ibb] * og.ccart[ic];
                        else
                            f = -og.ccart[ia] * og.ccart[ic];
                        gsum[dir] += coef * acc[m] * f;
                    }
                }
            }
        } else {
            // ---------------- C: solid harmonics on d, c, b in LDS; the a-transform accumulates in registers ----------------
            double *buf0 = reg, *buf1 = reg + nouts;
#pragma unroll
            for (int m = 0; m < HL_NPT; m++) {
                const int n = s + TPQ * m;
                if (n < nouts) buf0[n] = acc[m];
            }
            eri_group_sync<TPQ>();
            {
                const double *C = C2S + C2S_OFF[ld];
                const int x3 = ncas * ncb * ncc;
                for (int e = s; e < x3 * sd; e += TPQ) {
                    const int x = e / sd, md = e - x * sd;
                    double v = 0.0;
                    for (int c = 0; c < ncd; c++) v += C[md * ncd + c] * buf0[x * ncd + c];
                    buf1[e] = v;
                }
            }
            eri_group_sync<TPQ>();
            {
                const double *C = C2S + C2S_OFF[lc];
                const int y2 = ncas * ncb;
                for (int e = s; e < y2 * sc * sd; e += TPQ) {
                    const int md = e % sd, mc = (e / sd) % sc, y = e / (sd * sc);
                    double v = 0.0;
                    for (int c = 0; c < ncc; c++) v += C[mc * ncc + c] * buf1[(y * ncc + c) * sd + md];
                    buf0[e] = v;
                }
            }
            eri_group_sync<TPQ>();
            const int W = sc * sd;
            {
                const double *C = C2S + C2S_OFF[lb];
                for (int e = s; e < ncas * sb * W; e += TPQ) {
                    const int w = e % W, mb = (e / W) % sb, cal = e / (W * sb);
                    double v = 0.0;
                    for (int c = 0; c < ncb; c++) v += C[mb * ncb + c] * buf0[(cal * ncb + c) * W + w];
                    buf1[e] = v;
                }
            }
            eri_group_sync<TPQ>();
            {
                const double *C = C2S + C2S_OFF[la];
                const int R = sb * W;
                for (int e = s; e < nsph; e += TPQ) {  // lane s owns the elements e = s (mod TPQ) in every slice
                    const int ma = e / R, rest = e - ma * R;
                    double v = 0.0;
                    if (hc.full)
                        for (int c = 0; c < nca; c++) v += C[ma * nca + c] * buf1[c * R + rest];
                    else
                        v = C[ma * nca + slice] * buf1[rest];
                    sphl[e] += v;
                }
            }
        }
    }

    if constexpr (MODE == ERI_OUT_GRAD) {
        constexpr int WRED = TPQ < 64 ? TPQ : 64;
#pragma unroll
        for (int dir = 0; dir < 3; dir++)
#pragma unroll
            for (int o = WRED / 2; o > 0; o >>= 1) gsum[dir] += __shfl_xor(gsum[dir], o);
        const int a = ish % og.norig;
        double *gp = og.gpart + ((size_t)(blockIdx.x % og.nslot) * og.natm + og.sh_atom[a]) * 3;
        double *gk = og.gpart + ((size_t)(blockIdx.x % og.nslot) * og.natm + og.sh_atom[ksh]) * 3;  // gmode 1 only
        if (TPQ <= 64) {
            if (s == 0 && active)
                for (int dir = 0; dir < 3; dir++) {
                    atomicAdd(&gp[dir], gsum[dir]);
                    if (og.gmode == 1) atomicAdd(&gk[dir], -gsum[dir]);
                }
        } else {
            __syncthreads();
            if ((tid & 63) == 0)
                for (int dir = 0; dir < 3; dir++) lds[(tid >> 6) * 3 + dir] = gsum[dir];
            __syncthreads();
            if (tid == 0 && active)
                for (int dir = 0; dir < 3; dir++) {
                    const double v = lds[dir] + lds[3 + dir] + lds[6 + dir] + lds[9 + dir];
                    atomicAdd(&gp[dir], v);
                    if (og.gmode == 1) atomicAdd(&gk[dir], -v);
                }
        }
        return;
    } else {
        // ---------------- scatter of the spherical block ----------------
        const int ai = sh.ao_off[ish], aj = sh.ao_off[jsh], ak = sh.ao_off[ksh], al = sh.ao_off[lsh];
        const int oj2 = sa * sb, ok1 = oj2 + sc * sd, ok2 = ok1 + sa * sc, ok3 = ok2 + sa * sd, ok4 = ok3 + sb * sc,
                  njk = ok4 + sb * sd;
        if (active) {
            for (int e = s; e < nsph; e += TPQ) {
                const double v = sphl[e];
                const int md = e % sd, mc = (e / sd) % sc, mb = (e / (sd * sc)) % sb, ma = e / (sd * sc * sb);
                const int i = ai + ma, j = aj + mb, k = ak + mc, l = al + md;
                if constexpr (MODE == ERI_OUT_SCHWARZ) {
                    if (ma == mc && mb == md)
                        atomicMax(reinterpret_cast<unsigned long long *>(tiles) + (size_t)ib * 4, (unsigned long long)__double_as_longlong(fabs(v)));  // (four slots per pair: eri_core.hpp)
                } else
                if constexpr (MODE == ERI_OUT_JK) {
                    const double *D = og.dmat;
                    const size_t n = og.nao;
                    double *jk = reg + hc.off_jk;
                    atomicAdd(&jk[ma * sb + mb], v * D[(size_t)k * n + l]);  // ds_add_f64
                    atomicAdd(&jk[oj2 + mc * sd + md], v * D[(size_t)i * n + j]);
                    if (og.kacc) {
                        atomicAdd(&jk[ok1 + ma * sc + mc], v * D[(size_t)j * n + l]);
                        atomicAdd(&jk[ok2 + ma * sd + md], v * D[(size_t)j * n + k]);
                        atomicAdd(&jk[ok3 + mb * sc + mc], v * D[(size_t)i * n + l]);
                        atomicAdd(&jk[ok4 + mb * sd + md], v * D[(size_t)i * n + k]);
                    }
                } else if constexpr (MODE == ERI_OUT_TILES) {
                    // (one of the two equal-by-symmetry values of a quartet with a repeated shell: eri_core.hpp, bitwise reproducible store)
                    const bool twin = (ai == aj && mb > ma) || (ak == al && md > mc) || (ai == ak && aj == al && mc * sd + md > ma * sb + mb);
                    if (!twin) tile_put_all(tiles, i, j, k, l, v, og.st_lo, og.st_hi, og.st_nao);
                } else if constexpr (MODE == ERI_OUT_3C) {
                    const size_t io = i - og.ao0, jo = j - og.ao0, kx = k - og.aux0;
                    tiles[(io * og.nao + jo) * og.naux + kx] = v;
                    tiles[(jo * og.nao + io) * og.naux + kx] = v;
                } else {
                    tiles[(size_t)(i - og.aux0) * og.naux + (k - og.aux0)] = v;
                }
            }
        }
        if constexpr (MODE == ERI_OUT_JK) {
            eri_group_sync<TPQ>();
            if (active) {
                const double deg = (ish == jsh ? 0.5 : 1.0) * (ksh == lsh ? 0.5 : 1.0) * ((ish == ksh && jsh == lsh) ? 0.5 : 1.0);
                const size_t n = og.nao;
                const double *jk = reg + hc.off_jk;
                const int nend = og.kacc ? njk : ok1;
                for (int e = s; e < nend; e += TPQ) {
                    double *dst;
                    double f = deg;
                    if (e < oj2) { dst = og.jacc + (size_t)(ai + e / sb) * n + aj + e % sb; f *= 4.0; }
                    else if (e < ok1) { const int x = e - oj2; dst = og.jacc + (size_t)(ak + x / sd) * n + al + x % sd; f *= 4.0; }
                    else if (e < ok2) { const int x = e - ok1; dst = og.kacc + (size_t)(ai + x / sc) * n + ak + x % sc; }
                    else if (e < ok3) { const int x = e - ok2; dst = og.kacc + (size_t)(ai + x / sd) * n + al + x % sd; }
                    else if (e < ok4) { const int x = e - ok3; dst = og.kacc + (size_t)(aj + x / sc) * n + ak + x % sc; }
                    else { const int x = e - ok4; dst = og.kacc + (size_t)(aj + x / sd) * n + al + x % sd; }
                    atomicAdd(dst, f * jk[e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host: LDS layout and lane-group size of one class
// ---------------------------------------------------------------------------------------------
struct HlPlan {
    HlClass hc;
    int tpq;
    size_t lds_bytes;
};

inline bool hl_plan(int mode, int la, int lb, int lc, int ld, HlPlan &p) {
    const int nr = (la + lb + lc + ld) / 2 + 1;
    if (la > 5 || lb > 4 || lc > 4 || ld > 4 || la + lb > 9 || nr > RYS_NMAX) return false;
    const int nmax = la + lb, mmax = lc + ld, M1 = mmax + 1, nitem = 3 * nr;
    const int nca = ncart(la), ncb = ncart(lb), ncc = ncart(lc), ncd = ncart(ld);
    const int sa = 2 * la + 1, sb = 2 * lb + 1, sc = 2 * lc + 1, sd = 2 * ld + 1;
    const bool sphm = mode != ERI_OUT_GRAD;
    const int nsph = sphm ? sa * sb * sc * sd : 0;
    const int njk = mode == ERI_OUT_JK ? sa * sb + sc * sd + sa * sc + sa * sd + sb * sc + sb * sd : 0;
    for (int full = 1; full >= 0; full--) {
        const int ni = full ? la + 1 : 1, ncas = full ? nca : 1;
        const int gsz = (nmax + 1) * M1, hsz = ni * (lb + 1) * M1, Gsz = ni * (lb + 1) * (lc + 1) * (ld + 1);
        if (Gsz > 1023) continue;  // packed output indices (10 bits per direction)
        const int nouts = ncas * ncb * ncc * ncd;
        int core = nitem * (gsz + hsz + Gsz);
        if (sphm) {
            const int b1 = std::max(ncas * ncb * ncc * sd, ncas * sb * sc * sd);
            core = std::max(core, nouts + b1);
        }
        HlClass hc{la, lb, lc, ld, nr, full, 0, nitem * gsz, nitem * (gsz + hsz), core, core + 150, core + 150 + njk};
        hc.region = (core + 150 + njk + nsph) | 1;
        if (nitem * gsz > 4095 || nitem * hsz > 8191) continue;  // packed descriptors
        hc.tab_doubles = ((nitem * (hsz + Gsz) + 1) / 2 + 1) & ~1;
        const int tpqs[3] = {16, 64, 256};
        for (int t = 0; t < 3; t++) {
            const int tpq = tpqs[t];
            if (nouts > HL_NPT * tpq) continue;
            const size_t bytes = sizeof(double) * ((size_t)hc.region * (256 / tpq) + hc.tab_doubles);
            if (bytes > (size_t)HL_LDS_LIMIT) continue;
            if (mode == ERI_OUT_GRAD && tpq == 256 && hc.region < 16) hc.region = 17;
            p.hc = hc;
            p.tpq = tpq;
            p.lds_bytes = sizeof(double) * ((size_t)hc.region * (256 / tpq) + hc.tab_doubles);
            return true;
        }
    }
    return false;
}

template <int MODE>
static int launch_hl(double *tiles, const DevShells &ds, const DevPairs &db, const DevPairs &dk, int b0, int nb, int k0, int nk,
                     int same, const EriOut &og, int la, int lb, int lc, int ld, hipStream_t st, long long ntask_screened = -1) {
    if (nb == 0 || nk == 0) return 0;
    HlPlan p;
    if (!hl_plan(MODE == ERI_OUT_SCHWARZ ? ERI_OUT_TILES : MODE, la, lb, lc, ld, p)) {
        set_error("shell quartet class (" + std::to_string(la) + std::to_string(lb) + "|" + std::to_string(lc) + std::to_string(ld) +
                  ") is beyond the integral kernels (angular momentum above g)");
        return DQC_EINVAL;
    }
    // task map: dense rectangle / triangle; `same` = 2: the nb diagonal quartets; ntask_screened >= 0: og.toff's total
    const long long ntask = ntask_screened >= 0 ? ntask_screened : (same == 2 ? (long long)nb : (same ? (long long)nb * (nb + 1) / 2 : (long long)nb * nk));
    if (ntask == 0) return 0;
    const int qpb = 256 / p.tpq;
    const long long nblk = ((ntask + qpb - 1) / qpb + og.nparts - 1) / og.nparts;
#define DQC_HL(T)                                                                                                            \
    {                                                                                                                        \
        auto kern = eri_hl_kernel<T, MODE>;                                                                                   \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS_LIMIT);             \
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), p.lds_bytes, st, tiles, ds, db, dk, b0, nb, k0, nk, same, ntask, \
                           og, p.hc);                                                                                        \
    }
    if (p.tpq == 16) DQC_HL(16)
    else if (p.tpq == 64) DQC_HL(64)
    else DQC_HL(256)
#undef DQC_HL
    DQC_CHECK_LAUNCH();
    return 0;
}

// dqc_set_generic_eri(1) / DQC_ERI_GENERIC=1: every class through the runtime kernel (the test suite's cross-check of the two
// implementations)
inline bool hl_forced() { return generic_eri_forced(); }

}  // namespace dqc
