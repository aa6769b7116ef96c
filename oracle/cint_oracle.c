/*
 * oracle/cint_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement, in plain C, of the native integral arithmetic that the
 * reference (diffqc/dqc) obtains from the un-vendored dependency
 * `dqclibs>=0.1.0` (libcint + PySCF's libcgto) -- /root/reference/setup.py:59.
 * The library sources are absent from /root/reference, so this file restates
 * the *published* algorithms and anchors them on the reference's call sites:
 *
 *   int1e_ovlp_sph / int1e_kin_sph / int1e_nuc_sph via GTOint2c
 *        dqc/hamilton/intor/molintor.py:96-112, 624-644
 *   int2e_sph via GTOnr2e_fill_drv + GTOnr2e_fill_s4 (packed i>=j, k>=l)
 *        dqc/hamilton/intor/molintor.py:114-119, 667-688; symmetry.py:40-69
 *   fills4 (CSYMM)                     dqc/hamilton/intor/symmetry.py:55-64
 *   GTOval_sph / GTOval_ip_sph         dqc/hamilton/intor/gtoeval.py:196-239
 *
 * Inputs are libcint-style atm/bas/env tables exactly as LibcintWrapper builds
 * them (dqc/hamilton/intor/lcintwrap.py:37-86): bas = [iatom, l, nprim, nctr=1,
 * kappa, ptr_exp, ptr_coef, 0]; env coefficients already carry the radial
 * normalisation of CGTOBasis.wfnormalize_ (dqc/utils/datastruct.py:34-61).
 *
 * Algorithm: McMurchie-Davidson Hermite expansion (J. Comput. Phys. 26 (1978)
 * 218; Helgaker/Jorgensen/Olsen ch. 9) with the Boys function from its
 * convergent series / asymptotic form.  The HIP product path uses Rys
 * quadrature instead, so oracle and product share no integral code.
 *
 * Conventions (libcint): real solid harmonics normalised on the unit sphere,
 * AO order p = (x,y,z); d,f: m = -l..l.  The transformation tables below are
 * written out by hand from the closed forms (independent of the generated
 * table the HIP side uses).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LMAX 4               /* highest shell angular momentum supported (g for aux) */
#define LMAX1 (LMAX + 1)
#define NCART(l) (((l) + 1) * ((l) + 2) / 2)
#define MAXCART NCART(LMAX)
#define BAS_SLOTS 8
#define ATM_SLOTS 6
#define ATOM_OF 0
#define ANG_OF 1
#define NPRIM_OF 2
#define PTR_EXP 5
#define PTR_COEFF 6
#define PTR_COORD 1

/* ------------------------------------------------------------------ */
/* Boys function F_m(T), m = 0..mmax                                  */
/* ------------------------------------------------------------------ */
static void boys(int mmax, double T, double *F)
{
    if (T < 35.0) {
        /* series for the highest order, then downward recursion */
        double term = 1.0 / (2 * mmax + 1);
        double sum = term;
        for (int k = 1; k < 400; k++) {
            term *= 2.0 * T / (2 * mmax + 2 * k + 1);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        double emt = exp(-T);
        F[mmax] = emt * sum;
        for (int m = mmax; m > 0; m--)
            F[m - 1] = (2.0 * T * F[m] + emt) / (2 * m - 1);
    } else {
        /* asymptotic F_0 = sqrt(pi/T)/2 (error ~ exp(-T)), upward recursion */
        double emt = exp(-T);
        F[0] = 0.5 * sqrt(M_PI / T);
        for (int m = 0; m < mmax; m++)
            F[m + 1] = ((2 * m + 1) * F[m] - emt) / (2.0 * T);
    }
}

/* ------------------------------------------------------------------ */
/* Hermite expansion coefficients E^{ij}_t (one Cartesian direction)  */
/* E[i][j][t], i<=la, j<=lb, t<=i+j ; includes exp(-mu X_AB^2)        */
/* ------------------------------------------------------------------ */
#define EDIM (2 * LMAX + 3)  /* +2 so that kinetic can use j+2 */
typedef double Ecoef[LMAX1 + 2][LMAX1 + 2][EDIM];

static void hermite_E(int la, int lb, double a, double b, double XAB, Ecoef E)
{
    double p = a + b, mu = a * b / p;
    double XPA = -b / p * XAB;  /* P - A */
    double XPB = a / p * XAB;   /* P - B */
    double hp = 0.5 / p;
    memset(E, 0, sizeof(Ecoef));
    E[0][0][0] = exp(-mu * XAB * XAB);
    for (int i = 0; i <= la; i++) {
        if (i > 0) {
            for (int t = 0; t <= i; t++) {
                double v = XPA * E[i - 1][0][t];
                if (t > 0) v += hp * E[i - 1][0][t - 1];
                if (t + 1 <= i - 1) v += (t + 1) * E[i - 1][0][t + 1];
                E[i][0][t] = v;
            }
        }
        for (int j = 1; j <= lb; j++) {
            for (int t = 0; t <= i + j; t++) {
                double v = XPB * E[i][j - 1][t];
                if (t > 0) v += hp * E[i][j - 1][t - 1];
                if (t + 1 <= i + j - 1) v += (t + 1) * E[i][j - 1][t + 1];
                E[i][j][t] = v;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* Hermite Coulomb integrals R_{tuv} = R^0_{tuv}, t+u+v <= L          */
/* ------------------------------------------------------------------ */
#define RL (4 * LMAX + 1)
typedef double Rtens[RL][RL][RL];

static void hermite_R(int L, double alpha, double X, double Y, double Z, Rtens R)
{
    /* Rn[n][t][u][v] built downward in n; keep two layers via full 4-index scratch */
    static __thread double W[RL][RL][RL][RL];
    double F[RL + 1];
    boys(L, alpha * (X * X + Y * Y + Z * Z), F);
    double m2a = 1.0;
    for (int n = 0; n <= L; n++) { W[n][0][0][0] = m2a * F[n]; m2a *= -2.0 * alpha; }
    /* build by increasing total order N = t+u+v ; R^{n}_{tuv} needs R^{n+1} of order N-1 */
    for (int N = 1; N <= L; N++) {
        for (int n = 0; n <= L - N; n++) {
            for (int t = 0; t <= N; t++)
                for (int u = 0; u <= N - t; u++) {
                    int v = N - t - u;
                    double val;
                    if (t > 0) {
                        val = X * W[n + 1][t - 1][u][v];
                        if (t > 1) val += (t - 1) * W[n + 1][t - 2][u][v];
                    } else if (u > 0) {
                        val = Y * W[n + 1][t][u - 1][v];
                        if (u > 1) val += (u - 1) * W[n + 1][t][u - 2][v];
                    } else {
                        val = Z * W[n + 1][t][u][v - 1];
                        if (v > 1) val += (v - 1) * W[n + 1][t][u][v - 2];
                    }
                    W[n][t][u][v] = val;
                }
        }
    }
    for (int t = 0; t <= L; t++)
        for (int u = 0; u <= L - t; u++)
            for (int v = 0; v <= L - t - u; v++)
                R[t][u][v] = W[0][t][u][v];
}

/* ------------------------------------------------------------------ */
/* Cartesian component tables (libcint order: lx descending, then ly) */
/* ------------------------------------------------------------------ */
static void cart_powers(int l, int (*pw)[3])
{
    int n = 0;
    for (int lx = l; lx >= 0; lx--)
        for (int ly = l - lx; ly >= 0; ly--) {
            pw[n][0] = lx; pw[n][1] = ly; pw[n][2] = l - lx - ly; n++;
        }
}

/* real solid harmonics in Cartesian monomials, rows = spherical component
 * (libcint order), columns = Cartesian component (order above).          */
static const double C2S_S[1][1] = {{0.282094791773878143}};
static const double C2S_P[3][3] = {
    {0.488602511902919921, 0, 0},
    {0, 0.488602511902919921, 0},
    {0, 0, 0.488602511902919921}};
/* d cart: xx xy xz yy yz zz */
static const double C2S_D[5][6] = {
    {0, 1.092548430592079070, 0, 0, 0, 0},                                  /* xy   */
    {0, 0, 0, 0, 1.092548430592079070, 0},                                  /* yz   */
    {-0.315391565252520002, 0, 0, -0.315391565252520002, 0, 0.630783130505040012}, /* z2 */
    {0, 0, 1.092548430592079070, 0, 0, 0},                                  /* xz   */
    {0.546274215296039535, 0, 0, -0.546274215296039535, 0, 0}};             /* x2-y2 */
/* f cart: xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz */
static const double C2S_F[7][10] = {
    {0, 1.770130769779930531, 0, 0, 0, 0, -0.590043589926643510, 0, 0, 0},  /* y(3x2-y2) */
    {0, 0, 0, 0, 2.890611442640554055, 0, 0, 0, 0, 0},                      /* xyz */
    {0, -0.457045799464465739, 0, 0, 0, 0, -0.457045799464465739, 0, 1.828183197857862944, 0}, /* y(4z2-x2-y2) */
    {0, 0, -1.119528997770346170, 0, 0, 0, 0, -1.119528997770346170, 0, 0.746352665180230782}, /* z(2z2-3x2-3y2) */
    {-0.457045799464465739, 0, 0, -0.457045799464465739, 0, 1.828183197857862944, 0, 0, 0, 0}, /* x(4z2-x2-y2) */
    {0, 0, 1.445305721320277020, 0, 0, 0, 0, -1.445305721320277020, 0, 0},  /* z(x2-y2) */
    {0.590043589926643510, 0, 0, -1.770130769779930531, 0, 0, 0, 0, 0, 0}}; /* x(x2-3y2) */
/* g cart (15): xxxx xxxy xxxz xxyy xxyz xxzz xyyy xyyz xyzz xzzz yyyy yyyz yyzz yzzz zzzz */
static const double C2S_G[9][15] = {
    {0, 2.503342941796704538, 0, 0, 0, 0, -2.503342941796704530, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 5.310392309339791593, 0, 0, 0, 0, 0, 0, -1.770130769779930530, 0, 0, 0},
    {0, -0.946174695757560014, 0, 0, 0, 0, -0.946174695757560014, 0, 5.677048174545360108, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, -2.007139630671867500, 0, 0, 0, 0, 0, 0, -2.007139630671867500, 0, 2.676186174229156671, 0},
    {0.317356640745612911, 0, 0, 0.634713281491225822, 0, -2.538853125964903290, 0, 0, 0, 0,
     0.317356640745612911, 0, -2.538853125964903290, 0, 0.846284375321634430},
    {0, 0, -2.007139630671867500, 0, 0, 0, 0, -2.007139630671867500, 0, 2.676186174229156671, 0, 0, 0, 0, 0},
    {-0.473087347878780002, 0, 0, 0, 0, 2.838524087272680054, 0, 0, 0, 0, 0.473087347878780009, 0,
     -2.838524087272680050, 0, 0},
    {0, 0, 1.770130769779930531, 0, 0, 0, 0, -5.310392309339791590, 0, 0, 0, 0, 0, 0, 0},
    {0.625835735449176134, 0, 0, -3.755014412695056800, 0, 0, 0, 0, 0, 0, 0.625835735449176134, 0, 0, 0, 0}};

static const double *c2s_table(int l)
{
    switch (l) {
    case 0: return &C2S_S[0][0];
    case 1: return &C2S_P[0][0];
    case 2: return &C2S_D[0][0];
    case 3: return &C2S_F[0][0];
    case 4: return &C2S_G[0][0];
    }
    return NULL;
}

/* exposed so tests can cross-check against the generated HIP-side table */
void orc_cart2sph(int l, double *out /* (2l+1, ncart) */)
{
    memcpy(out, c2s_table(l), sizeof(double) * (2 * l + 1) * NCART(l));
}

/* ------------------------------------------------------------------ */
/* shell helpers                                                      */
/* ------------------------------------------------------------------ */
typedef struct {
    int l, nprim;
    const double *exps, *coefs, *r;
} Shell;

static Shell get_shell(int ish, const int *atm, const int *bas, const double *env)
{
    Shell s;
    const int *b = bas + ish * BAS_SLOTS;
    s.l = b[ANG_OF];
    s.nprim = b[NPRIM_OF];
    s.exps = env + b[PTR_EXP];
    s.coefs = env + b[PTR_COEFF];
    s.r = env + atm[b[ATOM_OF] * ATM_SLOTS + PTR_COORD];
    return s;
}

static void make_ao_loc(int nbas, const int *bas, int *ao_loc)
{
    ao_loc[0] = 0;
    for (int i = 0; i < nbas; i++)
        ao_loc[i + 1] = ao_loc[i] + 2 * bas[i * BAS_SLOTS + ANG_OF] + 1;
}

int orc_nao(int nbas, const int *bas)
{
    int n = 0;
    for (int i = 0; i < nbas; i++) n += 2 * bas[i * BAS_SLOTS + ANG_OF] + 1;
    return n;
}

/* out_sph[ms_a][ms_b] = sum C_a[ms_a][ca] C_b[ms_b][cb] cart[ca][cb] */
static void c2s_2index(int la, int lb, const double *cart, double *sph)
{
    int na = NCART(la), nb = NCART(lb), sa = 2 * la + 1, sb = 2 * lb + 1;
    const double *Ca = c2s_table(la), *Cb = c2s_table(lb);
    double tmp[(2 * LMAX + 1) * MAXCART];
    for (int i = 0; i < sa; i++)
        for (int cb = 0; cb < nb; cb++) {
            double v = 0;
            for (int ca = 0; ca < na; ca++) v += Ca[i * na + ca] * cart[ca * nb + cb];
            tmp[i * nb + cb] = v;
        }
    for (int i = 0; i < sa; i++)
        for (int j = 0; j < sb; j++) {
            double v = 0;
            for (int cb = 0; cb < nb; cb++) v += Cb[j * nb + cb] * tmp[i * nb + cb];
            sph[i * sb + j] = v;
        }
}

/* ------------------------------------------------------------------ */
/* one-electron integrals: which = 0 ovlp, 1 kin, 2 nuc               */
/* out: (nao, nao) row-major, out[i*nao+j] = <i|O|j>                  */
/* ------------------------------------------------------------------ */
static void int1e_pair(int which, Shell A, Shell B, int natm, const int *atm, const double *env,
                       const double *zs, double *sph)
{
    int la = A.l, lb = B.l, na = NCART(la), nb = NCART(lb);
    int pa[MAXCART][3], pb[MAXCART][3];
    cart_powers(la, pa); cart_powers(lb, pb);
    double cart[MAXCART * MAXCART];
    memset(cart, 0, sizeof(cart));
    double AB[3] = {A.r[0] - B.r[0], A.r[1] - B.r[1], A.r[2] - B.r[2]};
    static __thread Rtens R;
    for (int ip = 0; ip < A.nprim; ip++)
        for (int jp = 0; jp < B.nprim; jp++) {
            double a = A.exps[ip], b = B.exps[jp], p = a + b;
            double cc = A.coefs[ip] * B.coefs[jp];
            Ecoef Ex, Ey, Ez;
            hermite_E(la, lb + 2, a, b, AB[0], Ex);
            hermite_E(la, lb + 2, a, b, AB[1], Ey);
            hermite_E(la, lb + 2, a, b, AB[2], Ez);
            if (which >= 3) {
                /* int1e_r / int1e_rr about the common origin 0 (libcint int1e_r_sph, int1e_rr_sph; reference call site
                 * intor.int1e("r0" * n), dqc/hamilton/hcgto.py:117-125).  1D: <i| x^e |j> = sum_t E_t^{ij} M_t^e with the
                 * Hermite moments about 0: M_0^0 = 1, M_0^1 = P, M_1^1 = 1, M_0^2 = P^2 + 1/(2p), M_1^2 = 2 P, M_2^2 = 2
                 * (times sqrt(pi/p)).  which = 3 + d (x, y, z) or 6 + 3 d1 + d2 (second moments). */
                int e[3] = {0, 0, 0};
                if (which < 6) e[which - 3] = 1;
                else { e[(which - 6) / 3] += 1; e[(which - 6) % 3] += 1; }
                double P[3] = {(a * A.r[0] + b * B.r[0]) / p, (a * A.r[1] + b * B.r[1]) / p,
                               (a * A.r[2] + b * B.r[2]) / p};
                double pref = cc * pow(M_PI / p, 1.5);
                for (int ca = 0; ca < na; ca++)
                    for (int cb = 0; cb < nb; cb++) {
                        double v = 1.0;
                        for (int d = 0; d < 3; d++) {
                            int i = pa[ca][d], j = pb[cb][d];
                            double (*E)[LMAX1 + 2][EDIM] = d == 0 ? Ex : (d == 1 ? Ey : Ez);
                            double e0 = E[i][j][0], e1 = (i + j >= 1) ? E[i][j][1] : 0.0, e2 = (i + j >= 2) ? E[i][j][2] : 0.0;
                            double m;
                            if (e[d] == 0) m = e0;
                            else if (e[d] == 1) m = P[d] * e0 + e1;
                            else m = (P[d] * P[d] + 0.5 / p) * e0 + 2.0 * P[d] * e1 + 2.0 * e2;
                            v *= m;
                        }
                        cart[ca * nb + cb] += pref * v;
                    }
            } else if (which == 0 || which == 1) {
                double pref = cc * pow(M_PI / p, 1.5);
                for (int ca = 0; ca < na; ca++)
                    for (int cb = 0; cb < nb; cb++) {
                        int i = pa[ca][0], k = pa[ca][1], m = pa[ca][2];
                        int j = pb[cb][0], l = pb[cb][1], n = pb[cb][2];
                        double v;
                        if (which == 0) {
                            v = Ex[i][j][0] * Ey[k][l][0] * Ez[m][n][0];
                        } else {
                            /* T_ij (1D) = -1/2 [ j(j-1) S_{i,j-2} - 2b(2j+1) S_ij + 4b^2 S_{i,j+2} ] */
                            double Sx = Ex[i][j][0], Sy = Ey[k][l][0], Sz = Ez[m][n][0];
                            double Tx = -2.0 * b * b * Ex[i][j + 2][0] + b * (2 * j + 1) * Sx;
                            if (j >= 2) Tx -= 0.5 * j * (j - 1) * Ex[i][j - 2][0];
                            double Ty = -2.0 * b * b * Ey[k][l + 2][0] + b * (2 * l + 1) * Sy;
                            if (l >= 2) Ty -= 0.5 * l * (l - 1) * Ey[k][l - 2][0];
                            double Tz = -2.0 * b * b * Ez[m][n + 2][0] + b * (2 * n + 1) * Sz;
                            if (n >= 2) Tz -= 0.5 * n * (n - 1) * Ez[m][n - 2][0];
                            v = Tx * Sy * Sz + Sx * Ty * Sz + Sx * Sy * Tz;
                        }
                        cart[ca * nb + cb] += pref * v;
                    }
            } else {
                double P[3] = {(a * A.r[0] + b * B.r[0]) / p, (a * A.r[1] + b * B.r[1]) / p,
                               (a * A.r[2] + b * B.r[2]) / p};
                double pref = cc * 2.0 * M_PI / p;
                int L = la + lb;
                for (int ic = 0; ic < natm; ic++) {
                    const double *C = env + atm[ic * ATM_SLOTS + PTR_COORD];
                    double Z = zs ? zs[ic] : (double)atm[ic * ATM_SLOTS + 0];
                    hermite_R(L, p, P[0] - C[0], P[1] - C[1], P[2] - C[2], R);
                    for (int ca = 0; ca < na; ca++)
                        for (int cb = 0; cb < nb; cb++) {
                            int i = pa[ca][0], k = pa[ca][1], m = pa[ca][2];
                            int j = pb[cb][0], l = pb[cb][1], n = pb[cb][2];
                            double v = 0;
                            for (int t = 0; t <= i + j; t++)
                                for (int u = 0; u <= k + l; u++)
                                    for (int w = 0; w <= m + n; w++)
                                        v += Ex[i][j][t] * Ey[k][l][u] * Ez[m][n][w] * R[t][u][w];
                            cart[ca * nb + cb] -= Z * pref * v;
                        }
                }
            }
        }
    c2s_2index(la, lb, cart, sph);
}

/* zs: optional per-atom (possibly fractional) charges, or NULL to use atm[:,0]
 * (fractional-Z path of molintor.nuclattr, dqc/hamilton/intor/molintor.py:105-112) */
void orc_int1e(int which, double *out, const int *atm, int natm, const int *bas, int nbas,
               const double *env, const double *zs)
{
    int *ao_loc = (int *)malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(nbas, bas, ao_loc);
    int nao = ao_loc[nbas];
#pragma omp parallel for schedule(dynamic)
    for (int ish = 0; ish < nbas; ish++) {
        double sph[(2 * LMAX + 1) * (2 * LMAX + 1)];
        for (int jsh = 0; jsh < nbas; jsh++) {
            Shell A = get_shell(ish, atm, bas, env), B = get_shell(jsh, atm, bas, env);
            int1e_pair(which, A, B, natm, atm, env, zs, sph);
            int sa = 2 * A.l + 1, sb = 2 * B.l + 1;
            for (int i = 0; i < sa; i++)
                for (int j = 0; j < sb; j++)
                    out[(size_t)(ao_loc[ish] + i) * nao + ao_loc[jsh] + j] = sph[i * sb + j];
        }
    }
    free(ao_loc);
}

/* ------------------------------------------------------------------ */
/* two-electron integrals (ij|kl), spherical, one shell quartet        */
/* ------------------------------------------------------------------ */
/* Hermite-basis expansion of a primitive pair: Hc[ca*nb+cb][tuv-linear] */
#define NHERM(L) (((L) + 1) * ((L) + 2) * ((L) + 3) / 6)
#define MAXHERM NHERM(2 * LMAX)

static int herm_index[2 * LMAX + 1][2 * LMAX + 1][2 * LMAX + 1];
static int herm_tuv[MAXHERM][3];
static int herm_ready = 0;
static void init_herm(void)
{
    if (herm_ready) return;
    int n = 0;
    /* ordered by total degree so that NHERM(L) prefix = all t+u+v<=L */
    for (int N = 0; N <= 2 * LMAX; N++)
        for (int t = N; t >= 0; t--)
            for (int u = N - t; u >= 0; u--) {
                int v = N - t - u;
                herm_index[t][u][v] = n;
                herm_tuv[n][0] = t; herm_tuv[n][1] = u; herm_tuv[n][2] = v;
                n++;
            }
    herm_ready = 1;
}

typedef struct {
    double p, P[3];
    double *H; /* [ncart_a*ncart_b][nherm] includes coefficient product and exp prefactor */
} PrimPair;

static void build_prim_pairs(Shell A, Shell B, PrimPair *pp, double *Hbuf)
{
    int la = A.l, lb = B.l, na = NCART(la), nb = NCART(lb), nh = NHERM(la + lb);
    int pa[MAXCART][3], pb[MAXCART][3];
    cart_powers(la, pa); cart_powers(lb, pb);
    double AB[3] = {A.r[0] - B.r[0], A.r[1] - B.r[1], A.r[2] - B.r[2]};
    int n = 0;
    for (int ip = 0; ip < A.nprim; ip++)
        for (int jp = 0; jp < B.nprim; jp++, n++) {
            double a = A.exps[ip], b = B.exps[jp], p = a + b;
            Ecoef Ex, Ey, Ez;
            hermite_E(la, lb, a, b, AB[0], Ex);
            hermite_E(la, lb, a, b, AB[1], Ey);
            hermite_E(la, lb, a, b, AB[2], Ez);
            pp[n].p = p;
            for (int d = 0; d < 3; d++) pp[n].P[d] = (a * A.r[d] + b * B.r[d]) / p;
            pp[n].H = Hbuf + (size_t)n * na * nb * nh;
            double cc = A.coefs[ip] * B.coefs[jp];
            memset(pp[n].H, 0, sizeof(double) * na * nb * nh);
            for (int ca = 0; ca < na; ca++)
                for (int cb = 0; cb < nb; cb++) {
                    int i = pa[ca][0], k = pa[ca][1], m = pa[ca][2];
                    int j = pb[cb][0], l = pb[cb][1], q = pb[cb][2];
                    double *h = pp[n].H + (size_t)(ca * nb + cb) * nh;
                    for (int t = 0; t <= i + j; t++)
                        for (int u = 0; u <= k + l; u++)
                            for (int v = 0; v <= m + q; v++)
                                h[herm_index[t][u][v]] = cc * Ex[i][j][t] * Ey[k][l][u] * Ez[m][q][v];
                }
        }
}

/* out: spherical block [sa][sb][sc][sd] row-major */
static void eri_quartet(Shell A, Shell B, Shell C, Shell D, const PrimPair *bra, int nbra,
                        const PrimPair *ket, int nket, double *sph, double *work)
{
    int la = A.l, lb = B.l, lc = C.l, ld = D.l;
    int nab = NCART(la) * NCART(lb), ncd = NCART(lc) * NCART(ld);
    int Lb = la + lb, Lk = lc + ld, L = Lb + Lk;
    int nhb = NHERM(Lb), nhk = NHERM(Lk);
    double *cart = work;                 /* nab*ncd */
    double *W = cart + (size_t)nab * ncd; /* nhb*ncd */
    memset(cart, 0, sizeof(double) * nab * ncd);
    static __thread Rtens R;
    for (int ib = 0; ib < nbra; ib++)
        for (int ik = 0; ik < nket; ik++) {
            double p = bra[ib].p, q = ket[ik].p;
            double alpha = p * q / (p + q);
            double X = bra[ib].P[0] - ket[ik].P[0], Y = bra[ib].P[1] - ket[ik].P[1],
                   Z = bra[ib].P[2] - ket[ik].P[2];
            hermite_R(L, alpha, X, Y, Z, R);
            double pref = 2.0 * pow(M_PI, 2.5) / (p * q * sqrt(p + q));
            /* W[tuv][cd] = sum_{tau nu phi} (-1)^{tau+nu+phi} Hket[cd][tau nu phi] R[t+tau][u+nu][v+phi] */
            for (int hb = 0; hb < nhb; hb++) {
                int t = herm_tuv[hb][0], u = herm_tuv[hb][1], v = herm_tuv[hb][2];
                for (int cd = 0; cd < ncd; cd++) {
                    const double *hk = ket[ik].H + (size_t)cd * nhk;
                    double s = 0;
                    for (int h2 = 0; h2 < nhk; h2++) {
                        if (hk[h2] == 0.0) continue;
                        int tt = herm_tuv[h2][0], uu = herm_tuv[h2][1], vv = herm_tuv[h2][2];
                        double r = R[t + tt][u + uu][v + vv];
                        s += ((tt + uu + vv) & 1) ? -hk[h2] * r : hk[h2] * r;
                    }
                    W[(size_t)hb * ncd + cd] = s * pref;
                }
            }
            for (int ab = 0; ab < nab; ab++) {
                const double *hbp = bra[ib].H + (size_t)ab * nhb;
                double *dst = cart + (size_t)ab * ncd;
                for (int hb = 0; hb < nhb; hb++) {
                    double e = hbp[hb];
                    if (e == 0.0) continue;
                    const double *w = W + (size_t)hb * ncd;
                    for (int cd = 0; cd < ncd; cd++) dst[cd] += e * w[cd];
                }
            }
        }
    /* cart [ca][cb][cc][cd] -> sph, one index at a time */
    int n[4] = {NCART(la), NCART(lb), NCART(lc), NCART(ld)};
    int s[4] = {2 * la + 1, 2 * lb + 1, 2 * lc + 1, 2 * ld + 1};
    int ls[4] = {la, lb, lc, ld};
    double *src = cart, *dst = W; /* W buffer is large enough: see alloc */
    int dims[4] = {n[0], n[1], n[2], n[3]};
    for (int ax = 0; ax < 4; ax++) {
        const double *Cm = c2s_table(ls[ax]);
        int outer = 1, inner = 1;
        for (int k = 0; k < ax; k++) outer *= dims[k];
        for (int k = ax + 1; k < 4; k++) inner *= dims[k];
        int nc = n[ax], ns = s[ax];
        for (int o = 0; o < outer; o++)
            for (int m = 0; m < ns; m++)
                for (int in = 0; in < inner; in++) {
                    double v = 0;
                    for (int c = 0; c < nc; c++)
                        v += Cm[m * nc + c] * src[((size_t)o * nc + c) * inner + in];
                    dst[((size_t)o * ns + m) * inner + in] = v;
                }
        dims[ax] = ns;
        double *tmp = src; src = dst; dst = tmp;
    }
    memcpy(sph, src, sizeof(double) * s[0] * s[1] * s[2] * s[3]);
}

/* Packed s4 ERI: out[(ij),(kl)], ij = i(i+1)/2+j (i>=j), npair x npair, exactly the
 * layout GTOnr2e_fill_s4 produces (dqc/hamilton/intor/symmetry.py:42-49).
 * No screening (prescreen NULL, molintor.py:676). */
void orc_int2e_s4(double *out, const int *atm, int natm, const int *bas, int nbas, const double *env)
{
    (void)natm;
    init_herm();
    int *ao_loc = (int *)malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(nbas, bas, ao_loc);
    size_t nao = ao_loc[nbas];
    size_t npair = nao * (nao + 1) / 2;
    int nshp = nbas * (nbas + 1) / 2;
    /* precompute primitive-pair Hermite tables for all shell pairs i>=j */
    PrimPair **pps = (PrimPair **)calloc(nshp, sizeof(PrimPair *));
    double **hb = (double **)calloc(nshp, sizeof(double *));
    int *npp = (int *)calloc(nshp, sizeof(int));
#pragma omp parallel for schedule(dynamic)
    for (int ij = 0; ij < nshp; ij++) {
        int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
        while (i * (i + 1) / 2 > ij) i--;
        while ((i + 1) * (i + 2) / 2 <= ij) i++;
        int j = ij - i * (i + 1) / 2;
        Shell A = get_shell(i, atm, bas, env), B = get_shell(j, atm, bas, env);
        int np = A.nprim * B.nprim;
        size_t sz = (size_t)np * NCART(A.l) * NCART(B.l) * NHERM(A.l + B.l);
        pps[ij] = (PrimPair *)malloc(sizeof(PrimPair) * np);
        hb[ij] = (double *)malloc(sizeof(double) * sz);
        npp[ij] = np;
        build_prim_pairs(A, B, pps[ij], hb[ij]);
    }
#pragma omp parallel
    {
        size_t wsz = (size_t)MAXCART * MAXCART * MAXCART * MAXCART;
        double *work = (double *)malloc(sizeof(double) * (2 * wsz + (size_t)MAXHERM * MAXCART * MAXCART));
        double *sph = (double *)malloc(sizeof(double) * wsz);
#pragma omp for schedule(dynamic)
        for (int ij = nshp - 1; ij >= 0; ij--) {
            int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
            while (i * (i + 1) / 2 > ij) i--;
            while ((i + 1) * (i + 2) / 2 <= ij) i++;
            int j = ij - i * (i + 1) / 2;
            Shell A = get_shell(i, atm, bas, env), B = get_shell(j, atm, bas, env);
            int sa = 2 * A.l + 1, sb = 2 * B.l + 1;
            for (int kl = 0; kl <= ij; kl++) {
                int k = (int)((sqrt(8.0 * kl + 1) - 1) / 2);
                while (k * (k + 1) / 2 > kl) k--;
                while ((k + 1) * (k + 2) / 2 <= kl) k++;
                int l = kl - k * (k + 1) / 2;
                Shell C = get_shell(k, atm, bas, env), D = get_shell(l, atm, bas, env);
                int sc = 2 * C.l + 1, sd = 2 * D.l + 1;
                eri_quartet(A, B, C, D, pps[ij], npp[ij], pps[kl], npp[kl], sph, work);
                for (int a = 0; a < sa; a++)
                    for (int b = 0; b < sb; b++) {
                        size_t ia = ao_loc[i] + a, ib = ao_loc[j] + b;
                        if (ib > ia) continue;
                        size_t pij = ia * (ia + 1) / 2 + ib;
                        for (int c = 0; c < sc; c++)
                            for (int d = 0; d < sd; d++) {
                                size_t ic = ao_loc[k] + c, id = ao_loc[l] + d;
                                if (id > ic) continue;
                                size_t pkl = ic * (ic + 1) / 2 + id;
                                double v = sph[((a * sb + b) * sc + c) * sd + d];
                                out[pij * npair + pkl] = v;
                                out[pkl * npair + pij] = v;
                            }
                    }
            }
        }
        free(work); free(sph);
    }
    for (int ij = 0; ij < nshp; ij++) { free(pps[ij]); free(hb[ij]); }
    free(pps); free(hb); free(npp); free(ao_loc);
}

/* Packed s8 ERI for bases whose s4 matrix does not fit the host (naphthalene / cc-pVTZ: 58 GB as s4, 29 GB here):
 * out[P (P + 1) / 2 + Q] = (ij|kl), P = i(i+1)/2+j >= Q = k(k+1)/2+l over AO pairs -- the lower triangle of the matrix
 * orc_int2e_s4 produces, the same numbers (eri_quartet), no screening. */
void orc_int2e_s8(double *out, const int *atm, int natm, const int *bas, int nbas, const double *env)
{
    (void)natm;
    init_herm();
    int *ao_loc = (int *)malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(nbas, bas, ao_loc);
    int nshp = nbas * (nbas + 1) / 2;
    PrimPair **pps = (PrimPair **)calloc(nshp, sizeof(PrimPair *));
    double **hb = (double **)calloc(nshp, sizeof(double *));
    int *npp = (int *)calloc(nshp, sizeof(int));
#pragma omp parallel for schedule(dynamic)
    for (int ij = 0; ij < nshp; ij++) {
        int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
        while (i * (i + 1) / 2 > ij) i--;
        while ((i + 1) * (i + 2) / 2 <= ij) i++;
        int j = ij - i * (i + 1) / 2;
        Shell A = get_shell(i, atm, bas, env), B = get_shell(j, atm, bas, env);
        int np = A.nprim * B.nprim;
        size_t sz = (size_t)np * NCART(A.l) * NCART(B.l) * NHERM(A.l + B.l);
        pps[ij] = (PrimPair *)malloc(sizeof(PrimPair) * np);
        hb[ij] = (double *)malloc(sizeof(double) * sz);
        npp[ij] = np;
        build_prim_pairs(A, B, pps[ij], hb[ij]);
    }
#pragma omp parallel
    {
        size_t wsz = (size_t)MAXCART * MAXCART * MAXCART * MAXCART;
        double *work = (double *)malloc(sizeof(double) * (2 * wsz + (size_t)MAXHERM * MAXCART * MAXCART));
        double *sph = (double *)malloc(sizeof(double) * wsz);
#pragma omp for schedule(dynamic)
        for (int ij = nshp - 1; ij >= 0; ij--) {
            int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
            while (i * (i + 1) / 2 > ij) i--;
            while ((i + 1) * (i + 2) / 2 <= ij) i++;
            int j = ij - i * (i + 1) / 2;
            Shell A = get_shell(i, atm, bas, env), B = get_shell(j, atm, bas, env);
            int sa = 2 * A.l + 1, sb = 2 * B.l + 1;
            for (int kl = 0; kl <= ij; kl++) {
                int k = (int)((sqrt(8.0 * kl + 1) - 1) / 2);
                while (k * (k + 1) / 2 > kl) k--;
                while ((k + 1) * (k + 2) / 2 <= kl) k++;
                int l = kl - k * (k + 1) / 2;
                Shell C = get_shell(k, atm, bas, env), D = get_shell(l, atm, bas, env);
                int sc = 2 * C.l + 1, sd = 2 * D.l + 1;
                eri_quartet(A, B, C, D, pps[ij], npp[ij], pps[kl], npp[kl], sph, work);
                for (int a = 0; a < sa; a++)
                    for (int b = 0; b < sb; b++) {
                        size_t ia = ao_loc[i] + a, ib = ao_loc[j] + b;
                        if (ib > ia) continue;
                        size_t pij = ia * (ia + 1) / 2 + ib;
                        for (int c = 0; c < sc; c++)
                            for (int d = 0; d < sd; d++) {
                                size_t ic = ao_loc[k] + c, id = ao_loc[l] + d;
                                if (id > ic) continue;
                                size_t pkl = ic * (ic + 1) / 2 + id;
                                size_t hi = pij > pkl ? pij : pkl, lo = pij > pkl ? pkl : pij;
                                out[hi * (hi + 1) / 2 + lo] = sph[((a * sb + b) * sc + c) * sd + d];
                            }
                    }
            }
        }
        free(work); free(sph);
    }
    for (int ij = 0; ij < nshp; ij++) { free(pps[ij]); free(hb[ij]); }
    free(pps); free(hb); free(npp); free(ao_loc);
}

/* y = M x for the symmetric matrix held as the packed lower triangle above (the J contraction of the s8 store:
 * x = packed density with doubled off-diagonals, y = packed J; hcgto.py:204-214 on the packed form) */
void orc_symv_s8(double *y, const double *packed, const double *x, long long npair)
{
    for (long long p = 0; p < npair; p++) y[p] = 0.0;
#pragma omp parallel
    {
        double *acc = (double *)calloc((size_t)npair, sizeof(double));
#pragma omp for schedule(dynamic, 64)
        for (long long hi = 0; hi < npair; hi++) {
            const double *row = packed + (size_t)hi * (hi + 1) / 2;
            const double xh = x[hi];
            double s = 0.0;
            for (long long lo = 0; lo < hi; lo++) {
                s += row[lo] * x[lo];
                acc[lo] += row[lo] * xh;
            }
            acc[hi] += s + row[hi] * xh;
        }
#pragma omp critical
        for (long long p = 0; p < npair; p++) y[p] += acc[p];
        free(acc);
    }
}

/* Listed shell quartets (test infrastructure for bases whose packed matrix does not fit the host: naphthalene /
 * cc-pVTZ is 58 GB packed).  quartets: (nq, 4) shell indices (i, j, k, l); the spherical block
 * [sa][sb][sc][sd] of quartet q is written row-major at out + offs[q] -- the same numbers orc_int2e_s4 scatters. */
void orc_int2e_quartets(double *out, const long long *offs, const int *quartets, int nq, const int *atm, int natm,
                        const int *bas, int nbas, const double *env)
{
    (void)natm; (void)nbas;
    init_herm();
#pragma omp parallel
    {
        size_t wsz = (size_t)MAXCART * MAXCART * MAXCART * MAXCART;
        double *work = (double *)malloc(sizeof(double) * (2 * wsz + (size_t)MAXHERM * MAXCART * MAXCART));
#pragma omp for schedule(dynamic)
        for (int q = 0; q < nq; q++) {
            Shell S[4];
            for (int x = 0; x < 4; x++) S[x] = get_shell(quartets[4 * q + x], atm, bas, env);
            PrimPair *pp[2];
            double *hb[2];
            int np[2];
            for (int x = 0; x < 2; x++) {
                Shell A = S[2 * x], B = S[2 * x + 1];
                np[x] = A.nprim * B.nprim;
                pp[x] = (PrimPair *)malloc(sizeof(PrimPair) * np[x]);
                hb[x] = (double *)malloc(sizeof(double) * (size_t)np[x] * NCART(A.l) * NCART(B.l) * NHERM(A.l + B.l));
                build_prim_pairs(A, B, pp[x], hb[x]);
            }
            eri_quartet(S[0], S[1], S[2], S[3], pp[0], np[0], pp[1], np[1], out + offs[q], work);
            for (int x = 0; x < 2; x++) { free(pp[x]); free(hb[x]); }
        }
        free(work);
    }
}

/* ------------------------------------------------------------------ */
/* 3-centre and 2-centre 2-electron integrals (density fitting).
 * Reference call sites: dqc/df/dfmol.py:35-40 -> intor.coul2c / coul3c -> int2c2e("r12") / int3c2e("ar12")
 * (dqc/hamilton/intor/molintor.py:36-72, 121-130), i.e. libcint's int2c2e_sph and int3c2e_sph.
 * libcint evaluates (ij|k) as the 4-centre integral with the fourth function replaced by the unit
 * s-function (exponent 0, value 1): here that function is an explicit shell with exponent 0 and
 * coefficient sqrt(4 pi) (cancelling the l = 0 solid-harmonic factor 1/sqrt(4 pi)).
 * `bas` is the concatenated shell table (LibcintWrapper.concatenate, lcintwrap.py:299-370); the orbital shells
 * are [sh0, sh1) and the auxiliary shells [k0, k1). */
static const double UNIT_EXP[1] = {0.0};
static const double UNIT_COEF[1] = {3.5449077018110320546}; /* sqrt(4 pi) */

static Shell unit_shell(const double *r)
{
    Shell u;
    u.l = 0; u.nprim = 1; u.exps = UNIT_EXP; u.coefs = UNIT_COEF; u.r = r;
    return u;
}

static int range_nao(const int *bas, int s0, int s1)
{
    int n = 0;
    for (int i = s0; i < s1; i++) n += 2 * bas[i * BAS_SLOTS + ANG_OF] + 1;
    return n;
}

/* out (nao, nao, naux) row-major */
void orc_int3c2e(double *out, const int *atm, int natm, const int *bas, int nbas, const double *env,
                 int sh0, int sh1, int k0, int k1)
{
    (void)natm; (void)nbas;
    init_herm();
    const int nsh = sh1 - sh0, nk = k1 - k0;
    const size_t nao = range_nao(bas, sh0, sh1), naux = range_nao(bas, k0, k1);
    int *loc = (int *)malloc(sizeof(int) * (nsh + 1)), *kloc = (int *)malloc(sizeof(int) * (nk + 1));
    loc[0] = 0; kloc[0] = 0;
    for (int i = 0; i < nsh; i++) loc[i + 1] = loc[i] + 2 * bas[(sh0 + i) * BAS_SLOTS + ANG_OF] + 1;
    for (int i = 0; i < nk; i++) kloc[i + 1] = kloc[i] + 2 * bas[(k0 + i) * BAS_SLOTS + ANG_OF] + 1;
    /* ket tables: (aux shell, unit) */
    PrimPair **kp = (PrimPair **)calloc(nk, sizeof(PrimPair *));
    double **kh = (double **)calloc(nk, sizeof(double *));
    for (int k = 0; k < nk; k++) {
        Shell C = get_shell(k0 + k, atm, bas, env), U = unit_shell(C.r);
        kp[k] = (PrimPair *)malloc(sizeof(PrimPair) * C.nprim);
        kh[k] = (double *)malloc(sizeof(double) * (size_t)C.nprim * NCART(C.l) * NHERM(C.l));
        build_prim_pairs(C, U, kp[k], kh[k]);
    }
#pragma omp parallel
    {
        size_t wsz = (size_t)MAXCART * MAXCART * MAXCART * MAXCART;
        double *work = (double *)malloc(sizeof(double) * (2 * wsz + (size_t)MAXHERM * MAXCART * MAXCART));
        double *sph = (double *)malloc(sizeof(double) * wsz);
#pragma omp for schedule(dynamic)
        for (int ij = 0; ij < nsh * (nsh + 1) / 2; ij++) {
            int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
            while (i * (i + 1) / 2 > ij) i--;
            while ((i + 1) * (i + 2) / 2 <= ij) i++;
            int j = ij - i * (i + 1) / 2;
            Shell A = get_shell(sh0 + i, atm, bas, env), B = get_shell(sh0 + j, atm, bas, env);
            int np = A.nprim * B.nprim, sa = 2 * A.l + 1, sb = 2 * B.l + 1;
            PrimPair *bp = (PrimPair *)malloc(sizeof(PrimPair) * np);
            double *bh = (double *)malloc(sizeof(double) * (size_t)np * NCART(A.l) * NCART(B.l) * NHERM(A.l + B.l));
            build_prim_pairs(A, B, bp, bh);
            for (int k = 0; k < nk; k++) {
                Shell C = get_shell(k0 + k, atm, bas, env), U = unit_shell(C.r);
                int sc = 2 * C.l + 1;
                eri_quartet(A, B, C, U, bp, np, kp[k], C.nprim, sph, work);
                for (int a = 0; a < sa; a++)
                    for (int b = 0; b < sb; b++)
                        for (int c = 0; c < sc; c++) {
                            size_t ia = loc[i] + a, ib = loc[j] + b, ic = kloc[k] + c;
                            double v = sph[(a * sb + b) * sc + c];
                            out[(ia * nao + ib) * naux + ic] = v;
                            out[(ib * nao + ia) * naux + ic] = v;
                        }
            }
            free(bp); free(bh);
        }
        free(work); free(sph);
    }
    for (int k = 0; k < nk; k++) { free(kp[k]); free(kh[k]); }
    free(kp); free(kh); free(loc); free(kloc);
}

/* out (naux, naux) row-major */
void orc_int2c2e(double *out, const int *atm, int natm, const int *bas, int nbas, const double *env, int k0, int k1)
{
    (void)natm; (void)nbas;
    init_herm();
    const int nk = k1 - k0;
    const size_t naux = range_nao(bas, k0, k1);
    int *kloc = (int *)malloc(sizeof(int) * (nk + 1));
    kloc[0] = 0;
    for (int i = 0; i < nk; i++) kloc[i + 1] = kloc[i] + 2 * bas[(k0 + i) * BAS_SLOTS + ANG_OF] + 1;
    PrimPair **kp = (PrimPair **)calloc(nk, sizeof(PrimPair *));
    double **kh = (double **)calloc(nk, sizeof(double *));
    for (int k = 0; k < nk; k++) {
        Shell C = get_shell(k0 + k, atm, bas, env), U = unit_shell(C.r);
        kp[k] = (PrimPair *)malloc(sizeof(PrimPair) * C.nprim);
        kh[k] = (double *)malloc(sizeof(double) * (size_t)C.nprim * NCART(C.l) * NHERM(C.l));
        build_prim_pairs(C, U, kp[k], kh[k]);
    }
#pragma omp parallel
    {
        size_t wsz = (size_t)MAXCART * MAXCART * MAXCART * MAXCART;
        double *work = (double *)malloc(sizeof(double) * (2 * wsz + (size_t)MAXHERM * MAXCART * MAXCART));
        double *sph = (double *)malloc(sizeof(double) * wsz);
#pragma omp for schedule(dynamic)
        for (int k = 0; k < nk; k++) {
            Shell C = get_shell(k0 + k, atm, bas, env), UC = unit_shell(C.r);
            int sc = 2 * C.l + 1;
            for (int l = 0; l <= k; l++) {
                Shell D = get_shell(k0 + l, atm, bas, env), UD = unit_shell(D.r);
                int sd = 2 * D.l + 1;
                eri_quartet(C, UC, D, UD, kp[k], C.nprim, kp[l], D.nprim, sph, work);
                for (int c = 0; c < sc; c++)
                    for (int d = 0; d < sd; d++) {
                        size_t ic = kloc[k] + c, id = kloc[l] + d;
                        out[ic * naux + id] = sph[c * sd + d];
                        out[id * naux + ic] = sph[c * sd + d];
                    }
            }
        }
        free(work); free(sph);
    }
    for (int k = 0; k < nk; k++) { free(kp[k]); free(kh[k]); }
    free(kp); free(kh); free(kloc);
}

/* fills4: expand packed (npair,npair) to dense (n,n,n,n); symmetry.py:55-64 */
void orc_fills4(double *dense, const double *packed, int n)
{
    size_t N = n, npair = N * (N + 1) / 2;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (size_t j = 0; j < N; j++) {
            size_t ii = (size_t)i > j ? (size_t)i : j, jj = (size_t)i > j ? j : (size_t)i;
            size_t pij = ii * (ii + 1) / 2 + jj;
            double *dst = dense + ((size_t)i * N + j) * N * N;
            const double *src = packed + pij * npair;
            for (size_t k = 0; k < N; k++)
                for (size_t l = 0; l < N; l++) {
                    size_t kk = k > l ? k : l, ll = k > l ? l : k;
                    dst[k * N + l] = src[kk * (kk + 1) / 2 + ll];
                }
        }
}

/* ------------------------------------------------------------------ */
/* AO values on grid.  deriv = 0: out (nao, ngrid);                    */
/* deriv = 1: out (3, nao, ngrid) = d/dx, d/dy, d/dz  (GTOval_ip_sph); */
/* deriv = 2: out (nao, ngrid) = laplacian (GTOval_lapl_sph)           */
/* coords: (ngrid,3) row-major                                        */
/* ------------------------------------------------------------------ */
void orc_eval_gto(int deriv, double *out, const double *coords, int ngrid, const int *atm, int natm,
                  const int *bas, int nbas, const double *env)
{
    (void)natm;
    int *ao_loc = (int *)malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(nbas, bas, ao_loc);
    size_t nao = ao_loc[nbas];
    size_t G = ngrid;
#pragma omp parallel for schedule(dynamic)
    for (int ish = 0; ish < nbas; ish++) {
        Shell A = get_shell(ish, atm, bas, env);
        int l = A.l, nc = NCART(l), ns = 2 * l + 1;
        int pw[MAXCART][3];
        cart_powers(l, pw);
        const double *Cm = c2s_table(l);
        for (size_t g = 0; g < G; g++) {
            double x = coords[g * 3] - A.r[0], y = coords[g * 3 + 1] - A.r[1], z = coords[g * 3 + 2] - A.r[2];
            double r2 = x * x + y * y + z * z;
            double e0 = 0, e1 = 0, e2 = 0; /* sum c e, sum c (-2a) e, sum c 4a^2 e */
            for (int ip = 0; ip < A.nprim; ip++) {
                double a = A.exps[ip];
                double e = A.coefs[ip] * exp(-a * r2);
                e0 += e; e1 += -2.0 * a * e; e2 += 4.0 * a * a * e;
            }
            double xp[LMAX + 3], yp[LMAX + 3], zp[LMAX + 3];
            xp[0] = yp[0] = zp[0] = 1.0;
            for (int k = 1; k <= l + 2; k++) { xp[k] = xp[k - 1] * x; yp[k] = yp[k - 1] * y; zp[k] = zp[k - 1] * z; }
            double cv[MAXCART], cx[MAXCART], cy[MAXCART], cz[MAXCART], cl[MAXCART];
            for (int c = 0; c < nc; c++) {
                int i = pw[c][0], j = pw[c][1], k = pw[c][2];
                double mono = xp[i] * yp[j] * zp[k];
                cv[c] = mono * e0;
                if (deriv == 1) {
                    /* d/dx [x^i e] = i x^{i-1} e + x^i (-2a x) e */
                    cx[c] = (i ? i * xp[i - 1] : 0.0) * yp[j] * zp[k] * e0 + xp[i + 1] * yp[j] * zp[k] * e1;
                    cy[c] = (j ? j * yp[j - 1] : 0.0) * xp[i] * zp[k] * e0 + xp[i] * yp[j + 1] * zp[k] * e1;
                    cz[c] = (k ? k * zp[k - 1] : 0.0) * xp[i] * yp[j] * e0 + xp[i] * yp[j] * zp[k + 1] * e1;
                } else if (deriv == 2) {
                    /* d2/dx2 [x^i e^{-a x^2}] = i(i-1)x^{i-2} - 2a(2i+1)x^i + 4a^2 x^{i+2} */
                    double dxx = (i >= 2 ? i * (i - 1) * xp[i - 2] : 0.0) * e0 + (2 * i + 1) * xp[i] * e1 + xp[i + 2] * e2;
                    double dyy = (j >= 2 ? j * (j - 1) * yp[j - 2] : 0.0) * e0 + (2 * j + 1) * yp[j] * e1 + yp[j + 2] * e2;
                    double dzz = (k >= 2 ? k * (k - 1) * zp[k - 2] : 0.0) * e0 + (2 * k + 1) * zp[k] * e1 + zp[k + 2] * e2;
                    cl[c] = dxx * yp[j] * zp[k] + xp[i] * dyy * zp[k] + xp[i] * yp[j] * dzz;
                }
            }
            for (int m = 0; m < ns; m++) {
                size_t ao = ao_loc[ish] + m;
                double v = 0, vx = 0, vy = 0, vz = 0, vl = 0;
                for (int c = 0; c < nc; c++) {
                    double cf = Cm[m * nc + c];
                    if (cf == 0.0) continue;
                    v += cf * cv[c];
                    if (deriv == 1) { vx += cf * cx[c]; vy += cf * cy[c]; vz += cf * cz[c]; }
                    if (deriv == 2) vl += cf * cl[c];
                }
                if (deriv == 0) out[ao * G + g] = v;
                else if (deriv == 1) {
                    out[(0 * nao + ao) * G + g] = vx;
                    out[(1 * nao + ao) * G + g] = vy;
                    out[(2 * nao + ao) * G + g] = vz;
                } else out[ao * G + g] = vl;
            }
        }
    }
    free(ao_loc);
}

/* Boys function exposed for tests of the HIP-side Rys tables (sum of weights = F_0) */
void orc_boys(int mmax, double T, double *F) { boys(mmax, T, F); }

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
