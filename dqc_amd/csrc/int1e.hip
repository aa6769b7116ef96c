// int1e.hip -- overlap, kinetic and nuclear-attraction matrices on the device.
// Replaces GTOint2c(int1e_{ovlp,kin,nuc}_sph) (reference call site
// dqc/hamilton/intor/molintor.py:624-644; shortcuts :96-112).  O(nao^2) one-off work: one wave
// per unordered shell pair (see int1e_kernel), Cartesian blocks in private memory, solid-harmonic transform at the end.
//
// overlap / kinetic : 1D Obara-Saika recurrences for S_ij = int (x-A)^i (x-B)^j exp(...) dx
// nuclear attraction: Rys quadrature, V = -Z (2 pi/p) K_ab sum_r w_r Ix Iy Iz  (same root tables
//                     as the two-electron kernels)
#include "common.hpp"

namespace dqc {

constexpr int L1 = DQC_LMAX + 1;      // i index range
constexpr int L3 = DQC_LMAX + 3;      // j index range (+2 for the kinetic operator)

// s[i][j] for i<=la, j<=lb : overlap of x^i_A x^j_B exp(-a x_A^2 - b x_B^2) over one axis, WITHOUT the
// sqrt(pi/p) and exp(-mu AB^2) factors
DQC_DEV void overlap_1d(double s[L1][L3], int la, int lb, double PA, double PB, double hp) {
    s[0][0] = 1.0;
    for (int i = 0; i < la; i++) s[i + 1][0] = PA * s[i][0] + (i ? i * hp * s[i - 1][0] : 0.0);
    for (int j = 0; j < lb; j++)
        for (int i = 0; i <= la; i++)
            s[i][j + 1] = PB * s[i][j] + (j ? j * hp * s[i][j - 1] : 0.0) + (i ? i * hp * s[i - 1][j] : 0.0);
}

template <int N>
DQC_DEV void nuc_accumulate(double *cart, int la, int lb, int na, int nb, double p, const double *P,
                            const double *A, const double *AB, const double *C, double pref) {
    double X = p * ((P[0] - C[0]) * (P[0] - C[0]) + (P[1] - C[1]) * (P[1] - C[1]) + (P[2] - C[2]) * (P[2] - C[2]));
    double u[N], w[N];
    rys_roots<N>(X, u, w);
    for (int r = 0; r < N; r++) {
        double g[3][2 * DQC_LMAX + 1][L1];  // g[d][i][j] after HRR (i up to la+lb-j)
        const double b10 = 0.5 * (1.0 - u[r]) / p;
        for (int d = 0; d < 3; d++) {
            const double c00 = (P[d] - A[d]) - u[r] * (P[d] - C[d]);
            g[d][0][0] = 1.0;
            if (la + lb > 0) g[d][1][0] = c00;
            for (int n = 1; n < la + lb; n++) g[d][n + 1][0] = c00 * g[d][n][0] + n * b10 * g[d][n - 1][0];
            for (int j = 1; j <= lb; j++)
                for (int i = 0; i <= la + lb - j; i++) g[d][i][j] = g[d][i + 1][j - 1] + AB[d] * g[d][i][j - 1];
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
                        cart[ca * nb + cb] += wr * g[0][ax][bx] * g[1][ay][by] * g[2][az][bz];
                    }
            }
    }
}

// One WAVE per unordered shell pair (ish >= jsh; every operator here is real symmetric, the transposed block is mirrored): the
// 64 lanes split the (primitive pair) x (nucleus) combinations -- 1280 for a contracted (s|s) pair of a 20-atom molecule -- each
// into a private Cartesian block, the blocks are summed across the wave (shuffles) into LDS and the solid-harmonic transform is
// done by the lanes over the (m_a, m_b) outputs.  Round 4: the kernel had one THREAD per ordered pair (144 waves for a 20-atom
// cc-pVDZ molecule, nuclear attraction 4.1 ms -- a third of the whole ERI fill for an O(n^2) integral).
constexpr int I1_WPB = 4;  // waves (shell pairs) per block
__global__ __launch_bounds__(64 * I1_WPB) void int1e_kernel(int which, double *__restrict__ out, int nao, DevShells sh, int natm,
                             const double *__restrict__ atom_xyz, const double *__restrict__ atom_z) {
    constexpr int MC = (DQC_LMAX + 1) * (DQC_LMAX + 2) / 2;
    __shared__ double scart[I1_WPB][MC * MC];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * I1_WPB + wv;
    const long long npair = (long long)sh.nsh * (sh.nsh + 1) / 2;
    if (q >= npair) return;  // (wave-uniform: no block-level barrier below)
    int ish = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((long long)ish * (ish + 1) / 2 > q) ish--;
    while ((long long)(ish + 1) * (ish + 2) / 2 <= q) ish++;
    const int jsh = (int)(q - (long long)ish * (ish + 1) / 2);
    const int la = sh.l[ish], lb = sh.l[jsh];
    const int na = (la + 1) * (la + 2) / 2, nb = (lb + 1) * (lb + 2) / 2;
    const double A[3] = {sh.xyz[ish * 3], sh.xyz[ish * 3 + 1], sh.xyz[ish * 3 + 2]};
    const double B[3] = {sh.xyz[jsh * 3], sh.xyz[jsh * 3 + 1], sh.xyz[jsh * 3 + 2]};
    const double AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
    const double ab2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
    double cart[MC * MC];
    for (int i = 0; i < na * nb; i++) cart[i] = 0.0;

    const int npa = sh.nprim[ish], npb = sh.nprim[jsh], npp = npa * npb;
    const int ncomb = which == 2 ? npp * natm : npp;
    for (int cmb = lane; cmb < ncomb; cmb += 64) {
        const int pq = which == 2 ? cmb % npp : cmb, ic = which == 2 ? cmb / npp : 0;
        const int ip = pq / npb, jp = pq - ip * npb;
        {
            const double a = sh.exps[sh.prim_off[ish] + ip], b = sh.exps[sh.prim_off[jsh] + jp];
            const double cc = sh.coefs[sh.prim_off[ish] + ip] * sh.coefs[sh.prim_off[jsh] + jp];
            const double p = a + b, hp = 0.5 / p;
            const double K = exp(-a * b / p * ab2);
            double P[3];
            for (int d = 0; d < 3; d++) P[d] = (a * A[d] + b * B[d]) / p;
            if (which == 2) {
                const int nroots = (la + lb) / 2 + 1;
                {
                    const double pref = -atom_z[ic] * cc * K * 2.0 * M_PI / p;
                    const double *C = atom_xyz + ic * 3;
                    switch (nroots) {
                    case 1: nuc_accumulate<1>(cart, la, lb, na, nb, p, P, A, AB, C, pref); break;
                    case 2: nuc_accumulate<2>(cart, la, lb, na, nb, p, P, A, AB, C, pref); break;
                    case 3: nuc_accumulate<3>(cart, la, lb, na, nb, p, P, A, AB, C, pref); break;
                    case 4: nuc_accumulate<4>(cart, la, lb, na, nb, p, P, A, AB, C, pref); break;
                    default: nuc_accumulate<5>(cart, la, lb, na, nb, p, P, A, AB, C, pref); break;
                    }
                }
            } else {
                double s[3][L1][L3];
                for (int d = 0; d < 3; d++) overlap_1d(s[d], la, lb + 2, P[d] - A[d], P[d] - B[d], hp);
                const double pref = cc * K * pow(M_PI / p, 1.5);
                int ca = 0;
                for (int ax = la; ax >= 0; ax--)
                    for (int ay = la - ax; ay >= 0; ay--, ca++) {
                        const int az = la - ax - ay;
                        int cb = 0;
                        for (int bx = lb; bx >= 0; bx--)
                            for (int by = lb - bx; by >= 0; by--, cb++) {
                                const int bz = lb - bx - by;
                                const double sx = s[0][ax][bx], sy = s[1][ay][by], sz = s[2][az][bz];
                                double v;
                                if (which == 0) {
                                    v = sx * sy * sz;
                                } else if (which >= 3) {
                                    // multipole moments about the origin 0 (libcint int1e_r_sph / int1e_rr_sph; reference
                                    // intor.int1e("r0" * n), hcgto.py:117-125): x = (x - B) + B on the ket raises j,
                                    //   <i|x|j> = S(j+1) + B S(j),   <i|x^2|j> = S(j+2) + 2 B S(j+1) + B^2 S(j)
                                    int e[3] = {0, 0, 0};
                                    if (which < 6) e[which - 3] = 1;
                                    else { e[(which - 6) / 3] += 1; e[(which - 6) % 3] += 1; }
                                    const int bj[3] = {bx, by, bz}, ai[3] = {ax, ay, az};
                                    v = 1.0;
                                    for (int d = 0; d < 3; d++) {
                                        const double s0 = s[d][ai[d]][bj[d]], s1 = s[d][ai[d]][bj[d] + 1], s2 = s[d][ai[d]][bj[d] + 2];
                                        v *= e[d] == 0 ? s0 : (e[d] == 1 ? s1 + B[d] * s0 : s2 + 2.0 * B[d] * s1 + B[d] * B[d] * s0);
                                    }
                                } else {
                                    // -1/2 d2/dx2 on the ket: -2b^2 S(j+2) + b(2j+1) S(j) - j(j-1)/2 S(j-2)
                                    const double tx = -2.0 * b * b * s[0][ax][bx + 2] + b * (2 * bx + 1) * sx -
                                                      (bx >= 2 ? 0.5 * bx * (bx - 1) * s[0][ax][bx - 2] : 0.0);
                                    const double ty = -2.0 * b * b * s[1][ay][by + 2] + b * (2 * by + 1) * sy -
                                                      (by >= 2 ? 0.5 * by * (by - 1) * s[1][ay][by - 2] : 0.0);
                                    const double tz = -2.0 * b * b * s[2][az][bz + 2] + b * (2 * bz + 1) * sz -
                                                      (bz >= 2 ? 0.5 * bz * (bz - 1) * s[2][az][bz - 2] : 0.0);
                                    v = tx * sy * sz + sx * ty * sz + sx * sy * tz;
                                }
                                cart[ca * nb + cb] += pref * v;
                            }
                    }
            }
        }
    }
    // sum of the lanes' partial blocks (butterfly: every lane ends with the total; fixed order -> reproducible), into LDS
    for (int e = 0; e < na * nb; e++) {
        double v = cart[e];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == (e & 63)) scart[wv][e] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // solid-harmonic transform of both indices: lanes over the (m_a, m_b) outputs; the transposed block is mirrored
    const double *Ca = C2S + C2S_OFF[la], *Cb = C2S + C2S_OFF[lb];
    const int ia0 = sh.ao_off[ish], ib0 = sh.ao_off[jsh];
    const int sa = 2 * la + 1, sb = 2 * lb + 1;
    for (int o = lane; o < sa * sb; o += 64) {
        const int ma = o / sb, mb = o - ma * sb;
        double v = 0;
        for (int ca = 0; ca < na; ca++) {
            const double fa = Ca[ma * na + ca];
            if (fa == 0.0) continue;
            double t = 0;
            for (int cb = 0; cb < nb; cb++) t += Cb[mb * nb + cb] * scart[wv][ca * nb + cb];
            v += fa * t;
        }
        out[(size_t)(ia0 + ma) * nao + ib0 + mb] = v;
        if (ish != jsh) out[(size_t)(ib0 + mb) * nao + ia0 + ma] = v;  // (a diagonal pair writes each element once)
    }
}

}  // namespace dqc

extern "C" int dqc_int1e(int which, double *d_out, const int *atm, int natm, const int *bas, int nbas,
                         const double *env, int nenv, const double *zs, void *stream) {
    using namespace dqc;
    if (which < 0 || which > 14) {
        set_error("dqc_int1e: which must be 0 (ovlp), 1 (kin), 2 (nuc), 3-5 (r0: x, y, z) or 6-14 (r0r0: 6 + 3 d1 + d2)");
        return DQC_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, zs);
    if (rc) return rc;
    DevPool pool(st);  // stream-ordered scratch: this call only enqueues
    DevShells ds;
    if ((rc = upload_shells(ds, b, pool, st))) { set_error("dqc_int1e: device upload failed"); return rc; }
    double *d_xyz = nullptr, *d_z = nullptr;
    if ((rc = pool.upload(&d_xyz, b.atom_xyz, st)) || (rc = pool.upload(&d_z, b.atom_z, st))) {
        set_error("dqc_int1e: device upload failed");
        return rc;
    }
    const long long npair = (long long)nbas * (nbas + 1) / 2;  // one wave per unordered shell pair
    if (npair > 0) {
        hipLaunchKernelGGL(int1e_kernel, dim3((unsigned)((npair + I1_WPB - 1) / I1_WPB)), dim3(64 * I1_WPB), 0, st, which, d_out, b.nao, ds, natm,
                           d_xyz, d_z);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}
