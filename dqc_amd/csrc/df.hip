// df.hip -- 2-centre and 3-centre 2-electron integrals for density fitting.
// Replaces  intor.coul2c(auxbw) = int2c2e_sph  and  intor.coul3c(basisw, basisw, auxbw) = int3c2e_sph
// (reference call sites dqc/df/dfmol.py:35-40, dqc/hamilton/intor/molintor.py:36-72, 121-130, 624-665).
//
// libcint evaluates (ij|k) and (k|l) as 4-centre integrals whose missing functions are the unit s-function.  The
// same construction is used here: one extra shell with exponent 0 and coefficient sqrt(4 pi) (cancelling the l = 0
// solid-harmonic factor) is appended to the shell table, the ket "pair" list is (auxiliary shell, unit), and the
// Rys shell-quartet kernel of eri_core.hpp runs unchanged in its 3C / 2C output modes: <LA,LB,LC,0> and <LA,0,LC,0>.
#include "eri_generic.hpp"

namespace dqc {

struct DfSetup {
    Basis b;
    HostPairs orb, aux;
    DevPool pool;
    DevShells ds;
    DevPairs dorb{nullptr, nullptr, nullptr}, daux{nullptr, nullptr, nullptr};
    EriOut og{0, 0, 0, 0};
};

static int df_setup(DfSetup &s, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, int sh0,
                    int sh1, int k0, int k1, bool need_orb, hipStream_t st, const char *who) {
    int rc = parse_basis(s.b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    if ((rc = boys_table_ensure())) return rc;
    if (sh0 < 0 || sh1 > nbas || sh0 > sh1 || k0 < 0 || k1 > nbas || k0 > k1) {
        set_error(std::string(who) + ": shell ranges outside the table");
        return DQC_EINVAL;
    }
    auto ao_of = [&](int sh) { return sh < nbas ? s.b.shells[sh].ao_off : s.b.nao; };
    s.og.ao0 = ao_of(sh0);
    s.og.nao = ao_of(sh1) - ao_of(sh0);
    s.og.aux0 = ao_of(k0);
    s.og.naux = ao_of(k1) - ao_of(k0);
    // the unit shell
    HostShell u;
    u.atom = 0; u.l = 0; u.nprim = 1; u.ao_off = s.b.nao; u.prim_off = (int)s.b.exps.size();
    u.r[0] = u.r[1] = u.r[2] = 0.0;
    s.b.exps.push_back(0.0);
    s.b.coefs.push_back(3.5449077018110320546);  // sqrt(4 pi)
    s.b.shells.push_back(u);
    const int unit = nbas;
    if (need_orb) build_pairs(s.b, s.orb, sh0, sh1);
    build_pairs(s.b, s.aux, k0, k1, unit);
    if ((rc = upload_shells(s.ds, s.b, s.pool, st))) { set_error(std::string(who) + ": device upload failed"); return rc; }
    auto up = [&](HostPairs &hp, DevPairs &dp) {
        int *d_sh = nullptr, *d_off = nullptr;
        double *d_pp = nullptr;
        int r;
        if ((r = s.pool.upload(&d_sh, hp.sh, st)) || (r = s.pool.upload(&d_off, hp.pp_off, st)) ||
            (r = s.pool.upload(&d_pp, hp.pp, st)))
            return r;
        dp = DevPairs{d_sh, d_off, d_pp};
        return 0;
    };
    if (need_orb && (rc = up(s.orb, s.dorb))) { set_error(std::string(who) + ": device upload failed"); return rc; }
    if ((rc = up(s.aux, s.daux))) { set_error(std::string(who) + ": device upload failed"); return rc; }
    return 0;
}

template <int LA, int LB, int LC, int MODE>
static int launch_df_class(double *out, const DfSetup &s, hipStream_t st) {
    using Cfg = EriCfg<LA, LB, LC, 0>;
    const HostPairs &hb = MODE == ERI_OUT_3C ? s.orb : s.aux;
    const DevPairs &db = MODE == ERI_OUT_3C ? s.dorb : s.daux;
    const int cb = LA * (LA + 1) / 2 + LB, ck = LC * (LC + 1) / 2;
    const int nb = hb.cls_count[cb], nk = s.aux.cls_count[ck];
    if (nb == 0 || nk == 0 || hl_forced()) return 0;
    const long long ntask = (long long)nb * nk;
    const long long nblk = eri_num_blocks<Cfg>(nb, nk, ntask);
    auto kern = eri_kernel<LA, LB, LC, 0, MODE>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES, st, out, s.ds, db, s.daux, hb.cls_start[cb],
                       nb, s.aux.cls_start[ck], nk, 0, ntask, s.og);
    DQC_CHECK_LAUNCH();
    return 0;
}

template <int LC>
static int launch_3c_ket(double *out, const DfSetup &s, hipStream_t st) {
    int rc;
#define DQC_3C(LA, LB) \
    if ((rc = launch_df_class<LA, LB, LC, ERI_OUT_3C>(out, s, st))) return rc;
    DQC_3C(0, 0) DQC_3C(1, 0) DQC_3C(1, 1) DQC_3C(2, 0) DQC_3C(2, 1) DQC_3C(2, 2) DQC_3C(3, 0) DQC_3C(3, 1) DQC_3C(3, 2)
    DQC_3C(3, 3)
#undef DQC_3C
    return 0;
}

template <int LC>
static int launch_2c_ket(double *out, const DfSetup &s, hipStream_t st) {
    int rc;
#define DQC_2C(LA) \
    if ((rc = launch_df_class<LA, 0, LC, ERI_OUT_2C>(out, s, st))) return rc;
    DQC_2C(0) DQC_2C(1) DQC_2C(2) DQC_2C(3)
#undef DQC_2C
    return 0;
}

// classes with a g shell (orbital or auxiliary) through the runtime kernel (eri_generic.hpp); DQC_ERI_GENERIC=1: all classes
template <int MODE>
static int launch_df_generic(double *out, const DfSetup &s, hipStream_t st) {
    const HostPairs &hb = MODE == ERI_OUT_3C ? s.orb : s.aux;
    const DevPairs &db = MODE == ERI_OUT_3C ? s.dorb : s.daux;
    for (int la = 0; la <= DQC_LMAX; la++)
        for (int lb = 0; lb <= (MODE == ERI_OUT_3C ? la : 0); lb++)
            for (int lc = 0; lc <= DQC_LMAX; lc++) {
                if (!hl_forced() && la <= ERI_LMAX && lc <= ERI_LMAX) continue;
                const int cb = la * (la + 1) / 2 + lb, ck = lc * (lc + 1) / 2;
                int rc = launch_hl<MODE>(out, s.ds, db, s.daux, hb.cls_start[cb], hb.cls_count[cb], s.aux.cls_start[ck],
                                         s.aux.cls_count[ck], 0, s.og, la, lb, lc, 0, st);
                if (rc) return rc;
            }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// density-fitted Coulomb matrix from the stored integrals (DFMol.get_elrep, dfmol.py:60-79):
//     t_k = sum_ij D_ij (ij|k),   c = (k|l)^-1 t,   J_ij = sum_k (ij|k) c_k
// Both passes stream j3c once; (ij|k) = (ji|k), so only the rows i >= j of the (nao, nao, naux) array are read
// (half the bytes) and D enters as (D_ij + D_ji)(1 - delta_ij / 2).  HBM-bound: 2 x 4 nao (nao+1) naux bytes.
// ---------------------------------------------------------------------------------------------
DQC_DEV void tri_decode(long long q, int &i, int &j) {
    long long r = (long long)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > q) r--;
    while ((r + 1) * (r + 2) / 2 <= q) r++;
    i = (int)r;
    j = (int)(q - r * (r + 1) / 2);
}

constexpr int DF_ROWS = 64;  // triangular rows per block in the first pass (32: twice the atomics on the naux sums, 0.113 ms; 64: 0.087; 96: 0.089 -- C5 molecule)

__global__ __launch_bounds__(256) void df_rhs_kernel(double *__restrict__ t, const double *__restrict__ j3c,
                                                     const double *__restrict__ dm, int nao, int naux, long long npair) {
    __shared__ double sw[DF_ROWS];
    __shared__ long long soff[DF_ROWS];
    const long long q0 = (long long)blockIdx.x * DF_ROWS;
    if (threadIdx.x < DF_ROWS) {
        const long long q = q0 + threadIdx.x;
        double wgt = 0.0;
        long long off = 0;
        if (q < npair) {
            int i, j;
            tri_decode(q, i, j);
            wgt = (dm[(size_t)i * nao + j] + dm[(size_t)j * nao + i]) * (i == j ? 0.5 : 1.0);
            off = ((long long)i * nao + j) * naux;
        }
        sw[threadIdx.x] = wgt;
        soff[threadIdx.x] = off;
    }
    __syncthreads();
    if ((naux & 1) == 0) {  // rows are 16-byte aligned: two auxiliary functions per lane and load
        for (int k = 2 * threadIdx.x; k < naux; k += 512) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
            for (int r = 0; r < DF_ROWS; r++) {
                const double2 v = *reinterpret_cast<const double2 *>(j3c + soff[r] + k);
                a0 += sw[r] * v.x;
                a1 += sw[r] * v.y;
            }
            atomicAdd(&t[k], a0);
            atomicAdd(&t[k + 1], a1);
        }
        return;
    }
    for (int k = threadIdx.x; k < naux; k += 256) {
        double acc = 0.0;
#pragma unroll 8
        for (int r = 0; r < DF_ROWS; r++) acc += sw[r] * j3c[soff[r] + k];
        atomicAdd(&t[k], acc);
    }
}

// y = A x for a small dense (n, n) matrix: one wave per row
__global__ __launch_bounds__(256) void df_matvec_kernel(double *__restrict__ y, const double *__restrict__ a,
                                                        const double *__restrict__ x, int n) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    double acc = 0.0;
    for (int k = lane; k < n; k += 64) acc += a[(size_t)row * n + k] * x[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) y[row] = acc;
}

__global__ __launch_bounds__(256) void df_j_kernel(double *__restrict__ jmat, const double *__restrict__ j3c,
                                                   const double *__restrict__ c, int nao, int naux, long long npair) {
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= npair) return;
    int i, j;
    tri_decode(q, i, j);
    const double *row = j3c + ((size_t)i * nao + j) * naux;
    double acc = 0.0;
    if ((naux & 1) == 0) {
        // four loads of the row in flight per lane (the plain loop waited for every 16 bytes before asking for the next: a 9 KB row
        // was nine dependent round trips)
        int k = 2 * lane;
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (; k + 384 < naux; k += 512) {
            const double2 v0 = *reinterpret_cast<const double2 *>(row + k), v1 = *reinterpret_cast<const double2 *>(row + k + 128);
            const double2 v2 = *reinterpret_cast<const double2 *>(row + k + 256), v3 = *reinterpret_cast<const double2 *>(row + k + 384);
            const double2 c0 = *reinterpret_cast<const double2 *>(c + k), c1 = *reinterpret_cast<const double2 *>(c + k + 128);
            const double2 c2 = *reinterpret_cast<const double2 *>(c + k + 256), c3 = *reinterpret_cast<const double2 *>(c + k + 384);
            acc += v0.x * c0.x + v0.y * c0.y;
            a1 += v1.x * c1.x + v1.y * c1.y;
            a2 += v2.x * c2.x + v2.y * c2.y;
            a3 += v3.x * c3.x + v3.y * c3.y;
        }
        for (; k < naux; k += 128) {
            const double2 v = *reinterpret_cast<const double2 *>(row + k), cc = *reinterpret_cast<const double2 *>(c + k);
            acc += v.x * cc.x + v.y * cc.y;
        }
        acc += a1 + a2 + a3;
    } else
        for (int k = lane; k < naux; k += 64) acc += row[k] * c[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) {
        jmat[(size_t)i * nao + j] = acc;
        jmat[(size_t)j * nao + i] = acc;
    }
}

}  // namespace dqc

extern "C" {

int dqc_int3c2e(double *d_out, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, int sh0,
                int sh1, int k0, int k1, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    DfSetup s;
    int rc = df_setup(s, atm, natm, bas, nbas, env, nenv, sh0, sh1, k0, k1, true, st, "dqc_int3c2e");
    if (rc) return rc;
    if (s.og.nao == 0 || s.og.naux == 0) return DQC_OK;
    if ((rc = launch_3c_ket<0>(d_out, s, st)) || (rc = launch_3c_ket<1>(d_out, s, st)) ||
        (rc = launch_3c_ket<2>(d_out, s, st)) || (rc = launch_3c_ket<3>(d_out, s, st)) ||
        (rc = launch_df_generic<ERI_OUT_3C>(d_out, s, st)))
        return rc;
    DQC_HIP(hipStreamSynchronize(st));
    return DQC_OK;
}

int dqc_int2c2e(double *d_out, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, int k0,
                int k1, void *stream) {
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    DfSetup s;
    int rc = df_setup(s, atm, natm, bas, nbas, env, nenv, 0, 0, k0, k1, false, st, "dqc_int2c2e");
    if (rc) return rc;
    if (s.og.naux == 0) return DQC_OK;
    if ((rc = launch_2c_ket<0>(d_out, s, st)) || (rc = launch_2c_ket<1>(d_out, s, st)) ||
        (rc = launch_2c_ket<2>(d_out, s, st)) || (rc = launch_2c_ket<3>(d_out, s, st)) ||
        (rc = launch_df_generic<ERI_OUT_2C>(d_out, s, st)))
        return rc;
    DQC_HIP(hipStreamSynchronize(st));
    return DQC_OK;
}

int dqc_df_coulomb(double *d_j, const double *d_j3c, const double *d_inv_j2c, const double *d_dm_ao, int nao, int naux,
                   double *d_work, void *stream) {
    // d_work: 2 * naux doubles (t, c); d_j (nao, nao) is overwritten.  Enqueues only.
    using namespace dqc;
    hipStream_t st = (hipStream_t)stream;
    if (nao <= 0) return DQC_OK;
    if (naux <= 0) { DQC_HIP(hipMemsetAsync(d_j, 0, sizeof(double) * (size_t)nao * nao, st)); return DQC_OK; }
    const long long npair = (long long)nao * (nao + 1) / 2;
    double *t = d_work, *c = d_work + naux;
    DQC_HIP(hipMemsetAsync(t, 0, sizeof(double) * naux, st));
    hipLaunchKernelGGL(df_rhs_kernel, dim3((unsigned)((npair + DF_ROWS - 1) / DF_ROWS)), dim3(256), 0, st, t, d_j3c, d_dm_ao,
                       nao, naux, npair);
    DQC_CHECK_LAUNCH();
    hipLaunchKernelGGL(df_matvec_kernel, dim3((unsigned)((naux + 3) / 4)), dim3(256), 0, st, c, d_inv_j2c, t, naux);
    DQC_CHECK_LAUNCH();
    hipLaunchKernelGGL(df_j_kernel, dim3((unsigned)((npair + 3) / 4)), dim3(256), 0, st, d_j, d_j3c, c, nao, naux, npair);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
