// fock.hip -- the small-matrix ends of a restricted Fock build, two launches instead of ~16 (round 6)
//
// Around the tile stream (and the grid pass) a Fock build of the reference does, per density matrix, in torch.matmul / einsum calls
// (hcgto.py:204-241 J / K with the orbital-converter round trips of orbconverter.py:126-163; hf.py:182-201, ks.py:176-187):
//     D_ao = X D X^T                                  (AO density from the orthogonal-basis density)
//     J, K from the accumulators, symmetrised          (jk_finish_kernel)
//     E_J = 1/2 tr D_ao J,  E_K = -1/4 tr D_ao K
//     F2 = X^T (J - K / 2 + V_xc,ao) X, symmetrised    (+ the core Hamiltonian)
// Each of these is a few microseconds of work and was one launch (or three) of its own: at nao = 114 (benzene / cc-pVDZ) the build
// took 0.24 ms as a hipGraph around a 58 us tile pass.  Here:
//     fock_prep_kernel    one block per 16-row panel of D_ao: T = X_panel D (LDS), P = T X^T for the tiles on and right of the
//                         diagonal, written with their mirror images (the matrix comes out bitwise symmetric, no second pass);
//                         or P = L_panel L^T straight from the AO-basis orbital factor; zeroes the J / K accumulators
//     fock_finish_kernel  one block per 16-row panel of F2: M = J (+ J^T) - (K + K^T) / 2 + V formed on the fly from the accumulators,
//                         T = X_panel^T M (LDS), F2 = T X + core for the tiles on and right of the diagonal + mirrors; block 0 also
//                         forms the two traces in a fixed order (deterministic)
// fp64 MFMA (16x16x4) panels, 16 waves per block; operands come from L2 (the matrices are 0.1 - 1.4 MB).
#include "grid_common.hpp"

namespace dqc {

constexpr int FK_NT = 1024, FK_WAVES = FK_NT / 64;

// LDS row stride of a 16-row panel with K columns: == 2 (mod 32) doubles, so that the A-fragment reads sP[lr][k0 + kq]
// (bank = 2 lr + kq over a half-wave) are conflict-free
DQC_DEV int fk_stride(int K) { return ((K + 31) / 32) * 32 + 2; }

// out tile (16 x 16) = A_panel (16 x K, LDS, stride NS) . B, B[k][n] = bfun(k, n) for k < K (K padded to a multiple of 4 by the caller
// through zero columns of the panel); returns the accumulators (C layout: row = kq + 4 reg, col = lr)
template <class BF>
DQC_DEV v4d fk_tile(const double *sA, int NS, int K4, int lr, int kq, BF bfun) {
    // batches of U k-steps: all operand loads of a batch are issued before its first MFMA (bfun loads unconditionally from clamped
    // addresses and selects -- a guarded load would put the L2 latency in front of every MFMA)
    constexpr int U = 8;
    v4d acc{0, 0, 0, 0};
    const double *ap = sA + lr * NS + kq;
    int k0 = 0;
    for (; k0 + 4 * U <= K4; k0 += 4 * U) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a[u] = ap[k0 + 4 * u];
            b[u] = bfun(k0 + 4 * u + kq, lr);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = mfma_f64(a[u], b[u], acc);
    }
    if (k0 < K4) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool on = k0 + 4 * u < K4;
            a[u] = on ? ap[k0 + 4 * u] : 0.0;  // (the panel has 16 NS doubles: k0 + 4 u + kq stays inside for on == true only)
            b[u] = bfun(k0 + 4 * u + kq, lr);  // (clamped loads: always in range; a zero column of the panel meets it past K)
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = mfma_f64(a[u], b[u], acc);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// prep.  mode 0: D_ao = X Ds X^T with Ds = (D + D^T) / 2, D (north x north), X (nao x north) row-major.
//        mode 1: D_ao = L L^T with L = orb (ldo x rp) row-major, rows >= nao zero (the padded AO-basis factor of ao_orb2dm).
// work[0 : n2] <- D_ao zero padded to (npad x npad); work[n2 : (2 | 3) n2] <- 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FK_NT) void fock_prep_kernel(double *__restrict__ work, const double *__restrict__ dm,
                                                          const double *__restrict__ x, const double *__restrict__ orb, int rp, int nao,
                                                          int north, int npad, int with_k) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x, i0 = 16 * p;
    const size_t n2 = (size_t)npad * npad;
    // the accumulators of the tile stream
    {
        const size_t tot = (size_t)(with_k ? 2 : 1) * n2;
        for (size_t e = (size_t)blockIdx.x * FK_NT + tid; e < tot; e += (size_t)gridDim.x * FK_NT) work[n2 + e] = 0.0;
    }
    const int K = orb ? rp : north;   // inner dimension of the second product
    const int K4 = (K + 3) & ~3, NS = fk_stride(K4);
    double *sT = lds;                 // (16, NS): T = X_panel Ds   or   L_panel
    double *sX = lds + 16 * NS;       // (16, NSX): X_panel (mode 0 only)
    double *sD = lds + 2 * 16 * NS;   // (16, 17): the diagonal tile
    if (orb) {
        for (int e = tid; e < 16 * K4; e += FK_NT) {
            const int r = e / K4, c = e - r * K4;
            sT[r * NS + c] = (i0 + r < nao && c < rp) ? orb[(size_t)(i0 + r) * rp + c] : 0.0;
        }
        __syncthreads();
    } else {
        for (int e = tid; e < 16 * K4; e += FK_NT) {
            const int r = e / K4, c = e - r * K4;
            sX[r * NS + c] = (i0 + r < nao && c < north) ? x[(size_t)(i0 + r) * north + c] : 0.0;
        }
        __syncthreads();
        const int ntn = (north + 15) / 16;
        for (int jt = wave; jt < ntn; jt += FK_WAVES) {
            const v4d acc = fk_tile(sX, NS, K4, lr, kq, [&](int k, int n) {
                const int c = 16 * jt + n, kc = min(k, north - 1), cc = min(c, north - 1);
                const double val = 0.5 * (dm[(size_t)kc * north + cc] + dm[(size_t)cc * north + kc]);
                return (k < north && c < north) ? val : 0.0;
            });
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (16 * jt + lr < K4) sT[(kq + 4 * r) * NS + 16 * jt + lr] = acc[r];
        }
        __syncthreads();
    }
    // P = T B^T for the tiles jt >= p, B = X (mode 0) or L (mode 1), both (nao x K) row-major with row stride K
    const double *bm = orb ? orb : x;
    const int ntp = (npad + 15) / 16;
    for (int jt = p + wave; jt < ntp; jt += FK_WAVES) {
        const v4d acc = fk_tile(sT, NS, K4, lr, kq, [&](int k, int n) {
            const int j = 16 * jt + n;
            const double val = bm[(size_t)min(j, nao - 1) * K + min(k, K - 1)];
            return (k < K && j < nao) ? val : 0.0;
        });
        if (jt == p) {
#pragma unroll
            for (int r = 0; r < 4; r++) sD[(kq + 4 * r) * 17 + lr] = acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + kq + 4 * r, j = 16 * jt + lr;
                if (i < npad && j < npad) {
                    work[(size_t)i * npad + j] = acc[r];
                    work[(size_t)j * npad + i] = acc[r];
                }
            }
        }
    }
    __syncthreads();
    if (tid < 256) {
        const int a = tid >> 4, b = tid & 15, i = i0 + a, j = i0 + b;
        if (i < npad && j < npad) work[(size_t)i * npad + j] = 0.5 * (sD[a * 17 + b] + sD[b * 17 + a]);
    }
}

// ---------------------------------------------------------------------------------------------
// finish.  fock (north x north) = sym( X^T M X ) + core,  M = (Wj + Wj^T) - (Wk + Wk^T) / 2 + V   (AO basis, nao x nao);
// en[0] = 1/2 sum D_ao J, en[1] = -1/4 sum D_ao K (0 without K); jout (nao x nao, optional) <- J.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FK_NT) void fock_finish_kernel(double *__restrict__ fock, double *__restrict__ en, double *__restrict__ jout,
                                                            const double *__restrict__ work, const double *__restrict__ v, int ldv,
                                                            const double *__restrict__ core, const double *__restrict__ x, int nao, int north,
                                                            int npad, int with_k, const double *__restrict__ dscp) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[2][FK_WAVES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int q = blockIdx.x, a0 = 16 * q;
    const size_t n2 = (size_t)npad * npad;
    const double dsc = dscp ? *dscp : 0.0;
    const double *wj = work + n2, *wk = work + 2 * n2;
    auto jval = [&](int i, int j) { return det_value(wj[(size_t)i * npad + j], dsc) + det_value(wj[(size_t)j * npad + i], dsc); };
    auto kval = [&](int i, int j) { return det_value(wk[(size_t)i * npad + j], dsc) + det_value(wk[(size_t)j * npad + i], dsc); };
    const int KA = (nao + 3) & ~3, NSA = fk_stride(KA);       // first product: inner dimension nao
    double *sXt = lds;                // (16, NSA): X^T panel
    double *sT = lds + 16 * NSA;      // (16, NSA): T = X^T_panel M
    double *sD = lds + 2 * 16 * NSA;  // (16, 17)
    for (int e = tid; e < 16 * KA; e += FK_NT) {
        const int i = e >> 4, a = e & 15;  // (consecutive threads: consecutive columns of X)
        sXt[a * NSA + i] = (i < nao && a0 + a < north) ? x[(size_t)i * north + a0 + a] : 0.0;
    }
    __syncthreads();
    const int nta = (nao + 15) / 16;
    for (int jt = wave; jt < nta; jt += FK_WAVES) {
        const v4d acc = fk_tile(sXt, NSA, KA, lr, kq, [&](int i, int n) {
            const int j = 16 * jt + n, ic = min(i, nao - 1), jc = min(j, nao - 1);
            double m = jval(ic, jc);
            if (with_k) m -= 0.5 * kval(ic, jc);  // (block-uniform)
            if (v) m += v[(size_t)ic * ldv + jc];
            return (i < nao && j < nao) ? m : 0.0;
        });
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (16 * jt + lr < KA) sT[(kq + 4 * r) * NSA + 16 * jt + lr] = acc[r];
    }
    __syncthreads();
    const int ntn = (north + 15) / 16;
    for (int bt = q + wave; bt < ntn; bt += FK_WAVES) {
        const v4d acc = fk_tile(sT, NSA, KA, lr, kq, [&](int j, int n) {
            const int b = 16 * bt + n;
            const double val = x[(size_t)min(j, nao - 1) * north + min(b, north - 1)];
            return (j < nao && b < north) ? val : 0.0;
        });
        if (bt == q) {
#pragma unroll
            for (int r = 0; r < 4; r++) sD[(kq + 4 * r) * 17 + lr] = acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int a = a0 + kq + 4 * r, b = 16 * bt + lr;
                if (a < north && b < north) {
                    fock[(size_t)a * north + b] = acc[r] + (core ? core[(size_t)a * north + b] : 0.0);
                    fock[(size_t)b * north + a] = acc[r] + (core ? core[(size_t)b * north + a] : 0.0);
                }
            }
        }
    }
    __syncthreads();
    if (tid < 256) {
        const int a = tid >> 4, b = tid & 15, ia = a0 + a, ib = a0 + b;
        if (ia < north && ib < north)
            fock[(size_t)ia * north + ib] = 0.5 * (sD[a * 17 + b] + sD[b * 17 + a]) + (core ? core[(size_t)ia * north + ib] : 0.0);
    }
    if (q != 0) return;
    // block 0: the two traces (fixed order: thread-strided partial sums, wave reduction, waves added in order) and the J copy
    double sj = 0.0, sk = 0.0;
    for (int e = tid; e < nao * nao; e += FK_NT) {
        const int i = e / nao, j = e - i * nao;
        const double d = work[(size_t)i * npad + j], jv = jval(i, j);
        sj += d * jv;
        if (with_k) sk += d * kval(i, j);
        if (jout) jout[e] = jv;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sj += __shfl_down(sj, o);
        sk += __shfl_down(sk, o);
    }
    if (lane == 0) { red[0][wave] = sj; red[1][wave] = sk; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < FK_WAVES; w++) { a += red[0][w]; b += red[1][w]; }
        en[0] = 0.5 * a;
        en[1] = -0.25 * b;
    }
}

}  // namespace dqc

extern "C" {

int dqc_fock_max_nao(void) { return 448; }  // two 16-row panels of fk_stride(nao) doubles + the diagonal tile within 160 KB of LDS

int dqc_fock_prep(double *d_work, const double *d_dm, const double *d_x, const double *d_orb, int rp, int nao, int north, int with_k,
                  void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    if (nao > dqc_fock_max_nao() || north > nao || north <= 0) { set_error("dqc_fock_prep: needs 0 < north <= nao <= 448"); return DQC_EINVAL; }
    if (!d_orb && (!d_dm || !d_x)) { set_error("dqc_fock_prep: either the AO-basis factor or (dm, x)"); return DQC_EINVAL; }
    if (d_orb && (rp <= 0 || rp > nao + 16)) { set_error("dqc_fock_prep: factor width outside (0, nao + 16]"); return DQC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const int K = d_orb ? rp : north, K4 = (K + 3) & ~3, NS = ((K4 + 31) / 32) * 32 + 2;
    const size_t shm = sizeof(double) * (2 * 16 * (size_t)NS + 16 * 17);
    (void)hipFuncSetAttribute((const void *)fock_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(fock_prep_kernel, dim3((npad + 15) / 16), dim3(FK_NT), shm, st, d_work, d_dm, d_x, d_orb, rp, nao, north, npad, with_k);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

int dqc_fock_finish(double *d_fock, double *d_energies, double *d_j_ao, const double *d_work, const double *d_vxc_ao, int ldv,
                    const double *d_core, const double *d_x, int nao, int north, int with_k, void *stream) {
    using namespace dqc;
    if (nao <= 0) return DQC_OK;
    if (nao > dqc_fock_max_nao() || north > nao || north <= 0) { set_error("dqc_fock_finish: needs 0 < north <= nao <= 448"); return DQC_EINVAL; }
    if (!d_fock || !d_energies || !d_work || !d_x) { set_error("dqc_fock_finish: null argument"); return DQC_EINVAL; }
    if (d_vxc_ao && ldv < nao) { set_error("dqc_fock_finish: ldv < nao"); return DQC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int npad = (nao + DQC_TILE_B - 1) / DQC_TILE_B * DQC_TILE_B;
    const int KA = (nao + 3) & ~3, NSA = ((KA + 31) / 32) * 32 + 2;
    const size_t shm = sizeof(double) * (2 * 16 * (size_t)NSA + 16 * 17);
    const double *dscp = deterministic_mode() ? d_work + 3 * (size_t)npad * npad : nullptr;
    (void)hipFuncSetAttribute((const void *)fock_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(fock_finish_kernel, dim3((north + 15) / 16), dim3(FK_NT), shm, st, d_fock, d_energies, d_j_ao, d_work, d_vxc_ao, ldv,
                       d_core, d_x, nao, north, npad, with_k, dscp);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}

}  // extern "C"
