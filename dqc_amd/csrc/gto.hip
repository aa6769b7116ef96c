// gto.hip -- AO values (and gradients) on the grid.
// Replaces GTOval_sph / GTOval_ip_sph (reference call site dqc/hamilton/intor/gtoeval.py:196-239)
// with the (ngrid, nao) "to_transpose" layout HamiltonCGTO.setup_grid caches (hcgto.py:168, :179).
// No screening, like the reference (non0tab all ones, gtoeval.py:211-212).
//
// Mapping: one wave per 64 points, one lane per grid point, shells looped uniformly (shell data
// comes through scalar loads); a 64-point x 16-column LDS tile turns the per-lane row writes into 128-byte
// row segments before they go to HBM.
#include "common.hpp"
#include <type_traits>

namespace dqc {

// compile-time copy of the solid-harmonic tables: the per-l shell bodies below unroll over it and keep the non-zero terms only
namespace gto_ce {
#define C2S_QUAL constexpr
#include "cart2sph.inc"
#undef C2S_QUAL
}  // namespace gto_ce

// DERIV: 0 phi | 1 + gradient (4 components) | 2 + laplacian (5) | 3 + the six second derivatives xx xy xz yy yz zz (10,
// used by the GGA nuclear gradient)
typedef double gto_v2d __attribute__((ext_vector_type(2)));
template <int DERIV>
__global__ __launch_bounds__(64) void eval_gto_kernel(double *__restrict__ out, const double *__restrict__ coords,
                                                       int ngrid, int nao, int ld, DevShells sh, int colrange) {
    constexpr int NC = DERIV == 0 ? 1 : (DERIV == 1 ? 4 : (DERIV == 2 ? 5 : 10));
    // columns staged per flush.  The tile is what bounds the occupancy (one wave per block): 16 columns x 4 components = 35 KB
    // allowed 4 waves per CU and the kernel wrote at 1 TB/s (round 3); 8 columns let 8 waves per CU overlap their evaluation and
    // their stores.  The ten components of DERIV 3 had 4 columns (32-byte row segments): 5.1 ms for a 20-atom cc-pVDZ molecule;
    // with 8 (64-byte segments, 46 KB of tile, three waves per CU) 3.1 ms
    constexpr int GTO_CW = DERIV == 0 ? 16 : 8;  // (16 columns for the GGA form again in round 6, with the faster evaluation: no gain)
    __shared__ double tile[1][NC][64][GTO_CW + 1];
    constexpr int wave = 0;
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    if (g0 >= ngrid) return;
    const int g = min(g0 + lane, ngrid - 1);
    const double px = coords[g * 3], py = coords[g * 3 + 1], pz = coords[g * 3 + 2];
    const size_t cstride = (size_t)ngrid * ld;

    // grid.y splits the AO columns into ranges of `colrange` (a multiple of GTO_CW): a block evaluates the shells that reach into
    // its range and stores the columns inside it.  One wave used to walk ALL shells of its 64 points (a 20-atom cc-pVDZ molecule:
    // 96 shells, 236 exponentials per point) with 6-8 waves per CU: 1.0 ms = 2.2 TB/s of writes for the four GGA components.
    const int cbeg = blockIdx.y * colrange, cend = min(cbeg + colrange, ld);
    if (cbeg >= ld) return;
    int col0 = cbeg;  // first AO column held in the tile
    int nfill = 0;    // columns filled
    auto flush = [&](int ncols) {
        // tile[wave][c][p][j] -> out[c][g0+p][col0+j]; lanes sweep (p, j) with j fastest
        __syncthreads();
        if (ncols == GTO_CW) {  // a full tile: 16-byte stores (col0 is a multiple of GTO_CW and the row stride is even)
            for (int c = 0; c < NC; c++)
                for (int e = lane; e < 64 * (GTO_CW / 2); e += 64) {
                    const int p = e / (GTO_CW / 2), j = 2 * (e % (GTO_CW / 2));
                    if (g0 + p < ngrid)
                        // non-temporal: the array is 2-6 GB written once (round 6, rocprofv3: 0.820 -> 0.754 ms for the four GGA
                        // components of a 20-atom cc-pVDZ molecule, 2.12 -> 1.88 ms for the ten of the gradient code)
                        __builtin_nontemporal_store(gto_v2d{tile[wave][c][p][j], tile[wave][c][p][j + 1]},
                                                    reinterpret_cast<gto_v2d *>(out + c * cstride + (size_t)(g0 + p) * ld + col0 + j));
                }
        } else {
            for (int c = 0; c < NC; c++)
                for (int e = lane; e < 64 * GTO_CW; e += 64) {
                    int p = e / GTO_CW, j = e % GTO_CW;
                    if (j < ncols && g0 + p < ngrid)
                        out[c * cstride + (size_t)(g0 + p) * ld + col0 + j] = tile[wave][c][p][j];
                }
        }
        __syncthreads();
    };

    // One shell's columns, l a COMPILE-TIME constant (round 6): the Cartesian loops unroll, the solid-harmonic coefficients fold
    // into the code (zero terms vanish) and the powers x^lx are read from registers by constant index.  With l a run-time value
    // (rounds 1-5) every xp[lx] was a chain of selects over the power table and every term tested its coefficient:
    // the evaluation alone took 1.0 ms of the 1.1 ms launch (tools/ubench/_alt ablations: no stores 1.00 ms, no evaluation 0.84).
    auto shell = [&](auto lc, int a0, double x, double y, double z, double e0, double e1, double e2) {
        constexpr int l = decltype(lc)::value;
        double xp[l + 3], yp[l + 3], zp[l + 3];
        xp[0] = yp[0] = zp[0] = 1.0;
#pragma unroll
        for (int k = 1; k <= l + 2; k++) { xp[k] = xp[k - 1] * x; yp[k] = yp[k - 1] * y; zp[k] = zp[k - 1] * z; }
        constexpr int nc = (l + 1) * (l + 2) / 2, ns = 2 * l + 1;
#pragma unroll
        for (int m = 0; m < ns; m++) {
            if (a0 + m < cbeg || a0 + m >= cend) continue;  // (a shell straddling two ranges is evaluated by both blocks)
            double v = 0, vx = 0, vy = 0, vz = 0, vl = 0;
            double hxx = 0, hxy = 0, hxz = 0, hyy = 0, hyz = 0, hzz = 0;
            int c = 0;
#pragma unroll
            for (int lx = l; lx >= 0; lx--)
#pragma unroll
                for (int ly = l - lx; ly >= 0; ly--, c++) {
                    const double cf = gto_ce::C2S[gto_ce::C2S_OFF[l] + m * nc + c];
                    if (cf == 0.0) continue;  // (folded at compile time)
                    const int lz = l - lx - ly;
                    double mono = xp[lx] * yp[ly] * zp[lz];
                    v += cf * mono * e0;
                    if (DERIV) {
                        vx += cf * ((lx ? lx * xp[lx ? lx - 1 : 0] : 0.0) * yp[ly] * zp[lz] * e0 + xp[lx + 1] * yp[ly] * zp[lz] * e1);
                        vy += cf * ((ly ? ly * yp[ly ? ly - 1 : 0] : 0.0) * xp[lx] * zp[lz] * e0 + xp[lx] * yp[ly + 1] * zp[lz] * e1);
                        vz += cf * ((lz ? lz * zp[lz ? lz - 1 : 0] : 0.0) * xp[lx] * yp[ly] * e0 + xp[lx] * yp[ly] * zp[lz + 1] * e1);
                    }
                    if (DERIV == 2) {
                        // d2/dx2 [x^i exp(-a x^2)] = i(i-1) x^(i-2) - 2a(2i+1) x^i + 4a^2 x^(i+2), summed over primitives
                        const double dxx = (lx >= 2 ? lx * (lx - 1) * xp[lx >= 2 ? lx - 2 : 0] : 0.0) * e0 + (2 * lx + 1) * xp[lx] * e1 + xp[lx + 2] * e2;
                        const double dyy = (ly >= 2 ? ly * (ly - 1) * yp[ly >= 2 ? ly - 2 : 0] : 0.0) * e0 + (2 * ly + 1) * yp[ly] * e1 + yp[ly + 2] * e2;
                        const double dzz = (lz >= 2 ? lz * (lz - 1) * zp[lz >= 2 ? lz - 2 : 0] : 0.0) * e0 + (2 * lz + 1) * zp[lz] * e1 + zp[lz + 2] * e2;
                        vl += cf * (dxx * yp[ly] * zp[lz] + xp[lx] * dyy * zp[lz] + xp[lx] * yp[ly] * dzz);
                    }
                    if (DERIV == 3) {
                        const double dxx = (lx >= 2 ? lx * (lx - 1) * xp[lx >= 2 ? lx - 2 : 0] : 0.0) * e0 + (2 * lx + 1) * xp[lx] * e1 + xp[lx + 2] * e2;
                        const double dyy = (ly >= 2 ? ly * (ly - 1) * yp[ly >= 2 ? ly - 2 : 0] : 0.0) * e0 + (2 * ly + 1) * yp[ly] * e1 + yp[ly + 2] * e2;
                        const double dzz = (lz >= 2 ? lz * (lz - 1) * zp[lz >= 2 ? lz - 2 : 0] : 0.0) * e0 + (2 * lz + 1) * zp[lz] * e1 + zp[lz + 2] * e2;
                        // d2/dxdy [x^i y^j R(r^2)] = i j x^(i-1) y^(j-1) R + (i x^(i-1) y^(j+1) + j x^(i+1) y^(j-1)) R1 + x^(i+1) y^(j+1) R2
                        const double xm = lx ? lx * xp[lx ? lx - 1 : 0] : 0.0, ym = ly ? ly * yp[ly ? ly - 1 : 0] : 0.0, zm = lz ? lz * zp[lz ? lz - 1 : 0] : 0.0;
                        const double dxy = xm * ym * e0 + (xm * yp[ly + 1] + ym * xp[lx + 1]) * e1 + xp[lx + 1] * yp[ly + 1] * e2;
                        const double dxz = xm * zm * e0 + (xm * zp[lz + 1] + zm * xp[lx + 1]) * e1 + xp[lx + 1] * zp[lz + 1] * e2;
                        const double dyz = ym * zm * e0 + (ym * zp[lz + 1] + zm * yp[ly + 1]) * e1 + yp[ly + 1] * zp[lz + 1] * e2;
                        hxx += cf * dxx * yp[ly] * zp[lz];
                        hyy += cf * xp[lx] * dyy * zp[lz];
                        hzz += cf * xp[lx] * yp[ly] * dzz;
                        hxy += cf * dxy * zp[lz];
                        hxz += cf * dxz * yp[ly];
                        hyz += cf * dyz * xp[lx];
                    }
                }
            tile[wave][0][lane][nfill] = v;
            if (DERIV) {
                tile[wave][1][lane][nfill] = vx;
                tile[wave][2][lane][nfill] = vy;
                tile[wave][3][lane][nfill] = vz;
            }
            if (DERIV == 2) tile[wave][4][lane][nfill] = vl;
            if (DERIV == 3) {
                tile[wave][4][lane][nfill] = hxx;
                tile[wave][5][lane][nfill] = hxy;
                tile[wave][6][lane][nfill] = hxz;
                tile[wave][7][lane][nfill] = hyy;
                tile[wave][8][lane][nfill] = hyz;
                tile[wave][9][lane][nfill] = hzz;
            }
            nfill++;
            if (nfill == GTO_CW) {
                flush(GTO_CW);
                col0 += GTO_CW;
                nfill = 0;
            }
        }
    };
    static_assert(DQC_LMAX == 4, "one shell body per accepted l");

    for (int is = 0; is < sh.nsh; is++) {
        const int l = sh.l[is], np = sh.nprim[is], po = sh.prim_off[is];
        const int a0 = sh.ao_off[is];
        if (a0 + 2 * l + 1 <= cbeg || a0 >= cend) continue;  // (uniform: the shell has no column in this block's range)
        const double x = px - sh.xyz[is * 3], y = py - sh.xyz[is * 3 + 1], z = pz - sh.xyz[is * 3 + 2];
        const double r2 = x * x + y * y + z * z;
        double e0 = 0, e1 = 0, e2 = 0;
        for (int ip = 0; ip < np; ip++) {
            double a = sh.exps[po + ip];
            double e = sh.coefs[po + ip] * exp(-a * r2);
            e0 += e;
            e1 -= 2.0 * a * e;
            e2 += 4.0 * a * a * e;
        }
        switch (l) {  // (uniform)
        case 0: shell(std::integral_constant<int, 0>{}, a0, x, y, z, e0, e1, e2); break;
        case 1: shell(std::integral_constant<int, 1>{}, a0, x, y, z, e0, e1, e2); break;
        case 2: shell(std::integral_constant<int, 2>{}, a0, x, y, z, e0, e1, e2); break;
        case 3: shell(std::integral_constant<int, 3>{}, a0, x, y, z, e0, e1, e2); break;
        default: shell(std::integral_constant<int, 4>{}, a0, x, y, z, e0, e1, e2); break;
        }
    }
    // zero padding columns nao..ld-1 (ld = the arrays' row stride, dqc_ao_stride: ld - nao < 16)
    while (col0 + nfill < cend) {
        for (int c = 0; c < NC; c++) tile[wave][c][lane][nfill] = 0.0;
        nfill++;
        if (nfill == GTO_CW) {
            flush(GTO_CW);
            col0 += GTO_CW;
            nfill = 0;
        }
    }
    if (nfill) flush(nfill);
}

}  // namespace dqc

extern "C" int dqc_eval_gto(int deriv, double *d_out, const double *d_coords, int ngrid, const int *atm,
                            int natm, const int *bas, int nbas, const double *env, int nenv, void *stream) {
    using namespace dqc;
    if (deriv < 0 || deriv > 3) { set_error("dqc_eval_gto: deriv must be 0, 1, 2 or 3"); return DQC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    Basis b;
    int rc = parse_basis(b, atm, natm, bas, nbas, env, nenv, nullptr);
    if (rc) return rc;
    DevPool pool(st);  // stream-ordered scratch: this call only enqueues
    DevShells ds;
    if ((rc = upload_shells(ds, b, pool, st))) { set_error("dqc_eval_gto: device upload failed"); return rc; }
    if (ngrid > 0) {
        int nblk = (ngrid + 63) / 64;
        const int ld = dqc_ao_stride(b.nao);
        // column ranges (grid.y), every range a whole number of flush tiles.  Measured by rocprofv3: the four GGA components take
        // 1.17 ms split four ways against 1.01 ms unsplit (shells that straddle two ranges are evaluated twice, and the write
        // pattern, not the parallelism, bounds the kernel): only the ten-component form is split
        static const int nsplit3 = [] { const char *e = getenv("DQC_GTO_SPLIT"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
        const int cw = deriv == 0 ? 16 : 8, nsplit = (deriv == 3 && ld >= 64) ? nsplit3 : 1;
        const int colrange = ((ld + nsplit - 1) / nsplit + cw - 1) / cw * cw;
        const dim3 grid(nblk, (ld + colrange - 1) / colrange);
        const int ncomp = deriv == 0 ? 1 : (deriv == 1 ? 4 : (deriv == 2 ? 5 : 10));
        // the slack the grid kernels may read past the last row (dqc_ao_doubles): zeros
        const size_t body = (size_t)ncomp * (size_t)ngrid * ld, slack = dqc_ao_doubles(ncomp, ngrid, b.nao) - body;
        if (slack) DQC_HIP(hipMemsetAsync(d_out + body, 0, slack * sizeof(double), st));
        if (deriv == 0)
            hipLaunchKernelGGL(eval_gto_kernel<0>, grid, dim3(64), 0, st, d_out, d_coords, ngrid, b.nao, ld, ds, colrange);
        else if (deriv == 1)
            hipLaunchKernelGGL(eval_gto_kernel<1>, grid, dim3(64), 0, st, d_out, d_coords, ngrid, b.nao, ld, ds, colrange);
        else if (deriv == 2)
            hipLaunchKernelGGL(eval_gto_kernel<2>, grid, dim3(64), 0, st, d_out, d_coords, ngrid, b.nao, ld, ds, colrange);
        else
            hipLaunchKernelGGL(eval_gto_kernel<3>, grid, dim3(64), 0, st, d_out, d_coords, ngrid, b.nao, ld, ds, colrange);
        DQC_CHECK_LAUNCH();
    }
    return DQC_OK;
}
