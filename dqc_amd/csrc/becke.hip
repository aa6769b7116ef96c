// becke.hip -- Becke partition weights of a multi-centre integration grid, one lane per grid point.
//
// Replaces the reference's torch expression (dqc/grid/multiatoms_scheme.py:9-67, called from dqc/grid/becke_grid.py:18-60):
// for every point of atom a's grid
//     mu_ij = (r_j - r_i) / R_ij,   mu_ij <- mu_ij - a_ij (mu_ij^2 - 1)          (atomic-size adjustment, a_ij clamped to +-0.45)
//     s_ij  = 1/2 (1 - f(f(f(mu_ij)))),  f(x) = x (3 - x^2) / 2                  (three-fold cell function)
//     P_j   = prod_i s_ij, dropped where some mu_ij >= cut (the reference's sparsification, 0.74)
//     w     = P_a / sum_j P_j
// The reference (and dqc_amd.grid on the CPU) forms (natm, natm, ngrid_atom) temporaries atom by atom: ~500 elementwise
// launches moving 56 MB each for a 20-atom sg3 grid (12 ms).  Here a point keeps its own distances: the (i, j) loops are
// uniform across the wave (pair tables through scalar loads), nothing but the coordinates is read and one double is written.
#include "common.hpp"

namespace dqc {

__global__ __launch_bounds__(256) void becke_weights_kernel(double *__restrict__ w, const double *__restrict__ xyz,
                                                            const int *__restrict__ atom_off, const double *__restrict__ pos,
                                                            const double *__restrict__ inv_rij, const double *__restrict__ aij,
                                                            int natm, int ngrid, double cut) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngrid) return;
    const double x = xyz[3 * (size_t)g], y = xyz[3 * (size_t)g + 1], z = xyz[3 * (size_t)g + 2];
    int own = 0;
    while (own + 1 < natm && g >= atom_off[own + 1]) own++;  // the atom whose grid this point belongs to
    const double sdiag = 0.5 * (1.0 + 1e-12) + 0.5;  // the i == j factor of the reference expression (f(0) = 0, + eye / 2)
    double psum = 0.0, pown = 0.0;
    for (int j = 0; j < natm; j++) {
        const double dxj = x - pos[3 * j], dyj = y - pos[3 * j + 1], dzj = z - pos[3 * j + 2];
        const double rj = sqrt(dxj * dxj + dyj * dyj + dzj * dzj);
        double p = 1.0;
        bool keep = true;
        for (int i = 0; i < natm; i++) {
            if (i == j) { p *= sdiag; continue; }
            const double dx = x - pos[3 * i], dy = y - pos[3 * i + 1], dz = z - pos[3 * i + 2];
            const double ri = sqrt(dx * dx + dy * dy + dz * dz);
            double mu = (rj - ri) * inv_rij[i * natm + j];
            mu = mu - aij[i * natm + j] * (mu * mu - 1.0);
            keep = keep && (mu < cut);
            double f = mu;
#pragma unroll
            for (int k = 0; k < 3; k++) f = -0.5 * (f * (f * f - 3.0));
            p *= -0.5 * (f - (1.0 + 1e-12));
        }
        if (!keep) p = 0.0;
        psum += p;
        if (j == own) pown = p;
    }
    w[g] = pown / psum;
}

}  // namespace dqc

extern "C" int dqc_becke_weights(double *d_w, const double *d_xyz, const int *d_atom_off, const double *d_pos,
                                 const double *d_inv_rij, const double *d_aij, int natm, int ngrid, double cut, void *stream) {
    // d_xyz (ngrid, 3): the atoms' grids one after the other, atom a owning the points [d_atom_off[a], d_atom_off[a + 1]);
    // d_pos (natm, 3); d_inv_rij, d_aij (natm, natm): 1 / |R_i - R_j| (anything finite on the diagonal) and the size-adjustment
    // coefficients a_ij; d_w (ngrid) <- partition weights.  Enqueues only.
    using namespace dqc;
    if (ngrid <= 0 || natm <= 0) return DQC_OK;
    hipLaunchKernelGGL(becke_weights_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_w, d_xyz, d_atom_off,
                       d_pos, d_inv_rij, d_aij, natm, ngrid, cut);
    DQC_CHECK_LAUNCH();
    return DQC_OK;
}
