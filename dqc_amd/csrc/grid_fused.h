/* grid_fused.h -- entry points of the VARIANT library libdqc_amd_fused.so (grid_vxc.hip compiled with -DDQC_WITH_FUSED).
 * Not part of the product ABI (include/dqc_amd.h): the fused pass is 3x slower than the three launches it would replace
 *  and is kept as a measured negative result and a cross-check, exercised by its own test. */
#ifndef DQC_AMD_GRID_FUSED_H
#define DQC_AMD_GRID_FUSED_H
#ifdef __cplusplus
extern "C" {
#endif
/* ---- fused grid pass: density -> XC potentials -> Vxc matrix from ONE read of the AO matrix ------
 * = dqc_grid_density_lr + dqc_xc_eval + dqc_grid_vxc (HamiltonCGTO.get_vxc, hcgto.py:260-269 with 371-495) for a
 * restricted density given in factor form D = L L^T (d_orb (ld, norb_pad), d_orbt (norb_pad, ld) as for
 * dqc_grid_density_lr), d_ao (4, ngrid, ld), LDA / GGA functional ids.  Covered shapes: dqc_grid_fused_supported(nao,
 * norb_pad) != 0 (ld / 16 in {11, 13}, i.e. 145 <= nao <= 208 except 177..192, norb_pad <= 64); otherwise DQC_EINVAL.
 * Optional outputs (may be NULL): d_rho (ngrid), d_grho (3, ngrid), d_exc (1) = sum_g w_g e_xc(g). */
int dqc_grid_fused_supported(int nao, int norb_pad);
int dqc_grid_fused(double *d_vmat, double *d_rho, double *d_grho, double *d_exc, const double *d_ao, int ngrid, int nao,
                   const double *d_w, const double *d_orb, const double *d_orbt, int norb_pad, const int *ids,
                   const double *coefs, int nterm, void *stream);
#ifdef __cplusplus
}
#endif
#endif
