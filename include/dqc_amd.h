/*
 * dqc_amd.h -- C ABI of libdqc_amd.so: the MI355X (gfx950) native arithmetic behind DQC's
 * SCF Fock-build hot path.  These entry points are what the reference's ctypes layer for this
 * path would bind in place of dqclibs (libcint/libcgto) and pylibxc; INTEGRATION.md shows the
 * reference-side stubs.  All matrices are float64; all integer tables int32.
 *
 * Conventions
 *   - `atm`, `bas`, `env` are HOST pointers to libcint-style tables exactly as
 *     LibcintWrapper builds them (reference: dqc/hamilton/intor/lcintwrap.py:37-86):
 *     atm[natm][6] = {Z, ptr_xyz, 1, ptr_zeta, 0, 0}; bas[nbas][8] = {iatom, l, nprim, nctr=1,
 *     kappa, ptr_exp, ptr_coef, 0}; env = doubles.  nctr must be 1 (the reference splits general
 *     contractions upstream, dqc/api/loadbasis.py:72-82).  Spherical AOs, libcint order.
 *   - every other pointer named d_* is a DEVICE pointer (HBM); the caller owns all buffers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue work and
 *     return; they synchronise only where stated.
 *   - return value: 0 on success, negative DQC_E* on error; dqc_last_error() gives the text.
 *   - `ld` ("leading dimension") of the AO-indexed SQUARE device matrices (D, V, the factor pair) is
 *     dqc_padded_nao(nao), the smallest multiple of 16 >= nao; padding columns/rows are zero.
 *   - the AO-on-grid arrays (dqc_eval_gto output, the d_ao arguments) have their own row stride
 *     `lda` = dqc_ao_stride(nao) (nao rounded up to 8 doubles: 64-byte rows, no tile padding in HBM) and must be
 *     allocated with dqc_ao_doubles(ncomp, ngrid, nao) doubles: the kernels read whole 16-column tiles,
 *     i.e. up to ld - lda doubles past the end of the last row (dqc_eval_gto zeroes that slack).
 */
#ifndef DQC_AMD_H
#define DQC_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQC_OK 0
#define DQC_EINVAL (-1)   /* bad argument (unsupported l, nctr != 1, ...) */
#define DQC_EHIP (-2)     /* a HIP runtime call failed */
#define DQC_ENOMEM (-3)

const char *dqc_last_error(void);
int dqc_version(void);
/* number of spherical AOs described by bas  (CINTcgto_spheric summed; lcintwrap.py:376-383) */
int dqc_nao(const int *bas, int nbas);
/* leading dimension of the AO-indexed square device matrices */
int dqc_padded_nao(int nao);
/* row stride of the AO-on-grid arrays, and the doubles an array of ncomp components must hold */
int dqc_ao_stride(int nao);
size_t dqc_ao_doubles(int ncomp, int ngrid, int nao);

/* ---- one-electron integrals --------------------------------------------------------------
 * Replaces GTOint2c(int1e_{ovlp,kin,nuc}_sph, ...)  (dqc/hamilton/intor/molintor.py:624-644,
 * shortcuts :96-112).  which: 0 overlap, 1 kinetic, 2 nuclear attraction (-sum_A Z_A <i|1/r_A|j>).
 * zs: optional HOST array of natm (possibly fractional) charges, NULL = use atm[:,0]
 * (molintor.py:105-112).  d_out: (nao, nao) row-major, contiguous (ld = nao). */
int dqc_int1e(int which, double *d_out, const int *atm, int natm, const int *bas, int nbas,
              const double *env, int nenv, const double *zs, void *stream);

/* ---- two-electron integrals into blocked-s8 tiles ------------------------------------------
 * Replaces GTOnr2e_fill_drv(int2e_sph, GTOnr2e_fill_s4, prescreen=NULL, ...) + CSYMM().fills4
 * (molintor.py:667-688, symmetry.py:40-69).  Instead of the reference's packed (npair,npair)
 * buffer expanded to a dense nao^4 tensor, the integrals stay on the device in 8-fold-unique
 * TILE storage: AOs are grouped in blocks of 8; tile (I>=J, K>=L, IJ>=KL) holds the sub-tensor
 * g[(i,j)][(k,l)] contiguously, R(IJ) rows x C(KL) columns.  PACKED since round 3 (symmetry.py:40-69 packs s4; here the
 * 8-fold symmetry is packed at block AND element level): a diagonal block pair (I == J, resp. K == L) keeps only its
 * i >= j (k >= l) elements -- 36 rows (columns), index i (i + 1) / 2 + j -- any other pair all 64 (index 8 i + j).  Tile
 * (IJ, KL) starts at off(I, J) + R(IJ) * (64 KL - 28 K), off(I, J) = 8 I (64 I^3 + 16 I^2 + 129 I - 47) + 256 J (8 I^2 + I + 8 J + 8)
 * (the prefix sum of the tile sizes over the block pairs before (I, J)).  dqc_eri_store_doubles(nao) = what d_tiles must hold
 * (1.895 GB at nao 208; nao^4 / 8 doubles = 1.872 GB is the ideal, full 8^4 tiles were 2.024 GB); dqc_eri_tile_count = number
 * of tiles.  Fully overwritten; enqueues only. */
size_t dqc_eri_tile_count(int nao);
size_t dqc_eri_store_doubles(int nao);
int dqc_eri_fill_tiles(double *d_tiles, const int *atm, int natm, const int *bas, int nbas,
                       const double *env, int nenv, void *stream);
/* expand the tiles into the reference's dense (nao,nao,nao,nao) tensor (tests / small nao only) */
/* HOST only (no device call): the pair tables the fill / direct kernels are launched with.  merge != 0: s shells of one atom over the
 * same exponent list -- the contractions the reference splits a generally contracted shell into (dqc/api/loadbasis.py:72-82) --
 * evaluated as one group.  h_out[4] = groups, pairs, primitive pairs kept, primitive quartets over all unique (bra, ket) pairs. */
int dqc_eri_pair_stats(const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, int merge, long long *h_out);

int dqc_eri_tiles_to_dense(double *d_dense, const double *d_tiles, int nao, void *stream);

/* ---- J / K contraction ---------------------------------------------------------------------
 * Replaces einsum("ij,ijkl->kl", dm, el_mat) and einsum("il,ijkl->ijk").sum(-3)
 * (dqc/hamilton/hcgto.py:209, :234) in the AO basis.  d_dm: (nao,nao) contiguous, symmetric
 * part is used.  d_J: (nao,nao) <- J_kl = sum_ij D_ij (ij|kl), symmetrised.  d_K may be NULL;
 * otherwise <- K_jk = sum_il D_il (ij|kl), symmetrised (NOT scaled by -1/2).
 * d_work: dqc_jk_work_doubles(nao) doubles of scratch. */
size_t dqc_jk_work_doubles(int nao);
int dqc_jk_from_tiles(double *d_J, double *d_K, const double *d_tiles, const double *d_dm,
                      int nao, double *d_work, void *stream);

/* ---- the small-matrix ends of a restricted Fock build, fused (round 6; csrc/fock.hip) ----------
 * Around its tile pass and grid pass the reference's Fock build runs, per density matrix, a dozen nao^3 / nao^2 torch calls:
 * D_ao = X D X^T (orbconverter.py:126-163 unconvert_dm), the symmetrisation of J and K, the energy traces (hcgto.py:302-328),
 * X^T (J - K / 2 + V_xc) X (convert2, hcgto.py:204-241, ks.py:176-187) and the sum with the core Hamiltonian (hf.py:182-201).
 * Four entry points replace them around dqc's own kernels (six tile-parallel launches: one 4-wave block per 16 x 16 output tile):
 *   dqc_fock_prep           d_work <- symmetric AO density (zero padded) + zeroed J / K accumulators.  Either from the orthogonal-
 *                           basis density d_dm (north, north) [its symmetric part] and the orthogonaliser d_x (nao, north), or
 *                           (d_orb != NULL) from the AO-basis orbital factor of ao_orb2dm, d_orb (>= nao rows, rp columns,
 *                           row-major, rows >= nao zero): D_ao = L L^T.
 *   dqc_jk_stream_prepared  the pass over the ERI tiles on that work buffer (with_k: Coulomb + exchange), accumulators left in it
 *   dqc_fock_finish         d_fock (north, north) <- sym(X^T (J - K / 2 + V) X) + core, bitwise symmetric; d_energies[0] =
 *                           1/2 tr D_ao J, [1] = -1/4 tr D_ao K (0 without K); d_j_ao (nao, nao; may be NULL) <- J.
 *                           d_vxc_ao: symmetric AO-basis matrix with row stride ldv (dqc_grid_vxc's output) or NULL; d_core:
 *                           (north, north) or NULL.
 *   dqc_fock_factor         the AO-basis factor of D = C diag(w) C^T (ao_orb2dm, hcgto.py:272-281): L = X (C sqrt(w)) written zero
 *                           padded as d_orb (ld, rp) and its transpose d_orbt (rp, ld) -- the operand pair of dqc_grid_density_lr
 *                           and of dqc_fock_prep.  d_c: (north, r) orbitals with row stride ldc, d_w: r occupations >= 0.
 * nao <= dqc_fock_max_nao().  d_work: dqc_jk_work_doubles(nao) doubles, the same buffer through the calls of one build, in the
 * order prep, tile pass, finish (the buffer also carries the scratch matrices and the reduction ticket of dqc_fock_finish). */
int dqc_fock_max_nao(void);
int dqc_fock_factor(double *d_orb, double *d_orbt, const double *d_x, const double *d_c, int ldc, const double *d_w, int nao, int north,
                    int r, int ld, int rp, void *stream);
/* ao_orb2dm (hcgto.py:272-281) and its factor in one launch: d_dm (north, north) <- C diag(w) C^T, bitwise symmetric, and the padded
 * factor pair of dqc_fock_factor */
int dqc_fock_orb2dm(double *d_dm, double *d_orb, double *d_orbt, const double *d_x, const double *d_c, int ldc, const double *d_w, int nao,
                    int north, int r, int ld, int rp, void *stream);
int dqc_fock_prep(double *d_work, const double *d_dm, const double *d_x, const double *d_orb, int rp, int nao, int north, int with_k,
                  void *stream);
int dqc_jk_stream_prepared(const double *d_tiles, int nao, double *d_work, int with_k, void *stream);
/* the Kohn-Sham finish straight on the raw cross-block sums of dqc_grid_vxc_raw (no symmetrisation launch in between): vscale = the
 * scale that call returned (0: plain doubles) */
int dqc_fock_finish_vraw(double *d_fock, double *d_energies, double *d_work, const double *d_vxc_raw, int ldv, double vscale,
                         const double *d_core, const double *d_x, int nao, int north, void *stream);
int dqc_fock_finish(double *d_fock, double *d_energies, double *d_j_ao, double *d_work, const double *d_vxc_ao, int ldv,
                    const double *d_core, const double *d_x, int nao, int north, int with_k, void *stream);

/* Several density matrices in ONE pass over the tiles (unrestricted HF: J[D_u + D_d], K[2 D_u], K[2 D_d],
 * hcgto.py:238-241, hf.py:93-103; batched dm, base_hamilton.py:92-93).
 * d_dmJ (nj, nao, nao) -> d_J (nj, nao, nao);  d_dmK (nk, nao, nao) -> d_K (nk, nao, nao); either count may be 0.
 * Exchange matrices are processed two per pass (nk > 2: ceil(nk / 2) passes; all Coulomb ones ride on the first).
 * d_work: dqc_jk_multi_work_doubles(nao, nj, nk) doubles of scratch. */
size_t dqc_jk_multi_work_doubles(int nao, int nj, int nk);
int dqc_jk_from_tiles_multi(double *d_J, const double *d_dmJ, int nj, double *d_K, const double *d_dmK, int nk,
                            const double *d_tiles, int nao, double *d_work, void *stream);

/* ---- AO values on the grid -----------------------------------------------------------------
 * Replaces GTOval_sph / GTOval_ip_sph (dqc/hamilton/intor/gtoeval.py:196-239) with the
 * to_transpose=True layout of HamiltonCGTO.setup_grid (hcgto.py:168, :179).
 * deriv 0: d_out (ngrid, lda) = phi;  deriv 1: d_out (4, ngrid, lda) = phi, d/dx, d/dy, d/dz;
 * deriv 2: d_out (5, ngrid, lda) = the same plus the laplacian (GTOval_lapl_sph, hcgto.py:183-186).
 * deriv 3: d_out (10, ngrid, lda) = phi, gradient, then d2/dxx, dxy, dxz, dyy, dyz, dzz (GTOval_sph_deriv2 order without
 *          the duplicates; used by the GGA nuclear gradient).
 * d_coords: (ngrid, 3).  lda = dqc_ao_stride(nao); d_out holds dqc_ao_doubles(ncomp, ngrid, nao) doubles; padding columns and
 * the slack are written as zero. */
int dqc_eval_gto(int deriv, double *d_out, const double *d_coords, int ngrid, const int *atm,
                 int natm, const int *bas, int nbas, const double *env, int nenv, void *stream);

/* ---- density on the grid  (HamiltonCGTO._dm2densinfo, hcgto.py:371-443) ---------------------
 * d_ao: (ncomp, ngrid, lda) from dqc_eval_gto, ncomp = 1 (LDA) or 4 (GGA).  d_dm: (ld, ld)
 * symmetric AO density (zero padded).  d_rho: (ngrid).  d_grho: (3, ngrid) or NULL. */
int dqc_grid_density(double *d_rho, double *d_grho, const double *d_ao, int ncomp, int ngrid,
                     int nao, const double *d_dm, void *stream);


/* ---- exchange-correlation functional  (pylibxc LibXCFunctional.compute, unpolarised) --------
 * Replaces dqc/xc/libxc.py:40-85 + libxc_wrapper.py:380-413 for sums of libxc functionals
 * (dqc/xc/base_xc.py:197-268).  ids: HOST array of nterm functional ids (DQC_XC_*), coefs:
 * HOST array of nterm weights.  Outputs (any may be NULL): d_edens (n) energy per unit volume
 * (= zk*rho), d_vrho (n), d_vgrad (3,n) = 2*vsigma*grad rho  (libxc.py:239). */
#define DQC_XC_LDA_X 1
#define DQC_XC_LDA_C_VWN 7   /* VWN5 */
#define DQC_XC_LDA_C_PZ 9    /* Perdew-Zunger 81 */
#define DQC_XC_LDA_C_PW 12
#define DQC_XC_LDA_C_PW_MOD 13   /* PW92 with the full-precision constants (the LDA limit inside gga_c_pbe) */
#define DQC_XC_GGA_X_PBE 101
#define DQC_XC_GGA_X_PBE_R 102   /* revPBE (Zhang, Yang, PRL 80, 890 (1998)): kappa = 1.245 */
#define DQC_XC_GGA_X_B86 103     /* exchange functionals given by an enhancement factor (csrc/xc_funcs.hpp: x_enhancement) */
#define DQC_XC_GGA_X_B88 106
#define DQC_XC_GGA_X_G96 107
#define DQC_XC_GGA_X_PW86 108
#define DQC_XC_GGA_X_PW91 109
#define DQC_XC_GGA_X_OPTX 110
#define DQC_XC_GGA_X_WC 118
#define DQC_XC_GGA_X_PBE_SOL 116 /* PBEsol (Perdew et al., PRL 100, 136406 (2008)): mu = 10/81 */
#define DQC_XC_GGA_X_RPBE 117    /* RPBE (Hammer, Hansen, Norskov, PRB 59, 7413 (1999)): F = 1 + kappa (1 - exp(-mu s^2 / kappa)) */
#define DQC_XC_GGA_C_PBE 130
#define DQC_XC_GGA_C_LYP 131
#define DQC_XC_GGA_C_P86 132     /* Perdew 86 on PZ81 */
#define DQC_XC_GGA_C_PBE_SOL 133 /* PBEsol correlation: beta = 0.046 */
#define DQC_XC_MGGA_X_TPSS 202
#define DQC_XC_MGGA_C_TPSS 231
#define DQC_XC_MGGA_X_SCAN 263
#define DQC_XC_MGGA_C_SCAN 267
int dqc_xc_eval(double *d_edens, double *d_vrho, double *d_vgrad, const double *d_rho,
                const double *d_grho, int n, const int *ids, const double *coefs, int nterm,
                void *stream);

/* spin-polarised variant (polarised branches of dqc/xc/libxc.py:124-242): inputs rho_u, rho_d (n) and their
 * gradients (3,n) or NULL; outputs edens (n) = zk*(rho_u+rho_d), vrho_{u,d} (n), and
 * vgrad_u = 2 vsigma_uu grad_u + vsigma_ud grad_d, vgrad_d = 2 vsigma_dd grad_d + vsigma_ud grad_u  (libxc.py:205-215) */
/* the same pass with the grid quadrature of the energy density fused in (E_xc = sum_g w_g e_g, hcgto.py:320-328):
 * wavefront reductions, one partial per block, summed in a fixed order (no atomics: bit-reproducible).  d_exc:
 * DQC_XC_QUAD_DOUBLES doubles, the result in d_exc[0], the rest scratch; d_edens may be NULL. */
#define DQC_XC_QUAD_DOUBLES 1025
int dqc_xc_eval_quad(double *d_exc, double *d_edens, double *d_vrho, double *d_vgrad, const double *d_rho,
                     const double *d_grho, const double *d_w, int n, const int *ids, const double *coefs, int nterm,
                     void *stream);
int dqc_xc_eval_pol(double *d_edens, double *d_vrho_u, double *d_vrho_d, double *d_vgrad_u, double *d_vgrad_d,
                    const double *d_rho_u, const double *d_rho_d, const double *d_grho_u, const double *d_grho_d,
                    int n, const int *ids, const double *coefs, int nterm, void *stream);

/* meta-GGA variant (CalcMGGALibXCUnpol, dqc/xc/libxc_wrapper.py; inputs rho, grad rho, tau -- the supported
 * functionals do not depend on the laplacian, so vlapl = 0): adds d_vtau (n).  Terms may mix LDA/GGA ids with
 * DQC_XC_MGGA_X_SCAN, DQC_XC_MGGA_X_TPSS, DQC_XC_MGGA_C_SCAN and DQC_XC_MGGA_C_TPSS. */
int dqc_xc_eval_mgga(double *d_edens, double *d_vrho, double *d_vgrad, double *d_vtau, const double *d_rho,
                     const double *d_grho, const double *d_tau, int n, const int *ids, const double *coefs,
                     int nterm, void *stream);

/* spin-polarised meta-GGA CORRELATION terms (CalcMGGALibXCPol, libxc_wrapper.py:221-378; DQC_XC_MGGA_C_SCAN): they depend
 * on rho_u, rho_d, |grad(rho_u + rho_d)|^2 and tau_u + tau_d only, so vsigma = (v, 2 v, v) for (uu, ud, dd) and both spins
 * share  d_vgrad (3,n) = 2 v grad(rho_u + rho_d)  (= 2 vsigma_ss grad_s + vsigma_ud grad_s', libxc.py:205-215) and
 * d_vtau (n).  (Polarised meta-GGA EXCHANGE goes through dqc_xc_eval_mgga by the spin-scaling relation.) */
int dqc_xc_eval_mgga_pol(double *d_edens, double *d_vrho_u, double *d_vrho_d, double *d_vgrad, double *d_vtau,
                         const double *d_rho_u, const double *d_rho_d, const double *d_grho_u, const double *d_grho_d,
                         const double *d_tau_u, const double *d_tau_d, int n, const int *ids, const double *coefs, int nterm,
                         void *stream);

/* the same with ONE GRADIENT POTENTIAL PER SPIN: DQC_XC_MGGA_C_SCAN and DQC_XC_MGGA_C_TPSS (TPSS correlation depends on
 * sigma_uu, sigma_ud, sigma_dd separately: v_grad,u = 2 v_uu grad n_u + v_ud grad n_d, dqc/xc/libxc.py:205-215); d_vgrad_u,
 * d_vgrad_d (3, n) each, d_vtau (n) shared. */
int dqc_xc_eval_mgga_pol2(double *d_edens, double *d_vrho_u, double *d_vrho_d, double *d_vgrad_u, double *d_vgrad_d,
                          double *d_vtau, const double *d_rho_u, const double *d_rho_d, const double *d_grho_u,
                          const double *d_grho_d, const double *d_tau_u, const double *d_tau_d, int n, const int *ids,
                          const double *coefs, int nterm, void *stream);

/* ---- density-fitting integrals  (DFMol.build, dqc/df/dfmol.py:24-58) --------------------------
 * intor.coul2c(auxbw) = int2c2e_sph and intor.coul3c(basisw, basisw, auxbw) = int3c2e_sph
 * (dqc/hamilton/intor/molintor.py:36-72, 121-130).  atm/bas/env are the CONCATENATED host tables of
 * LibcintWrapper.concatenate (dqc/hamilton/intor/lcintwrap.py:299-370): orbital shells [sh0, sh1), auxiliary
 * shells [k0, k1) -- the `shell_idxs` of the two sub-wrappers.  Outputs are row-major device arrays:
 * d_j3c (nao, nao, naux) and d_j2c (naux, naux).  Shells up to g. */
int dqc_int3c2e(double *d_j3c, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv,
                int sh0, int sh1, int k0, int k1, void *stream);
int dqc_int2c2e(double *d_j2c, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv,
                int k0, int k1, void *stream);

/* density-fitted Coulomb matrix (DFMol.get_elrep, dfmol.py:60-79), AO basis:
 *   t_k = sum_ij D_ij (ij|k),  c = inv_j2c t,  J_ij = sum_k (ij|k) c_k.
 * d_j (nao, nao) is overwritten; d_work holds 2 * naux doubles.  Only enqueues on `stream`. */
int dqc_df_coulomb(double *d_j, const double *d_j3c, const double *d_inv_j2c, const double *d_dm_ao, int nao,
                   int naux, double *d_work, void *stream);

/* ---- nuclear gradients of the SCF energy   ------------------------------------
 * The reference differentiates through its "ip" derivative integrals (molintor.py:463-500) and the implicit-function
 * backward of the SCF fixed point (scf_qccalc.py:63-67, 109-113); at convergence that is
 *   dE/dR_A = sum D dh - sum W dS + (2e derivative term) + dE_xc + dE_nn,
 * and these two entry points return the integral-derivative terms contracted with the densities (no derivative
 * tensor is stored).  Densities are in the CARTESIAN AO basis: D_cart = T^T D_ao T with T = dqc_cart2sph_matrix
 * (HOST array (nao, ncart), ncart = dqc_ncart).  Both ADD into d_grad (natm, 3) and synchronise the stream.
 *   dqc_int1e_grad: 2 sum_{a in A} sum_b [D_ab (d_A a|T+V|b) - W_ab (d_A a|b)]  + Hellmann-Feynman term of every nucleus
 *   dqc_eri_grad  : sum_{a in A} sum_bcd (d_A a b|c d) [2 jscale D_ab D_cd - kscale D_ac D_bd]
 *                   (RHF: (1, 1); RKS: (1, 0); UHF: (1, 0) with the total density plus (0, 2) with each spin density)
 * Shells up to g (companions up to h). */
int dqc_ncart(const int *bas, int nbas);
int dqc_cart2sph_matrix(double *h_out, const int *bas, int nbas);
int dqc_int1e_grad(double *d_grad, const double *d_dcart, const double *d_wcart, const int *atm, int natm,
                   const int *bas, int nbas, const double *env, int nenv, const double *zs, void *stream);
int dqc_eri_grad(double *d_grad, const double *d_dcart, double jscale, double kscale, const int *atm, int natm, const int *bas,
                 int nbas, const double *env, int nenv, void *stream);
/* gradient of the density-fitted Coulomb energy E_J = 1/2 t^T M^-1 t (dfmol.py:60-79 differentiated):
 *   d_grad += sum D_ij c_k d(ij|k) - 1/2 sum c_k c_l d(k|l),  c = M^-1 t,
 * over the concatenated tables of dqc_int3c2e; d_dcart (ncart, ncart) / d_ccart (ncart): density matrix and fit
 * coefficients in the Cartesian basis of ALL shells of the table (T^T . T with T = dqc_cart2sph_matrix of the whole table;
 * zero outside the orbital block / the auxiliary segment).  d_grad has one row per atom OF THE TABLE (the concatenated
 * table lists the molecule's atoms twice: the caller folds the two halves).  Shells up to g. */
int dqc_df_grad(double *d_grad, const double *d_dcart, const double *d_ccart, const int *atm, int natm, const int *bas,
                int nbas, const double *env, int nenv, int sh0, int sh1, int k0, int k1, void *stream);

/* ---- Becke partition weights of the multi-centre grid  (dqc/grid/multiatoms_scheme.py:9-67, becke_grid.py:18-60) ----------
 * d_xyz (ngrid, 3): the atoms' grids one after the other, atom a owning points [d_atom_off[a], d_atom_off[a + 1]) (natm + 1
 * ints on the device); d_pos (natm, 3) nuclei; d_inv_rij, d_aij (natm, natm): 1 / |R_i - R_j| (finite on the diagonal) and the
 * atomic-size adjustment a_ij = clamp(u / (u^2 - 1), +-0.45), u = (rad_j - rad_i) / (rad_j + rad_i); cut: cell functions with
 * mu >= cut are dropped (the reference's 0.74).  d_w (ngrid) <- P_own / sum_j P_j.  Enqueues only. */
int dqc_becke_weights(double *d_w, const double *d_xyz, const int *d_atom_off, const double *d_pos,
                      const double *d_inv_rij, const double *d_aij, int natm, int ngrid, double cut, void *stream);
/* Backward of dqc_becke_weights (the grid-response term of an XC nuclear gradient; the reference differentiates its torch expression
 * dqc/grid/multiatoms_scheme.py:9-67 by autograd): d_cw (ngrid) = dL/dw; d_gpos (natm, 3) += sum_g cw dw_g/dR (explicit dependence
 * on the nuclei), d_gxyz (ngrid, 3) = cw dw_g/dr_g; d_scratch natm * ngrid doubles.  Enqueues only. */
int dqc_becke_weights_grad(double *d_gpos, double *d_gxyz, double *d_scratch, const double *d_cw, const double *d_xyz,
                           const int *d_atom_off, const double *d_pos, const double *d_inv_rij, const double *d_aij,
                           int natm, int ngrid, double cut, void *stream);

/* ---- occupied-space projector without an eigensolver  (the `diagonalize` + `ao_orb2dm` step, hf.py:105-113, 227-247) --
 * Trace-correcting purification X <- X^2 | 2X - X^2 (by the sign of tr X - nocc), one fused fp64-MFMA launch per
 * iteration, frozen once max|X^2 - X| < tol; no host decision (hipGraph-capturable).  d_x (ld, ld): X0 = (emax I - F) /
 * (emax - emin) zero padded to ld (multiple of 16) on entry, the projector on return; d_tmp (ld, ld) scratch;
 * d_state: 2 * (iters + 2) doubles, on return trace[k] = tr X_k and (from offset iters + 2) idem[k] = max|X_k^2 - X_k|. */
int dqc_purify_tc2(double *d_x, double *d_tmp, int ld, double nocc, int iters, double tol, double *d_state,
                   void *stream);
/* the same sequence as ONE persistent launch whose worker blocks share one XCD (one L2: no write-back between iterations);
 * ld <= 256; d_ctl: 4 unsigned ints of scratch, d_ctl[2] != 0 afterwards = the kernel gave up (see csrc/purify.hip) and d_x is not
 * a projector -- the caller's idempotency check catches it like a purification that did not converge. */
int dqc_purify_tc2_persist(double *d_x, double *d_tmp, int ld, double nocc, int iters, double tol, double *d_state,
                           unsigned *d_ctl, void *stream);
/* F -> P in ONE launch: Gershgorin bounds, X0, the TC2 iterations, two McWeeny steps, symmetrisation and the error
 * d_err[0] = max |P^2 - P| + |tr P - nocc| (large when the purification did not converge or the kernel gave up).  d_fock, d_p: (n, n)
 * contiguous, n <= 256; d_work: dqc_projector_work_doubles(n, iters) doubles.  Replaces `diagonalize` + the projector part of
 * `ao_orb2dm` (dqc/qccalc/hf.py:105-113, 227-247) for uniform occupations in an orthonormal basis. */
size_t dqc_projector_work_doubles(int n, int iters);
int dqc_projector_tc2(double *d_p, double *d_err, const double *d_fock, int n, double nocc, int iters, double tol, double *d_work,
                      void *stream);

/* Orthonormal basis of the range of a projector (the orbitals `ao_orb2dm` wants, hcgto.py:272-281, without an eigensolver):
 * d_y (n, r) = P . Omega with full column rank, d_g (r, r) = Y^T Y  ->  d_q (n, r) = Y C^-T with G = C C^T, so Q^T Q = 1 and
 * Q Q^T = P.  One launch (Cholesky in LDS + row-wise forward substitution); r <= 132.  Enqueues only. */
int dqc_orth_factor(double *d_q, const double *d_y, const double *d_g, int n, int r, void *stream);

/* ---- the same two steps for a BATCH of molecules iterated in lockstep (the reference runs one SCF loop per
 * molecule, scf_qccalc.py:109-113; here nmol same-size problems share every launch, blockIdx.y = molecule) ----
 * d_x, d_tmp (nmol, ld, ld); d_state (nmol, 2 * (iters + 2)); every matrix freezes on its own. */
int dqc_purify_tc2_batched(double *d_x, double *d_tmp, int ld, int nmol, double nocc, int iters, double tol,
                           double *d_state, void *stream);
/* d_y (nmol, n, r), d_g (nmol, r, r) -> d_q (nmol, n, r) */
int dqc_orth_factor_batched(double *d_q, const double *d_y, const double *d_g, int n, int r, int nmol, void *stream);
/* Pulay (DIIS) coefficients of nmol independent problems on the device: d_gram (nmol, nhist, nhist) scalar products of the stored
 * error vectors [F, D] (first m slots valid), d_c (nmol, nhist) <- c with sum c = 1 minimising |sum c_i e_i| -- the
 * minimum-norm least-squares solution of the bordered system (Gram block normalised to a unit largest diagonal), as
 * numpy.linalg.lstsq gives it in the one-molecule driver.  nhist <= 16.  Replaces the fixed-point mixer of
 * scf_qccalc.py:109-113 (xitorch Broyden) -- any convergent mixer has the same fixed point. */
int dqc_diis_solve(double *d_c, const double *d_gram, int nmol, int nhist, int m, void *stream);
/* the same with the number of valid slots min(*d_count, nhist) read from DEVICE memory: the step counter of an SCF loop that
 * replays as one hipGraph per iteration (dqc_amd/devscf.py) */
int dqc_diis_solve_dev(double *d_c, const double *d_gram, int nmol, int nhist, const long long *d_count, void *stream);

/* ---- Vxc matrix  (HamiltonCGTO._get_vxc_from_potinfo, hcgto.py:445-495) ----------------------
 * d_vmat (ld, ld) <- sym( sum_g w_g phi_ga [ vrho_g phi_gb + sum_d 2 vgrad_dg d_d phi_gb ] ),
 * AO basis.  d_vgrad may be NULL (LDA; then ncomp may be 1).  The matrix is overwritten. */
int dqc_grid_vxc(double *d_vmat, const double *d_ao, int ncomp, int ngrid, int nao,
                 const double *d_w, const double *d_vrho, const double *d_vgrad, void *stream);

/* density from the orbital factor: every SCF density matrix is D = C_occ diag(n) C_occ^T
 * (HamiltonCGTO.ao_orb2dm, hcgto.py:272-281), i.e. D = L L^T with L = C_occ sqrt(n).  Same outputs as
 * dqc_grid_density(D) at 2 r / nao of its GEMM flops.  d_orb (ld, norb_pad) = L in the AO basis, zero padded,
 * d_orbt (norb_pad, ld) its transpose; norb_pad = dqc_padded_norb(r) (0: r too wide, use the dense entry point). */
int dqc_padded_norb(int norb);
int dqc_grid_density_lr(double *d_rho, double *d_grho, const double *d_ao, int ncomp, int ngrid, int nao,
                        const double *d_orb, const double *d_orbt, int norb_pad, void *stream);
/* Both spin densities of an unrestricted Kohn-Sham build from ONE pass over the AO matrix (reference: SpinParam.apply_fcn over
 * HamiltonCGTO._dm2densinfo, dqc/hamilton/hcgto.py:260-269, 371-418 -- one pass per spin there).  d_orb (ld, 2 norb_pad_spin) =
 * [L_u | L_d] row-major with every channel zero padded to norb_pad_spin columns (16, 32, 48 or 64), d_orbt its transpose;
 * d_rho (2, ngrid), d_grho (2, 3, ngrid); GGA form only.  Enqueues only. */
int dqc_grid_density_lr_pol(double *d_rho, double *d_grho, const double *d_ao, int ncomp, int ngrid, int nao,
                            const double *d_orb, const double *d_orbt, int norb_pad_spin, void *stream);

/* Grid sums of the XC part of a nuclear gradient, GGA / meta-GGA, one pass over the deriv-3 AO array (reference: autograd through
 * eval_gradgto and _dm2densinfo, dqc/hamilton/intor/gtoeval.py:173-193, dqc/hamilton/hcgto.py:371-443; written out in
 * dqc_amd/gradient.py).  d_ao (10, ngrid, lda): value, 3 gradients, xx xy xz yy yz zz; d_b = Phi D and d_c0..2 = d_i Phi D, (ngrid, ldb)
 * each; d_u (3, ngrid) the gradient part of the potential, d_grho (3, ngrid), d_vtau (ngrid) or NULL.  Outputs: d_q (ngrid, 3) the
 * per-point term of the atoms that carry the points, d_perao (nao, 3) the per-basis-function term (zeroed here).  Enqueues only. */
int dqc_grid_xc_gradient_terms(double *d_q, double *d_perao, const double *d_ao, int ngrid, int nao, const double *d_b,
                               const double *d_c0, const double *d_c1, const double *d_c2, int ldb, const double *d_w,
                               const double *d_vrho, const double *d_u, const double *d_grho, const double *d_vtau, void *stream);

/* meta-GGA densities of D = L L^T from ONE pass over the four AO components: rho, grad rho (3, ngrid) and
 * tau = 1/2 sum_d sum_r (d_d Phi . L)_r^2 (hcgto.py:398-438 with D in factor form) -- four rank-r GEMMs, no row-dot epilogue;
 * replaces dqc_grid_density_lr + three value-only passes over the gradient components.  norb_pad <= 96. */
int dqc_grid_density_lr_tau(double *d_rho, double *d_grho, double *d_tau, const double *d_ao, int ncomp, int ngrid, int nao,
                            const double *d_orb, int norb_pad, void *stream);

/* "pair" forms used by the meta-GGA branches (hcgto.py:420-438, 473-489), both on single-component (ngrid, lda)
 * arrays:  d_out_g = sum_ij a_gi D_ij b_gj   and   d_vmat = sym( sum_g w_g v_g a_ga b_gb ). */
int dqc_grid_density_pair(double *d_out, const double *d_ao_a, const double *d_ao_b, int ngrid, int nao,
                          const double *d_dm, void *stream);
int dqc_grid_vxc_pair(double *d_vmat, const double *d_ao_a, const double *d_ao_b, int ngrid, int nao,
                      const double *d_w, const double *d_v, void *stream);

/* dqc_grid_vxc without its closing symmetrisation: d_vmat (ld, ld) <- the raw sums M; V = (M + M^T) / 2 on the first nao rows / columns
 * (fixed-point integers of scale *h_scale in deterministic mode, *h_scale = 0 otherwise) -- consumed by dqc_fock_finish_vraw */
int dqc_grid_vxc_raw(double *d_vmat, const double *d_ao, int ncomp, int ngrid, int nao, const double *d_w, const double *d_vrho,
                     const double *d_vgrad, double *h_scale, void *stream);

/* ---- the tile store spread over several GPUs (one molecule on N ranks: 8 x 288 GB hold the tiles of ~1200 basis functions) ----
 * Tiles follow each other in the order (IJ, KL <= IJ); a rank keeps the tiles [tile_begin, tile_end) in a buffer of
 * dqc_eri_tile_offset(nao, tile_end) - dqc_eri_tile_offset(nao, tile_begin) doubles.  dqc_eri_fill_tiles_part evaluates every
 * shell quartet (a quartet's integrals scatter over several tiles) and writes only the slice; dqc_jk_from_tiles_part streams the
 * slice and returns the PARTIAL d_J / d_K of its tiles -- the caller adds the ranks' parts (all_reduce).  tile_end = -1 in the
 * fill: to the end of the store.  The deterministic mode needs the whole store. */
long long dqc_eri_tile_offset(int nao, long long tile);
int dqc_eri_fill_tiles_part(double *d_tiles_part, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv,
                            long long tile_begin, long long tile_end, void *stream);
int dqc_jk_from_tiles_part(double *d_J, double *d_K, const double *d_tiles_part, const double *d_dm, int nao, double *d_work,
                           long long tile_begin, long long tile_end, void *stream);

/* ---- direct SCF: J and K straight from the shell quartets, nothing stored  ("direct / recompute above") ----
 * The stored tile form needs ~nao^4 bytes (2 GB at nao 208, 31 GB at 412, beyond one GPU near nao 740); this entry point
 * re-evaluates every unique shell quartet (the Rys kernel of dqc_eri_fill_tiles) and contracts it with the density on the fly:
 *   d_J (nao, nao) <- sum_kl (ij|kl) D_kl,   d_K (nao, nao) <- sum_jl (ij|kl) D_jl   (plain K, not -K/2; d_K may be NULL),
 * D = the symmetric part of d_dm (nao, nao), AO basis.  Same einsum strings as dqc_jk_from_tiles (hcgto.py:209, 234) at the
 * cost of one integral evaluation per call.  fp64 atomics (not covered by the deterministic mode).  Enqueues only. */
int dqc_jk_direct(double *d_J, double *d_K, const double *d_dm, const int *atm, int natm, const int *bas, int nbas,
                  const double *env, int nenv, void *stream);

/* ---- direct SCF with integral screening: a context that keeps the pair tables and the Schwarz bounds on the device -------
 * The reference passes prescreen = NULL to libcint (dqc/hamilton/intor/molintor.py:667-688) because it stores the whole
 * tensor; a direct build pays for every quartet in every iteration, so here the Cauchy-Schwarz bound |(ab|cd)| <= Q_ab Q_cd,
 * Q_ab = sqrt(max |(ab|ab)|), decides what is evaluated:
 *   dqc_direct_create  parses the tables, evaluates the diagonal quartets (one device pass, one host synchronisation), sorts the
 *                      shell pairs of every angular-momentum class by Q and leaves tables + bounds resident on the device;
 *   dqc_direct_jk      J and K of one density as dqc_jk_direct, skipping every shell quartet whose contribution is bounded by
 *                      tau:  Q_ab Q_cd max(4 |D_ab|, 4 |D_cd|, |D_ac|, |D_ad|, |D_bc|, |D_bd|) < tau  (block maxima of D; the
 *                      exchange blocks only when d_K is given).  Quartets are dropped at launch (prefix of the Q-sorted partner
 *                      list, with the maxima of |D| over the blocks -- by angular momentum -- that a class of quartets can
 *                      touch) and per quartet inside the kernel.  tau = 0: every quartet (bit-for-bit the
 *                      unscreened sums up to the order of the atomics).  One 8-byte device->host read per call (max |D| sets the
 *                      launch sizes).  J and K are linear in D: hand over density DIFFERENCES to let the screening bite as the
 *                      SCF converges (HamiltonMI355 does).  The error of an element of J / K is bounded by tau times the number
 *                      of skipped quartets that touch it; tau = 1e-13 keeps SCF energies within 1e-10 Ha of the unscreened path
 *                      on the systems of the test suite.
 *   dqc_direct_stats   unique shell quartets / quartets launched / max |D| of the last dqc_direct_jk call (host scalars);
 *   dqc_direct_bounds  host copies of the bounds and the shell pairs they belong to (tests). */
int dqc_direct_create(void **ctx, const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, void *stream);
int dqc_direct_jk(void *ctx, double *d_J, double *d_K, const double *d_dm, double tau, void *stream);
/* one molecule over several GPUs ("intra-molecule sharding"): rank `part` of `nparts` evaluates every nparts-th
 * block of shell quartets of the (identically screened) pass and gets the PARTIAL sums d_J, d_K; the caller adds the parts up
 * (all_reduce over RCCL: n^2 doubles per matrix -- latency-bound, 5 MB at nao 824).  Every rank holds its own context. */
int dqc_direct_jk_part(void *ctx, double *d_J, double *d_K, const double *d_dm, double tau, int part, int nparts, void *stream);
int dqc_direct_stats(void *ctx, long long *quartets_total, long long *quartets_launched, double *dmax);
int dqc_direct_npairs(void *ctx);
int dqc_direct_bounds(void *ctx, double *h_q, int *h_shells);
/* the same, plus (h_group, npairs, optional) the index of the GROUP pair every shell pair belongs to: s shells of one atom over the
 * same exponents are evaluated together and screened with the largest bound of their members (round 5) */
int dqc_direct_bounds_groups(void *ctx, double *h_q, int *h_shells, int *h_group);
int dqc_direct_destroy(void *ctx);

/* ---- deterministic mode ----------------------------------------------------------------------
 * The Fock build sums over blocks with fp64 atomics (J / K accumulators, split-K Vxc partials, the trace of the purification
 * iterate): the order of the additions, hence the last bits of the result, vary from run to run, while the reference's CPU
 * path is deterministic.  dqc_set_deterministic(1) switches those sums to fixed-point 64-bit INTEGER atomics (associative:
 * bit-identical results whatever the order; contributions are rounded to 2^-k with k chosen per call from a rigorous bound
 * of the sum: 2 max(ii|ii) sum|D_ij| for J / K, 2^14 for V).  Process-wide, returns the previous setting. */
int dqc_set_deterministic(int on);
int dqc_get_deterministic(void);

/* ---- integral-kernel selection (diagnostic) ---------------------------------------------------
 * Shell-quartet classes s ... f run compile-time instantiations of the Rys kernel; classes holding a g shell (an h companion
 * in the gradients) run the runtime-angular-momentum kernel (csrc/eri_generic.hpp).  dqc_set_generic_eri(1) sends EVERY class
 * of dqc_eri_fill_tiles / dqc_jk_direct / dqc_int3c2e / dqc_int2c2e / dqc_eri_grad / dqc_df_grad through the runtime kernel, so
 * that the two implementations can be compared on the same basis.  Process-wide (environment: DQC_ERI_GENERIC=1), returns the
 * previous setting. */
int dqc_set_generic_eri(int on);

/* ---- compute-unit partitions (round 6) --------------------------------------------------------
 * The restricted Kohn-Sham build has two independent halves per density matrix: the Coulomb stream over the ERI tiles (HBM-bound,
 * no matrix-core work; reference: the J einsum of hcgto.py:204-227) and the grid pass (density, functional, Vxc; matrix-core-bound;
 * hcgto.py:371-495).  The Vxc kernel holds 408 of the 512 VGPRs of every SIMD and 128 of the 160 KB of LDS of the CUs it runs on; the other hot kernels'
 * blocks do not fit beside it, so two launches on ordinary streams never share the chip.  dqc_stream_create_partition makes a HIP stream whose kernels run only on CUs [cu_begin, cu_end) of every XCD
 * (hipExtStreamCreateWithCUMask): the host layer gives the grid pass most of each XCD and the Coulomb streams of OTHER molecules
 * the rest, so the tile stream rides in the HBM bandwidth the matrix-core-bound kernels leave.  Kernels that size their launch by
 * the CU count (one block per CU) ask dqc_stream_cus(stream).  priority is reserved (0). */
int dqc_device_cu_count(void);
int dqc_stream_create_partition(void **stream_out, int cu_begin, int cu_end, int priority);
int dqc_stream_destroy(void *stream);
int dqc_stream_cus(void *stream);
/* Cap on the compute units the one-block-per-CU Vxc kernels occupy (0: none; environment DQC_VXC_CUS).  Their blocks hold most of the VGPRs and LDS
 * of a CU for the whole launch; a cap below the CU count leaves the rest of the chip to what other streams have queued (the
 * HBM-bound Coulomb and density passes of other molecules of a batch).  Process-wide, returns the previous setting. */
int dqc_set_vxc_cus(int ncu);

/* ---- micro-benchmarks used by bench.py to price the roofline on the box it runs on ---------- */
int dqc_probe_stream_read(const double *d_buf, size_t n, double *d_out, void *stream);
/* fp64 MFMA (16x16x4) issue-rate probe: 2048 waves x 8 accumulators x iters MFMAs; d_out: 131072 doubles */
int dqc_probe_mfma_f64(double *d_out, int iters, void *stream);

#ifdef __cplusplus
}
#endif
#endif
