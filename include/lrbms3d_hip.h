/*
 * lrbms3d_hip.h -- C ABI of the 3D / P2 hot path in liblrbms_hip.so (BASELINE.json config 5: 3D diffusion, 8x8x8
 * subdomains, SWIPDG p = 2, local basis dim 30).
 *
 * The reference binds the 2D / P1 operators only (python/dune/pylrbms/discretize_elliptic_block_swipdg.py:22-23; :195 uses
 * x[0], x[1]), so these entry points have no reference call site of their own: each one is the 3D / P2 counterpart of the
 * 2D entry point named beside it (include/lrbms_hip.h, which cites the reference line it replaces), with the same rules:
 * caller-allocated device buffers, plain pointers and sizes, int return codes (LRBMS_OK / LRBMS_E_*), explicit hipStream_t.
 *
 * Geometry: Kuhn triangulation (6 tetrahedra per cube), every subdomain a translate of one template of kx x ky x kz cubes,
 * every element a translate of one of 6 reference tetrahedra.  All local integrals are contractions
 *     block[e][c] = sum_k sample[e][k] * TABLE[type(e)][k][c]
 * of coefficient samples (host-evaluated data functions at the quadrature points, as in 2D) with reference tables that the
 * host builds once (pylrbms_amd/grid3d.py) and the context keeps on the device.
 *
 *   S, S_ext     local / local + halo subdomains;  n_T elements, n = 10 n_T DG DoFs, n_rt RT0 DoFs per subdomain
 *   sides        0 = z-, 1 = y-, 2 = x-, 3 = x+, 4 = y+, 5 = z+ ; slots 0..6 = sides 0..2, self (3), sides 3..5
 *   ncf          faces per side, nbf = 6 ncf side faces ("sf" = side * ncf + pos);  nvs nodes per side,  nb boundary nodes
 *   Q, N, QN     affine components, local basis size, Q * N;  columns (q, j) of flux images are q-major
 */
#ifndef LRBMS3D_HIP_H
#define LRBMS3D_HIP_H

#include <stdint.h>

#include "lrbms_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrbms3_ctx lrbms3_ctx;

typedef struct {
  int32_t n_T, n_rt, ncf, nvs, n_nodes, nb, nbel, nsel, nbd;
  int32_t nA, nB, nC, nFs, nFf;                   /* points of the rules: system volume, product volume, estimator volume,
                                                     system face, flux face */
  int32_t o_fs, o_ff, o_c, lam_stride;            /* lambda_q record: volA | 4 x Fs | 4 x Ff | volC */
  int32_t hat_stride, f_stride;                   /* lambda_hat and f records: volB | volC;  lambda_bar record: volB */
  double volume, kmin;                            /* |T|; smallest eigenvalue of the symmetric part of kappa */
  const int32_t *elem_type;                       /* [n_T] 0..5 */
  const int32_t *up_face;                         /* [n_T][4] faces whose inner neighbour has the higher element index (-1 padded) */
  const int32_t *order;                           /* [n_T] traversal order of the element loops (a permutation; cache locality only) */
  const int32_t *nb_elem, *nb_out, *face_pos, *tsign, *elem_rt;   /* [n_T][4]: inner neighbour or -(1+side); the neighbour's
                                                     element across a side face; position on the side; RT0 orientation; RT0 DoF */
  const int32_t *rt_e0, *rt_f0, *rt_e1, *rt_f1;   /* [n_rt] the (<= 2) own elements of an RT0 face */
  const int32_t *side_elem, *side_face, *side_elem_out, *side_face_out;   /* [6][ncf] */
  const int32_t *dof_node;                        /* [n] Lagrange node of a DG DoF */
  const int32_t *node_ptr, *node_dofs;            /* CSR node -> own DoFs: [n_nodes + 1], [n] */
  const int32_t *node_mask, *node_count;          /* [n_nodes] sides a node lies on (bit a); size of its averaging patch */
  const int32_t *side_nodes;                      /* [6][nvs] node at position p of side a (-1 padded) */
  const int32_t *sn_ptr, *sn_dofs;                /* CSR (side, pos) -> the NEIGHBOUR's DoFs at that node: [6 nvs + 1], [...] */
  const int32_t *dof_bslot;                       /* [n] compact index (0 .. nbd-1) of a DoF on a boundary node, else -1 */
  const int32_t *bn_ptr, *bn_slots;               /* CSR boundary node -> compact indices of its own DoFs: [nb + 1], [nbd] */
  const int32_t *bnodes, *bnode_sides;            /* [nb] boundary nodes; [nb][3] their (side * nvs + pos) memberships, -1 padded */
  const int32_t *bel_elem, *bel_bnode;            /* [nbel] elements with a boundary node; [nbel][10] boundary-node index per DoF or -1 */
  const int32_t *sel_elem, *sel_sf;               /* [nsel] elements with a side face; [nsel][4] side-face index per face or -1 */
  const double *divc;                             /* [6][4] |f| / |T| */
  const double *TV, *TE, *TAA;                    /* [6][nA|nB|nC][100]  w |T| grad phi_i . kappa grad phi_j */
  const double *TFo, *TFn, *TFb;                  /* [6][4][nFs][100]    inner face (own, own) / (own, neighbour); Dirichlet face */
  const double *TPo, *TPn, *TPb;                  /* [6][4][nFs][100]    their penalty parts alone (local energy product) */
  const double *TC, *TCb;                         /* [6][4][nFf][10]     flux coefficients: inner / Dirichlet face */
  const double *TPH, *TM;                         /* [6][nB][10] w |T| phi_i;  [6][100] mass */
  const double *TB, *TAB;                         /* [6][nC][16] w |T| psi_f . kappa^-1 psi_g;  [6][nC][40] w |T| grad phi_i . psi_f */
  const double *WB, *WC;                          /* [nB], [nC] w |T| */
} lrbms3_mesh_desc;

int lrbms3_ctx_create(int device, lrbms3_ctx** out);
int lrbms3_ctx_destroy(lrbms3_ctx* ctx);
const char* lrbms3_last_error(lrbms3_ctx* ctx);

/* Launch policy of the library (no numerical convention).  The library reads no environment variable: these options are the
 * only switches; tests use them to reach every kernel path at small sizes.  2D: lrbms_ctx_set_option.
 *   LRBMS3_OPT_KSPLIT          0 (default): workgroups per (subdomain, operator) of the projection kernels chosen from the launch
 *                              size (1 from ~400 workgroups per kernel on, i.e. at config 5's 512 subdomains); 1 .. 8: forced
 *                              (1 = the in-kernel epilogues, > 1 = partial tiles + k3_pg_combine)
 *   LRBMS3_OPT_SERIAL          1: every kernel of the pass on the caller's stream (kernel statistics of a serial pass)
 *   LRBMS3_OPT_WAVES           0 (default): 4 / 8 waves per workgroup of the projection kernels; else that many
 *   LRBMS3_OPT_ESTIMATE_VALU   1: VALU form of the batched estimate (cross-check of the matrix-core form)
 *   LRBMS3_OPT_SOLVE_VALU      1: VALU form of the batched solver's panel matvec (cross-check)
 *   LRBMS3_OPT_FOM_COARSE      0: full-order solves without the coarse level whatever lrbms3_fom_coarse_space set */
#define LRBMS3_OPT_KSPLIT 1
#define LRBMS3_OPT_SERIAL 2
#define LRBMS3_OPT_WAVES 3
#define LRBMS3_OPT_ESTIMATE_VALU 4
#define LRBMS3_OPT_SOLVE_VALU 5
#define LRBMS3_OPT_FOM_COARSE 6
int lrbms3_ctx_set_option(lrbms3_ctx* ctx, int32_t option, int32_t value);

/* Template, tables, neighbour table nbr [S][7] (index into the S_ext ordering per slot, -1 = none, nbr[s][3] == s) and
 * phys [S_ext] (bit a set: side a lies on the physical boundary).   2D: lrbms_mesh_upload. */
int lrbms3_mesh_upload(lrbms3_ctx* ctx, const lrbms3_mesh_desc* desc, int32_t S, int32_t S_ext, const int32_t* nbr,
                       const int32_t* phys);

/* -- offline assembly ------------------------------------------------------------------------------------------------- */
/* SWIPDG system (sigma = 20 / 38, beta = 1/2).  lam [Q][S_ext][n_T][lam_stride];
 *   A_diag [Q][S][n_T][5][100]  block-ELL: 0 = (e, e), 1 + f = (e, inner neighbour across face f)
 *   A_cpl  [Q][S][6][ncf][100]  block (own side element, neighbour's element) per coupling face, zero on physical sides
 * 2D: lrbms_assemble_swipdg. */
int lrbms3_assemble_system(lrbms3_ctx* ctx, int32_t Q, const double* lam, double* A_diag, double* A_cpl, void* stream);
/* f_smp [S][n_T][f_stride], lhat [S][n_T][hat_stride] -> b [S][n], f2 [S], ceps [S], bdiv [S][n_T] = int_T f.  2D: lrbms_assemble_rhs. */
int lrbms3_assemble_rhs(lrbms3_ctx* ctx, const double* f_smp, const double* lhat, double* b, double* f2, double* ceps,
                        double* bdiv, void* stream);
/* lbar [S][n_T][nB] -> ebar [S][n_T][100] (E_ii element blocks);  Aaa [Q][Q][S][n_T][100], Aab [Q][S][n_T][10][4],
 * Bbb [S][n_T][4][4] (RT0 orientation signs folded in).  2D: lrbms_assemble_products. */
int lrbms3_assemble_products(lrbms3_ctx* ctx, int32_t Q, const double* lam, const double* lbar, const double* lhat, double* ebar,
                             double* Aaa, double* Aab, double* Bbb, void* stream);
/* Local energy product of every subdomain (reference: local_energy_dg_product_i, discretize_elliptic_block_swipdg.py:651-677, the
 * product the local bases are orthonormalised in, reductor.py:19-31): sum_q theta_bar_q [volume term of lambda_q + the PENALTY
 * terms of the faces inside the subdomain + the Dirichlet penalty, with the inside coefficient, on every face of the subdomain
 * boundary (coupling or physical: all-Dirichlet boundary info on the subdomain layer, :658)].  theta_bar [Q] host;
 * P_diag [S][n_T][5][100] block-ELL like A_diag (no coupling blocks: the product is local).  2D: P_diag of lrbms_assemble_products. */
int lrbms3_assemble_energy_product(lrbms3_ctx* ctx, int32_t Q, const double* theta_bar, const double* lam, double* P_diag,
                                   void* stream);

/* Y [S][n][M] = P_diag X (the local energy product applied to M vectors per subdomain): Gram-Schmidt of the reductor.
 * 2D: lrbms_blockell_apply. */
int lrbms3_energy_product_apply(lrbms3_ctx* ctx, int32_t M, const double* P_diag, const double* X, double* Y, void* stream);

/* Cf [Q][S_ext][n_T][4][10]: contribution of the element's own DoFs to the RT0 DoF of its face f (unsigned; the kernels
 * apply the orientation).  2D: lrbms_assemble_flux. */
int lrbms3_assemble_flux(lrbms3_ctx* ctx, int32_t Q, const double* lam, double* Cf, void* stream);

/* -- project + estimate-offline (the timed region) ----------------------------------------------------------------------- */
/* One pass over all local subdomains: Oswald interpolation error and RT0 flux reconstruction of the local bases, Galerkin
 * projection of the system and of the estimator operators, in the FACTORED layout (cf. lrbms_project_estimate_fused_factored):
 *   V [S_ext][n][N] (halo filled)
 *   B_sys [Q][S][7][N][N], rhs_red [S][N]
 *   G_nc [S][N][N], G_bb, G_rdd [S][QN][QN], G_ab [Q][S][N][QN], G_aa [Q][Q][S][N][N], r_fd [S][QN]      the [self, self] blocks
 *   Rb, Yb, Dp [S][nbf][QN], Xab [Q][S][nbf][N]     per side face: neighbour's flux image; rows of B R_self, vol div div R_self,
 *                                                   A_ab^T V at the face
 *   As [S][6][nvs][N], Cn [S][nb][N]                per side node: neighbour's share of the vertex average; -P^T E W_self
 * so that for coefficients u (own u_s, neighbours u_a), ur = (theta_q u)_q, zf = Rb ur_a (per side face), z = sum_a As u_a:
 *   nc  = u_s^T G_nc u_s + 2 z^T Cn u_s + sum_e z_e^T ebar_e z_e
 *   df  = ur^T G_bb ur + 2 zf^T Yb ur + sum_e zf_e^T Bbb_e zf_e + 2 sum_q theta_q (u_s^T G_ab_q ur + zf^T Xab_q u_s) + sum theta theta u_s^T G_aa u_s
 *   rdd = ur^T G_rdd ur + 2 zf^T Dp ur + sum_e |T| (div zf_e)^2,   rfd = r_fd^T ur + sum_e bdiv_e div zf_e
 * work >= lrbms3_work_size doubles (flux image R_self [S][n_rt][QN], vertex averages [S][n_nodes][N], rows of E W_self at
 * the boundary DoFs [S][nbd][N]). */
int64_t lrbms3_work_size(lrbms3_ctx* ctx, int32_t Q, int32_t N);
int lrbms3_project_estimate(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* A_diag, const double* A_cpl,
                            const double* b, const double* ebar, const double* Aaa, const double* Aab, const double* Bbb,
                            const double* bdiv, const double* Cf, double* work, double* B_sys, double* rhs_red, double* G_nc,
                            double* G_bb, double* G_rdd, double* G_ab, double* G_aa, double* r_fd, double* Rb, double* Yb,
                            double* Dp, double* Xab, double* As, double* Cn, void* stream);

/* The same pass in two halves for a sharded run that overlaps its halo exchange with compute: phase 1 reads the first S
 * (rank-local) slabs of V only -- everything but the neighbours' shares; phase 2 needs the halo slabs and writes Rb, As and the
 * coupling blocks of B_sys.  phase 0 == lrbms3_project_estimate; 1 followed by 2 is bit-identical to 0 (same buffers, same
 * work).  2D: lrbms_project_estimate_fused_phase. */
int lrbms3_project_estimate_phase(lrbms3_ctx* ctx, int32_t phase, int32_t Q, int32_t N, const double* V, const double* A_diag,
                                  const double* A_cpl, const double* b, const double* ebar, const double* Aaa, const double* Aab,
                                  const double* Bbb, const double* bdiv, const double* Cf, double* work, double* B_sys,
                                  double* rhs_red, double* G_nc, double* G_bb, double* G_rdd, double* G_ab, double* G_aa,
                                  double* r_fd, double* Rb, double* Yb, double* Dp, double* Xab, double* As, double* Cn, void* stream);

/* Per-kernel device timing of the pass, as lrbms_kernel_timing / lrbms_kernel_timing_read. */
int lrbms3_kernel_timing(lrbms3_ctx* ctx, int32_t enable);
int lrbms3_kernel_timing_read(lrbms3_ctx* ctx, char* names, int64_t names_cap, double* ms, int32_t cap, int32_t* count);

/* -- online ---------------------------------------------------------------------------------------------------------------- */
/* Local estimator terms from the factored operators.  theta [Q] host; u [S_ext][N]; eta_loc [3][S]: nc, r (scaled by
 * (1/pi^2) / c_eps h^2), df -- squared quantities.  2D: lrbms_reduced_estimate_factored. */
int lrbms3_reduced_estimate(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* u, const double* G_nc,
                            const double* G_bb, const double* G_rdd, const double* G_ab, const double* G_aa, const double* r_fd,
                            const double* Rb, const double* Yb, const double* Dp, const double* Xab, const double* As,
                            const double* Cn, const double* ebar, const double* Bbb, const double* bdiv, const double* f2,
                            const double* ceps, double hdiam, double* eta_loc, void* stream);

/* Throughput form: nmu parameters at once (passes of 8): theta [nmu][Q] host, u [S_ext][N][nmu] (parameter fastest, the layout
 * lrbms3_reduced_solve_batch returns), eta_loc [3][S][nmu].  Every operator and factor is read once per pass of 8.
 * 2D: lrbms_reduced_estimate_batch_factored. */
int lrbms3_reduced_estimate_batch(lrbms3_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* u,
                                  const double* G_nc, const double* G_bb, const double* G_rdd, const double* G_ab, const double* G_aa,
                                  const double* r_fd, const double* Rb, const double* Yb, const double* Dp, const double* Xab,
                                  const double* As, const double* Cn, const double* ebar, const double* Bbb, const double* bdiv,
                                  const double* f2, const double* ceps, double hdiam, double* eta_loc, void* stream);

/* (sum_q theta_q B_sys_q) u = rhs_red by block-Jacobi preconditioned CG on the 7-slot block-sparse reduced system (S_ext == S).
 * info[0] = iterations, info[1] = final relative residual (host, may be NULL).  2D: lrbms_reduced_solve. */
int64_t lrbms3_reduced_solve_work_size(lrbms3_ctx* ctx, int32_t N);
int lrbms3_reduced_solve(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, const double* rhs_red,
                         double* work, double* u, double rtol, int32_t max_iter, double* info, void* stream);

/* Throughput form of the reduced solve: nmu <= 64 parameters at once (N <= 32), in groups of <= 16.  theta [nmu][Q] host;
 * u [S][N][nmu] (parameter fastest).  Inside a group every projected block is read once per CG iteration for all its parameters
 * (panel matvec on the matrix cores), independent CG scalars per parameter; preconditioner: the inverse diagonal blocks at the
 * group-mean theta, or the prebuilt two-level one (lrbms3_reduced_precond_use).  The (<= 4) groups run on the caller's stream and
 * the library's three side streams, launches interleaved: their kernels are latency-bound and share the chip.  info[0] =
 * iterations (of the slowest group), info[1] = worst relative residual.  2D: lrbms_reduced_solve_batch. */
int64_t lrbms3_reduced_solve_batch_work_size(lrbms3_ctx* ctx, int32_t N, int32_t nmu);
int lrbms3_reduced_solve_batch(lrbms3_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* B_sys,
                               const double* rhs_red, double* work, double* u, double rtol, int32_t max_iter, double* info,
                               void* stream);

/* Coarse level of lrbms3_reduced_solve_batch's preconditioner, built once per reduced model (2D: lrbms_reduced_precond_build /
 * _use): the Galerkin problem on the FIRST local basis vector of every subdomain (the constant the reductor starts every basis
 * with, reference reductor.py:29-31), A0[s][t] = sum_q theta_q B_sys_q[s][slot of t][0][0], inverted densely (rocSOLVER) at the
 * reference parameter theta; added to the inverse diagonal blocks it halves the iteration count at 8^3 subdomains.  Any SPD
 * preconditioner is admissible, so one build serves every parameter.
 *   lrbms3_reduced_precond_build   theta [Q] host; work: lrbms3_reduced_precond_work_size doubles; pc: lrbms3_reduced_precond_size
 *                                  doubles, caller-owned, filled ([S][S] coarse inverse, then the inverse diagonal blocks
 *                                  [S][N][N] at the same theta); LRBMS_E_INVALID if the coarse matrix is not positive definite
 *   lrbms3_reduced_precond_use     subsequent lrbms3_reduced_solve_batch calls with basis size N use pc (it must stay alive);
 *                                  NULL: the inverse diagonal blocks alone */
int64_t lrbms3_reduced_precond_size(lrbms3_ctx* ctx, int32_t N);
int64_t lrbms3_reduced_precond_work_size(lrbms3_ctx* ctx, int32_t N);
int lrbms3_reduced_precond_build(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, double* work,
                                 double* pc, void* stream);
int lrbms3_reduced_precond_use(lrbms3_ctx* ctx, int32_t N, const double* pc);

/* Snapshot generation: A(mu) x = b on the never-assembled block operator (S_ext == S), CG with a two-level additive
 * preconditioner: the inverse 10 x 10 element blocks plus, if lrbms3_fom_coarse_space was called, the Galerkin coarse problem on
 * nc functions per subdomain (dense (nc S)^2 inverse by rocSOLVER per solve; the 2D solver's coarse space are the subdomain
 * indicator functions).  b, x [S][n]; work: lrbms3_fom_solve_work_size doubles; info[0] = iterations, info[1] = final relative
 * residual (host, may be NULL); LRBMS_E_NOT_CONVERGED above rtol after max_iter.  2D: lrbms_fom_solve
 * (DuneDiscretization._solve, discretize_elliptic_block_swipdg.py:219-225, has no 3D counterpart in the reference).
 *
 * lrbms3_fom_coarse_space: Phi [n][nc], the values of the nc <= 4 coarse functions at the local DoFs -- functions of the
 * subdomain-local coordinates, hence one table for all subdomains (the host mirror passes 1, x, y, z: P1 per subdomain);
 * nc = 0 switches the coarse level off.  Needs the mesh.  The dense coarse problem is capped at 8 192 unknowns: beyond it only
 * the first function is used, beyond 8 192 subdomains none. */
int lrbms3_fom_coarse_space(lrbms3_ctx* ctx, int32_t nc, const double* Phi);
/* keep != 0: the next lrbms3_fom_solve leaves its dense coarse inverse in the context and the following ones reuse it whatever
 * their parameter (any SPD preconditioner is admissible; saves the ~10 ms factorisation per snapshot at config 5) until keep = 0
 * frees it or the coarse space changes. */
int lrbms3_fom_precond_keep(lrbms3_ctx* ctx, int32_t keep);
int64_t lrbms3_fom_solve_work_size(lrbms3_ctx* ctx);
int lrbms3_fom_solve(lrbms3_ctx* ctx, int32_t Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                     double* work, double* x, double rtol, int32_t max_iter, double* info, void* stream);

/* y [S][n][M] = sum_q theta_q (A_diag_q x_s + sum_sides A_cpl_q x_neighbour), x [S_ext][n][M].  2D: lrbms_fom_apply. */
int lrbms3_fom_apply(lrbms3_ctx* ctx, int32_t Q, int32_t M, const double* theta, const double* A_diag, const double* A_cpl,
                     const double* x, double* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRBMS3D_HIP_H */
