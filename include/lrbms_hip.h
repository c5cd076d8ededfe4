/*
 * lrbms_hip.h -- C ABI of liblrbms_hip.so: the MI355X (gfx950) hot path of dune-community/pylrbms.
 *
 * Drop-in boundary.  In the reference the lower boundary of this path is a set of
 * pybind11 objects from dune-gdt / dune-xt (factories returning operators with .assemble()/.matrix(),
 * walkers, Matrix/Vector with the buffer protocol).  Here it is flat device arrays:
 *
 *   - the caller (Python/torch, or any host) allocates EVERY input/output buffer in device memory and passes
 *     raw pointers + sizes; the library never frees caller memory ("caller-allocated, callee fills", as the
 *     caller-supplied matrix `ll` at reference discretize_elliptic_block_swipdg.py:402-406);
 *   - library-owned state (the subdomain template, the neighbour table) lives in an opaque lrbms_ctx;
 *   - every entry point returns int (0 = LRBMS_OK, < 0 = LRBMS_E_*); no C++ exception crosses the ABI;
 *     lrbms_last_error(ctx) gives the message;
 *   - every launch goes to the hipStream_t passed as `stream` (void*; NULL = default stream); calls on one
 *     ctx are not re-entrant, different ctxs are independent.
 *
 * All floating point data is fp64, all index data int32.  Array shapes are C-contiguous, written [a][b][c].
 *   S      = number of subdomains owned by this ctx ("local")
 *   S_ext  = S + halo subdomains (neighbours owned by other ranks); local subdomains come first
 *   n_T, n = 3 n_T, n_rt, ncf : elements / DG DoFs / RT0 DoFs per subdomain, faces per subdomain side
 *   Q      = affine diffusion components, N = local reduced basis size (uniform), C = 5 Q N, W = 5 N
 *   slots  : neighbourhood slots in sorted order 0=S 1=W 2=self 3=E 4=N; sides 0=S 1=W 2=E 3=N
 *   NS=16  : samples per element: 7 volume points (Radon rule) then 3 Gauss points on each of the 3 faces,
 *            face f parametrised from local vertex f+1 to f+2
 */
#ifndef LRBMS_HIP_H
#define LRBMS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRBMS_OK 0
#define LRBMS_E_INVALID (-1)   /* bad argument / shape */
#define LRBMS_E_HIP (-2)       /* a HIP runtime call failed */
#define LRBMS_E_STATE (-3)     /* mesh not uploaded yet */
#define LRBMS_E_NOT_CONVERGED (-4)

#define LRBMS_NS 16

typedef struct lrbms_ctx lrbms_ctx;

/* Host-side description of the subdomain template (pylrbms_amd.grid.SubdomainTemplate) and of the
 * local neighbour table.  Replaces what the reference gets from make_cube_dd_subdomains_grid
 * (grid.py:18-30), make_block_dg_space / make_rt_space (block_swipdg.py:543-546) and
 * compute_pattern / compute_coupling_pattern (:548-568). */
typedef struct {
  int32_t kx, ky, n_T, n_rt, n_vertices, ncf;
  double hx, hy;
  double kappa[4];              /* constant 2x2 diffusion tensor, row-major */
  const int32_t* nb_elem;       /* [n_T][3]  >=0 inner neighbour element, <0: -(1+side) */
  const int32_t* nb_face;       /* [n_T][3]  local face index in the inner neighbour */
  const int32_t* nb_elem_out;   /* [n_T][3]  element in the neighbouring subdomain (side faces) */
  const int32_t* nb_face_out;   /* [n_T][3] */
  const int32_t* elem_side_pos; /* [n_T][3]  position of a side face along its side */
  const int32_t* elem_rt;       /* [n_T][3]  RT0 DoF of (element, face) */
  const int32_t* face_sign;     /* [n_T][3]  +1/-1 orientation (S/W sides -1, overridden on the domain boundary) */
  const int32_t* dof_vertex;    /* [n]       lattice vertex of a DG DoF */
  const int32_t* vdof_ptr;      /* [n_vertices+1] CSR vertex -> DoFs */
  const int32_t* vdof_idx;      /* [n] */
  const int32_t* rt_e0;         /* [n_rt] primary element of an RT0 face ... */
  const int32_t* rt_f0;         /* [n_rt] ... and its local face */
  const int32_t* rt_e1;         /* [n_rt] secondary element (own subdomain if inner, neighbour's if side, -1) */
  const int32_t* rt_f1;         /* [n_rt] */
  const int32_t* rt_side;       /* [n_rt] -1 inner, else side */
  const int32_t* side_elem;     /* [4][ncf] our element at position pos of a side */
  const int32_t* side_elem_out; /* [4][ncf] the neighbour's element there */
  const int32_t* side_count;    /* [4] */
  int32_t ntouch;               /* row length of touch_elem */
  const int32_t* touch_elem;    /* [4][ntouch] elements with a vertex on the side, -1 padded */
  const int32_t* touch_count;   /* [4] */
  const double* grad;           /* [n_T][3][2] grad phi_i */
  const double* area;           /* [n_T] */
  const double* normal;         /* [n_T][3][2] outward unit normals */
  const double* face_len;       /* [n_T][3] */
  const double* points;         /* [n_T][3][2] vertex coordinates relative to the subdomain origin */
} lrbms_mesh_desc;

/* -- context ------------------------------------------------------------------------------------------- */
int lrbms_ctx_create(int device, lrbms_ctx** out);
int lrbms_ctx_destroy(lrbms_ctx* ctx);
const char* lrbms_last_error(lrbms_ctx* ctx);
const char* lrbms_version(void);

/* Upload template + neighbour table.  nbr [S][5]: index into the S_ext ordering per slot, -1 = none,
 * nbr[s][2] == s.  (K0; reference grid.py:8-42, block_swipdg.py:66-70,78,393,421.) */
int lrbms_mesh_upload(lrbms_ctx* ctx, const lrbms_mesh_desc* desc, int32_t S, int32_t S_ext, const int32_t* nbr);

/* -- quadrature ------------------------------------------------------------------------------------------ */
/* One rule per integrand of the path.  dune-gdt integrates every local integrand with the rule of order
 * (integrand order + over_integrate); the over_integrate arguments are in the reference tree
 * (discretize_elliptic_block_swipdg.py:247,267,327,347,369,405,519,655,660,782).  The host chooses the rules
 * (pylrbms_amd/quadrature.py: the reference's orders by default), samples the data functions at their points and hands
 * both over; the kernels do the arithmetic.  Triangle rules in barycentric coordinates, weights summing to 1; edge rules
 * on [0, 1] from local vertex f + 1 to f + 2 of the element, ascending and symmetric about 1/2 (the neighbour across the
 * face sees the points in reverse order).  The o_ and _stride fields lay out the sample records per element:
 *   lambda_q      [Q][S_ext][n_T][lam_stride]   system_volume pts | 3 faces x nfs system face pts (face f: inner rule if the
 *                                               template has a neighbour element across it, else the coupling rule, which also
 *                                               serves Dirichlet boundary faces) | 3 x energy_face pts | 3 x flux_face pts |
 *                                               energy_volume pts
 *   lambda_q (df) [Q][S][n_T][lamdf_stride]     df_aa pts | df_ab pts
 *   lambda_hat    [S][n_T][lhat_stride]         df_aa | df_ab | df_bb | ceps pts
 *   f             [S][n_T][f_stride]            rhs pts | f2 pts
 *   lambda_bar    [S][n_T][lbar_stride]         elliptic_bar pts */
#define LRBMS_MAXQV 16
#define LRBMS_MAXQF 4
typedef struct {
  int32_t n, pad;
  double w[LRBMS_MAXQV];
  double b[LRBMS_MAXQV][3];
} lrbms_tri_rule;
typedef struct {
  int32_t n, pad;
  double w[LRBMS_MAXQF];
  double t[LRBMS_MAXQF];
} lrbms_edge_rule;
typedef struct {
  lrbms_tri_rule system_volume, energy_volume, elliptic_bar, rhs, f2, ceps, df_aa, df_ab, df_bb;
  lrbms_edge_rule system_inner_face, system_coupling_face, energy_face, flux_face;
  int32_t nfs, o_sysv, o_sysf, o_enf, o_flf, o_env, lam_stride;
  int32_t o_aa, o_ab, lamdf_stride;
  int32_t o_haa, o_hab, o_hbb, o_hceps, lhat_stride;
  int32_t o_frhs, o_ff2, f_stride, lbar_stride, pad_;
} lrbms_quadrature;
/* Copied into the context (host pointer); must be called before any lrbms_assemble_* function. */
int lrbms_set_quadrature(lrbms_ctx* ctx, const lrbms_quadrature* quad);

/* -- offline assembly (not in the timed project+estimate region) ---------------------------------------- */
/* K1-K3: SWIPDG system per affine component; replaces make_elliptic_swipdg_affine_factor_matrix_operator
 * + the coupling / boundary assemblers (block_swipdg.py:399-437) and the block axpy of :475-497.
 *   lam    [Q][S_ext][n_T][lam_stride]  samples of lambda_q (record layout above)
 *   A_diag [Q][S][n_T][4][9]    block-ELL: block 0 = (e,e), block 1+f = (e, inner neighbour across face f)
 *   A_cpl  [Q][S][4][ncf][9]    block (ii, neighbour at side) per coupling face: rows side_elem, cols side_elem_out */
int lrbms_assemble_swipdg(lrbms_ctx* ctx, int32_t Q, const double* lam, double* A_diag, double* A_cpl, void* stream);

/* K5 + scalars: replaces make_l2_volume_vector_functional (block_swipdg.py:518-521), apply_l2_product,
 * min_diffusion_eigenvalue (:776-783).
 *   f_smp [S][n_T][f_stride], lhat [S][n_T][lhat_stride]  ->  b [S][n], f2 [S] = ||f||^2_{L2(Omega_ii)}, ceps [S] */
int lrbms_assemble_rhs(lrbms_ctx* ctx, const double* f_smp, const double* lhat, double* b, double* f2, double* ceps,
                       void* stream);

/* K6 + K9: local products and estimator operators (block_swipdg.py:319-378, :644-691, :722-729).
 *   theta_bar [Q]                          host pointer, theta_q(mu_bar)
 *   lam [Q][S_ext][n_T][lam_stride], lam_df [Q][S][n_T][lamdf_stride], lbar [S][n_T][lbar_stride], lhat [S][n_T][lhat_stride]
 *   P_diag [S][n_T][4][9]                  energy product (elliptic + penalty at mu_bar), block-ELL
 *   ebar   [S][n_T]                        int_T lambda_bar         (E_ii = ebar * stiffness template)
 *   caa    [Q][Q][S][n_T]                  int_T lambda_q lambda_q' / lambda_hat
 *   Aab    [Q][S][n_T][3][3]               df_ab element blocks [i][f]
 *   Bbb    [S][n_T][3][3]                  df_bb element blocks [f][g] */
int lrbms_assemble_products(lrbms_ctx* ctx, int32_t Q, const double* theta_bar, const double* lam, const double* lam_df,
                            const double* lbar, const double* lhat, double* P_diag, double* ebar, double* caa, double* Aab,
                            double* Bbb, void* stream);

/* K8 (assembly half): coefficient rows of the RT0 diffusive-flux reconstruction
 * (RS2017_apply_diffusive_flux_reconstruction_in_neighborhood, block_swipdg.py:165-169).
 *   F [Q][S][n_rt][6]: r_e = sum_i F[..][i] v(e0, i) + F[..][3+i] v(e1, i) */
int lrbms_assemble_flux(lrbms_ctx* ctx, int32_t Q, const double* lam, double* F, void* stream);

/* -- project + estimate-offline (the timed region: K7 + K8 + P1 + P2) ------------------------------------ */
/* K7: OswaldInterpolationErrorOperator.apply (block_swipdg.py:83-122), target-major.
 *   V [S_ext][n][N]  ->  Wt [S][n][5 N], column (slot, j) = component ii of oi_kk.apply(V_kk)[j] */
int lrbms_oswald_apply(lrbms_ctx* ctx, int32_t N, const double* V, double* Wt, void* stream);

/* K8: FluxReconstructionOperator.apply (block_swipdg.py:148-176), target-major.
 *   V [S_ext][n][N], F  ->  Rt [S][n_rt][5 Q N], column (slot, q, j) */
int lrbms_flux_reconstruct(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* F, const double* V, double* Rt,
                           void* stream);

/* P1: Galerkin projection of the system (project_system over d.operator / d.rhs / products; reductor.py:70).
 *   work   >= Q S n N doubles of scratch
 *   B_sys  [Q][S][5][N][N]   block (ii, neighbour in slot) of V^T A_q V, zero where no neighbour
 *   rhs_red[S][N], E_red [S][N][N] (energy product), M_red [S][N][N] (L2) */
int lrbms_project_system(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* A_diag,
                         const double* A_cpl, const double* P_diag, const double* b, double* work, double* B_sys,
                         double* rhs_red, double* E_red, double* M_red, void* stream);

/* P2: projected estimator operators nc_i, r_fd_i, r_dd_i, df_aa_i, df_bb_i, df_ab_i
 * (block_swipdg.py:733-770 projected at reductor.py:70).
 *   work     >= lrbms_estimator_work_size(...) doubles of scratch
 *   G_nc  [S][W][W]; r_fd [S][C]; G_ab [Q][S][N][C]; G_aa [Q][Q][S][N][N]
 *   G_rdd, G_bb [S][9][QN][QN]  block-compact: the (5 QN)^2 operator of subdomain ii only couples the own flux image
 *     with itself and with each neighbour image (supported on the shared side faces), so only 9 of its 25 blocks are
 *     non-zero (the reference keeps such operators as BlockOperators with None blocks, block_swipdg.py:336-338):
 *     block 0 = [self,self], block 1 + side = [a,self] (= [self,a]^T), block 5 + side = [a,a]; rows / columns (q, j) */
int64_t lrbms_estimator_work_size(lrbms_ctx* ctx, int32_t Q, int32_t N);
int lrbms_estimator_grams(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* Wt, const double* Rt,
                          const double* ebar, const double* caa, const double* Aab, const double* Bbb,
                          const double* b, double* work, double* G_nc, double* r_fd, double* G_rdd, double* G_bb,
                          double* G_ab, double* G_aa, void* stream);

/* Fused form of the whole timed region (K7 + K8 + P1 + P2) that exploits the support of the neighbour images:
 * same outputs as lrbms_oswald_apply + lrbms_flux_reconstruct + lrbms_project_system + lrbms_estimator_grams, but the
 * padded image bases Wt / Rt are never materialised.  Supported for N <= 64 and Q N <= 128 on templates whose tables
 * fit the LDS: lrbms_fused_supported answers for the dense output layout (this entry point),
 * lrbms_fused_factored_supported for the factored one (lrbms_project_estimate_fused_factored below), which needs less
 * LDS and also takes large templates (k_c = 16).  work >= lrbms_fused_work_size doubles. */
int lrbms_fused_supported(lrbms_ctx* ctx, int32_t Q, int32_t N);
/* Row length (doubles) of F_nc [S][4][nvs][.]: 2 N + 4 nvs, plus N with LRBMS_OPT_OSWALD_VERTEX_PATCH (the diagonal subdomain's
 * share of the vertex average at the two corner vertices a side carries: A_a | C_a | M_a0 .. M_a3 | A_diag). */
int32_t lrbms_fused_fnc_ld(lrbms_ctx* ctx, int32_t N);
/* v_mfma_f64_16x16x4_f64 instructions (2 048 flops each, padding included) the dense projection kernel of the fused pass
 * executes per subdomain for this shape under the context's launch options (measurement only: roofline of bench.py). */
int64_t lrbms_fused_mfma_per_subdomain(lrbms_ctx* ctx, int32_t Q, int32_t N);
int lrbms_fused_factored_supported(lrbms_ctx* ctx, int32_t Q, int32_t N);
int64_t lrbms_fused_work_size(lrbms_ctx* ctx, int32_t Q, int32_t N);
int lrbms_project_estimate_fused(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* F,
                                 const double* A_diag, const double* A_cpl, const double* P_diag, const double* b,
                                 const double* ebar, const double* caa, const double* Aab, const double* Bbb,
                                 double* work, double* B_sys, double* rhs_red, double* E_red, double* M_red,
                                 double* G_nc, double* r_fd, double* G_rdd, double* G_bb, double* G_ab, double* G_aa,
                                 void* stream);

/* The same pass in two halves for a sharded run that overlaps its halo exchange with compute (one exchange step per
 * pass): phase 1 reads only the first S (rank-local) slabs of V -- compact flux image and vertex
 * averages of the own basis, the dense self blocks: B_sys diagonal, E_red, M_red, G_aa, G_ab[:, self], rhs_red,
 * G_bb / G_rdd [self, self], r_fd[self], G_nc[self, self]; phase 2 needs the halo slabs and writes every block that
 * involves a neighbour.  phase 0 == lrbms_project_estimate_fused; 1 followed by 2 is bit-identical to 0.  Same
 * buffers (including `work`) must be passed to both halves.  phase 3 / 4 split phase 1 into its preparation kernels and
 * its dense kernels: phase 2 depends on 3 and on the halo only, so it may run on another stream beside 4 -- and beside the dense
 * kernels of a phase-1 call: phases 1 and 3 record a library-owned event behind the preparation on their stream, and phase 2 makes
 * ITS stream wait for that event before its first kernel (the caller orders only the halo in front of phase 2).
 * phase 5 = 1 and 2 in ONE call, for a rank with few subdomains whose step is bound by the host: phase 1 goes on `stream`, phase 2 on
 * library stream 0 (lrbms_ctx_aux_stream(ctx, 0)) BEHIND whatever the caller has queued there -- the wait for the halo exchange and
 * its unpack -- and behind the preparation; both are joined into `stream` before the call returns (one fork and one join per step).
 * `stream` must not be library stream 0.  The same kernels with the same arguments: bit-identical to 1 followed by 2. */
/* The i-th (0..2) library-owned HIP stream of the context (a hipStream_t).  A host that overlaps its halo exchange with
 * the pass runs the halo-dependent phase on stream 0 (HIP multiplexes streams onto few hardware queues: a further host
 * stream may land on the queue of the caller's stream and serialise behind the dense kernels). */
void* lrbms_ctx_aux_stream(lrbms_ctx* ctx, int32_t i);

/* Conventions the reference tree does not determine, switchable per context so that a value pinned
 * later against the reference can be matched without touching the kernels.  Set before lrbms_assemble_* / the passes.
 *   LRBMS_OPT_OSWALD_ZERO_ON_SUBDOMAIN_BOUNDARY  0 (default): the Oswald interpolant vanishes on the physical boundary;
 *       1: on the whole boundary of the subdomain -- discretize_elliptic_block_swipdg.py:108-113 passes an all-Dirichlet
 *       boundary info on the subdomain layer to apply_oswald_interpolation_operator
 *   LRBMS_OPT_ACCUMULATE_COUPLING_ACROSS_Q       0 (default): one coupling matrix per affine component; 1: component q
 *       carries the coupling terms of all components q' <= q, as discretize_elliptic_block_swipdg.py:551-565 (matrices
 *       allocated once) with :581-583 (assembled into for every lambda) would if the assembler does not zero them
 *   LRBMS_OPT_OSWALD_VERTEX_PATCH                0 (default): the Oswald average at a vertex runs over the elements of the
 *       subdomain and of its FACE neighbours (HEAD: grid.neighborhood_of, discretize_elliptic_block_swipdg.py:78-113); 1: over
 *       every element at the vertex -- at a cross point also the elements of the diagonal subdomain -- the reading that
 *       reproduces the nonconformity value the reference prints (linearelliptic_block_swipdg_decomp.py:41: 1.66e-01).
 *       Factored layout of the fused pass only (rows of F_nc grow by N columns, lrbms_fused_fnc_ld); the dense layout and the
 *       unfused kernels have five slots per neighbourhood and refuse it.  Sharded grids (S_ext > S): the diagonal subdomains must
 *       be halo slabs and lrbms_set_diagonal_neighbours must name them.  Frozen once a fused pass has run on the context
 *       (LRBMS_E_STATE): it decides the row length of F_nc buffers the caller has sized */
#define LRBMS_OPT_OSWALD_VERTEX_PATCH 9
#define LRBMS_OPT_OSWALD_ZERO_ON_SUBDOMAIN_BOUNDARY 1
#define LRBMS_OPT_ACCUMULATE_COUPLING_ACROSS_Q 2
/* Launch policy of the library (no numerical convention; every setting gives the same results up to the summation order the
 * parity tests bound).  The library reads no environment variable: these options are the only switches.
 *   LRBMS_OPT_STREAMS          -1 (default): the fused pass forks its small kernels over the library streams below 192
 *                              subdomains per rank; 0: never; 1: always
 *   LRBMS_OPT_F1_KSPLIT        0 (default): workgroups per subdomain of the dense projection kernel chosen from S; 1, 2, 4: forced
 *   LRBMS_OPT_F1_FORM          0 (default): the leanest form of that kernel the shape allows (k_f1w, the rank-2 form of the
 *                              df_aa / df_ab groups: Q = 2, even N in [34, 40], with LRBMS_OPT_PREP_LDS; k_f1v: other even N <= 40
 *                              at Q = 2; else k_f1u; N > 48 or Q > 2: k_f1); 1: the producer / consumer form k_f1 for every shape;
 *                              2: neither k_f1w nor k_f1v; 3: no k_f1w (cross-checks of the forms against each other)
 *   LRBMS_OPT_COARSE           coarse level of the reduced solvers' preconditioner: 1 (default) hand-written block-tridiagonal
 *                              factorisation where the band allows it; 0: none (block-Jacobi); 2: rocSOLVER always
 *   LRBMS_OPT_SOLVE_VALU       1: VALU form of the batched solver's panel matvec (cross-check of the matrix-core form)
 *   LRBMS_OPT_ESTIMATE_VALU    1: VALU form of the batched estimate (dense layout only; cross-check)
 *   LRBMS_OPT_PREP_LDS         1 (default): the preparation sweeps of the fused pass (flux image, vertex averages) run from one copy of
 *                              the subdomain's basis slab in LDS whenever it fits (k_prep_lds), with G_nc[self, self] folded into the
 *                              same kernel; 2: the LDS form without that fold (k_f3 computes G_nc[self, self]); 0: the two streaming
 *                              sweeps.  With more subdomains than CUs k_prep_lds runs one persistent workgroup per CU that takes its
 *                              subdomains one after the other and prefetches the next slab into registers; 3: as 1 with one workgroup
 *                              per subdomain at every count (the same bits; cross-check) */
#define LRBMS_OPT_STREAMS 3
#define LRBMS_OPT_F1_KSPLIT 4
#define LRBMS_OPT_F1_FORM 5
#define LRBMS_OPT_COARSE 6
#define LRBMS_OPT_SOLVE_VALU 7
#define LRBMS_OPT_ESTIMATE_VALU 8
#define LRBMS_OPT_PREP_LDS 10
int lrbms_ctx_set_option(lrbms_ctx* ctx, int32_t option, int32_t value);

/* LRBMS_OPT_OSWALD_VERTEX_PATCH on sharded grids: nbr_diag [S][4] (host) = index into the S_ext slabs of the diagonal neighbour
 * at corner 0 SW, 1 SE, 2 NW, 3 NE of every local subdomain, or -1.  lrbms_mesh_upload derives the table from nbr where the side
 * neighbour in between is local; a rank of a sharded grid whose diagonal neighbours are halo slabs of their own (they then need
 * the rows of the elements at the shared cross point only) hands it over here.  Must follow lrbms_mesh_upload. */
int lrbms_set_diagonal_neighbours(lrbms_ctx* ctx, const int32_t* nbr_diag);

/* Incremental re-projection after online enrichment.  The reference re-reduces everything after a round of local enrichment
 * (online_enrichment.py:49-58 calls reductor.reduce() after reductor.py:75-78 enrich_local); but the projected operators of target
 * subdomain ii depend on the bases of ii and of its neighbours only, so after a round that changed the bases of the subdomains M
 * only the targets M + neighbours(M) change.  lrbms_fused_set_subset restricts every following call of the fused pass
 * (lrbms_project_estimate_fused, _factored, _phase) to the `count` LOCAL subdomains listed in `subset` (host array, strictly
 * ascending, each in [0, S)); the pass then writes the rows of exactly those subdomains into the SAME output buffers (strides as
 * for all S subdomains) and leaves every other row untouched -- bit-identical to what the whole pass writes there.  The list is
 * copied (ordered on the library's copy stream before it returns); count == 0 (subset may be NULL) lifts the restriction. */
int lrbms_fused_set_subset(lrbms_ctx* ctx, const int32_t* subset, int32_t count);

/* Per-kernel device timing of the fused pass (measurement only; the reference has wall-clock prints around
 * rd.solve / rd.estimate, python/scripts/linearelliptic_block_swipdg_decomp.py:67-75).  While enabled, every kernel of
 * lrbms_project_estimate_fused(_phase) is bracketed by a HIP event pair on the stream it is launched on;
 * lrbms_kernel_timing_read synchronises the device and returns the kernels of the passes since the last read:
 * names (newline-separated, caller buffer of names_cap bytes), ms [cap], *count entries. */
int lrbms_kernel_timing(lrbms_ctx* ctx, int32_t enable);
int lrbms_kernel_timing_read(lrbms_ctx* ctx, char* names, int64_t names_cap, double* ms, int32_t cap, int32_t* count);

int lrbms_project_estimate_fused_phase(lrbms_ctx* ctx, int32_t phase, int32_t Q, int32_t N, const double* V, const double* F,
                                       const double* A_diag, const double* A_cpl, const double* P_diag, const double* b,
                                       const double* ebar, const double* caa, const double* Aab, const double* Bbb, double* work,
                                       double* B_sys, double* rhs_red, double* E_red, double* M_red, double* G_nc, double* r_fd,
                                       double* G_rdd, double* G_bb, double* G_ab, double* G_aa, void* stream);

/* FACTORED layout of the projected estimator operators (default of the Python host side).  The image of a neighbour's
 * basis on the target subdomain lives on the <= ncf side faces only, so every block of df_bb_i, r_dd_i, df_ab_i
 * (discretize_elliptic_block_swipdg.py:747,762-770) that involves a neighbour slot a has rank <= ncf.  Instead of the
 * dense side blocks (205 KB per side at config 3: 1 GB of writes per pass and of reads per estimate) the pass returns
 *   G_rdd_self, G_bb_self [S][QN][QN]   the [self, self] blocks,       G_ab_self [Q][S][N][QN]   the self columns,
 *   F_side [S][4][ncf][4 QN + 4]         one row per side face p:  Ra | Yb | Dp | Xab (q, i) | sc0 sc1 sc2 0  with
 *       G_bb[a, self] = Ra^T Yb    G_bb[a, a]  = Ra^T diag(sc0) Ra    G_ab^q[:, a] = Xab_q^T Ra
 *       G_rdd[a, self] = Ra^T Dp   G_rdd[a, a] = Ra^T diag(sc1) Ra    r_fd[a] = sc2^T Ra  (r_fd itself stays dense [S][5QN])
 * The same holds for the nonconformity operator nc_i (:733-745): the Oswald image of neighbour a lives on the nvs =
 * max(nvx, nvy) vertices of side a, W_a = -P_a A_a with A_a the vertex averages, so the pass returns
 *   G_nc_self [S][N][N]                  the [self, self] block,
 *   F_nc [S][4][nvs][2 N + 4 nvs]        one row per side vertex:  A_a | C_a | M_a0 M_a1 M_a2 M_a3  with
 *       G_nc[a, self] = A_a^T C_a  (= G_nc[self, a]^T)      G_nc[a, b] = A_a^T M_ab A_b
 *   (C_a = -P_a^T E W_self, M_ab = P_a^T E P_b; 26 KB per subdomain at config 3 instead of 115 KB of side blocks).
 * (the reference keeps such operators as BlockOperators whose missing blocks are None, block_swipdg.py:336-338: a
 * factored block is the same idea one step further).  lrbms_fside_size / lrbms_fnc_size: doubles of F_side / F_nc.
 * `phase` as in lrbms_project_estimate_fused_phase (0 = whole pass).  The dense entry points above produce the same blocks;
 * tests compare both. */
int64_t lrbms_fside_size(lrbms_ctx* ctx, int32_t Q, int32_t N);
int64_t lrbms_fnc_size(lrbms_ctx* ctx, int32_t N);
int lrbms_project_estimate_fused_factored(lrbms_ctx* ctx, int32_t phase, int32_t Q, int32_t N, const double* V, const double* F,
                                          const double* A_diag, const double* A_cpl, const double* P_diag, const double* b,
                                          const double* ebar, const double* caa, const double* Aab, const double* Bbb, double* work,
                                          double* B_sys, double* rhs_red, double* E_red, double* M_red, double* G_nc_self,
                                          double* r_fd, double* G_rdd_self, double* G_bb_self, double* G_ab_self, double* G_aa,
                                          double* F_side, double* F_nc, void* stream);
/* E1 on the factored layout (same arguments otherwise as lrbms_reduced_estimate / _batch). */
int lrbms_reduced_estimate_factored(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* u,
                                    const double* G_nc_self, const double* r_fd, const double* G_rdd_self, const double* G_bb_self,
                                    const double* G_ab_self, const double* G_aa, const double* F_side, const double* F_nc,
                                    const double* f2, const double* ceps, double hdiam, double* eta_loc, void* stream);
int lrbms_reduced_estimate_batch_factored(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* u,
                                          const double* G_nc_self, const double* r_fd, const double* G_rdd_self,
                                          const double* G_bb_self, const double* G_ab_self, const double* G_aa, const double* F_side,
                                          const double* F_nc, const double* f2, const double* ceps, double hdiam, double* eta_loc,
                                          void* stream);

/* -- online --------------------------------------------------------------------------------------------- */
/* E1: EstimatorBase._estimate_elliptic on reduced coefficients (estimators.py:45-112), per-subdomain part.
 *   theta [Q] host; u [S_ext][N]; f2, ceps [S]; hdiam scalar
 *   eta_loc [3][S]: nc, r (scaled by (1/pi^2)/c_eps h^2), df -- squared quantities exactly as estimators.py:71-91 */
int lrbms_reduced_estimate(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* u,
                           const double* G_nc, const double* r_fd, const double* G_rdd, const double* G_bb,
                           const double* G_ab, const double* G_aa, const double* f2, const double* ceps, double hdiam,
                           double* eta_loc, void* stream);

/* E1, throughput form: nmu <= 64 reduced solutions per call, in passes of <= 16 over the same arrays; theta [nmu][Q] host,
 * u [S_ext][N][nmu] (mu fastest, the layout lrbms_reduced_solve_batch returns), eta_loc [3][S][nmu].  Every projected
 * operator is read once per pass of 16. */
int lrbms_reduced_estimate_batch(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* u,
                                 const double* G_nc, const double* r_fd, const double* G_rdd, const double* G_bb,
                                 const double* G_ab, const double* G_aa, const double* f2, const double* ceps, double hdiam,
                                 double* eta_loc, void* stream);

/* O1: rd.solve(mu): (sum_q theta_q B_sys_q) u = rhs_red by block-Jacobi preconditioned CG on the block-sparse
 * reduced system (single rank: S_ext == S), two-level preconditioner (see lrbms_reduced_precond_build).
 * work >= lrbms_reduced_solve_work_size doubles.
 * Returns LRBMS_E_NOT_CONVERGED if the relative residual is above rtol after max_iter.  info[0] = iterations,
 * info[1] = final relative residual (host pointers, may be NULL). */
int64_t lrbms_reduced_solve_work_size(lrbms_ctx* ctx, int32_t N);
int lrbms_reduced_solve(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys,
                        const double* rhs_red, double* work, double* u, double rtol, int32_t max_iter, double* info,
                        void* stream);

/* O1, throughput form: nmu <= 64 parameters per call, in groups of <= 32 (calls of <= 16 parameters: one group of <= 16; with
 * LRBMS_OPT_SOLVE_VALU: groups of <= 16).  theta [nmu][Q] host; u [S][N][nmu] (mu fastest).
 * Inside a group every projected block is read once per CG iteration for all its parameters (panel matvec on the matrix
 * cores: a 16- or 32-column panel), independent CG scalars per parameter.  The groups run on the caller's stream and the library's
 * side streams, launches interleaved iteration by iteration: their kernels are latency-bound and share the chip (the host does
 * not thread: one call, one caller -- a ctx is not re-entrant).  One preconditioner per call: the prebuilt one
 * (lrbms_reduced_precond_use) or inverse diagonal blocks + coarse level at the mean theta of the call.
 * info[0] = iterations (of the slowest group), info[1] = worst relative residual. */
int64_t lrbms_reduced_solve_batch_work_size(lrbms_ctx* ctx, int32_t N, int32_t nmu);
int lrbms_reduced_solve_batch(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* B_sys,
                              const double* rhs_red, double* work, double* u, double rtol, int32_t max_iter, double* info,
                              void* stream);

/* Preconditioner of the reduced solves, built once per reduced model.  Both reduced solvers precondition with the
 * inverse diagonal blocks plus a coarse level on the first local basis vector of every subdomain (the constant the
 * reference starts every basis with, reductor.py:29-31); without the calls below they build it per call at (the batch
 * mean of) theta, which costs about as much as a batched solve (a dense S x S factorisation).  Any SPD preconditioner is
 * admissible, so one built at a reference parameter serves a whole parameter range:
 *   lrbms_reduced_precond_build   theta [Q] host (reference parameter); work >= lrbms_reduced_solve_work_size doubles;
 *                                 pc: lrbms_reduced_precond_size doubles, caller-owned, filled
 *   lrbms_reduced_precond_use     subsequent lrbms_reduced_solve / _batch calls on this context with basis size N use pc
 *                                 (pc must stay alive; NULL: back to per-call preconditioners) */
int64_t lrbms_reduced_precond_size(lrbms_ctx* ctx, int32_t N);
int lrbms_reduced_precond_build(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, double* work,
                                double* pc, void* stream);
int lrbms_reduced_precond_use(lrbms_ctx* ctx, int32_t N, const double* pc);

/* -- snapshot generation (SURVEY.md section 8f "next" #2) -------------------------------------------------- */
/* DuneDiscretization._solve (block_swipdg.py:219-225; ISTL bicgstab.ilut in the reference driver,
 * online_adaptive_lrbms.py:71): A(mu) x = b for the full-order block operator, never assembled: matvec on the block-ELL
 * data (A_diag, A_cpl), CG with the 3x3 element blocks as block-Jacobi preconditioner (the operator is SPD).
 *   theta [Q] host; b, x [S][n]; work: lrbms_fom_solve_work_size doubles (device); info (host, may be NULL):
 *   iterations, final relative residual.  LRBMS_E_NOT_CONVERGED above rtol after max_iter.  Needs S_ext == S. */
int64_t lrbms_fom_solve_work_size(lrbms_ctx* ctx);
int lrbms_fom_solve(lrbms_ctx* ctx, int32_t Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                    double* work, double* x, double rtol, int32_t max_iter, double* info, void* stream);

/* -- parabolic LRBMS (SURVEY.md section 8f "next" #3) --------------------------------------------------------- */
/* InstationaryDuneDiscretization._solve (discretize_parabolic_block_swipdg.py:28-40) with pyMOR's
 * ImplicitEulerTimeStepper(nt) (:87), mass = block L2 product (:49-59):
 *   (M + dt A(mu)) u_{k+1} = M u_k + dt b,  k = 0 .. nt-1,  all nt steps in one call (the step operator is combined once,
 *   every step is a warm-started CG with the kernels of lrbms_fom_solve).
 *   theta [Q] host; U [nt+1][S][n]: U[0] = initial value (input; zero in the reference, :82), U[1..nt] written;
 *   work: lrbms_fom_solve_work_size doubles; info (host, may be NULL): CG iterations over all steps, worst final residual
 *   relative to |M u_k + dt b|.  LRBMS_E_NOT_CONVERGED if a step misses rtol within max_iter.  Needs S_ext == S. */
int lrbms_fom_implicit_euler(lrbms_ctx* ctx, int32_t Q, const double* theta, double dt, int32_t nt, const double* A_diag,
                             const double* A_cpl, const double* b, double* work, double* U, double rtol, int32_t max_iter,
                             double* info, void* stream);

/* d.l2_product.apply_inverse(Y).pairwise_dot(Y) per subdomain (time-stepping residual of ParabolicEstimator.estimate,
 * estimators.py:146-148; r_l2_i of discretize_parabolic_block_swipdg.py:72-74 applied to M^-1 Y):
 *   Y [S][n][L];  out [S][L] = y^T M_s^-1 y  (the P1 mass matrix is inverted element by element in closed form). */
int lrbms_mass_inverse_norm2(lrbms_ctx* ctx, int32_t L, const double* Y, double* out, void* stream);

/* Elliptic-reconstruction terms of the parabolic estimator (estimators.py:65-68, :80-83; operators r_ud_i / r_l2_i of
 * discretize_parabolic_block_swipdg.py:65-74).  The reference never evaluates them (the branch starts with `assert False`,
 * estimators.py:64); they are provided for `ParabolicEstimator(elliptic_reconstruction=True)`.
 *   lrbms_div_apply      Rt [S][n_rt][C] -> mode 0: D [S][n_T][C], the divergence of the RT0 columns (one value per element);
 *                        mode 1: M_s Div_s Rt_s [S][n][C] (the operator inside r_ud_s)
 *   lrbms_div_pairing    full order: out [S][L] = g^T Div U_r with U_r = sum_{slot,q} theta_q Rt[:, (slot, q, l)]
 *                        (D from mode 0, C = 5 Q L; G [S][n][L]); r_ud_s(M^-1 g, U_r), the mass matrices cancel
 *   lrbms_reduced_reconstruction_terms   reduced: out [L][S] = y^T M_red^-1 y - b^T M_red^-1 b - 2 (M_red^-1 (y - b))^T G_ud ur
 *                        with y = (A_red(mu) u_l)_s, b = rhs_red[s], G_ud [S][N][5 Q N] the projected r_ud_s, U [L][S][N];
 *                        work: lrbms_reduced_time_residual_work_size doubles */
int lrbms_div_apply(lrbms_ctx* ctx, int32_t C, int32_t mode, const double* Rt, double* out, void* stream);
int lrbms_div_pairing(lrbms_ctx* ctx, int32_t Q, int32_t L, const double* theta, const double* D, const double* G, double* out,
                      void* stream);
int lrbms_reduced_reconstruction_terms(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t L, const double* theta, const double* B_sys,
                                       const double* M_red, const double* rhs_red, const double* G_ud, const double* U,
                                       double* work, double* out, void* stream);

/* The reduced counterpart of lrbms_fom_implicit_euler on the projected operators:
 *   (M_red + dt sum_q theta_q B_sys_q) u_{k+1} = M_red u_k + dt rhs_red.
 *   B_sys [Q][S][5][N][N], M_red [S][N][N], rhs_red [S][N] as written by the projection; U [nt+1][S][N] (U[0] input);
 *   work: lrbms_reduced_solve_work_size doubles.  N <= 64, S_ext == S. */
int lrbms_reduced_implicit_euler(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, double dt, int32_t nt,
                                 const double* B_sys, const double* M_red, const double* rhs_red, double* work, double* U,
                                 double rtol, int32_t max_iter, double* info, void* stream);

/* Time-stepping residual with d = rd (estimators.py:146-148): out [L][S] = y^T M_red[s]^-1 y,
 * y = (sum_q theta_q B_sys_q dU_l)_s, for L vectors dU [L][S][N]. */
int64_t lrbms_reduced_time_residual_work_size(lrbms_ctx* ctx, int32_t N);
int lrbms_reduced_time_residual(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t L, const double* theta, const double* B_sys,
                                const double* M_red, const double* dU, double* work, double* out, void* stream);

/* -- online enrichment (SURVEY.md section 8f "next" #1) ---------------------------------------------------- */
/* Dirichlet correction blocks of the neighbourhood problems: on every coupling face of subdomain s, the boundary-form
 * diagonal block minus the inner-face block already contained in A_diag
 * (make_elliptic_swipdg_matrix_operator_on_neighborhood with the all-Dirichlet local_boundary_info,
 * block_swipdg.py:240-247, :794-795).   D_corr [Q][S][4][ncf][9], zero where there is no neighbour. */
int lrbms_assemble_dirichlet_correction(lrbms_ctx* ctx, int32_t Q, const double* lam, double* D_corr, void* stream);

/* DuneDiscretization.solve_for_local_correction (block_swipdg.py:227-316) for nmark marked subdomains at once:
 * SWIPDG on N(ii) = ii + face neighbours with Dirichlet outer boundary, rhs = L2 functional of f, restricted to ii.
 *   theta [Q] host, marked [nmark] host (subdomain indices), b [S][n], work (device, lrbms_local_correction_work_size
 *   doubles), corr [nmark][n] out.  One workgroup per marked subdomain runs a block-Jacobi PCG in LDS / registers.
 *   info (host, may be NULL) [nmark][2] = iterations, final relative residual.
 * Returns LRBMS_E_NOT_CONVERGED if any neighbourhood is above rtol after max_iter.  Needs S_ext == S. */
int64_t lrbms_local_correction_work_size(lrbms_ctx* ctx, int32_t nmark);
int lrbms_local_correction_solve(lrbms_ctx* ctx, int32_t Q, const double* theta, int32_t nmark, const int32_t* marked,
                                 const double* A_diag, const double* A_cpl, const double* D_corr, const double* b,
                                 double* work, double* corr, double rtol, int32_t max_iter, double* info, void* stream);

/* -- helpers used by the host shim and the parity tests ------------------------------------------------- */
/* y [S][n][M] = blockELL(A [S][n_T][4][9]) x [S][n][M]   (diagonal blocks only, no coupling) */
int lrbms_blockell_apply(lrbms_ctx* ctx, int32_t M, const double* A, const double* x, double* y, void* stream);
/* y [S][n][M] = sum_q theta_q (A_diag_q x_s + sum_sides A_cpl_q[side] x_neighbour): the full-order block operator
 * (BlockOperator of block_swipdg.py:500-507 applied to a block vector); x [S_ext][n][M] with halo filled; theta host. */
int lrbms_fom_apply(lrbms_ctx* ctx, int32_t Q, int32_t M, const double* theta, const double* A_diag, const double* A_cpl,
                    const double* x, double* y, void* stream);
/* G[b] (Mx x My) = alpha * X[b]^T diag(rowscale) Y[b]; X [batch][K][ldx], Y [batch][K][ldy]; fp64 MFMA */
int lrbms_gemm_tn(lrbms_ctx* ctx, int32_t batch, int32_t K, int32_t Mx, int32_t My, const double* X, int64_t sx,
                  int32_t ldx, const double* Y, int64_t sy, int32_t ldy, double* G, int64_t sg, int32_t ldg,
                  const double* rowscale, double alpha, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRBMS_HIP_H */
