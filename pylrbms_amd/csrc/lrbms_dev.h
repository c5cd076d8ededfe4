// Internal definitions shared by the HIP translation units of liblrbms_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/lrbms_hip.h"

// Device view of the subdomain template (all pointers into one ctx-owned device allocation).
struct Tmpl {
  int kx, ky, nT, n, nrt, nv, nvx, nvy, ncf;
  double hx, hy;
  double kappa[4], kinv[4], kmin;
  const int *nb_elem, *nb_face, *nb_elem_out, *nb_face_out, *elem_side_pos, *elem_rt, *face_sign;
  const int *dof_vertex, *vdof_ptr, *vdof_idx;
  const int *rt_e0, *rt_f0, *rt_e1, *rt_f1, *rt_side;
  const int *side_elem, *side_elem_out, *side_count;
  int ntouch;
  const int *touch_elem, *touch_count;
  const double *grad, *area, *normal, *face_len, *points;
  // derived on the device at mesh upload (build_template_tables, fused.hip): template data every subdomain shares
  const double* stiff;       // [nT][9]            K_T[i][j] = grad phi_i . kappa grad phi_j
  const double* mass9;       // [nT][9]            M_T[i][j] = |T| / 12 (1 + delta_ij)   (P1 mass block)
  const double* zero64;      // [9 nT + 64] zeros: operand lanes of k_f1v that must read 0.0 walk through here
  const double* lgz;         // [nT][2][3]         L^T G_T (kappa = L L^T, G_T = the gradients of the three P1 shape functions): K_T = (L^T G_T)^T (L^T G_T)
  const double* hab;         // [nT][2][3]         L^-1 (G_T G_T^T)^-1 G_T: A_ab,T = G_T^T W_T  =>  W'_T = hab_T A_ab,T = L^-1 W_T (k_f1w)
  const int* touch_vtx;      // [4][ntouch][3]     lattice vertex of local DoF i of the p-th element touching side sd
  const int* touch_pos;      // [4][ntouch][3][4]  its position along side sd', or -1 if it is not on that side
  const int* touch_mask;     // [4][ntouch]        bit sd' set if the element has a vertex on side sd'
  const int* touch_rlist;    // [4][nvs][9]        count, then the rows 3 p + k (p-th touching element, DoF k) that sit on side vertex pos (count -1: > 8 rows)
  const int* sface_i;        // [4][ncf][12]       side face p of side sd: element T, its face f_p on the side, RT0 rows of its faces 0..2,
                             //                    template signs of those faces, the side a face lies on (or -1); 0-padded
  const double* sface_d;     // [4][ncf][4]        |e_g| (g = 0..2), |T|
  // conventions the reference tree leaves open (lrbms_ctx_set_option; defaults = DESIGN.md section 3)
  int opt_oswald_subdomain;      // 1: the Oswald interpolant vanishes on the WHOLE subdomain boundary (block_swipdg.py:108-113 read
                                 //    literally: all-Dirichlet boundary info on the subdomain layer), 0: on the physical boundary only
  int opt_oswald_vertex;         // 1: the Oswald vertex patch is every element at the vertex (at a cross point also the elements of the
                                 //    DIAGONAL subdomain), the reading that reproduces the reference's printed nonconformity value
                                 //    (linearelliptic_block_swipdg_decomp.py:41); 0: the elements of the subdomain and its FACE neighbours
                                 //    (HEAD: grid.neighborhood_of, block_swipdg.py:78-113).  Factored layout, one rank only.
  int opt_accumulate_coupling;   // 1: coupling matrices accumulate across the affine components q (block_swipdg.py:551-565 vs
                                 //    :581-583), 0: one coupling matrix per component
  // incremental re-projection (lrbms_fused_set_subset): the fused pass runs over the sub_count local subdomains sub_list[0 ..] only --
  // one int32 indirection in front of "workgroup index -> subdomain"; array strides and the neighbour table are untouched.
  // nullptr: every subdomain (workgroup b works on subdomain b).  Set in the launcher's COPY of the template, never in ctx->t.
  const int* sub_list;
  int sub_count;
  // diagonal neighbours [S][4] (corner 0 SW, 1 SE, 2 NW, 3 NE; index into the S_ext slabs or -1): read with
  // LRBMS_OPT_OSWALD_VERTEX_PATCH only.  mesh_upload derives it from nbr where that is possible (side neighbour local);
  // lrbms_set_diagonal_neighbours replaces it (sharded grids: the diagonal subdomain is a halo slab of its own).
  const int* nbr_diag;
};

// the subdomain workgroup-index b of a fused-pass kernel works on
__device__ inline int subdomain_of(const Tmpl& t, int b) { return t.sub_list ? t.sub_list[b] : b; }

// Library-owned side streams, one set per device, shared by every context of the process (2D and 3D alike).  HIP maps streams
// round-robin onto a few hardware queues (4 by default); a second context with side streams of its own lands on queues the
// first one (or the caller's stream) already uses and its "concurrent" chains then run one after another -- measured: the
// config-5 pass 2.49 ms instead of 2.20 ms when a 2D context with three streams of its own was alive in the process.
// Reference-counted; the stream is destroyed with its last user.  Thread-safe.
hipStream_t lrbms_side_stream_acquire(int device, int i);   // i in [0, 3); nullptr on failure
void lrbms_side_stream_release(int device, int i);

struct lrbms_ctx {
  int device = 0;
  int num_cus = 0;                // compute units of the device (queried on first use)
  bool has_mesh = false;
  Tmpl t{};
  int S = 0, S_ext = 0;
  int* nbr = nullptr;             // [S][5] device
  std::vector<int32_t> nbr_host;  // the same on the host (argument checks)
  std::vector<void*> owned;       // device allocations to free
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};   // library-owned streams: independent small kernels run concurrently
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_prep = nullptr;   // fused pass, phases 1 / 3: recorded behind the preparation of the own basis; phase 2 waits for it
  hipEvent_t prep_done = nullptr; // ... or for the fork event of that call, recorded at the same point (whichever the last call used)
  // coarse level of the reduced solvers' preconditioner (online.hip): rocBLAS handle and S x S scratch, created on first use
  void* blas = nullptr;
  double* coarse = nullptr;       // [2][S][S] + info
  long coarse_cap = 0;
  // per-kernel device timing of the fused pass (lrbms_kernel_timing): HIP event pairs on the stream each kernel runs on
  bool ktime = false;
  struct KTimer { const char* name; hipEvent_t e0, e1; bool used; };
  std::vector<KTimer> ktimers;
  int ktime_n = 0;
  // K-split of k_f1u (fused.hip): per-workgroup partial tiles and one arrival counter per subdomain, allocated on first use
  double* ksp_part = nullptr;
  long ksp_part_cap = 0;
  int* ksp_ticket = nullptr;
  long ksp_ticket_cap = 0;
  // W'^q_T = hab_T A_ab,T^q [Q][S][nT][2][3], the rank-2 factors of the df_ab element blocks k_f1w multiplies with the flux rows: written
  // by lrbms_assemble_products beside Aab (wab_src = the Aab it belongs to); a pass that is handed another Aab forms them itself
  double* wab = nullptr;
  long wab_cap = 0;
  const double* wab_src = nullptr;
  int wab_Q = 0;
  bool pass_ran = false;              // a fused pass has run: conventions that change buffer shapes (the vertex patch: F_nc rows) are frozen
  bool diag_explicit = false;         // lrbms_set_diagonal_neighbours was called (needed with the vertex patch when S_ext > S)
  int* subset = nullptr;              // lrbms_fused_set_subset: device copy of the list (ctx-owned), subset_n == 0: no restriction
  int subset_n = 0, subset_cap = 0;
  lrbms_quadrature* qdev = nullptr;   // device copy of the quadrature (lrbms_set_quadrature), read by the assembly kernels
  // launch policy (lrbms_ctx_set_option, LRBMS_OPT_STREAMS ...): the library reads no environment variable
  int opt_streams = -1, opt_f1_ksplit = 0, opt_f1_legacy = 0, opt_coarse = 1, opt_solve_valu = 0, opt_estimate_valu = 0, opt_prep_lds = 1;
  const double* user_pc = nullptr;   // prebuilt preconditioner the reduced solves use (lrbms_reduced_precond_use), caller-owned
  int user_pc_N = 0;
  std::string err;
};

// SWIPDG constants (dune-gdt elliptic-ipdg.hh: inner_sigma / boundary_sigma for polorder <= 1), beta = 1/(d-1) = 1
#define SIGMA_INNER 8.0
#define SIGMA_BOUNDARY 14.0

__host__ __device__ inline int side_to_slot(int side) { return side < 2 ? side : side + 1; }
__host__ __device__ inline int slot_to_side(int slot) { return slot < 2 ? slot : slot - 1; }  // slot != 2

static inline int lrbms_fail(lrbms_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define LRBMS_HIP_CHECK(ctx, expr)                                                                  \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return lrbms_fail(ctx, LRBMS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

#define LRBMS_REQUIRE_MESH(ctx)                                                      \
  do {                                                                               \
    if (!(ctx)) return LRBMS_E_INVALID;                                              \
    if (!(ctx)->has_mesh) return lrbms_fail(ctx, LRBMS_E_STATE, "mesh not uploaded"); \
  } while (0)

#define LRBMS_LAUNCH_CHECK(ctx) LRBMS_HIP_CHECK(ctx, hipGetLastError())

// RAII scope around one kernel launch: when timing is enabled (lrbms_kernel_timing) records an event pair on the
// kernel's own stream; otherwise costs one branch.
struct KScope {
  lrbms_ctx* ctx;
  hipStream_t st;
  int idx;
  KScope(lrbms_ctx* c, const char* name, hipStream_t s) : ctx(c), st(s), idx(-1) {
    if (!c->ktime) return;
    if (c->ktime_n == (int)c->ktimers.size()) {
      lrbms_ctx::KTimer k{name, nullptr, nullptr, false};
      if (hipEventCreate(&k.e0) != hipSuccess || hipEventCreate(&k.e1) != hipSuccess) return;
      c->ktimers.push_back(k);
    }
    idx = c->ktime_n++;
    c->ktimers[idx].name = name;
    c->ktimers[idx].used = true;
    (void)hipEventRecord(c->ktimers[idx].e0, st);
  }
  ~KScope() {
    if (idx >= 0) (void)hipEventRecord(ctx->ktimers[idx].e1, st);
  }
};

int build_template_tables(lrbms_ctx* ctx);
int launch_wab(lrbms_ctx* ctx, int Q, const double* Aab, double* Wab, hipStream_t st);   // fused.hip: W' = hab A_ab
long f1_mfma_per_subdomain(lrbms_ctx* ctx, int Q, int N);   // fused.hip: executed MFMAs of the dense projection kernel
// dense coarse level of the Krylov preconditioners (online.hip)
int coarse_begin(lrbms_ctx* ctx, double** A0_out, hipStream_t st);
int coarse_finish(lrbms_ctx* ctx, const double** A0inv_out, hipStream_t st);
int launch_coarse_apply(lrbms_ctx* ctx, int N, int nmu, const double* A0inv, const double* r, double* z, double* prz, hipStream_t st);   // fused.hip; called at the end of lrbms_mesh_upload

// launchers implemented in the other translation units
int launch_assemble_swipdg(lrbms_ctx*, int Q, const double* lam, double* A_diag, double* A_cpl, hipStream_t);
int launch_assemble_rhs(lrbms_ctx*, const double* f_smp, const double* lhat, double* b, double* f2, double* ceps, hipStream_t);
int launch_assemble_products(lrbms_ctx*, int Q, const double* theta_bar_dev, const double* lam, const double* lam_df,
                             const double* lbar, const double* lhat, double* P_diag, double* ebar, double* caa, double* Aab,
                             double* Bbb, hipStream_t);
int launch_assemble_flux(lrbms_ctx*, int Q, const double* lam, double* F, hipStream_t);
int launch_oswald(lrbms_ctx*, int N, const double* V, double* Wt, hipStream_t);
int launch_flux(lrbms_ctx*, int Q, int N, const double* F, const double* V, double* Rt, hipStream_t);
int launch_blockell_apply(lrbms_ctx*, int S, int M, const double* A, long sA, const double* x, double* y, hipStream_t);
int launch_fom_apply(lrbms_ctx*, int Q, int M, const double* theta, const double* A_diag, const double* A_cpl,
                     const double* x, double* y, hipStream_t);
int launch_gemm_tn(lrbms_ctx*, int batch, int K, int Mx, int My, const double* X, long sx, int ldx, const double* Y,
                   long sy, int ldy, double* G, long sg, int ldg, const double* rowscale, double alpha, hipStream_t);
int launch_project_system(lrbms_ctx*, int Q, int N, const double* V, const double* A_diag, const double* A_cpl,
                          const double* P_diag, const double* b, double* work, double* B_sys, double* rhs_red,
                          double* E_red, double* M_red, hipStream_t);
int launch_estimator_grams(lrbms_ctx*, int Q, int N, const double* V, const double* Wt, const double* Rt,
                           const double* ebar, const double* caa, const double* Aab, const double* Bbb, const double* b,
                           double* work, double* G_nc, double* r_fd, double* G_rdd, double* G_bb, double* G_ab,
                           double* G_aa, hipStream_t);
int launch_reduced_estimate(lrbms_ctx*, int Q, int N, const double* theta_dev, const double* u, const double* G_nc,
                            const double* r_fd, const double* G_rdd, const double* G_bb, const double* G_ab,
                            const double* G_aa, const double* Fside, const double* Fnc, const double* f2, const double* ceps,
                            double hdiam, double* eta_loc, hipStream_t);
int launch_reduced_solve(lrbms_ctx*, int Q, int N, const double* theta, const double* B_sys, const double* rhs_red,
                         double* work, double* u, double rtol, int max_iter, double* info, hipStream_t);
