// 3D / P2 hot path (BASELINE.json config 5) for gfx950: C ABI of include/lrbms3d_hip.h.
//
// Kuhn triangulation: every element is a translate of one of six reference tetrahedra, so every local integral is a
// contraction of coefficient samples with a reference table (pylrbms_amd/grid3d.py builds the tables on the host, once):
//     block[e][c] = sum_k sample[e][k] * TABLE[type(e)][k][c].
// The pass (lrbms3_project_estimate) is built from one MFMA kernel template, k3_pg<KIND>: G = sum_items X^T (L Y) with the
// element-local apply L Y and the projection both on the fp64 matrix cores, operands straight from global memory (no LDS
// staging, no VALU arithmetic in the loop), one wave per item, one workgroup per (subdomain, operator).  Images of a NEIGHBOUR's
// basis live on the side faces / side nodes of the target subdomain only: they are returned as factors (Rb, Yb, Dp, Xab, As, Cn)
// and the estimate kernel consumes the factors (header).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include <rocsolver/rocsolver.h>

#include "../../include/lrbms3d_hip.h"

// the process-wide side streams (capi.hip; see lrbms_dev.h): shared with the 2D contexts so that all of them fit the hardware queues
hipStream_t lrbms_side_stream_acquire(int device, int i);
void lrbms_side_stream_release(int device, int i);

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

struct T3 {
  int S, S_ext, nT, n, nrt, ncf, nbf, nvs, nnodes, nb, nbel, nsel, nbd;
  int nA, nB, nC, nFs, nFf, o_fs, o_ff, o_c, lam_stride, hat_stride, f_stride;
  double volume, kmin;
  const int *nbr, *phys;
  const int *elem_type, *up_face, *order, *nb_elem, *nb_out, *face_pos, *tsign, *elem_rt, *rt_e0, *rt_f0, *rt_e1, *rt_f1;
  const int *side_elem, *side_face, *side_elem_out, *side_face_out;
  const int *dof_node, *node_ptr, *node_dofs, *node_mask, *node_count, *side_nodes, *sn_ptr, *sn_dofs;
  const int *dof_bslot, *bn_ptr, *bn_slots, *bnodes, *bnode_sides, *bel_elem, *bel_bnode, *sel_elem, *sel_sf;
  const double *divc, *TV, *TE, *TAA, *TFo, *TFn, *TFb, *TPo, *TPn, *TPb, *TC, *TCb, *TPH, *TM, *TB, *TAB, *WB, *WC;
  const double* TSP;     // [6][nA + 8 nFs][100] diagonal block of the local energy product: TV | per face (TPo | TPb)
  const double* TSD;     // [6][nA + 8 nFs][100] system diagonal block: TV | per face (TFo | TFb)  (built at mesh upload)
  const double* zeros;   // [64] zeros: target of the loads of padding lanes
};

}  // namespace

struct lrbms3_ctx {
  int device = 0;
  bool has_mesh = false;
  T3 t{};
  std::vector<void*> owned;
  std::vector<int32_t> nbr_host;
  // coarse space of the full-order solver (lrbms3_fom_coarse_space): nc functions per subdomain, values at the local DoFs
  int fom_nc = 0;
  double* fom_phi = nullptr;      // [n][4] device, zero-padded columns
  bool fom_keep = false;          // lrbms3_fom_precond_keep: the coarse inverse of a solve stays in the context for the next ones
  double* fom_pc = nullptr;       // [M][M] device (owned), valid for fom_pc_M = nc S unknowns of fom_pc_nc functions
  long fom_pc_M = 0;
  int fom_pc_nc = 0;
  void* blas = nullptr;           // rocBLAS handle (dense coarse inverses), created on first use
  const double* user_pc = nullptr;   // coarse inverse the batched reduced solve uses (lrbms3_reduced_precond_use), caller-owned
  int user_pc_N = 0;
  hipStream_t aux[3] = {nullptr, nullptr, nullptr}; // library-owned streams: the flux chain and the Oswald chain of the pass; the
                                                    // groups of the batched reduced solve (all three)
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  double* thbar = nullptr;         // [8] device: theta(mu_bar) of lrbms3_assemble_energy_product
  double* pg_part = nullptr;       // K-split partial results of the k3_pg kernels (library-owned, grown on demand)
  long pg_part_cap = 0;
  // launch policy (lrbms3_ctx_set_option): the library reads no environment variable
  int opt_ksplit = 0, opt_serial = 0, opt_waves = 0, opt_estimate_valu = 0, opt_solve_valu = 0, opt_fom_coarse = 1;
  bool side_padding = false;       // some side has fewer faces than ncf (unequal cubes per direction): padded factor rows exist
  bool ktime = false;
  struct KTimer { const char* name; hipEvent_t e0, e1; };
  std::vector<KTimer> ktimers;
  int ktime_n = 0;
  std::string err;
};

namespace {

int fail3(lrbms3_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define HIP3(ctx, expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return fail3(ctx, LRBMS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define REQUIRE3(ctx)                                                                \
  do {                                                                               \
    if (!(ctx)) return LRBMS_E_INVALID;                                              \
    if (!(ctx)->has_mesh) return fail3(ctx, LRBMS_E_STATE, "mesh not uploaded");     \
  } while (0)
#define LAUNCH3(ctx) HIP3(ctx, hipGetLastError())

struct KScope3 {
  lrbms3_ctx* ctx;
  hipStream_t st;
  int idx;
  KScope3(lrbms3_ctx* c, const char* name, hipStream_t s) : ctx(c), st(s), idx(-1) {
    if (!c->ktime) return;
    if (c->ktime_n == (int)c->ktimers.size()) {
      lrbms3_ctx::KTimer k{name, nullptr, nullptr};
      if (hipEventCreate(&k.e0) != hipSuccess || hipEventCreate(&k.e1) != hipSuccess) return;
      c->ktimers.push_back(k);
    }
    idx = c->ktime_n++;
    c->ktimers[idx].name = name;
    (void)hipEventRecord(c->ktimers[idx].e0, st);
  }
  ~KScope3() {
    if (idx >= 0) (void)hipEventRecord(ctx->ktimers[idx].e1, st);
  }
};

__device__ inline int side_slot(int side) { return side < 3 ? side : side + 1; }

// RT0 orientation of (element, face) in subdomain s: +1 = outward from this element
__device__ inline int sgn3(const T3& t, int s, int e, int f) {
  const int nb = t.nb_elem[e * 4 + f];
  if (nb < 0 && ((t.phys[s] >> (-(nb + 1))) & 1)) return 1;
  return t.tsign[e * 4 + f];
}

// ------------------------------------------------------------------------------------------------- offline assembly
__device__ inline double contract(const double* __restrict__ smp, const double* __restrict__ tab, int K, int C, int c) {
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += smp[k] * tab[(long)k * C + c];
  return acc;
}

// f2 = ||f||^2, ceps = min lambda_hat * kmin per subdomain: one workgroup, fixed-order tree
__global__ __launch_bounds__(256) void k3_scalars(T3 t, const double* __restrict__ f_smp, const double* __restrict__ lhat,
                                                  double* __restrict__ f2, double* __restrict__ ceps) {
  __shared__ double sa[256], sb[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  double acc = 0.0, mn = INFINITY;
  for (int e = tid; e < t.nT; e += 256) {
    const double* rf = f_smp + ((long)s * t.nT + e) * t.f_stride;
    const double* rh = lhat + ((long)s * t.nT + e) * t.hat_stride;
    for (int k = 0; k < t.nB; ++k) {
      acc += t.WB[k] * rf[k] * rf[k];
      mn = fmin(mn, rh[k]);
    }
  }
  sa[tid] = acc;
  sb[tid] = mn;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      sa[tid] += sa[tid + w];
      sb[tid] = fmin(sb[tid], sb[tid + w]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    f2[s] = sa[0];
    ceps[s] = sb[0] * t.kmin;
  }
}

// Products and estimator operators on the matrix cores.  For the elements of ONE type the contraction
//     out[e][c] = sum_k W(e, k) TABLE[type][k][c]
// is a GEMM of the sample rows with the reference table: A operand = W (lane: element li, quadrature point lk), B operand = the
// table chunk staged in LDS once per workgroup (64 elements = 4 waves x 16), accumulators = 16 elements x 16 entries per tile.
// (The first version had one workgroup per element re-read the whole 173 KB table from L2 and divide per term: 15.8 ms at
// config 5.)  OP: 0 ebar (W = lambda_bar), 1 A_aa pair (W = lambda_q lambda_q' / lambda_hat), 2 A_ab (W = lambda_q / lambda_hat,
// orientation sign of column f at the store), 3 B_bb (W = 1 / lambda_hat, signs of row and column).
constexpr int ASM_KC = 32;        // quadrature points per staged table chunk

//   4 rhs b (W = f at rule B, table w |T| phi_i), 5 int_T f (W = f at rule C, table = the weights), 6 diagonal block of the SWIPDG
//   system (K = volume rule + 4 faces x (inner-face table | Dirichlet table), the weight of the table that does not apply to the
//   face set to zero), 7 block towards the neighbour across face `fq2` (inner: A_diag slot 1 + f; coupling face: A_cpl; zero else).
//   8 / 9 the same two for the LOCAL ENERGY PRODUCT (block_swipdg.py:651-677): penalty parts of the face tables only, every face of
//   the subdomain boundary (coupling or physical) a Dirichlet face with the inside coefficient, no coupling blocks; the sample is
//   sum_q theta_bar_q lambda_q (every term is linear in lambda; theta_bar [Q] on the device through `lhat`).
template <int OP, int NCT>
__global__ __launch_bounds__(256) void k3_asm(T3 t, int Q, int q, int q2, const double* __restrict__ lam, const double* __restrict__ lbar,
                                              const double* __restrict__ lhat, double* __restrict__ out, double* __restrict__ out_mirror) {
  __shared__ double Ts[ASM_KC][NCT * 16 + 4];
  constexpr int C = OP == 2 ? 40 : (OP == 3 ? 16 : (OP == 4 ? 10 : (OP == 5 ? 1 : 100)));
  const int ty = blockIdx.x % 6, grp = blockIdx.x / 6, s = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int ncube = t.nT / 6;
  const int cube = grp * 64 + wave * 16 + li;                        // this lane's element (A operand row)
  const int e = (cube < ncube ? cube : ncube - 1) * 6 + ty;
  const int fq = q2;                                                 // OP 7: the face
  const int K = OP == 0 || OP == 4 ? t.nB : (OP == 6 || OP == 8 ? t.nA + 8 * t.nFs : (OP == 7 || OP == 9 ? t.nFs : t.nC));
  const double* tab;
  if (OP == 0) tab = t.TE + (long)ty * K * C;
  if (OP == 1) tab = t.TAA + (long)ty * K * C;
  if (OP == 2) tab = t.TAB + (long)ty * K * C;
  if (OP == 3) tab = t.TB + (long)ty * K * C;
  if (OP == 4) tab = t.TPH + (long)ty * K * C;
  if (OP == 5) tab = t.WC;
  if (OP == 6) tab = t.TSD + (long)ty * K * C;
  if (OP == 7) tab = t.TFn + ((long)(ty * 4 + fq) * K) * C;
  if (OP == 8) tab = t.TSP + (long)ty * K * C;
  if (OP == 9) tab = t.TPn + ((long)(ty * 4 + fq) * K) * C;
  const long se = (long)s * t.nT + e;
  // OP 0: lambda_bar; 1-3: lambda_hat at rule C; 4 / 5: f at rule B / C (passed as lbar)
  const double* w0 = OP == 0 ? lbar + se * t.nB : (OP == 4 ? lbar + se * t.f_stride : (OP == 5 ? lbar + se * t.f_stride + t.nB
                                                                                        : lhat + se * t.hat_stride + t.nB));
  const double* rec = lam + (((long)q * t.S_ext + s) * t.nT + e) * t.lam_stride;                   // lambda_q record of the element
  const double* w1 = rec + t.o_c;                                                                  // ... at rule C
  const double* w2 = lam + (((long)q2 * t.S_ext + s) * t.nT + e) * t.lam_stride + t.o_c;
  bool bnd[4] = {false, false, false, false};                        // OP 6 / 7: face on the physical boundary
  if (OP == 6 || OP == 7) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int nb = t.nb_elem[e * 4 + f];
      bnd[f] = nb < 0 && ((t.phys[s] >> (-(nb + 1))) & 1);
    }
  }
  if (OP == 8 || OP == 9) {          // local product: every face of the subdomain boundary is a Dirichlet face
#pragma unroll
    for (int f = 0; f < 4; ++f) bnd[f] = t.nb_elem[e * 4 + f] < 0;
  }
  // OP 8 / 9: lambda at mu_bar = sum_q theta_bar_q lambda_q at point `off` of the element's records
  auto lam_bar = [&](int off) {
    double v = 0.0;
    for (int qq = 0; qq < Q; ++qq) v += lhat[qq] * lam[(((long)qq * t.S_ext + s) * t.nT + e) * t.lam_stride + off];
    return v;
  };
  d4 acc[NCT];
#pragma unroll
  for (int j = 0; j < NCT; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += ASM_KC) {
    __syncthreads();
    for (int i = tid; i < ASM_KC * NCT * 16; i += 256) {
      const int kk = i / (NCT * 16), c = i - kk * NCT * 16;
      Ts[kk][c] = (k0 + kk < K && c < C) ? tab[(long)(k0 + kk) * C + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < ASM_KC / 4; ++ks) {
      const int k = k0 + 4 * ks + lk;
      double w = 0.0;
      if (k < K) {
        if (OP == 0 || OP == 4 || OP == 5) w = w0[k];
        else if (OP == 1) w = w1[k] * w2[k] / w0[k];
        else if (OP == 2) w = w1[k] / w0[k];
        else if (OP == 3) w = 1.0 / w0[k];
        else if (OP == 6) {
          if (k < t.nA) {
            w = rec[k];
          } else {                       // face f: nFs points against the inner-face table, then the same points against the Dirichlet table
            const int kf = k - t.nA, f = kf / (2 * t.nFs), r2 = kf - f * 2 * t.nFs, dir = r2 >= t.nFs, pt = dir ? r2 - t.nFs : r2;
            const bool bf = f == 0 ? bnd[0] : (f == 1 ? bnd[1] : (f == 2 ? bnd[2] : bnd[3]));
            w = (bf == (dir != 0)) ? rec[t.o_fs + f * t.nFs + pt] : 0.0;
          }
        } else if (OP == 8) {
          if (k < t.nA) {
            w = lam_bar(k);
          } else {
            const int kf = k - t.nA, f = kf / (2 * t.nFs), r2 = kf - f * 2 * t.nFs, dir = r2 >= t.nFs, pt = dir ? r2 - t.nFs : r2;
            const bool bf = f == 0 ? bnd[0] : (f == 1 ? bnd[1] : (f == 2 ? bnd[2] : bnd[3]));
            w = (bf == (dir != 0)) ? lam_bar(t.o_fs + f * t.nFs + pt) : 0.0;
          }
        } else if (OP == 9) {
          const bool bf = fq == 0 ? bnd[0] : (fq == 1 ? bnd[1] : (fq == 2 ? bnd[2] : bnd[3]));
          w = bf ? 0.0 : lam_bar(t.o_fs + fq * t.nFs + k);
        } else {
          const bool bf = fq == 0 ? bnd[0] : (fq == 1 ? bnd[1] : (fq == 2 ? bnd[2] : bnd[3]));
          w = bf ? 0.0 : rec[t.o_fs + fq * t.nFs + k];
        }
      }
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, Ts[4 * ks + lk][j * 16 + li], acc[j], 0, 0, 0);
    }
  }
  // D layout: lane holds elements lk + 4 r of the wave's 16, entry j * 16 + li
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cb = grp * 64 + wave * 16 + lk + 4 * r;
    if (cb >= ncube) continue;
    const int eo = cb * 6 + ty;
    const long so = (long)s * t.nT + eo;
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      const int c = j * 16 + li;
      if (c >= C) continue;
      double v = acc[j][r];
      if (OP == 2) v *= (double)sgn3(t, s, eo, c & 3);
      if (OP == 3) v *= (double)(sgn3(t, s, eo, c >> 2) * sgn3(t, s, eo, c & 3));
      if (OP == 6) {
        out[(((long)q * t.S + s) * t.nT + eo) * 500 + c] = v;
      } else if (OP == 8) {
        out[((long)s * t.nT + eo) * 500 + c] = v;
      } else if (OP == 9) {
        out[((long)s * t.nT + eo) * 500 + (1 + fq) * 100 + c] = t.nb_elem[eo * 4 + fq] >= 0 ? v : 0.0;
      } else if (OP == 7) {
        const int nb = t.nb_elem[eo * 4 + fq];
        out[(((long)q * t.S + s) * t.nT + eo) * 500 + (1 + fq) * 100 + c] = nb >= 0 ? v : 0.0;
        if (nb < 0) {
          const int side = -(nb + 1);
          out_mirror[((((long)q * t.S + s) * 6 + side) * t.ncf + t.face_pos[eo * 4 + fq]) * 100 + c] = ((t.phys[s] >> side) & 1) ? 0.0 : v;
        }
      } else {
        out[so * C + c] = v;
        if (OP == 1 && out_mirror) out_mirror[so * C + c] = v;
      }
    }
  }
}

__global__ __launch_bounds__(64) void k3_assemble_flux(T3 t, const double* __restrict__ lam, double* __restrict__ Cf) {
  const int e = blockIdx.x, s = blockIdx.y, q = blockIdx.z, c = threadIdx.x;
  if (c >= 40) return;
  const int f = c / 10, i = c - f * 10;
  const int ty = t.elem_type[e];
  const int nb = t.nb_elem[e * 4 + f];
  const bool bnd = nb < 0 && ((t.phys[s] >> (-(nb + 1))) & 1);
  const double* lf = lam + (((long)q * t.S_ext + s) * t.nT + e) * t.lam_stride + t.o_ff + f * t.nFf;
  const double* tab = (bnd ? t.TCb : t.TC) + ((long)(ty * 4 + f) * t.nFf) * 10;
  Cf[(((long)q * t.S_ext + s) * t.nT + e) * 40 + c] = contract(lf, tab, t.nFf, 10, i);
}

// ------------------------------------------------------------------------------------------------- pass: preparation
// R_self [S][n_rt][QN]: RT0 flux image of the own basis on the own faces;  Rb [S][nbf][QN]: the neighbour's share on the side faces.
// One wave per row: the row's elements, orientation and the ten flux coefficients per (element, q) are wave-uniform and come
// through the scalar cache; the only vector-memory instructions are the loads of the basis rows (the address unit, not the
// HBM, bounds these kernels).
constexpr int FLUX_R = 1;     // rows per wave and step (4 rows with their loads grouped was measured: slower)
// rows a wave handles one after the other (measured at config 5, FLUX_LOOP 1 / 4 / 8 / 16: k3_flux 362 / 381 / 387 / 398 us -- it is bound
// by L2 traffic, every element's rows are read by its four faces; NODE_LOOP 1 / 4 / 8 / 16: k3_node_avg 232 / 192 / 192 / 198 us)
#ifndef FLUX_LOOP
#define FLUX_LOOP 1
#endif
#ifndef NODE_LOOP
#define NODE_LOOP 4
#endif

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The row blocks of one subdomain read overlapping basis
// rows, so a 1D grid is decoded such that all blocks of a subdomain get ids that are equal modulo 8 (one XCD) and close in time:
// chunks of 8 subdomains, inside a chunk the subdomain is the fast index.  Returns false for the padding of the last chunk.
__device__ inline bool xcd_block(int nblk, int S, int& xblk, int& s) {
  const int L = blockIdx.x, chunk = L / (8 * nblk), in = L - chunk * 8 * nblk;
  s = chunk * 8 + (in & 7);
  xblk = in >> 3;
  return s < S;
}
inline unsigned xcd_grid(int nblk, int S) { return (unsigned)((S + 7) / 8 * 8 * nblk); }

__global__ __launch_bounds__(256) void k3_flux(T3 t, int Q, int N, int rbeg, int rend, const double* __restrict__ V,
                                               const double* __restrict__ Cf, double* __restrict__ Rs, double* __restrict__ Rb) {
  const int QN = Q * N, nrows = rend;          // rows [rbeg, rend) of [own faces | side faces]
  int s, xblk;
  if (!xcd_block((rend - rbeg + 4 * FLUX_R * FLUX_LOOP - 1) / (4 * FLUX_R * FLUX_LOOP), t.S, xblk, s)) return;
  const int lane = threadIdx.x & 63, half = lane >> 5, jl = lane & 31;
  for (int rep = 0; rep < FLUX_LOOP; ++rep) {      // a wave lives for FLUX_LOOP rows: these sweeps are bound by the wave launch rate
  const int row0 = __builtin_amdgcn_readfirstlane(rbeg + ((xblk * FLUX_LOOP + rep) * 4 + (threadIdx.x >> 6)) * FLUX_R);
  if (row0 >= nrows) return;
  // The (<= 2) elements of a face are dealt to the two halves of the wave: lanes 0..31 take the first, lanes 32..63 the second,
  // lane = basis column.  Every basis row is loaded once per face (ten instead of twenty vector-memory instructions), both affine
  // components are formed in the lane from wave-uniform coefficients, and the halves meet through one lane exchange.
  const double* c0[FLUX_R];
  const double* c1[FLUX_R];
  double sgh[FLUX_R];
  long vrow[FLUX_R];
  double* dst[FLUX_R];
  bool on[FLUX_R];
#pragma unroll
  for (int r = 0; r < FLUX_R; ++r) {
    const int row = row0 + r < nrows ? row0 + r : nrows - 1;
    on[r] = row0 + r < nrows;
    int es[2], fs[2], ss[2], sg[2];
    if (row < t.nrt) {
      es[0] = t.rt_e0[row]; fs[0] = t.rt_f0[row]; es[1] = t.rt_e1[row]; fs[1] = t.rt_f1[row];
      ss[0] = ss[1] = s;
      sg[0] = sgn3(t, s, es[0], fs[0]);
      sg[1] = es[1] >= 0 ? sgn3(t, s, es[1], fs[1]) : 0;
    } else {
      const int sf = row - t.nrt;
      const int t2 = t.nbr[s * 7 + side_slot(sf / t.ncf)];
      es[0] = t.side_elem_out[sf]; fs[0] = t.side_face_out[sf]; es[1] = -1; fs[1] = 0;
      ss[0] = ss[1] = t2;
      sg[0] = sg[1] = 0;
      if (t2 >= 0 && es[0] >= 0) sg[0] = t.tsign[es[0] * 4 + fs[0]];      // a coupling face of the neighbour: its template orientation
      else es[0] = -1;
    }
    if (es[0] < 0) ss[0] = 0, es[0] = 0, fs[0] = 0;                        // nothing to add: sign 0, any valid address
    if (es[1] < 0) ss[1] = ss[0], es[1] = es[0], fs[1] = fs[0];
    c0[r] = Cf + ((long)ss[0] * t.nT + es[0]) * 40 + fs[0] * 10;
    c1[r] = Cf + ((long)ss[1] * t.nT + es[1]) * 40 + fs[1] * 10;
    sgh[r] = (double)(half ? sg[1] : sg[0]);
    vrow[r] = half ? ((long)ss[1] * t.n + es[1] * 10) : ((long)ss[0] * t.n + es[0] * 10);
    dst[r] = row < t.nrt ? Rs + ((long)s * t.nrt + row) * QN : Rb + ((long)s * t.nbf + row - t.nrt) * QN;
  }
  const long qstride = (long)t.S_ext * t.nT * 40;
  for (int j0 = 0; j0 < N; j0 += 32) {
    const int j = j0 + jl, jc = j < N ? j : N - 1;
    double vv[FLUX_R][10];
#pragma unroll
    for (int r = 0; r < FLUX_R; ++r) {
      const double* v = V + vrow[r] * N + jc;
#pragma unroll
      for (int i = 0; i < 10; ++i) vv[r][i] = v[(long)i * N];
    }
#pragma unroll
    for (int r = 0; r < FLUX_R; ++r)
      for (int q = 0; q < Q; ++q) {
        const double* a0 = c0[r] + q * qstride;
        const double* a1 = c1[r] + q * qstride;
        double s0 = 0.0, s1 = 0.0;          // both coefficient rows are wave-uniform (scalar loads); the lane keeps its half's sum
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          s0 += a0[i] * vv[r][i];
          s1 += a1[i] * vv[r][i];
        }
        const double acc = (half ? s1 : s0) * sgh[r];
        const double other = __shfl_xor(acc, 32);
        if (half == 0 && j < N && on[r]) dst[r][q * N + j] = acc + other;
      }
  }
  }
}

// (Measured and dropped, round 3: the own faces from element chunks staged in LDS -- 24 elements per workgroup, rows and flux
// coefficients loaded once with 16-byte loads, only the faces towards a later chunk read their second element from global memory:
// 268 + 70 us (side faces) against 363 us for this kernel, but the PASS went from 2.19 to 2.29 ms: with 73 KB of LDS and 128 VGPRs
// per workgroup the sweep no longer fits beside the MFMA kernels of the other two chains, and hiding under them is worth more than
// the 25 us.)
// Avg [S][n_nodes][N]: own share of the Oswald node average (0 on the physical boundary: the interpolant vanishes there);
// As [S][6][nvs][N]: the neighbours' shares at the side nodes.  One wave per node, DoF lists through the scalar cache.
__global__ __launch_bounds__(256) void k3_node_avg(T3 t, int N, int rbeg, int rend, const double* __restrict__ V,
                                                   double* __restrict__ Avg, double* __restrict__ As) {
  int s, xblk;
  if (!xcd_block((rend - rbeg + 4 * NODE_LOOP - 1) / (4 * NODE_LOOP), t.S, xblk, s)) return;          // rows [rbeg, rend) of [own nodes | side nodes]
  const int j = threadIdx.x & 63;
  const int jc = j < N ? j : N - 1;
  const int phys = t.phys[s];
  for (int rep = 0; rep < NODE_LOOP; ++rep) {
  const int row = __builtin_amdgcn_readfirstlane(rbeg + (xblk * NODE_LOOP + rep) * 4 + (threadIdx.x >> 6));
  if (row >= rend) return;
  int p0 = 0, p1 = 0, node, src = s;
  const int* list;
  if (row < t.nnodes) {
    node = row;
    list = t.node_dofs;
    if (!(t.node_mask[node] & phys)) p0 = t.node_ptr[row], p1 = t.node_ptr[row + 1];
  } else {
    const int sp = row - t.nnodes;
    node = t.side_nodes[sp];
    src = t.nbr[s * 7 + side_slot(sp / t.nvs)];
    list = t.sn_dofs;
    if (node >= 0 && src >= 0 && !(t.node_mask[node] & phys)) p0 = t.sn_ptr[sp], p1 = t.sn_ptr[sp + 1];
  }
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (p1 > p0) {
    const double* v = V + (long)src * t.n * N + jc;
    int p = p0;
    for (; p + 3 < p1; p += 4) {
      a0 += v[(long)list[p] * N];
      a1 += v[(long)list[p + 1] * N];
      a2 += v[(long)list[p + 2] * N];
      a3 += v[(long)list[p + 3] * N];
    }
    for (; p < p1; ++p) a0 += v[(long)list[p] * N];
    a0 = ((a0 + a1) + (a2 + a3)) / (double)t.node_count[node];
  }
  if (j < N) {
    if (row < t.nnodes) Avg[((long)s * t.nnodes + row) * N + j] = a0;
    else As[((long)s * 6 * t.nvs + row - t.nnodes) * N + j] = a0;
  }
  }
}

// ------------------------------------------------------------------------------------------------- pass: projections
enum { G_SYS = 0, G_AAA = 1, G_NC = 2, G_AB = 3, G_BB = 4, G_CPL = 6 };

struct GA {
  T3 t;
  int Q, N;
  const double *V, *A_diag, *A_cpl, *ebar, *Aaa, *Aab, *Bbb, *Rs, *Avg;
  double* out;
  double* Zb;   // NC: rows of E W_self at the boundary DoFs [S][nbd][N]
  double *out2, *out3;     // BB: G_rdd [S][QN][QN], r_fd [S][QN] (accumulated beside G_bb: they share every operand)
  const double* bdiv;
  const double* b;        // SYS: right-hand side [S][n]; rhs_red = V^T b rides on the X operands of the q = 0 blocks
  double* out4;           // SYS: rhs_red [S][N]
  double *Yb, *Dp, *Xab;  // BB: rows of B R_self and |T| div div R_self at the side faces; AB: A_ab^T V at the side faces
  int ksplit;             // > 1: the items of a (subdomain, operator) are dealt to ksplit workgroups, partial results go to `part`
  double* part;           //      [batch][ksplit][pg_part_size] and k3_pg_combine sums them in a fixed order
};

// ------------------------------------------------------------------------------------------------- pass: MFMA pipeline
// G = sum_items X^T (L Y), everything on the matrix cores, operands straight from global memory (L2 / L1 resident):
//   Z  = L Y      L [mz x ky] element block as the A operand (lane: row l&15, k l>>4), Y rows gathered as the B operand
//   G += X^T Z    the accumulator layout of Z (lane: rows (l>>4) + 4 r, column l&15) IS the B-operand layout of k-step r,
//                 so Z never leaves the registers; X rows as the A operand
// One wave per item (element / side face), NW waves per workgroup = one (subdomain, operator); the waves' tiles are summed in a
// fixed order through the LDS at the end.  RT x CT output tiles of 16 x 16 per wave.
template <int KIND> struct PGT {};
template <> struct PGT<G_SYS> { static constexpr int MZ = 10, KY = 50; };
template <> struct PGT<G_AAA> { static constexpr int MZ = 10, KY = 10; };
template <> struct PGT<G_NC> { static constexpr int MZ = 10, KY = 10; };
template <> struct PGT<G_CPL> { static constexpr int MZ = 10, KY = 10; };
template <> struct PGT<G_AB> { static constexpr int MZ = 4, KY = 10; };    // contracted through the 4 faces: W = A_ab^T V_e first
template <> struct PGT<G_BB> { static constexpr int MZ = 4, KY = 4; };

// side-face index (side * ncf + pos) of face f of element e, or -1; `has`: the neighbour across that side exists
__device__ inline int side_face_of(const T3& t, int s, int e, int f, bool& has) {
  const int nb = t.nb_elem[e * 4 + f];
  has = false;
  if (nb >= 0) return -1;
  const int side = -(nb + 1);
  has = t.nbr[s * 7 + side_slot(side)] >= 0;
  return side * t.ncf + t.face_pos[e * 4 + f];
}

constexpr int pg_max_threads(int tiles) { return tiles <= 4 ? 1024 : (tiles <= 8 ? 512 : 256); }   // VGPR budget 128 / 256 / 512

// Block id -> (subdomain, operator, part).  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2); all operator
// blocks (and all parts) of one subdomain read the same basis rows, so they get ids that differ by multiples of 8 inside one chunk
// of 8 subdomains: same XCD, launched together.
template <int KIND>
struct PGBlock {
  int b, s, q, q2, Mx, My, t2, side, part;
  double* out;
  __device__ bool decode(const GA& a, int x) {
    const T3& t = a.t;
    const int N = a.N, Q = a.Q, QN = Q * N, ks = a.ksplit;
    const int nops = KIND == G_SYS || KIND == G_AB ? Q : (KIND == G_AAA ? Q * (Q + 1) / 2 : (KIND == G_CPL ? 6 * Q : 1));
    const int chunk = x / (8 * nops * ks), within = x - chunk * 8 * nops * ks;
    const int op = (within >> 3) / ks, sx = chunk * 8 + (within & 7);
    part = (within >> 3) - op * ks;
    if (sx >= t.S) return false;
    b = KIND == G_CPL ? ((op / 6) * t.S + sx) * 6 + op % 6 : op * t.S + sx;
    q = q2 = t2 = side = 0;
    if (KIND == G_SYS) {
      q = b / t.S; s = b - q * t.S; Mx = My = N;
      out = a.out + (((long)q * t.S + s) * 7 + 3) * N * N;
    } else if (KIND == G_CPL) {
      side = b % 6;
      const int qs = b / 6;
      q = qs / t.S; s = qs - q * t.S; Mx = My = N;
      t2 = t.nbr[s * 7 + side_slot(side)];
      out = a.out + (((long)q * t.S + s) * 7 + side_slot(side)) * N * N;
    } else if (KIND == G_AAA) {       // one block per pair q <= q2 (row-major); the block (q2, q) is the transpose
      int p = b / t.S;
      s = b - p * t.S; Mx = My = N;
      while (p >= Q - q) p -= Q - q, ++q;
      q2 = q + p;
      out = a.out + (((long)q * Q + q2) * t.S + s) * N * N;
    } else if (KIND == G_NC) {
      s = b; Mx = My = N;
      out = a.out + (long)b * N * N;
    } else if (KIND == G_AB) {
      q = b / t.S; s = b - q * t.S; Mx = N; My = QN;
      out = a.out + (long)b * N * QN;
    } else {
      s = b; Mx = My = QN;
      out = a.out + (long)b * QN * QN;
    }
    return true;
  }
};

// doubles of one partial result: the output matrix (BB: G_bb and G_rdd) plus one row (SYS: rhs_red, BB: r_fd)
template <int KIND>
__host__ __device__ inline long pg_part_size(int N, int QN) {
  return KIND == G_BB ? 2L * QN * QN + QN : (KIND == G_AB ? (long)N * QN : (KIND == G_SYS ? (long)N * N + N : (long)N * N));
}

__device__ inline double ld8(const double* base, unsigned byte_off) {   // uniform base + 32-bit lane offset: no 64-bit VALU
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}

// On this chip the f64 MFMA stream of one wave and the VALU work of the other waves of the SIMD do not overlap (DESIGN.md
// section 5.1), so the address arithmetic per item is kept minimal: every operand address is (uniform base of the item) +
// (32-bit lane offset), and the lane offsets are (element term: one multiply per neighbour slot and item) + (lane constant
// computed once in front of the loop).  The K index of the apply is scheduled so that a k-step reads ONE neighbour slot:
// rows j = 0..3 and 4..7 of every slot as full steps, the rows 8, 9 of two slots paired into one step.
__device__ inline double2 ld16(const double* base, unsigned byte_off) {
  return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + byte_off);
}

// WIDE (even tile counts and even N): lane li owns the adjacent output columns li * CT + ct (rows li * RT + rt) instead of
// ct * 16 + li, and the two full k-steps of a slot read the rows j = 2 lk and 2 lk + 1, so a lane's operands of two tiles / two
// k-steps are 16 contiguous bytes: half as many vector-memory instructions per item.  The address unit of a CU serves its four
// SIMDs at a fixed number of cycles per instruction, and with 45 eight-byte loads per 38 MFMAs it, not the matrix pipe, set
// the pace.
template <int KIND, int RT, int CT, bool WIDE>
__global__ __launch_bounds__(pg_max_threads(KIND == G_BB ? (RT > 2 ? 8 : 4) : RT * CT * (KIND == G_NC || KIND == G_SYS ? 2 : 1))) void k3_pg(GA a) {
  extern __shared__ double lds[];   // [RT * CT][256]
  constexpr int MZ = PGT<KIND>::MZ, KY = PGT<KIND>::KY, KR = (MZ + 3) / 4;
  constexpr int NG = KY == 50 ? 5 : 1;                       // neighbour slots whose rows the apply reads
  constexpr int KS = KY == 4 ? 1 : 2 * NG + (NG + 1) / 2;    // k-steps: 13 for SYS, 3 for the 10-row kinds, 1 for the 4-row kinds
  constexpr bool NCK = KIND == G_NC, FACEK = KIND == G_AB || KIND == G_BB;
  const T3& t = a.t;
  const int N = a.N, Q = a.Q, QN = Q * N;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = blockDim.x >> 6;
  PGBlock<KIND> B;
  if (!B.decode(a, blockIdx.x)) return;
  const int b = B.b, s = B.s, q = B.q, q2 = B.q2, Mx = B.Mx, My = B.My, t2 = B.t2, side = B.side, part = B.part;
  double* out = B.out;
  int nitems = KIND == G_CPL ? t.ncf : t.nT;
  if (KIND == G_CPL && t2 < 0) {        // no neighbour on that side: the block is zero (the combine kernel skips it too)
    if (part == 0)
      for (int i = tid; i < Mx * My; i += blockDim.x) out[i] = 0.0;
    return;
  }
  const double* Vs = a.V + (long)s * t.n * N;
  const double* Yb = KIND == G_CPL ? a.V + (long)t2 * t.n * N : Vs;
  const double* Av = a.Avg + (long)s * t.nnodes * N;
  const double* Rss = a.Rs + (long)s * t.nrt * QN;
  const double* Lall;               // element blocks of this (subdomain, operator): item e at Lall + e * LSTRIDE
  constexpr int LSTRIDE = KIND == G_SYS ? 500 : (KIND == G_AB ? 40 : (KIND == G_BB ? 16 : 100));
  if (KIND == G_SYS) Lall = a.A_diag + ((long)q * t.S + s) * t.nT * 500;
  if (KIND == G_AAA) Lall = a.Aaa + (((long)q * Q + q2) * t.S + s) * t.nT * 100;
  if (KIND == G_NC) Lall = a.ebar + (long)s * t.nT * 100;
  if (KIND == G_AB) Lall = a.Aab + ((long)q * t.S + s) * t.nT * 40;
  if (KIND == G_BB) Lall = a.Bbb + (long)s * t.nT * 16;
  if (KIND == G_CPL) Lall = a.A_cpl + (((long)q * t.S + s) * 6 + side) * t.ncf * 100;

  // ---- lane constants
  const unsigned rowb = (unsigned)N * 8u, erow = 10u * rowb;            // bytes per DoF row / per element of the basis slab
  const int lic = li < MZ ? li : MZ - 1;
  unsigned yc[KS][CT];        // Y: (row j of the step) * rowb + column
  unsigned lc[KS];            // L: offset inside the item's block(s)
  int sA[KS];                 // neighbour slot this lane reads in step k (lane constant)
  bool pad[KS];               // lane is K padding in this step
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    int g, j;
    bool pd = false;
    if (KY == 4) {
      g = 0; j = lk;
    } else if (k < 2 * NG) {
      g = k >> 1; j = WIDE ? 2 * lk + (k & 1) : 4 * (k & 1) + lk;
    } else {                   // rows 8, 9 of slot gA (lanes lk < 2) and of slot gA + 1 (lanes lk >= 2)
      const int gA = 2 * (k - 2 * NG);
      g = lk < 2 ? gA : gA + 1;
      j = 8 + (lk & 1);
      if (g >= NG) g = gA, pd = true;
    }
    sA[k] = g; pad[k] = pd;
    if (KIND == G_SYS) lc[k] = 8u * (g * 100 + lic * 10 + j);
    else if (KIND == G_AB) lc[k] = 8u * (lic * 4 + j);
    else if (KIND == G_BB) lc[k] = 8u * (j * 4 + lic);                                        // symmetric: read transposed
    else lc[k] = 8u * (lic * 10 + j);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      int col;
      if (WIDE) {              // pairs of columns stay inside the (even) row: the last pair is repeated for the padding lanes
        const int c0 = li * CT + (ct & ~1);
        col = (c0 + 1 < My ? c0 : My - 2) + (ct & 1);
      } else {
        const int col0 = ct * 16 + li;
        col = col0 < My ? col0 : My - 1;
      }
      yc[k][ct] = (FACEK ? 0u : (unsigned)j * rowb) + 8u * col;
    }
  }
  unsigned xc[KR][RT];
  bool xin[KR];
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    const int row0 = 4 * r + lk;
    xin[r] = row0 < MZ;
    const int row = xin[r] ? row0 : MZ - 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int col;
      if (WIDE) {
        const int c0 = li * RT + (rt & ~1);
        col = (c0 + 1 < Mx ? c0 : Mx - 2) + (rt & 1);
      } else {
        const int col0 = rt * 16 + li;
        col = col0 < Mx ? col0 : Mx - 1;
      }
      xc[r][rt] = (KIND == G_BB ? 0u : (unsigned)row * rowb) + 8u * col;
    }
  }
  const double* zero = t.zeros + lane;

  d4 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  // index tables of an item (fetched one item ahead): element, its neighbours (SYS), node / RT0 numbers of this lane's rows
  struct Idx {
    int e, eo;
    int nb[4], up[4];
    int aux[KR], auy[KR];
  };
  const int last = nitems - 1;
  auto load_idx = [&](int item, Idx& ix) {
    item = item < last ? item : last;
    if (KIND != G_CPL) item = t.order[item];          // element visited at this position of the traversal
    ix.e = item;
    ix.eo = 0;
    if (KIND == G_CPL) {
      const int sp = side * t.ncf + item;
      ix.e = t.side_elem[sp];
      ix.eo = t.side_elem_out[sp];
    }
    if (KIND == G_SYS) {
      const int4 v = *reinterpret_cast<const int4*>(t.nb_elem + item * 4);
      ix.nb[0] = v.x; ix.nb[1] = v.y; ix.nb[2] = v.z; ix.nb[3] = v.w;
      const int4 u = *reinterpret_cast<const int4*>(t.up_face + item * 4);
      ix.up[0] = u.x; ix.up[1] = u.y; ix.up[2] = u.z; ix.up[3] = u.w;
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      ix.aux[r] = 0;
      ix.auy[r] = 0;
      if (NCK) {
        const int row = 4 * r + lk;
        ix.aux[r] = t.dof_node[item * 10 + (row < 10 ? row : 9)];                      // node of X row 4 r + lk
        const int j = r < 2 ? (WIDE ? 2 * lk + r : 4 * r + lk) : 8 + (lk & 1);          // node of the Y row of k-step r
        ix.auy[r] = t.dof_node[item * 10 + j];
      }
    }
    if (FACEK) ix.aux[0] = t.elem_rt[item * 4 + lk];
  };

  // G_bb: second accumulator set (G_rdd, upper tile triangle like G_bb) and the r_fd row
  constexpr bool BBK = KIND == G_BB;
  d4 accd[BBK ? RT : 1][BBK ? CT : 1];
  double fd[BBK ? CT : 1];
  if constexpr (BBK) {
    static_assert(KIND != G_BB || RT == CT, "G_bb is square");
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) accd[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < CT; ++j) fd[j] = 0.0;
  }
  // B_sys: lane constants of the symmetric form (below); rhs_red accumulators of the q = 0 blocks
  double racc[KIND == G_SYS ? RT : 1];
#pragma unroll
  for (int i = 0; i < (KIND == G_SYS ? RT : 1); ++i) racc[i] = 0.0;
  unsigned sl_full, sy_full[2][CT], sl_rem, sy_rem[CT];
  if constexpr (KIND == G_SYS) {
    sl_full = 8u * (lic * 10 + (WIDE ? 2 * lk : lk));                 // rows j = 2 lk, 2 lk + 1 (one 16-byte load) / j = lk, 4 + lk
    sl_rem = 8u * (lic * 10 + 8 + (lk & 1));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      int col;
      if (WIDE) {
        const int c0 = li * CT + (ct & ~1);
        col = (c0 + 1 < N ? c0 : N - 2) + (ct & 1);
      } else {
        const int col0 = ct * 16 + li;
        col = col0 < N ? col0 : N - 1;
      }
#pragma unroll
      for (int par = 0; par < 2; ++par) sy_full[par][ct] = (unsigned)(WIDE ? 2 * lk + par : 4 * par + lk) * rowb + 8u * col;
      sy_rem[ct] = (unsigned)(8 + (lk & 1)) * rowb + 8u * col;
    }
  }
  // G_ab: lane constants of its own form (below)
  unsigned abl[3], abv[3][RT];
  bool abpad[3];
  if constexpr (KIND == G_AB) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = k < 2 ? (WIDE ? 2 * lk + k : 4 * k + lk) : 8 + (lk & 1);
      abpad[k] = k == 2 && lk >= 2;
      abl[k] = 8u * (j * 4 + (li < 4 ? li : 3));            // A_ab[e][j][f = li]: rows f >= 4 of the product are never used
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        int col;
        if (WIDE) {
          const int c0 = li * RT + (rt & ~1);
          col = (c0 + 1 < N ? c0 : N - 2) + (rt & 1);
        } else {
          const int col0 = rt * 16 + li;
          col = col0 < N ? col0 : N - 1;
        }
        abv[k][rt] = (unsigned)j * rowb + 8u * col;
      }
    }
  }

  // the items of this part: a contiguous range of the traversal
  const int plen = (nitems + a.ksplit - 1) / a.ksplit, pbeg = part * plen;
  const int pend = pbeg + plen < nitems ? pbeg + plen : nitems;
  double* P = a.ksplit > 1 ? a.part + ((long)b * a.ksplit + part) * pg_part_size<KIND>(N, QN) : nullptr;
  Idx ix1, ix2;
  load_idx(pbeg + wave, ix1);
  for (int item0 = pbeg + wave; item0 < pend; item0 += NW) {
    const int item = item0 < last ? item0 : last;
    if constexpr (KIND == G_SYS) {
      // Symmetric form: A[e', e] = A[e, e']^T, so with H = sum_e V_e^T (1/2 A_ee V_e + sum_{e' > e} A[e, e'] V_e') the projection is
      // B = H + H^T (epilogue).  Per element only the diagonal block and the blocks towards the neighbours with the HIGHER index
      // are read (t.up_face, on average 1.75 of 3.5 inner faces): 55 % of the A_diag bytes and of the neighbour rows, and
      // 2 s + ceil(s / 2) k-steps for s = 1 + #upper slots instead of 13.  Slots are wave-uniform, skipped slots cost nothing.
      const int e = ix1.e;
      const double* Lb = Lall + (long)e * LSTRIDE;
      const unsigned ex = (unsigned)e * erow;
      const int nsl = 1 + (ix1.up[0] >= 0) + (ix1.up[1] >= 0) + (ix1.up[2] >= 0) + (ix1.up[3] >= 0);
      unsigned ebs[5], bos[5];
      ebs[0] = ex;
      bos[0] = 0u;
#pragma unroll
      for (int g = 1; g < 5; ++g) {
        const int f = ix1.up[g - 1], fc = f < 0 ? 0 : f;
        const int ee = fc == 0 ? ix1.nb[0] : (fc == 1 ? ix1.nb[1] : (fc == 2 ? ix1.nb[2] : ix1.nb[3]));
        ebs[g] = (unsigned)(f < 0 ? e : ee) * erow;
        bos[g] = f < 0 ? 0u : 800u * (1 + fc);
      }
      double lop[13], yv[CT][13], xop[RT][KR];
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        if (g < nsl) {                                    // wave-uniform
          if (WIDE) {
            const double2 v = ld16(Lb, bos[g] + sl_full);
            lop[2 * g] = v.x;
            lop[2 * g + 1] = v.y;
          } else {
            lop[2 * g] = ld8(Lb, bos[g] + sl_full);
            lop[2 * g + 1] = ld8(Lb, bos[g] + sl_full + 32u);
          }
#pragma unroll
          for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              if (WIDE) {
                if ((ct & 1) == 0) {
                  const double2 v = ld16(Vs, ebs[g] + sy_full[par][ct]);
                  yv[ct][2 * g + par] = v.x;
                  yv[ct + 1 < CT ? ct + 1 : ct][2 * g + par] = v.y;
                }
              } else {
                yv[ct][2 * g + par] = ld8(Vs, ebs[g] + sy_full[par][ct]);
              }
            }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {                       // rows 8, 9 of slots 2 r (lanes lk < 2) and 2 r + 1 (lanes lk >= 2)
        if (2 * r < nsl) {
          const int gB = 2 * r + 1 < 5 ? 2 * r + 1 : 2 * r;
          const bool hasB = 2 * r + 1 < nsl;
          const unsigned eo = (lk < 2 || !hasB) ? ebs[2 * r] : ebs[gB];
          const unsigned bo = (lk < 2 || !hasB) ? bos[2 * r] : bos[gB];
          lop[10 + r] = (lk >= 2 && !hasB) ? *zero : ld8(Lb, bo + sl_rem);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            if (WIDE) {
              if ((ct & 1) == 0) {
                const double2 v = ld16(Vs, eo + sy_rem[ct]);
                yv[ct][10 + r] = v.x;
                yv[ct + 1 < CT ? ct + 1 : ct][10 + r] = v.y;
              }
            } else {
              yv[ct][10 + r] = ld8(Vs, eo + sy_rem[ct]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if (WIDE) {
            if ((rt & 1) == 0) {
              const double2 v = xin[r] ? ld16(Vs, ex + xc[r][rt]) : *reinterpret_cast<const double2*>(t.zeros);
              xop[rt][r] = v.x;
              xop[rt + 1 < RT ? rt + 1 : rt][r] = v.y;
            }
          } else {
            xop[rt][r] = xin[r] ? ld8(Vs, ex + xc[r][rt]) : *zero;
          }
        }
      double bv[KR];                                      // q = 0 (wave-uniform): the element's entries of b for rhs_red = V^T b
#pragma unroll
      for (int r = 0; r < KR; ++r) bv[r] = (q == 0 && xin[r]) ? a.b[(long)s * t.n + e * 10 + 4 * r + lk] : 0.0;
      load_idx(item0 + NW, ix2);
      ix1 = ix2;
      lop[0] *= 0.5;                                      // the diagonal block enters H with the factor 1/2
      lop[1] *= 0.5;
      lop[10] *= lk < 2 ? 0.5 : 1.0;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 5; ++g)
          if (g < nsl) {
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[2 * g], yv[ct][2 * g], z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[2 * g + 1], yv[ct][2 * g + 1], z, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (2 * r < nsl) z = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[10 + r], yv[ct][10 + r], z, 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < KR; ++r) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[rt][r], z[r], acc[rt][ct], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < KR; ++r) racc[rt] += xop[rt][r] * bv[r];
      continue;
    }
    if constexpr (KIND == G_BB) {
      // Three operators of the four RT0 rows R_e of an element, one set of loads:
      //   G_bb  += R_e^T (B_bb R_e)                        apply (A operand B_bb, symmetric) + projection, upper tile triangle
      //   G_rdd += |T| (d^T R_e)^T (d^T R_e)               d_f = +-|f| / |T|: z1 = d^T R_e is ONE row, so the projection is a
      //                                                    k = 1 product of z1 with itself (lanes lk = 0 carry it)
      //   r_fd  += (int_T f) z1                            a VALU accumulation of the same row
      const int e = ix1.e;
      const double* Lb = Lall + (long)e * LSTRIDE;
      const unsigned rb = (unsigned)ix1.aux[0] * ((unsigned)QN * 8u);
      const int ty = t.elem_type[e];
      double dd[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) dd[f] = sgn3(t, s, e, f) * t.divc[ty * 4 + f];
      const double bd = a.bdiv[(long)s * t.nT + e];
      const double lopB = ld8(Lb, 8u * (lk * 4 + (li < 4 ? li : 3)));
      double re[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if (WIDE) {
          if ((ct & 1) == 0) {
            const double2 v = ld16(Rss, rb + yc[0][ct]);
            re[ct] = v.x;
            re[ct + 1 < CT ? ct + 1 : ct] = v.y;
          }
        } else {
          re[ct] = ld8(Rss, rb + yc[0][ct]);
        }
      }
      load_idx(item0 + NW, ix2);
      ix1 = ix2;
      const double dsel = lk == 0 ? dd[0] : (lk == 1 ? dd[1] : (lk == 2 ? dd[2] : dd[3]));
      const double l1 = li < 4 ? dsel : 0.0;              // A operand of z1: rows 0..3 = d^T (every k lane then holds the row)
      bool shas;
      const int sfi = side_face_of(t, s, e, lk, shas);    // this lane's face f = lk as a side face
      double zb[CT], z1r[CT], z1[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const d4 z0 = (d4){0.0, 0.0, 0.0, 0.0};
        zb[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(lopB, re[ct], z0, 0, 0, 0)[0];      // rows f = lk of B_bb R_e
        z1r[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, re[ct], z0, 0, 0, 0)[0];       // d^T R_e, replicated in the four k lanes
        z1[ct] = lk == 0 ? z1r[ct] : 0.0;                                                 // ... as a k = 1 operand
        fd[ct] += bd * z1[ct];
      }
      if (sfi >= 0) {      // side-face factors (read by the estimate): Yb = row of B R_self, Dp = |T| d_f (d^T R_self); 0 if no neighbour
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int col = WIDE ? li * CT + ct : ct * 16 + li;
          if (col < My) {
            a.Yb[((long)s * t.nbf + sfi) * QN + col] = shas ? zb[ct] : 0.0;
            a.Dp[((long)s * t.nbf + sfi) * QN + col] = shas ? t.volume * dsel * z1r[ct] : 0.0;
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const double a1 = t.volume * z1[rt];
#pragma unroll
        for (int ct = rt; ct < CT; ++ct) {
          acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(re[rt], zb[ct], acc[rt][ct], 0, 0, 0);
          accd[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, z1[ct], accd[rt][ct], 0, 0, 0);
        }
      }
      continue;
    }
    if constexpr (KIND == G_AB) {
      // G_ab[q] = sum_e V_e^T A_ab R_e = sum_e (A_ab^T V_e)^T R_e: contracted through the FOUR faces instead of the ten DoFs.
      //   W = A_ab^T V_e   [4 x N]    A operand A_ab^T (rows f), B operand the rows of V_e: 3 k-steps per row tile
      //   G += W^T R_e                 A operand = accumulator element 0 of W (lane: f = lk, i = li), B operand the four RT0 rows
      // 3 RT + RT CT = 14 MFMAs per element at config 5 instead of 28.
      const int e = ix1.e;
      const double* Lb = Lall + (long)e * LSTRIDE;
      const unsigned ex = (unsigned)e * erow, rb = (unsigned)ix1.aux[0] * ((unsigned)QN * 8u);
      bool shas;
      const int sfi = side_face_of(t, s, e, lk, shas);    // this lane's face f = lk as a side face
      double lop[3], vv[RT][3], re[CT];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lop[k] = abpad[k] ? *zero : ld8(Lb, abl[k]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if (WIDE) {
            if ((rt & 1) == 0) {
              const double2 v = ld16(Vs, ex + abv[k][rt]);
              vv[rt][k] = v.x;
              vv[rt + 1 < RT ? rt + 1 : rt][k] = v.y;
            }
          } else {
            vv[rt][k] = ld8(Vs, ex + abv[k][rt]);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if (WIDE) {
          if ((ct & 1) == 0) {
            const double2 v = ld16(Rss, rb + yc[0][ct]);
            re[ct] = v.x;
            re[ct + 1 < CT ? ct + 1 : ct] = v.y;
          }
        } else {
          re[ct] = ld8(Rss, rb + yc[0][ct]);
        }
      }
      load_idx(item0 + NW, ix2);
      ix1 = ix2;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 3; ++k) z = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[k], vv[rt][k], z, 0, 0, 0);
        if (sfi >= 0) {        // side-face factor Xab_q = (A_ab^T V_e) at the face (0 if there is no neighbour)
          const int col = WIDE ? li * RT + rt : rt * 16 + li;
          if (col < N) a.Xab[(((long)q * t.S + s) * t.nbf + sfi) * N + col] = shas ? z[0] : 0.0;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[0], re[ct], acc[rt][ct], 0, 0, 0);
      }
      continue;
    }
    // ---- operand loads of the item, all issued before the first MFMA (no arithmetic on loaded values here)
    const bool on = KIND != G_CPL || ix1.e >= 0;
    const int e = on ? ix1.e : 0;
    const double* Lb = Lall + (long)(KIND == G_CPL ? item : e) * LSTRIDE;
    unsigned eb[NG];                          // byte offset of the rows of the slot's element
    if (FACEK) {
      eb[0] = (unsigned)ix1.aux[0] * ((unsigned)QN * 8u);
    } else if (KIND == G_CPL) {
      eb[0] = (unsigned)(on ? ix1.eo : 0) * erow;
    } else {
      eb[0] = (unsigned)e * erow;
      if (KIND == G_SYS) {
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          const int ee = ix1.nb[g - 1];
          eb[g] = (unsigned)(ee < 0 ? e : ee) * erow;     // side face: no inner neighbour, its block is zero
        }
      }
    }
    const unsigned ex = (unsigned)e * erow;
    double lop[KS], yv[CT][KS], xop[RT][KR];
    int dnode[NCK ? KS : 1];
    if (NCK) {
#pragma unroll
      for (int k = 0; k < KS; ++k) dnode[NCK ? k : 0] = ix1.auy[NCK ? k : 0];
    }
    double yav[NCK ? CT : 1][NCK ? KS : 1], xav[NCK ? RT : 1][NCK ? KR : 1];
    int bslot[NCK ? KR : 1];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      unsigned eo = eb[0];
      if (NG > 1) {
        if (k < 2 * NG) eo = eb[k >> 1];
        else eo = (lk < 2 || 2 * (k - 2 * NG) + 1 >= NG) ? eb[2 * (k - 2 * NG)] : eb[(2 * (k - 2 * NG) + 1) % NG];
      }
      if (WIDE && KY != 4 && k < 2 * NG) {          // full steps: rows j = 2 lk, 2 lk + 1 of the slot in one 16-byte load
        if ((k & 1) == 0) {
          const double2 v = on ? ld16(Lb, lc[k]) : *reinterpret_cast<const double2*>(t.zeros);
          lop[k] = v.x;
          lop[k + 1 < KS ? k + 1 : k] = v.y;
        }
      } else {
        lop[k] = (pad[k] || !on) ? *zero : ld8(Lb, lc[k]);
      }
      unsigned nd = 0;
      if (NCK) nd = (unsigned)dnode[NCK ? k : 0] * rowb;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if (WIDE) {
          if ((ct & 1) == 0) {
            const double2 v = ld16(FACEK ? Rss : Yb, eo + yc[k][ct]);
            yv[ct][k] = v.x;
            yv[ct + 1 < CT ? ct + 1 : ct][k] = v.y;
            if (NCK) {
              const double2 w = ld16(Av, nd + 8u * (yc[k][ct] % rowb / 8u));
              yav[NCK ? ct : 0][NCK ? k : 0] = w.x;
              yav[NCK ? (ct + 1 < CT ? ct + 1 : ct) : 0][NCK ? k : 0] = w.y;
            }
          }
        } else {
          yv[ct][k] = ld8(FACEK ? Rss : Yb, eo + yc[k][ct]);
          if (NCK) yav[NCK ? ct : 0][NCK ? k : 0] = ld8(Av, nd + 8u * (yc[k][ct] % rowb / 8u));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      unsigned nd = 0;
      if (NCK) {
        nd = (unsigned)ix1.aux[r] * rowb;
        bslot[NCK ? r : 0] = t.dof_bslot[e * 10 + (4 * r + lk < MZ ? 4 * r + lk : MZ - 1)];
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const unsigned off = (KIND == G_BB ? eb[0] : ex) + xc[r][rt];
        if (WIDE) {
          if ((rt & 1) == 0) {
            const double2 v = xin[r] ? ld16(KIND == G_BB ? Rss : Vs, off) : *reinterpret_cast<const double2*>(t.zeros);
            xop[rt][r] = v.x;
            xop[rt + 1 < RT ? rt + 1 : rt][r] = v.y;
            if (NCK) {
              const double2 w = xin[r] ? ld16(Av, nd + 8u * (xc[r][rt] % rowb / 8u)) : *reinterpret_cast<const double2*>(t.zeros);
              xav[NCK ? rt : 0][NCK ? r : 0] = w.x;
              xav[NCK ? (rt + 1 < RT ? rt + 1 : rt) : 0][NCK ? r : 0] = w.y;
            }
          }
        } else {
          xop[rt][r] = xin[r] ? ld8(KIND == G_BB ? Rss : Vs, off) : *zero;
          if (NCK) xav[NCK ? rt : 0][NCK ? r : 0] = xin[r] ? ld8(Av, nd + 8u * (xc[r][rt] % rowb / 8u)) : *zero;
        }
      }
    }
    load_idx(item0 + NW, ix2);
    ix1 = ix2;
    if (NCK) {
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) yv[ct][k] -= yav[NCK ? ct : 0][NCK ? k : 0];
#pragma unroll
      for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xop[rt][r] -= xav[NCK ? rt : 0][NCK ? r : 0];
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < KS; ++k) z = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[k], yv[ct][k], z, 0, 0, 0);
      if (NCK) {      // rows of E W_self at boundary DoFs: the side-node factors Cn are summed from them (k3_side_nc)
        const int col = WIDE ? li * CT + ct : ct * 16 + li;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
          const int row = 4 * r + lk;
          if (row < MZ && col < My && bslot[NCK ? r : 0] >= 0) a.Zb[((long)s * t.nbd + bslot[NCK ? r : 0]) * N + col] = z[r];
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < KR; ++r) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[rt][r], z[r], acc[rt][ct], 0, 0, 0);
    }
  }
  if constexpr (BBK) {
    // fixed-order sum over the waves of both accumulator sets (upper tile triangle) and of the r_fd row, then the stores with the
    // mirrored tiles
    constexpr int NTRI = RT * (RT + 1) / 2;
    double* lfd = lds + 2 * NTRI * 256;
    for (int w = NW - 1; w > 0; --w) {
      __syncthreads();
      if (wave == w) {
        int tix = 0;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = i; j < CT; ++j, ++tix)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              lds[(tix * 4 + r) * 64 + lane] = acc[i][j][r];
              lds[((NTRI + tix) * 4 + r) * 64 + lane] = accd[i][j][r];
            }
#pragma unroll
        for (int j = 0; j < CT; ++j) lfd[j * 64 + lane] = fd[j];
      }
      __syncthreads();
      if (wave == w - 1) {
        int tix = 0;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = i; j < CT; ++j, ++tix)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[i][j][r] += lds[(tix * 4 + r) * 64 + lane];
              accd[i][j][r] += lds[((NTRI + tix) * 4 + r) * 64 + lane];
            }
#pragma unroll
        for (int j = 0; j < CT; ++j) fd[j] += lfd[j * 64 + lane];
      }
    }
    if (wave == 0 && P) {                  // K-split: both matrices in full (mirrored) and the r_fd row as a partial result
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = i; j < CT; ++j) {
          const int col = WIDE ? li * CT + j : j * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = WIDE ? (lk + 4 * r) * RT + i : i * 16 + lk + 4 * r;
            if (row < Mx && col < My) {
              P[(long)row * My + col] = acc[i][j][r];
              P[(long)QN * QN + (long)row * My + col] = accd[i][j][r];
              if (i != j) {
                P[(long)col * My + row] = acc[i][j][r];
                P[(long)QN * QN + (long)col * My + row] = accd[i][j][r];
              }
            }
          }
        }
      if (lk == 0) {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int col = WIDE ? li * CT + j : j * 16 + li;
          if (col < My) P[2L * QN * QN + col] = fd[j];
        }
      }
      return;
    }
    if (wave == 0) {
      double* o2 = a.out2 + (long)s * QN * QN;
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = i; j < CT; ++j) {
          const int col = WIDE ? li * CT + j : j * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = WIDE ? (lk + 4 * r) * RT + i : i * 16 + lk + 4 * r;
            if (row < Mx && col < My) {
              out[(long)row * My + col] = acc[i][j][r];
              o2[(long)row * My + col] = accd[i][j][r];
              if (i != j) {
                out[(long)col * My + row] = acc[i][j][r];
                o2[(long)col * My + row] = accd[i][j][r];
              }
            }
          }
        }
      if (lk == 0) {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int col = WIDE ? li * CT + j : j * 16 + li;
          if (col < My) a.out3[(long)s * QN + col] = fd[j];
        }
      }
    }
    return;
  }
  // ---- fixed-order sum over the waves: acc_0 + (acc_1 + (... + acc_{NW-1}))
  for (int w = NW - 1; w > 0; --w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[((i * CT + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wave == w - 1) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += lds[((i * CT + j) * 4 + r) * 64 + lane];
    }
  }
  if constexpr (KIND == G_SYS) {       // B = H + H^T through the LDS (Mx My <= RT CT 256 doubles)
    if (q == 0) {                      // rhs_red: sum over the four k lanes, then over the waves in a fixed order
      __shared__ double rsh[16][RT * 16];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        double v = racc[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lk == 0) rsh[wave][i * 16 + li] = v;
      }
      __syncthreads();
      if (wave == 0 && lk == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          double v = 0.0;
          for (int ww = 0; ww < NW; ++ww) v += rsh[ww][i * 16 + li];
          const int col = WIDE ? li * RT + i : i * 16 + li;
          if (col < N) (P ? P[(long)N * N + col] : a.out4[(long)s * N + col]) = v;
        }
      }
    }
    if (P) {                                // K-split: H itself is the partial result, the combine kernel forms H + H^T
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const int col = WIDE ? li * CT + j : j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = WIDE ? (lk + 4 * r) * RT + i : i * 16 + lk + 4 * r;
              if (row < Mx && col < My) P[(long)row * My + col] = acc[i][j][r];
            }
          }
      }
      return;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int col = WIDE ? li * CT + j : j * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = WIDE ? (lk + 4 * r) * RT + i : i * 16 + lk + 4 * r;
            if (row < Mx && col < My) lds[row * My + col] = acc[i][j][r];
          }
        }
    }
    __syncthreads();
    for (int i = tid; i < Mx * My; i += blockDim.x) {
      const int row = i / My, col = i - row * My;
      out[i] = lds[i] + lds[col * My + row];
    }
    return;
  }
  if (wave == 0 && P) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int col = WIDE ? li * CT + j : j * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = WIDE ? (lk + 4 * r) * RT + i : i * 16 + lk + 4 * r;
          if (row < Mx && col < My) P[(long)row * My + col] = acc[i][j][r];
        }
      }
    return;
  }
  if (wave == 0) {
    double* mirror = nullptr;       // AAA: block (q2, q) = transpose of block (q, q2)
    if (KIND == G_AAA && q != q2) mirror = a.out + (((long)q2 * Q + q) * t.S + s) * N * N;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int col = WIDE ? li * CT + j : j * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = WIDE ? (lk + 4 * r) * RT + i : i * 16 + lk + 4 * r;
          if (row < Mx && col < My) {
            out[(long)row * My + col] = acc[i][j][r];
            if (mirror) mirror[(long)col * My + row] = acc[i][j][r];
          }
        }
      }
  }
}

// K-split: sum of the partial results of a (subdomain, operator) in the fixed order part = 0 .. ksplit - 1, then the epilogue of
// the kind (B_sys = H + H^T and rhs_red; mirrored A_aa block; G_bb, G_rdd, r_fd)
template <int KIND>
__global__ __launch_bounds__(256) void k3_pg_combine(GA a) {
  PGBlock<KIND> B;
  GA a1 = a;
  a1.ksplit = 1;
  if (!B.decode(a1, blockIdx.x)) return;
  const T3& t = a.t;
  const int N = a.N, Q = a.Q, QN = Q * N, ks = a.ksplit;
  if (KIND == G_CPL && B.t2 < 0) return;
  const long psz = pg_part_size<KIND>(N, QN);
  const double* P = a.part + (long)B.b * ks * psz;
  const int Mx = B.Mx, My = B.My, nent = Mx * My;
  for (int i = threadIdx.x; i < nent; i += 256) {
    double v = 0.0;
    for (int p = 0; p < ks; ++p) v += P[p * psz + i];
    if (KIND == G_SYS) {
      const int row = i / My, col = i - row * My;
      double vt = 0.0;
      for (int p = 0; p < ks; ++p) vt += P[p * psz + (long)col * My + row];
      B.out[i] = v + vt;
    } else {
      B.out[i] = v;
      if (KIND == G_AAA && B.q != B.q2) {
        const int row = i / My, col = i - row * My;
        a.out[(((long)B.q2 * Q + B.q) * t.S + B.s) * N * N + (long)col * My + row] = v;
      }
      if (KIND == G_BB) {
        double v2 = 0.0;
        for (int p = 0; p < ks; ++p) v2 += P[p * psz + (long)QN * QN + i];
        a.out2[(long)B.s * QN * QN + i] = v2;
      }
    }
  }
  if (KIND == G_SYS && B.q == 0)
    for (int i = threadIdx.x; i < N; i += 256) {
      double v = 0.0;
      for (int p = 0; p < ks; ++p) v += P[p * psz + (long)N * N + i];
      a.out4[(long)B.s * N + i] = v;
    }
  if (KIND == G_BB)
    for (int i = threadIdx.x; i < QN; i += 256) {
      double v = 0.0;
      for (int p = 0; p < ks; ++p) v += P[p * psz + 2L * QN * QN + i];
      a.out3[(long)B.s * QN + i] = v;
    }
}

template <int KIND, int RT, int CT>
void launch_pg(const GA& a, int batch, int nw, hipStream_t st) {
  const int batch1 = batch / a.t.S * ((a.t.S + 7) / 8 * 8);     // whole chunks of 8 subdomains (PGBlock: XCD-aware block ids)
  batch = batch1 * a.ksplit;
  constexpr int maxt = pg_max_threads(KIND == G_BB ? (RT > 2 ? 8 : 4) : RT * CT * (KIND == G_NC || KIND == G_SYS ? 2 : 1));   // NC also holds the node averages of its operands; BB: two upper tile triangles
  if (nw * 64 > maxt) nw = maxt / 64;
  constexpr bool EVEN = RT % 2 == 0 && CT % 2 == 0;
  const size_t ldsb = sizeof(double) * (KIND == G_BB ? RT * (RT + 1) * 256 + CT * 64 : RT * CT * 256);   // BB: two upper triangles + r_fd
  if (EVEN && a.N % 2 == 0)
    hipLaunchKernelGGL((k3_pg<KIND, RT, CT, EVEN>), dim3(batch), dim3(64 * nw), ldsb, st, a);
  else
    hipLaunchKernelGGL((k3_pg<KIND, RT, CT, false>), dim3(batch), dim3(64 * nw), ldsb, st, a);
  if (a.ksplit > 1) hipLaunchKernelGGL((k3_pg_combine<KIND>), dim3(batch1), dim3(256), 0, st, a);
}

// tile shapes: square (rt == ct) for everything but AB, where ct = tiles of Q N >= rt = tiles of N
template <int KIND>
int dispatch_pg(const GA& a, int batch, int rt, int ct, int nw, hipStream_t st) {
#define PGCASE(R, C)                                 \
  if (rt == R && ct == C) {                          \
    launch_pg<KIND, R, C>(a, batch, nw, st);         \
    return 0;                                        \
  }
  if constexpr (KIND != G_AB) { PGCASE(1, 1) PGCASE(2, 2) PGCASE(3, 3) PGCASE(4, 4) }
  if constexpr (KIND == G_AB) { PGCASE(1, 1) PGCASE(1, 2) PGCASE(1, 3) PGCASE(1, 4) PGCASE(2, 2) PGCASE(2, 3) PGCASE(2, 4) PGCASE(3, 3) PGCASE(3, 4) PGCASE(4, 4) }
#undef PGCASE
  return -1;
}

// Cn [S][nb][N] = -(P^T E W_self) at the boundary nodes: sum of the rows k3_pg<NC> left in Zb over the DoFs of the node.
// One wave per node, its slot list through the scalar cache.
__global__ __launch_bounds__(256) void k3_side_nc(T3 t, int N, const double* __restrict__ Zb, double* __restrict__ Cn) {
  const int s = blockIdx.y, c = threadIdx.x & 63;
  const int bn = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (bn >= t.nb) return;
  const int cc = c < N ? c : N - 1;
  const double* z = Zb + (long)s * t.nbd * N + cc;
  const int p0 = t.bn_ptr[bn], p1 = t.bn_ptr[bn + 1];
  double a0 = 0.0, a1 = 0.0;
  int p = p0;
  for (; p + 1 < p1; p += 2) {
    a0 += z[(long)t.bn_slots[p] * N];
    a1 += z[(long)t.bn_slots[p + 1] * N];
  }
  if (p < p1) a0 += z[(long)t.bn_slots[p] * N];
  if (c < N) Cn[((long)s * t.nb + bn) * N + c] = -(a0 + a1);
}

// ------------------------------------------------------------------------------------------------- online: estimate
struct QV { double v[8]; };

__device__ inline double block_sum(double v, double* red) {   // fixed-order tree over 256 threads; red [256]
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  return red[0];
}

struct EA {
  const double *u, *G_nc, *G_bb, *G_rdd, *G_ab, *G_aa, *r_fd, *Rb, *Yb, *Dp, *Xab, *As, *Cn, *ebar, *Bbb, *bdiv, *f2, *ceps;
  double hdiam;
  double* eta;
};

__global__ __launch_bounds__(256) void k3_estimate(T3 t, int Q, int N, QV th, EA a) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, QN = Q * N;
  double* us = lds;                 // [7][N] coefficients on the neighbourhood (0 where no neighbour)
  double* ur = us + 7 * N;          // [QN]   theta_q u_self
  double* zf = ur + QN;             // [nbf]  flux of the neighbours on the side faces
  double* z = zf + t.nbf;           // [nb]   neighbours' share of the node averages
  double* red = z + t.nb;           // [256]
  for (int i = tid; i < 7 * N; i += 256) {
    const int s2 = t.nbr[s * 7 + i / N];
    us[i] = s2 >= 0 ? a.u[(long)s2 * N + i % N] : 0.0;
  }
  __syncthreads();
  const double* u0 = us + 3 * N;
  for (int i = tid; i < QN; i += 256) ur[i] = th.v[i / N] * u0[i % N];
  for (int sf = tid; sf < t.nbf; sf += 256) {
    const double* ua = us + side_slot(sf / t.ncf) * N;
    const double* r = a.Rb + ((long)s * t.nbf + sf) * QN;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) {
      double aq = 0.0;
      for (int j = 0; j < N; ++j) aq += r[q * N + j] * ua[j];
      acc += th.v[q] * aq;
    }
    zf[sf] = acc;
  }
  for (int bn = tid; bn < t.nb; bn += 256) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) {
      const int sp = t.bnode_sides[bn * 3 + k];
      if (sp < 0) continue;
      const double* ua = us + side_slot(sp / t.nvs) * N;
      const double* r = a.As + ((long)s * 6 * t.nvs + sp) * N;
      for (int j = 0; j < N; ++j) acc += r[j] * ua[j];
    }
    z[bn] = acc;
  }
  __syncthreads();
  // ---- nonconformity
  double p_nc = 0.0;
  for (int r = tid; r < N; r += 256) {
    const double* g = a.G_nc + ((long)s * N + r) * N;
    double d = 0.0;
    for (int j = 0; j < N; ++j) d += g[j] * u0[j];
    p_nc += u0[r] * d;
  }
  for (int bn = tid; bn < t.nb; bn += 256) {
    const double* g = a.Cn + ((long)s * t.nb + bn) * N;
    double d = 0.0;
    for (int j = 0; j < N; ++j) d += g[j] * u0[j];
    p_nc += 2.0 * z[bn] * d;
  }
  for (int k = tid; k < t.nbel; k += 256) {
    const int e = t.bel_elem[k];
    const double* E = a.ebar + ((long)s * t.nT + e) * 100;
    double ze[10];
    for (int i = 0; i < 10; ++i) {
      const int bn = t.bel_bnode[k * 10 + i];
      ze[i] = bn >= 0 ? z[bn] : 0.0;
    }
    for (int i = 0; i < 10; ++i) {
      if (ze[i] == 0.0) continue;
      double d = 0.0;
      for (int j = 0; j < 10; ++j) d += E[i * 10 + j] * ze[j];
      p_nc += ze[i] * d;
    }
  }
  const double nc = block_sum(p_nc, red);
  // ---- flux terms
  double p_bb = 0.0, p_dd = 0.0, p_fd = 0.0, p_ab = 0.0, p_aa = 0.0;
  for (int r = tid; r < QN; r += 256) {
    const double* gb = a.G_bb + ((long)s * QN + r) * QN;
    const double* gd = a.G_rdd + ((long)s * QN + r) * QN;
    double db = 0.0, dd = 0.0;
    for (int j = 0; j < QN; ++j) {
      db += gb[j] * ur[j];
      dd += gd[j] * ur[j];
    }
    p_bb += ur[r] * db;
    p_dd += ur[r] * dd;
    p_fd += a.r_fd[(long)s * QN + r] * ur[r];
  }
  for (int r = tid; r < N; r += 256) {
    for (int q = 0; q < Q; ++q) {
      const double* g = a.G_ab + (((long)q * t.S + s) * N + r) * QN;
      double d = 0.0;
      for (int j = 0; j < QN; ++j) d += g[j] * ur[j];
      p_ab += th.v[q] * u0[r] * d;
      for (int q2 = 0; q2 < Q; ++q2) {
        const double* ga = a.G_aa + ((((long)q * Q + q2) * t.S + s) * N + r) * N;
        double da = 0.0;
        for (int j = 0; j < N; ++j) da += ga[j] * u0[j];
        p_aa += th.v[q] * th.v[q2] * u0[r] * da;
      }
    }
  }
  for (int sf = tid; sf < t.nbf; sf += 256) {
    const double zz = zf[sf];
    if (zz == 0.0) continue;
    const double* yb = a.Yb + ((long)s * t.nbf + sf) * QN;
    const double* dp = a.Dp + ((long)s * t.nbf + sf) * QN;
    double db = 0.0, dd = 0.0;
    for (int j = 0; j < QN; ++j) {
      db += yb[j] * ur[j];
      dd += dp[j] * ur[j];
    }
    p_bb += 2.0 * zz * db;
    p_dd += 2.0 * zz * dd;
    for (int q = 0; q < Q; ++q) {
      const double* xa = a.Xab + (((long)q * t.S + s) * t.nbf + sf) * N;
      double d = 0.0;
      for (int j = 0; j < N; ++j) d += xa[j] * u0[j];
      p_ab += th.v[q] * zz * d;
    }
  }
  for (int k = tid; k < t.nsel; k += 256) {
    const int e = t.sel_elem[k], ty = t.elem_type[e];
    const double* B = a.Bbb + ((long)s * t.nT + e) * 16;
    double ze[4], dv = 0.0;
    for (int f = 0; f < 4; ++f) {
      const int sf = t.sel_sf[k * 4 + f];
      ze[f] = sf >= 0 ? zf[sf] : 0.0;
      dv += sgn3(t, s, e, f) * t.divc[ty * 4 + f] * ze[f];
    }
    for (int f = 0; f < 4; ++f)
      for (int g = 0; g < 4; ++g) p_bb += ze[f] * B[f * 4 + g] * ze[g];
    p_dd += t.volume * dv * dv;
    p_fd += a.bdiv[(long)s * t.nT + e] * dv;
  }
  const double bb = block_sum(p_bb, red);
  const double dd = block_sum(p_dd, red);
  const double fd = block_sum(p_fd, red);
  const double ab = block_sum(p_ab, red);
  const double aa = block_sum(p_aa, red);
  if (tid == 0) {
    const double pi = 3.14159265358979323846;
    a.eta[s] = nc;
    a.eta[t.S + s] = (a.f2[s] - 2.0 * fd + dd) * (1.0 / (pi * pi)) / a.ceps[s] * a.hdiam * a.hdiam;
    a.eta[2 * t.S + s] = bb + 2.0 * ab + aa;
  }
}

// Batched form: MB = 8 parameters per workgroup pass.  Threads = (worker w = tid >> 3, parameter m = tid & 7): the eight
// parameter lanes of a worker read the same factor row (one address, broadcast) and their own coefficient column from LDS, so
// every projected operator and factor is read once per batch.  u [S_ext][N][nmu] (parameter fastest), eta [3][S][nmu].
constexpr int EST_MB = 8;
struct TB8 { double v[EST_MB][8]; };

// sum_j g[j] * u[j * 8]: a row of a projected operator (global) against a coefficient column in the LDS (parameter-fastest, 8 per
// row), for the eight parameter lanes of a worker (consecutive lanes, same g, same control flow).  The lanes would all load the
// same address -- 8 useful bytes per lane-group and instruction, and the address unit, not the memory, sets the pace -- so lane m
// loads entry j0 + m of every 8-entry chunk (all chunks first: n <= 64) and the group passes the values around by lane shuffles.
__device__ inline double dot_row8(const double* __restrict__ g, const double* u, int n) {
  const int lane = threadIdx.x & 63, m = lane & 7, base = lane & ~7;
  double mine[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) mine[c] = (c * 8 < n && c * 8 + m < n) ? g[c * 8 + m] : 0.0;
  double d = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c * 8 >= n) break;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double gk = __shfl(mine[c], base + k);
      if (c * 8 + k < n) d += gk * u[(c * 8 + k) * 8];
    }
  }
  return d;
}

__global__ __launch_bounds__(256) void k3_estimate_batch(T3 t, int Q, int N, int nmu, int m0, TB8 th, EA a) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, w = tid >> 3, m = tid & 7, QN = Q * N, NWK = 32;
  const int mb = nmu - m0 < EST_MB ? nmu - m0 : EST_MB;      // parameters of this pass
  double* us = lds;                       // [7][N][8]
  double* ur = us + 7 * N * 8;            // [QN][8]
  double* zf = ur + QN * 8;               // [nbf][8]
  double* z = zf + t.nbf * 8;             // [nb][8]
  double* red = z + t.nb * 8;             // [256]
  for (int i = tid; i < 7 * N * 8; i += 256) {
    const int mm = i & 7, row = i >> 3, s2 = t.nbr[s * 7 + row / N];
    us[i] = (s2 >= 0 && mm < mb) ? a.u[((long)s2 * N + row % N) * nmu + m0 + mm] : 0.0;
  }
  __syncthreads();
  const double* u0 = us + 3 * N * 8 + m;          // own coefficients of parameter m: u0[j * 8]
  for (int i = tid; i < QN * 8; i += 256) {
    const int mm = i & 7, c = i >> 3;
    ur[i] = th.v[mm][c / N] * us[(3 * N + c % N) * 8 + mm];
  }
  for (int sf = w; sf < t.nbf; sf += NWK) {
    const double* ua = us + side_slot(sf / t.ncf) * N * 8 + m;
    const double* r = a.Rb + ((long)s * t.nbf + sf) * QN;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) {
      double aq = 0.0;
      aq = dot_row8(r + q * N, ua, N);
      acc += th.v[m][q] * aq;
    }
    zf[sf * 8 + m] = acc;
  }
  for (int bn = w; bn < t.nb; bn += NWK) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) {
      const int sp = t.bnode_sides[bn * 3 + k];
      if (sp < 0) continue;
      const double* ua = us + side_slot(sp / t.nvs) * N * 8 + m;
      const double* r = a.As + ((long)s * 6 * t.nvs + sp) * N;
      acc += dot_row8(r, ua, N);
    }
    z[bn * 8 + m] = acc;
  }
  __syncthreads();
  const double* urm = ur + m;
  double p_nc = 0.0, p_bb = 0.0, p_dd = 0.0, p_fd = 0.0, p_ab = 0.0, p_aa = 0.0;
  for (int r = w; r < N; r += NWK) {
    const double* g = a.G_nc + ((long)s * N + r) * N;
    const double d = dot_row8(g, u0, N);
    p_nc += u0[r * 8] * d;
    for (int q = 0; q < Q; ++q) {
      const double* gab = a.G_ab + (((long)q * t.S + s) * N + r) * QN;
      const double dab = dot_row8(gab, urm, QN);
      p_ab += th.v[m][q] * u0[r * 8] * dab;
      for (int q2 = 0; q2 < Q; ++q2) {
        const double* ga = a.G_aa + ((((long)q * Q + q2) * t.S + s) * N + r) * N;
        const double da = dot_row8(ga, u0, N);
        p_aa += th.v[m][q] * th.v[m][q2] * u0[r * 8] * da;
      }
    }
  }
  for (int bn = w; bn < t.nb; bn += NWK) {
    const double* g = a.Cn + ((long)s * t.nb + bn) * N;
    const double d = dot_row8(g, u0, N);
    p_nc += 2.0 * z[bn * 8 + m] * d;
  }
  for (int k = w; k < t.nbel; k += NWK) {
    const int e = t.bel_elem[k];
    const double* E = a.ebar + ((long)s * t.nT + e) * 100;
    double ze[10];
    for (int i = 0; i < 10; ++i) {
      const int bn = t.bel_bnode[k * 10 + i];
      ze[i] = bn >= 0 ? z[bn * 8 + m] : 0.0;
    }
    for (int i = 0; i < 10; ++i) {
      double d = 0.0;
      for (int j = 0; j < 10; ++j) d += E[i * 10 + j] * ze[j];
      p_nc += ze[i] * d;
    }
  }
  for (int r = w; r < QN; r += NWK) {
    const double* gb = a.G_bb + ((long)s * QN + r) * QN;
    const double* gd = a.G_rdd + ((long)s * QN + r) * QN;
    const double db = dot_row8(gb, urm, QN), dd = dot_row8(gd, urm, QN);
    p_bb += urm[r * 8] * db;
    p_dd += urm[r * 8] * dd;
    p_fd += a.r_fd[(long)s * QN + r] * urm[r * 8];
  }
  for (int sf = w; sf < t.nbf; sf += NWK) {
    const double zz = zf[sf * 8 + m];
    const double* yb = a.Yb + ((long)s * t.nbf + sf) * QN;
    const double* dp = a.Dp + ((long)s * t.nbf + sf) * QN;
    const double db = dot_row8(yb, urm, QN), dd = dot_row8(dp, urm, QN);
    p_bb += 2.0 * zz * db;
    p_dd += 2.0 * zz * dd;
    for (int q = 0; q < Q; ++q) {
      const double* xa = a.Xab + (((long)q * t.S + s) * t.nbf + sf) * N;
      const double d = dot_row8(xa, u0, N);
      p_ab += th.v[m][q] * zz * d;
    }
  }
  for (int k = w; k < t.nsel; k += NWK) {
    const int e = t.sel_elem[k], ty = t.elem_type[e];
    const double* B = a.Bbb + ((long)s * t.nT + e) * 16;
    double ze[4], dv = 0.0;
    for (int f = 0; f < 4; ++f) {
      const int sf = t.sel_sf[k * 4 + f];
      ze[f] = sf >= 0 ? zf[sf * 8 + m] : 0.0;
      dv += sgn3(t, s, e, f) * t.divc[ty * 4 + f] * ze[f];
    }
    for (int f = 0; f < 4; ++f)
      for (int g = 0; g < 4; ++g) p_bb += ze[f] * B[f * 4 + g] * ze[g];
    p_dd += t.volume * dv * dv;
    p_fd += a.bdiv[(long)s * t.nT + e] * dv;
  }
  // sums over the 32 workers per parameter, fixed order
  double tot[6];
  const double part[6] = {p_nc, p_bb, p_dd, p_fd, p_ab, p_aa};
  for (int k = 0; k < 6; ++k) {
    __syncthreads();
    red[tid] = part[k];
    __syncthreads();
    double acc = 0.0;
    if (tid < 8)
      for (int ww = 0; ww < NWK; ++ww) acc += red[ww * 8 + tid];
    tot[k] = acc;
  }
  if (tid < mb) {
    const double pi = 3.14159265358979323846;
    const long o = (long)s * nmu + m0 + tid;
    a.eta[o] = tot[0];
    a.eta[(long)t.S * nmu + o] = (a.f2[s] - 2.0 * tot[3] + tot[2]) * (1.0 / (pi * pi)) / a.ceps[s] * a.hdiam * a.hdiam;
    a.eta[2L * t.S * nmu + o] = tot[1] + 2.0 * tot[4] + tot[5];
  }
}

// ---- the same estimate for 16 parameters per pass on the matrix cores.  Every dense term is x^T (G y) with G a projected operator
// or a factor (rows x cols, row-major in global memory) and x, y coefficient panels [.][16] in the LDS (one column per parameter):
// a wave takes 16-row strips of G, T = G y is one accumulator tile (A operand: the strip straight from global memory, lane =
// (row l & 15, k l >> 4), four k-steps of loads in flight; B operand: y rows from the LDS, lane = (parameter l & 15, k)), and the
// lane sums its four rows of x . T for ITS parameter.  G is read once per pass for all 16 parameters with full-width loads (the
// VALU kernel above: eight lanes per loaded entry).  The element-local quadratic terms and the node sums stay on the VALU.
constexpr int EST16 = 16;
struct TB16 { double v[EST16][8]; };

// tile of T = G y for the 16-row strip at row0 (rows clamped into [0, rows)); y [cols][16] in the LDS; cols <= 64.  All loads of
// the strip are issued before the first MFMA (a load -> MFMA chain per k-step exposes one memory round trip each).
__device__ inline d4 est_strip(const double* __restrict__ G, int ld, int row0, int rows, int cols, const double* y, int li, int lk) {
  const int row = row0 + li < rows ? row0 + li : rows - 1;
  const double* g = G + (long)row * ld;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  for (int c0 = 0; c0 < cols; c0 += 32) {                            // eight k-steps of loads in flight (sixteen spill at 128 VGPRs)
    double av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + 4 * u + lk;
      av[u] = c0 + 4 * u < cols ? g[c < cols ? c : cols - 1] : 0.0;  // (c0 + 4 u < cols: wave-uniform)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c0 + 4 * u < cols) {
        const int c = c0 + 4 * u + lk;
        const double bv = c < cols ? y[c * 16 + li] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv, acc, 0, 0, 0);
      }
  }
  return acc;
}

// sum_rows x[row][m] T[row][m] over the strip's valid rows for this lane's parameter
__device__ inline double est_dot(const d4& T, const double* x, int row0, int rows, int li, int lk) {
  double v = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + lk + 4 * r;
    if (row < rows) v += x[row * 16 + li] * T[r];
  }
  return v;
}

// sum_j g[j] u[j * 16] for the 16 parameter lanes of a worker (consecutive lanes, same g, same control flow; n <= 64): lane m loads
// entry 16 c + m of every chunk, the group passes the values around by lane shuffles (see dot_row8)
__device__ inline double dot_row16(const double* __restrict__ g, const double* u, int n) {
  const int lane = threadIdx.x & 63, m = lane & 15, base = lane & ~15;
  double mine[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) mine[c] = (c * 16 < n && c * 16 + m < n) ? g[c * 16 + m] : 0.0;
  double d = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c * 16 >= n) break;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double gk = __shfl(mine[c], base + k);
      if (c * 16 + k < n) d += gk * u[(c * 16 + k) * 16];
    }
  }
  return d;
}

constexpr int EST_NW = 16;     // waves per workgroup of k3_estimate_batch16 (<= 128 VGPRs: sixteen k-steps of loads per strip would spill)

__global__ __launch_bounds__(64 * EST_NW) void k3_estimate_batch16(T3 t, int Q, int N, int nmu, int m0, TB16 th, EA a) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4, QN = Q * N;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = tid >> 4, m = tid & 15, NWK = 4 * EST_NW;     // VALU parts: workers x 16 parameters
  const int mb = nmu - m0 < EST16 ? nmu - m0 : EST16;
  double* us = lds;                        // [7][N][16]  coefficients of the neighbourhood
  double* ur = us + 7 * N * 16;            // [QN][16]    theta_q u_own
  double* zf = ur + QN * 16;               // [nbf][16]   neighbours' flux traces at the side faces
  double* z = zf + t.nbf * 16;             // [nb][16]    neighbours' node averages at the boundary nodes
  double* red = z + t.nb * 16;             // [EST_NW][6][16] parts of the waves + [NWK][16] r_fd parts of the workers
  for (int i = tid; i < 7 * N * 16; i += 64 * EST_NW) {
    const int mm = i & 15, row = i >> 4, s2 = t.nbr[s * 7 + row / N];
    us[i] = (s2 >= 0 && mm < mb) ? a.u[((long)s2 * N + row % N) * nmu + m0 + mm] : 0.0;
  }
  __syncthreads();
  const double* u0p = us + 3 * N * 16;     // own coefficients [N][16]
  for (int i = tid; i < QN * 16; i += 64 * EST_NW) {
    const int mm = i & 15, c = i >> 4;
    ur[i] = th.v[mm][c / N] * u0p[(c % N) * 16 + mm];
  }
  double thq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) thq[q] = q < Q ? th.v[li][q] : 0.0;
  // the strips of a term are dealt to the waves round-robin, every term starting one wave further (terms have 2 .. 25 strips)
  int rot = 0;
  auto first_strip = [&]() { return 16 * ((wave + EST_NW - (rot++ % EST_NW)) % EST_NW); };
  // zf = sum_q theta_q Rb_q u_a, side by side (the rows of a side read the coefficients of ONE neighbour)
  for (int side = 0; side < 6; ++side) {
    const double* ua = us + side_slot(side) * N * 16;
    for (int row0 = first_strip(); row0 < t.ncf; row0 += 16 * EST_NW) {
      d4 zt = (d4){0.0, 0.0, 0.0, 0.0};
      for (int q = 0; q < Q; ++q) {
        const d4 T = est_strip(a.Rb + ((long)s * t.nbf + side * t.ncf) * QN + q * N, QN, row0, t.ncf, N, ua, li, lk);
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[r] += thq[q] * T[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + lk + 4 * r;
        if (row < t.ncf) zf[(side * t.ncf + row) * 16 + li] = zt[r];
      }
    }
  }
  // z = sum over the (<= 3) sides of a boundary node of As u_a: side by side as MFMA strips over the side's nodes, each tile added to
  // the rows of its boundary nodes (a side holds a node once: no two lanes meet; the sides follow each other, so the order of the
  // additions is fixed).  (One worker per node with its rows loaded cooperatively: 78 us of 246 -- chains of index -> row -> 30
  // shuffles per node.)
  int* sp2bn = reinterpret_cast<int*>(red + EST_NW * 6 * 16 + 4 * EST_NW * 16);      // [6 nvs] side node -> boundary node
  for (int i = tid; i < 6 * t.nvs; i += 64 * EST_NW) sp2bn[i] = -1;
  for (int i = tid; i < t.nb * 16; i += 64 * EST_NW) z[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < 3 * t.nb; i += 64 * EST_NW) {
    const int sp = t.bnode_sides[i];
    if (sp >= 0) sp2bn[sp] = i / 3;
  }
  __syncthreads();
  for (int side = 0; side < 6; ++side) {
    const double* ua = us + side_slot(side) * N * 16;
    for (int row0 = first_strip(); row0 < t.nvs; row0 += 16 * EST_NW) {
      const d4 T = est_strip(a.As + ((long)s * 6 * t.nvs + side * t.nvs) * N, N, row0, t.nvs, N, ua, li, lk);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + lk + 4 * r;
        const int bn = row < t.nvs ? sp2bn[side * t.nvs + row] : -1;
        if (bn >= 0) z[bn * 16 + li] += T[r];
      }
    }
    __syncthreads();
  }
  double p_nc = 0.0, p_bb = 0.0, p_dd = 0.0, p_fd = 0.0, p_ab = 0.0, p_aa = 0.0;      // per lane: its parameter, part of the rows
  // ---- dense terms on the matrix cores
  for (int row0 = first_strip(); row0 < N; row0 += 16 * EST_NW) {
    p_nc += est_dot(est_strip(a.G_nc + (long)s * N * N, N, row0, N, N, u0p, li, lk), u0p, row0, N, li, lk);
    for (int q = 0; q < Q; ++q) {
      p_ab += thq[q] * est_dot(est_strip(a.G_ab + ((long)q * t.S + s) * N * QN, QN, row0, N, QN, ur, li, lk), u0p, row0, N, li, lk);
      for (int q2 = 0; q2 < Q; ++q2)
        p_aa += thq[q] * thq[q2] *
                est_dot(est_strip(a.G_aa + (((long)q * Q + q2) * t.S + s) * N * N, N, row0, N, N, u0p, li, lk), u0p, row0, N, li, lk);
    }
  }
  for (int row0 = first_strip(); row0 < t.nb; row0 += 16 * EST_NW)
    p_nc += 2.0 * est_dot(est_strip(a.Cn + (long)s * t.nb * N, N, row0, t.nb, N, u0p, li, lk), z, row0, t.nb, li, lk);
  for (int row0 = first_strip(); row0 < QN; row0 += 16 * EST_NW) {
    p_bb += est_dot(est_strip(a.G_bb + (long)s * QN * QN, QN, row0, QN, QN, ur, li, lk), ur, row0, QN, li, lk);
    p_dd += est_dot(est_strip(a.G_rdd + (long)s * QN * QN, QN, row0, QN, QN, ur, li, lk), ur, row0, QN, li, lk);
  }
  for (int row0 = first_strip(); row0 < t.nbf; row0 += 16 * EST_NW) {
    p_bb += 2.0 * est_dot(est_strip(a.Yb + (long)s * t.nbf * QN, QN, row0, t.nbf, QN, ur, li, lk), zf, row0, t.nbf, li, lk);
    p_dd += 2.0 * est_dot(est_strip(a.Dp + (long)s * t.nbf * QN, QN, row0, t.nbf, QN, ur, li, lk), zf, row0, t.nbf, li, lk);
    for (int q = 0; q < Q; ++q)
      p_ab += thq[q] * est_dot(est_strip(a.Xab + ((long)q * t.S + s) * t.nbf * N, N, row0, t.nbf, N, u0p, li, lk), zf, row0, t.nbf, li, lk);
  }
  // ---- element-local quadratic terms, one element per wave at a time, on the matrix cores as well:
  // boundary elements: ze^T E ze with ze = z at the element's boundary nodes (0 elsewhere); T = E ze is a 10 x 16 tile, K = 10
  // (four elements per wave in flight: one at a time is a chain of memory round trips -- index, block, LDS -- per element)
  constexpr int EU = 4;
  for (int k0 = wave * EU; k0 < t.nbel; k0 += EST_NW * EU) {
    double av[EU][3], bv[EU][3], xr[EU][3];
#pragma unroll
    for (int x = 0; x < EU; ++x) {
      const int k = k0 + x < t.nbel ? k0 + x : t.nbel - 1;
      const int e = t.bel_elem[k];
      const double* E = a.ebar + ((long)s * t.nT + e) * 100 + (li < 10 ? li : 9) * 10;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int c = 4 * u + lk, cc = c < 10 ? c : 9;
        av[x][u] = E[cc];
        const int bn = t.bel_bnode[k * 10 + cc];
        bv[x][u] = (c < 10 && bn >= 0 && k0 + x < t.nbel) ? z[bn * 16 + li] : 0.0;
        xr[x][u] = bv[x][u];                                       // rows lk + 4 u of ze: the same entries as the B operand
      }
    }
#pragma unroll
    for (int x = 0; x < EU; ++x) {
      d4 T = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 3; ++u) T = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x][u], bv[x][u], T, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 3; ++r) p_nc += xr[x][r] * T[r];       // (rows >= 10 and missing nodes carry ze = 0)
    }
  }
  // side elements: ze^T B ze, the divergence of the side part and its pairings; ze = zf at the element's side faces (4 x 4 block)
  for (int k0 = wave * EU; k0 < t.nsel; k0 += EST_NW * EU) {
    double av[EU], ze[EU], cf[EU], bd[EU];
#pragma unroll
    for (int x = 0; x < EU; ++x) {
      const bool on = k0 + x < t.nsel;
      const int k = on ? k0 + x : t.nsel - 1;
      const int e = t.sel_elem[k], ty = t.elem_type[e];
      const int sf = t.sel_sf[k * 4 + lk];
      ze[x] = (on && sf >= 0) ? zf[sf * 16 + li] : 0.0;            // face lk, parameter li
      av[x] = a.Bbb[((long)s * t.nT + e) * 16 + (li < 4 ? li : 3) * 4 + lk];
      cf[x] = sgn3(t, s, e, lk) * t.divc[ty * 4 + lk];
      bd[x] = a.bdiv[(long)s * t.nT + e];
    }
#pragma unroll
    for (int x = 0; x < EU; ++x) {
      const d4 T = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], ze[x], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      p_bb += ze[x] * T[0];                                         // row lk of B ze (rows 0 .. 3 sit in register 0)
      double dv = cf[x] * ze[x];
      dv += __shfl_xor(dv, 16);
      dv += __shfl_xor(dv, 32);
      if (lk == 0) {
        p_dd += t.volume * dv * dv;
        p_fd += bd[x] * dv;
      }
    }
  }
  // sum over the four k lanes of a parameter
  double part[6] = {p_nc, p_bb, p_dd, p_fd, p_ab, p_aa};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    part[k] += __shfl_xor(part[k], 16);
    part[k] += __shfl_xor(part[k], 32);
  }
  double v_fd = 0.0;                       // r_fd . ur: thread = (worker w, parameter m)
  for (int r = w; r < QN; r += NWK) v_fd += a.r_fd[(long)s * QN + r] * ur[r * 16 + m];
  double* vred = red + EST_NW * 6 * 16;    // [NWK][16]
  vred[w * 16 + m] = v_fd;
  if (lk == 0)
#pragma unroll
    for (int k = 0; k < 6; ++k) red[(wave * 6 + k) * 16 + li] = part[k];
  __syncthreads();
  if (tid < mb) {                          // totals per parameter, fixed order
    double tot[6];
    for (int k = 0; k < 6; ++k) {
      double v = 0.0;
      for (int ww = 0; ww < EST_NW; ++ww) v += red[(ww * 6 + k) * 16 + tid];
      tot[k] = v;
    }
    for (int ww = 0; ww < NWK; ++ww) tot[3] += vred[ww * 16 + tid];
    const double pi = 3.14159265358979323846;
    const long o = (long)s * nmu + m0 + tid;
    a.eta[o] = tot[0];
    a.eta[(long)t.S * nmu + o] = (a.f2[s] - 2.0 * tot[3] + tot[2]) * (1.0 / (pi * pi)) / a.ceps[s] * a.hdiam * a.hdiam;
    a.eta[2L * t.S * nmu + o] = tot[1] + 2.0 * tot[4] + tot[5];
  }
}

// ------------------------------------------------------------------------------------------------- online: reduced solve
__global__ __launch_bounds__(256) void k3_combine(long per_q, int Q, QV th, const double* __restrict__ B, double* __restrict__ Amu) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_q) return;
  double acc = 0.0;
  for (int q = 0; q < Q; ++q) acc += th.v[q] * B[q * per_q + i];
  Amu[i] = acc;
}

// inverse of the SPD diagonal blocks by Gauss-Jordan elimination in LDS (no pivoting)
__global__ __launch_bounds__(256) void k3_block_inverse(int N, const double* __restrict__ Amu, double* __restrict__ Dinv) {
  extern __shared__ double lds[];   // [N][2N + 1]
  const int s = blockIdx.x, tid = threadIdx.x, ld = 2 * N + 1;
  const double* A = Amu + ((long)s * 7 + 3) * N * N;
  for (int i = tid; i < N * 2 * N; i += 256) {
    const int r = i / (2 * N), c = i - r * 2 * N;
    // a zero diagonal entry = a zero-padded basis column (ragged local bases: its whole row and column are zero): identity there,
    // the padded unknown stays exactly 0 (2D: k_block_inverse)
    lds[r * ld + c] = c < N ? ((c == r && A[r * N + c] == 0.0) ? 1.0 : A[r * N + c]) : (c - N == r ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int p = 0; p < N; ++p) {
    const double ip = 1.0 / lds[p * ld + p];
    __syncthreads();
    for (int c = tid; c < 2 * N; c += 256) lds[p * ld + c] *= ip;
    __syncthreads();
    for (int i = tid; i < N * 2 * N; i += 256) {
      const int r = i / (2 * N), c = i - r * 2 * N;
      if (r != p && c != p) lds[r * ld + c] -= lds[r * ld + p] * lds[p * ld + c];
    }
    __syncthreads();
    for (int r = tid; r < N; r += 256)
      if (r != p) lds[r * ld + p] = 0.0;
    __syncthreads();
  }
  for (int i = tid; i < N * N; i += 256) Dinv[(long)s * N * N + i] = lds[(i / N) * ld + N + i % N];
}

// scal: [0] rz_old, [1] rz_new, [2] pAp, [3] rr, [4] bb;  partial arrays [S]
// init: x = 0, r = b, z = Dinv r, partial rz, rr
__global__ __launch_bounds__(64) void k3_pcg_init(int N, const double* __restrict__ rhs, const double* __restrict__ Dinv,
                                                  double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                  double* __restrict__ p, double* __restrict__ prz, double* __restrict__ prr) {
  __shared__ double rs[64], red[64];
  const int s = blockIdx.x, i = threadIdx.x;
  const double ri = i < N ? rhs[(long)s * N + i] : 0.0;
  rs[i] = ri;
  __syncthreads();
  double zi = 0.0;
  if (i < N) {
    const double* D = Dinv + ((long)s * N + i) * N;
    for (int j = 0; j < N; ++j) zi += D[j] * rs[j];
    x[(long)s * N + i] = 0.0;
    r[(long)s * N + i] = ri;
    z[(long)s * N + i] = zi;
    p[(long)s * N + i] = 0.0;
  }
  red[i] = ri * zi;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (i < w) red[i] += red[i + w];
    __syncthreads();
  }
  if (i == 0) prz[s] = red[0];
  __syncthreads();
  red[i] = ri * ri;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (i < w) red[i] += red[i + w];
    __syncthreads();
  }
  if (i == 0) prr[s] = red[0];
}

__device__ inline double sum_partials(const double* __restrict__ part, int S, double* red) {   // 64 threads, fixed order
  const int i = threadIdx.x;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // eight loads in flight per thread
  int k = i;
  for (; k + 7 * 64 < S; k += 8 * 64) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[k + u * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += v[u];
  }
  for (; k < S; k += 64) a[0] += part[k];
  const double acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  red[i] = acc;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (i < w) red[i] += red[i + w];
    __syncthreads();
  }
  const double v = red[0];
  __syncthreads();
  return v;
}

// direction + matvec: p_new = z + beta p_old on the neighbourhood (own part stored), Ap = sum_slot A[s][slot] p_new[slot]
__global__ __launch_bounds__(64) void k3_pcg_matvec(T3 t, int N, int first, const double* __restrict__ Amu,
                                                    const double* __restrict__ z, const double* __restrict__ p_old,
                                                    double* __restrict__ p_new, double* __restrict__ Ap,
                                                    const double* __restrict__ prz_new, const double* __restrict__ prz_old,
                                                    double* __restrict__ ppap) {
  extern __shared__ double lds[];   // [7][N] + 64
  double* red = lds + 7 * N;
  const int s = blockIdx.x, i = threadIdx.x, S = t.S;
  double beta = 0.0;
  if (!first) {
    const double a = sum_partials(prz_new, S, red), b = sum_partials(prz_old, S, red);
    beta = a / b;
  }
  for (int k = i; k < 7 * N; k += 64) {
    const int s2 = t.nbr[s * 7 + k / N];
    lds[k] = s2 >= 0 ? z[(long)s2 * N + k % N] + beta * p_old[(long)s2 * N + k % N] : 0.0;
  }
  __syncthreads();
  double acc = 0.0;
  if (i < N) {
    p_new[(long)s * N + i] = lds[3 * N + i];
    for (int slot = 0; slot < 7; ++slot) {
      if (t.nbr[s * 7 + slot] < 0) continue;
      const double* row = Amu + (((long)s * 7 + slot) * N + i) * N;
      const double* ps = lds + slot * N;
      for (int j = 0; j < N; ++j) acc += row[j] * ps[j];
    }
    Ap[(long)s * N + i] = acc;
  }
  __syncthreads();
  red[i] = i < N ? acc * lds[3 * N + i] : 0.0;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (i < w) red[i] += red[i + w];
    __syncthreads();
  }
  if (i == 0) ppap[s] = red[0];
}

// x += alpha p, r -= alpha Ap, z = Dinv r; partial r.z and r.r
__global__ __launch_bounds__(64) void k3_pcg_update(int S, int N, const double* __restrict__ Dinv, const double* __restrict__ p,
                                                    const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ z, const double* __restrict__ prz_cur,
                                                    const double* __restrict__ ppap, double* __restrict__ prz_out,
                                                    double* __restrict__ prr) {
  __shared__ double rs[64], red[64];
  const int s = blockIdx.x, i = threadIdx.x;
  const double rz = sum_partials(prz_cur, S, red), pap = sum_partials(ppap, S, red);
  const double alpha = rz / pap;
  double ri = 0.0;
  if (i < N) {
    x[(long)s * N + i] += alpha * p[(long)s * N + i];
    ri = r[(long)s * N + i] - alpha * Ap[(long)s * N + i];
    r[(long)s * N + i] = ri;
  }
  rs[i] = ri;
  __syncthreads();
  double zi = 0.0;
  if (i < N) {
    const double* D = Dinv + ((long)s * N + i) * N;
    for (int j = 0; j < N; ++j) zi += D[j] * rs[j];
    z[(long)s * N + i] = zi;
  }
  red[i] = ri * zi;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (i < w) red[i] += red[i + w];
    __syncthreads();
  }
  if (i == 0) prz_out[s] = red[0];
  __syncthreads();
  red[i] = ri * ri;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (i < w) red[i] += red[i + w];
    __syncthreads();
  }
  if (i == 0) prr[s] = red[0];
}

__global__ __launch_bounds__(256) void k3_reduce1(int S, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  // eight loads in flight per thread (a plain `acc += part[k]` loop waits one L2 round trip per entry: 11 us for 10 k entries)
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int k = tid;
  for (; k + 7 * 256 < S; k += 8 * 256) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[k + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += v[u];
  }
  for (; k < S; k += 256) a[0] += part[k];
  red[tid] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) out[0] = red[0];
}

// ------------------------------------------------------------------------------------------------- online: batched reduced solve
// nmu <= 16 parameters at once: every projected block is read once per iteration for the whole batch, one block-Jacobi
// preconditioner at the batch-mean theta (any SPD preconditioner is admissible), independent CG scalars per parameter.
// Vectors [S][N][nmu] (parameter fastest).  Threads (i = row, m = parameter).
struct TB { double v[16][8]; };     // theta [nmu][Q]

// scal [5][16]: rz_old, rz_new, pAp, rr, bb per parameter;  partial arrays [S][16]
__global__ __launch_bounds__(256) void k3b_reduce(int S, int nmu, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double red[256];
  const int tid = threadIdx.x, m = tid & 15, g = tid >> 4;
  double acc = 0.0;
  if (m < nmu)
    for (int k = g; k < S; k += 16) acc += part[(long)k * 16 + m];
  red[tid] = acc;
  __syncthreads();
  for (int w = 128; w >= 16; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid < 16) out[tid] = red[tid];
}

//   x: this group's columns of the caller's solution array, x[(s N + i) ldx + m]
__global__ __launch_bounds__(512) void k3b_init(int N, int nmu, int ldx, const double* __restrict__ rhs, const double* __restrict__ Dinv,
                                                double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                double* __restrict__ p, double* __restrict__ prz, double* __restrict__ prr) {
  extern __shared__ double lds[];      // [N] rhs + [N][16] products
  const int s = blockIdx.x, tid = threadIdx.x, i = tid >> 4, m = tid & 15;
  for (int k = tid; k < N; k += 512) lds[k] = rhs[(long)s * N + k];
  __syncthreads();
  double ri = 0.0, zi = 0.0;
  const bool on = i < N && m < nmu;
  if (on) {
    ri = lds[i];
    const double* D = Dinv + ((long)s * N + i) * N;
    for (int j = 0; j < N; ++j) zi += D[j] * lds[j];
    const long d = ((long)s * N + i) * nmu + m;
    x[((long)s * N + i) * ldx + m] = 0.0;
    r[d] = ri; z[d] = zi; p[d] = 0.0;
  }
  double* pr = lds + N;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (i < 32) pr[i * 16 + m] = on ? (pass ? ri * ri : ri * zi) : 0.0;
    __syncthreads();
    if (tid < 16) {
      double a = 0.0;
      for (int k = 0; k < N; ++k) a += pr[k * 16 + tid];
      (pass ? prr : prz)[(long)s * 16 + tid] = a;
    }
  }
}

// Coarse level of the batched reduced solve (nullptr members: block-Jacobi alone): y0 [S][16] = A0^-1 r0 of the last residual,
// prc_* [S][16] its contributions r0 . y0 to r.z -- summed with the fine partials wherever r.z is needed.
struct CoarseB {
  const double* y0;
  const double* prc_new;
  const double* prc_old;
};

// The CG scalars of an iteration are sums over the workgroups' partials [S][16] of the previous kernel; every workgroup (512
// threads) forms them itself, in the same fixed order, so an iteration is two launches (no reduction kernels in between).
// Returns the sum for parameter tid & 15 in every thread; buf [32][16] doubles of LDS.
__device__ inline double sum_partials16(const double* __restrict__ part, int S, double* buf, int tid) {
  const int m = tid & 15, g = tid >> 4, ng = blockDim.x >> 4;     // 16 or 32 groups (256 / 512 threads)
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // eight loads in flight (a plain loop waits one L2 round trip per entry)
  int k = g;
  for (; k + 7 * ng < S; k += 8 * ng) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + u * ng) * 16 + m];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += v[u];
  }
  for (; k < S; k += ng) a[0] += part[(long)k * 16 + m];
  const double acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  buf[g * 16 + m] = acc;
  __syncthreads();
  double v = 0.0;
  for (int k = 0; k < ng; ++k) v += buf[k * 16 + m];
  __syncthreads();
  return v;
}

// The same direction + matvec on the matrix cores (N <= 32): Ap [rows i][16 parameters] = sum_q sum_slot B_q[slot] (theta_q . p[slot])
// -- the A operand is a 16-row strip of a projected block straight from global memory (lane: row l & 15, columns 2 (l >> 4) and
// + 1 of the wave's 8-column band: one 16-byte load for two k-steps), the B operand the direction of the slot from the LDS with the
// parameter weight theta_qm folded in (a lane owns ONE parameter, l & 15).  Four waves, wave w owns the column band [8 w, 8 w + 8) of
// every block, so the A_mu strips are read exactly once; the waves' tiles meet in the LDS in a fixed order.
template <int RT>
__global__ __launch_bounds__(256) void k3b_matvec_mfma(T3 t, int Q, int N, int nmu, int first, TB th, const double* __restrict__ B,
                                                       const double* __restrict__ z, const double* __restrict__ p_old,
                                                       double* __restrict__ p_new, double* __restrict__ Ap,
                                                       const double* __restrict__ prz_new, const double* __restrict__ prz_old,
                                                       double* __restrict__ ppap, CoarseB cl) {
  extern __shared__ double lds[];      // [7][32][16] direction (rows >= N zero) + [4][RT][256] partial tiles + [32][16] sums
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4, S = t.S;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* dir = lds;
  double* part = lds + 7 * 32 * 16;
  double* buf = part + 4 * RT * 256;
  double bm = 0.0;
  if (!first) {
    double rz_new = sum_partials16(prz_new, S, buf, tid), rz_old = sum_partials16(prz_old, S, buf, tid);
    if (cl.y0) {
      rz_new += sum_partials16(cl.prc_new, S, buf, tid);
      rz_old += sum_partials16(cl.prc_old, S, buf, tid);
    }
    bm = rz_old == 0.0 ? 0.0 : rz_new / rz_old;
  }
  for (int k = tid; k < 7 * 32 * 16; k += 256) {
    const int slot = k >> 9, j = (k >> 4) & 31, mm = k & 15;
    const int s2 = t.nbr[s * 7 + slot];
    double v = 0.0;
    if (s2 >= 0 && mm < nmu && j < N) {
      const long d = ((long)s2 * N + j) * nmu + mm;
      double zt = z[d];
      if (cl.y0 && j == 0) zt += cl.y0[s2 * 16 + mm];              // z = Dinv r + e_0 y0: the coarse correction
      v = first ? zt : zt + bm * p_old[d];
    }
    dir[k] = v;
  }
  __syncthreads();
  for (int k = tid; k < N * 16; k += 256) {
    const int i = k >> 4, mm = k & 15;
    if (mm < nmu) p_new[((long)s * N + i) * nmu + mm] = dir[(3 * 32 + i) * 16 + mm];
  }
  d4 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = (d4){0.0, 0.0, 0.0, 0.0};
  const int c0 = 8 * wave + 2 * lk;                  // this lane's two columns of the band (clamped into the matrix: the matching
  const bool even = (N & 1) == 0;                    // direction rows are zero beyond N)
  const int ca = c0 < N ? c0 : N - 1, cb = c0 + 1 < N ? c0 + 1 : N - 1, cpair = c0 + 1 < N ? c0 : N - 2;
  if (8 * wave < N) {
    double thq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) thq[q] = q < Q ? th.v[li][q] : 0.0;
    for (int slot = 0; slot < 7; ++slot) {
      if (t.nbr[s * 7 + slot] < 0) continue;         // wave-uniform
      const double p0 = dir[(slot * 32 + c0) * 16 + li], p1 = dir[(slot * 32 + c0 + 1) * 16 + li];
      for (int q = 0; q < Q; ++q) {
        const double b0 = thq[q] * p0, b1 = thq[q] * p1;
        const double* blk = B + (((long)q * S + s) * 7 + slot) * N * N;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = rt * 16 + li < N ? rt * 16 + li : N - 1;
          double a0, a1;
          if (even) {
            const double2 v = *reinterpret_cast<const double2*>(blk + row * N + cpair);
            a0 = v.x; a1 = v.y;
          } else {
            a0 = blk[row * N + ca]; a1 = blk[row * N + cb];
          }
          acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[rt], 0, 0, 0);
          acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[rt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[((wave * RT + rt) * 4 + r) * 64 + lane] = acc[rt][r];
  __syncthreads();
  double pp = 0.0;
  if (wave < RT) {                                   // wave rt finishes tile rt: rows rt * 16 + lk + 4 r, parameter li
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += part[((w * RT + wave) * 4 + r) * 64 + lane];
      const int row = wave * 16 + lk + 4 * r;
      if (row < N && li < nmu) {
        Ap[((long)s * N + row) * nmu + li] = v;
        pp += v * dir[(3 * 32 + row) * 16 + li];
      }
    }
    pp += __shfl_xor(pp, 16);
    pp += __shfl_xor(pp, 32);
    if (lk == 0) buf[wave * 16 + li] = pp;
  }
  __syncthreads();
  if (tid < 16) {
    double a = 0.0;
    for (int rt = 0; rt < RT; ++rt) a += buf[rt * 16 + tid];
    ppap[(long)s * 16 + tid] = a;
  }
}

// direction + matvec: p_new = z + beta_m p_old on the neighbourhood (own part stored), Ap = sum_q theta_qm sum_slot B_q p_new;
// beta_m = r.z (prz_new) / previous r.z (prz_old)
__global__ __launch_bounds__(512) void k3b_matvec(T3 t, int Q, int N, int nmu, int first, TB th, const double* __restrict__ B,
                                                  const double* __restrict__ z, const double* __restrict__ p_old,
                                                  double* __restrict__ p_new, double* __restrict__ Ap, const double* __restrict__ prz_new,
                                                  const double* __restrict__ prz_old, double* __restrict__ ppap, CoarseB cl) {
  extern __shared__ double lds[];      // [7][N][16] direction + [32][16] products
  const int s = blockIdx.x, tid = threadIdx.x, i = tid >> 4, m = tid & 15, S = t.S;
  double bm = 0.0;                     // (512 % 16 == 0: a thread fills entries of its own parameter only)
  if (!first) {
    double* buf = lds + 7 * N * 16;
    double rz_new = sum_partials16(prz_new, S, buf, tid), rz_old = sum_partials16(prz_old, S, buf, tid);
    if (cl.y0) {
      rz_new += sum_partials16(cl.prc_new, S, buf, tid);
      rz_old += sum_partials16(cl.prc_old, S, buf, tid);
    }
    bm = rz_old == 0.0 ? 0.0 : rz_new / rz_old;                    // a converged parameter (r = 0) stays put
  }
  for (int k = tid; k < 7 * N * 16; k += 512) {
    const int slot = k / (N * 16), j = (k >> 4) % N, mm = k & 15;
    const int s2 = t.nbr[s * 7 + slot];
    double v = 0.0;
    if (s2 >= 0 && mm < nmu) {
      const long d = ((long)s2 * N + j) * nmu + mm;
      double zt = z[d];
      if (cl.y0 && j == 0) zt += cl.y0[s2 * 16 + mm];
      v = first ? zt : zt + bm * p_old[d];
    }
    lds[k] = v;
  }
  __syncthreads();
  double acc = 0.0;
  const bool on = i < N && m < nmu;
  if (on) {
    p_new[((long)s * N + i) * nmu + m] = lds[(3 * N + i) * 16 + m];
    for (int q = 0; q < Q; ++q) {
      double aq = 0.0;
      for (int slot = 0; slot < 7; ++slot) {
        if (t.nbr[s * 7 + slot] < 0) continue;
        const double* row = B + ((((long)q * S + s) * 7 + slot) * N + i) * N;
        const double* ps = lds + slot * N * 16 + m;
        for (int j = 0; j < N; ++j) aq += row[j] * ps[j * 16];
      }
      acc += th.v[m][q] * aq;
    }
    Ap[((long)s * N + i) * nmu + m] = acc;
  }
  double* pr = lds + 7 * N * 16;
  __syncthreads();
  if (i < 32) pr[i * 16 + m] = on ? acc * lds[(3 * N + i) * 16 + m] : 0.0;
  __syncthreads();
  if (tid < 16) {
    double a = 0.0;
    for (int k = 0; k < N && k < 32; ++k) a += pr[k * 16 + tid];
    ppap[(long)s * 16 + tid] = a;
  }
}

__global__ __launch_bounds__(512) void k3b_update(int N, int nmu, int ldx, const double* __restrict__ Dinv, const double* __restrict__ p,
                                                  const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r,
                                                  double* __restrict__ z, int S, const double* __restrict__ prz_cur,
                                                  const double* __restrict__ prc_cur, const double* __restrict__ ppap,
                                                  double* __restrict__ prz, double* __restrict__ prr) {
  extern __shared__ double lds[];      // [N][16] residual + [32][16] products
  const int s = blockIdx.x, tid = threadIdx.x, i = tid >> 4, m = tid & 15;
  const bool on = i < N && m < nmu;
  double ri = 0.0, zi = 0.0;
  double rz = sum_partials16(prz_cur, S, lds + N * 16, tid);
  if (prc_cur) rz += sum_partials16(prc_cur, S, lds + N * 16, tid);
  const double pap = sum_partials16(ppap, S, lds + N * 16, tid);
  if (on) {
    const double alpha = pap != 0.0 ? rz / pap : 0.0;               // a converged parameter (r = 0) stays put
    const long d = ((long)s * N + i) * nmu + m;
    x[((long)s * N + i) * ldx + m] += alpha * p[d];
    ri = r[d] - alpha * Ap[d];
    r[d] = ri;
  }
  if (i < N) lds[i * 16 + m] = ri;
  __syncthreads();
  if (on) {
    const double* D = Dinv + ((long)s * N + i) * N;
    for (int j = 0; j < N; ++j) zi += D[j] * lds[j * 16 + m];
    z[((long)s * N + i) * nmu + m] = zi;
  }
  double* pr = lds + N * 16;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (i < 32) pr[i * 16 + m] = on ? (pass ? ri * ri : ri * zi) : 0.0;
    __syncthreads();
    if (tid < 16) {
      double a = 0.0;
      for (int k = 0; k < N && k < 32; ++k) a += pr[k * 16 + tid];
      (pass ? prr : prz)[(long)s * 16 + tid] = a;
    }
  }
}

// ---- coarse level of the reduced solvers: the Galerkin problem on the FIRST local basis vector of every subdomain (the constant the
// reductor starts every basis with), A0[s][t] = A_mu[s][slot of t][0][0] -- a 7-point S x S matrix, inverted densely once per
// reduced model at a reference parameter (lrbms3_reduced_precond_build; any SPD preconditioner is admissible for the other mu).
__global__ __launch_bounds__(256) void k3r_coarse_fill(T3 t, int Q, int N, QV th, const double* __restrict__ B, double* __restrict__ A0,
                                                       double* __restrict__ Id) {
  const int idx = blockIdx.x * 256 + threadIdx.x, S = t.S;
  if (idx >= S * 7) return;
  const int s = idx / 7, slot = idx - s * 7;
  if (slot == 3) Id[(long)s * S + s] = 1.0;
  const int tt = slot == 3 ? s : t.nbr[s * 7 + slot];
  if (tt < 0) return;
  double v = 0.0;
  for (int q = 0; q < Q; ++q) v += th.v[q] * B[(((long)q * S + s) * 7 + slot) * N * N];
  A0[(long)s * S + tt] = v;
}

// y0 [S][16] = A0inv r0, r0[t][m] = r[t][0][m], on the matrix cores: one 16-row tile of A0inv per workgroup, sixteen waves split
// K = S and keep eight k-steps of operands in flight (a load -> MFMA chain per k-step is one L2 round trip each: 25 us);
// prc [S][16] = r0 . y0 (the coarse part of r.z)
__global__ __launch_bounds__(1024) void k3b_coarse_apply(int S, int N, int nmu, const double* __restrict__ A0inv, const double* __restrict__ r,
                                                         double* __restrict__ y0, double* __restrict__ prc) {
  __shared__ double part[16][4][64];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * 16, row = row0 + li < S ? row0 + li : S - 1;
  const double* arow = A0inv + (long)row * S;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  const int ksteps = (S + 3) / 4;
  for (int k0 = wave; k0 < ksteps; k0 += 16 * 8) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = 4 * (k0 + 16 * u) + lk, cc = c < S ? c : S - 1;
      a[u] = arow[cc];
      b[u] = r[((long)cc * N) * nmu + (li < nmu ? li : 0)];
      if (c >= S || li >= nmu) b[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) part[wave][q][lane] = acc[q];
  __syncthreads();
  if (wave < 4) {                                     // wave q finishes accumulator register q: rows lk + 4 q, column li
    double v = 0.0;
    for (int w = 0; w < 16; ++w) v += part[w][wave][lane];
    const int rr = row0 + lk + 4 * wave;
    if (rr < S) {
      const double r0 = li < nmu ? r[((long)rr * N) * nmu + li] : 0.0;
      y0[rr * 16 + li] = li < nmu ? v : 0.0;
      prc[rr * 16 + li] = li < nmu ? v * r0 : 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------- full-order apply
__global__ __launch_bounds__(256) void k3_fom_apply(T3 t, int Q, int M, QV th, const double* __restrict__ A_diag,
                                                    const double* __restrict__ A_cpl, const double* __restrict__ x,
                                                    double* __restrict__ y) {
  const int e = blockIdx.x, s = blockIdx.y;
  for (int w = threadIdx.x; w < 10 * M; w += 256) {
    const int i = w / M, m = w - i * M;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) {
      double aq = 0.0;
      const double* A = A_diag + ((((long)q * t.S + s) * t.nT + e) * 5) * 100 + i * 10;
      for (int slot = 0; slot < 5; ++slot) {
        int ee = e, ss = s;
        const double* L = A + slot * 100;
        if (slot > 0) {
          ee = t.nb_elem[e * 4 + slot - 1];
          if (ee < 0) {
            if (A_cpl == nullptr) continue;      // local operator (energy product): no coupling blocks
            const int side = -(ee + 1);
            ss = t.nbr[s * 7 + side_slot(side)];
            if (ss < 0) continue;
            L = A_cpl + ((((long)q * t.S + s) * 6 + side) * t.ncf + t.face_pos[e * 4 + slot - 1]) * 100 + i * 10;
            ee = t.nb_out[e * 4 + slot - 1];
          }
        }
        const double* xv = x + ((long)ss * t.n + ee * 10) * M + m;
        for (int j = 0; j < 10; ++j) aq += L[j] * xv[(long)j * M];
      }
      acc += th.v[q] * aq;
    }
    y[((long)s * t.n + e * 10 + i) * M + m] = acc;
  }
}

// ------------------------------------------------------------------------------------------------- full-order solve (snapshots)
// A(mu) x = b on the never-assembled block operator: CG with the 10 x 10 element blocks as block-Jacobi preconditioner.  The
// operator is combined once per solve (Amu = sum_q theta_q A_q); an iteration is three kernels + three one-block reductions,
// all scalars stay on the device.  EPB elements (10 rows each) per workgroup.
constexpr int FOM_EPB = 24;
constexpr int FOM_MAX_COARSE = 8192;      // largest dense coarse problem of the full-order solver (2 x 0.5 GB of work, ~0.1 s to invert)

// inverse of the diagonal 10 x 10 blocks (SPD): Gauss-Jordan, one thread per element, the block in LDS
__global__ __launch_bounds__(64) void k3f_block_inverse(T3 t, const double* __restrict__ Amu, double* __restrict__ Dinv) {
  __shared__ double a[64][101];
  const int s = blockIdx.y, e = blockIdx.x * 64 + threadIdx.x;
  if (e >= t.nT) return;
  double* m = a[threadIdx.x];
  const double* src = Amu + (((long)s * t.nT + e) * 5) * 100;
  for (int i = 0; i < 100; ++i) m[i] = src[i];
  double inv[100];
  for (int i = 0; i < 100; ++i) inv[i] = (i / 10 == i % 10) ? 1.0 : 0.0;
  for (int p = 0; p < 10; ++p) {
    const double ip = 1.0 / m[p * 10 + p];
    for (int c = 0; c < 10; ++c) {
      m[p * 10 + c] *= ip;
      inv[p * 10 + c] *= ip;
    }
    for (int r = 0; r < 10; ++r) {
      if (r == p) continue;
      const double f = m[r * 10 + p];
      for (int c = 0; c < 10; ++c) {
        m[r * 10 + c] -= f * m[p * 10 + c];
        inv[r * 10 + c] -= f * inv[p * 10 + c];
      }
    }
  }
  // the inverse of an SPD block is symmetric: its upper triangle is stored, packed row by row (56 doubles per block) -- the update
  // kernel of the CG streams these blocks every iteration and is bound by their bytes
  double* dst = Dinv + ((long)s * t.nT + e) * 56;
  int k = 0;
  for (int i = 0; i < 10; ++i)
    for (int j = i; j < 10; ++j) dst[k++] = inv[i * 10 + j];
  dst[55] = 0.0;
}

// scal: [0], [1] r.z of the last two updates (alternating)  [2] pAp  [3] rr  [4] bb
// p = z + beta p  (beta = rz_new / rz_old, 0 in the first iteration)
//   with the coarse level: z + R0^T y0 in the place of z (y0 [S][nc], Phi [n][4])
__global__ __launch_bounds__(256) void k3f_dir(long total, int first, int cur, const double* __restrict__ scal,
                                               const double* __restrict__ z, double* __restrict__ p, int n, int nc,
                                               const double* __restrict__ Phi, const double* __restrict__ y0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const double beta = first ? 0.0 : scal[cur] / scal[cur ^ 1];       // r.z of the last two updates alternate between slots 0 and 1
  double zi = z[i];
  if (nc > 0) {
    const long s = i / n;
    const int dof = (int)(i - s * n);
    for (int k = 0; k < nc; ++k) zi += Phi[dof * 4 + k] * y0[s * nc + k];
  }
  p[i] = zi + beta * p[i];
}

// y = Amu p on the block-ELL + coupling data; partial p.y per workgroup.
// Bound by the 0.8 GB of 10 x 10 blocks it streams, so (1) every block is read by ONE 16-byte-per-lane load of 50 lanes (800
// contiguous bytes; lane l holds A[i][2c], A[i][2c + 1], i = l / 5, c = l % 5 -- a thread per row reading 80-byte rows costs a
// cache-line lookup per lane and instruction, which, not the HBM, then sets the pace) and the products are summed across lanes;
// (2) the operator is symmetric, A[e'][e] = A[e][e']^T: inside a subdomain only the blocks towards the neighbour with the HIGHER
// element index are read from their own place (sum over the row: lanes l .. l + 4), the element on the other side reads the same
// block and sums over the columns (lanes l, l + 5, ...).  The two readers are workgroups of the same subdomain, and the 1D grid
// is decoded so that all workgroups of a subdomain run on one XCD (xcd_block): the second read comes from that XCD's L2 and HBM
// delivers 61 % of the block bytes.  One wave per element at a time, four waves per workgroup.
__global__ __launch_bounds__(256) void k3f_matvec(T3 t, int nbx, const double* __restrict__ Amu, const double* __restrict__ Cmu,
                                                  const double* __restrict__ p, double* __restrict__ y, double* __restrict__ part) {
  __shared__ double red[4];
  int s, bx;
  if (!xcd_block(nbx, t.S, bx, s)) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool act = lane < 50;
  const int li = act ? lane : 49, row = li / 5, c2 = 2 * (li - row * 5);
  const double* As = Amu + (long)s * t.nT * 500;
  const double* ps = p + (long)s * t.n;
  double py = 0.0;
  const int e1 = (bx + 1) * FOM_EPB < t.nT ? (bx + 1) * FOM_EPB : t.nT;
  for (int e = bx * FOM_EPB + wave; e < e1; e += 4) {
    const int4 nb = *reinterpret_cast<const int4*>(t.nb_elem + e * 4);
    const int nbv[4] = {nb.x, nb.y, nb.z, nb.w};
    double yr = 0.0, yc0 = 0.0, yc1 = 0.0;            // row sums (valid in lanes 5 i), column sums (valid in lanes 0 .. 4)
    {                                                 // diagonal block
      const double2 a = *reinterpret_cast<const double2*>(As + (long)e * 500 + li * 2);
      const double2 pv = *reinterpret_cast<const double2*>(ps + e * 10 + c2);
      yr = act ? a.x * pv.x + a.y * pv.y : 0.0;
    }
    // (the faces are wave-uniform branches; making every load unconditional -- faces without a neighbour pointing at the own block
    // with weight zero -- was measured: 136 instead of 126 us, the kernel is within 25 % of its HBM time either way)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int ee = nbv[f];                          // wave-uniform
      if (ee >= 0 && ee < e) {                        // the lower element's block towards e, used transposed
        const int4 nb2 = *reinterpret_cast<const int4*>(t.nb_elem + ee * 4);
        const int fb = nb2.x == e ? 0 : (nb2.y == e ? 1 : (nb2.z == e ? 2 : 3));
        const double2 a = *reinterpret_cast<const double2*>(As + (long)ee * 500 + (1 + fb) * 100 + li * 2);
        const double pr = ps[ee * 10 + row];
        yc0 += act ? a.x * pr : 0.0;
        yc1 += act ? a.y * pr : 0.0;
        continue;
      }
      const double* L;
      const double* pn;
      if (ee >= 0) {
        L = As + (long)e * 500 + (1 + f) * 100;
        pn = ps + ee * 10;
      } else {
        const int side = -(ee + 1), ss = t.nbr[s * 7 + side_slot(side)];
        if (ss < 0) continue;
        L = Cmu + (((long)s * 6 + side) * t.ncf + t.face_pos[e * 4 + f]) * 100;
        pn = p + (long)ss * t.n + t.nb_out[e * 4 + f] * 10;
      }
      const double2 a = *reinterpret_cast<const double2*>(L + li * 2);
      const double2 pv = *reinterpret_cast<const double2*>(pn + c2);
      yr += act ? a.x * pv.x + a.y * pv.y : 0.0;
    }
    // row sums: lanes l .. l + 4 (fixed order), result in lanes 5 i
    double r1 = yr + __shfl_down(yr, 1);
    r1 = r1 + __shfl_down(r1, 2);
    const double rs = r1 + __shfl_down(yr, 4);
    // column sums: lanes l, l + 5, ..., l + 45 (lanes >= 50 contribute zeros), result in lanes 0 .. 4
    double c0 = yc0 + __shfl_down(yc0, 5), c1 = yc1 + __shfl_down(yc1, 5);
    c0 += __shfl_down(c0, 10); c1 += __shfl_down(c1, 10);
    c0 += __shfl_down(c0, 20); c1 += __shfl_down(c1, 20);
    c0 += __shfl_down(yc0 + __shfl_down(yc0, 5), 40); c1 += __shfl_down(yc1 + __shfl_down(yc1, 5), 40);
    // entry t of y_e in lane t < 10: row sum of row t + column sum of column t
    const int tl = lane < 10 ? lane : 0;
    const double ya = __shfl(rs, 5 * tl), yb0 = __shfl(c0, tl >> 1), yb1 = __shfl(c1, tl >> 1);
    if (lane < 10) {
      const double v = ya + ((lane & 1) ? yb1 : yb0);
      y[(long)s * t.n + e * 10 + lane] = v;
      py += v * ps[e * 10 + lane];
    }
  }
  for (int o = 8; o > 0; o >>= 1) py += __shfl_down(py, o);          // lanes 0 .. 9 (others are zero)
  if (lane == 0) red[wave] = py;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)s * nbx + bx] = (red[0] + red[1]) + (red[2] + red[3]);
}

// x += alpha p, r -= alpha y, z = Dinv r; partials r.z and r.r per workgroup (init: x = 0, r = b: alpha = 0 with p = y = any)
__global__ __launch_bounds__(256) void k3f_update(T3 t, int init, int cur, const double* __restrict__ scal, const double* __restrict__ Dinv,
                                                  const double* __restrict__ p, const double* __restrict__ y, const double* __restrict__ b,
                                                  double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                  double* __restrict__ prz, double* __restrict__ prr, int nc,
                                                  const double* __restrict__ Phi, double* __restrict__ pr0) {
  __shared__ double rs[256], red[256];
  const int s = blockIdx.y, tid = threadIdx.x;
  const int el = tid / 10, i = tid - el * 10, e = blockIdx.x * FOM_EPB + el;
  const bool on = el < FOM_EPB && e < t.nT;
  const long d = (long)s * t.n + e * 10 + i;
  double ri = 0.0;
  if (on) {
    if (init) {
      ri = b[d];
      x[d] = 0.0;
    } else {
      const double alpha = scal[cur] / scal[2];
      x[d] += alpha * p[d];
      ri = r[d] - alpha * y[d];
    }
    r[d] = ri;
  }
  rs[tid] = ri;
  __syncthreads();
  double zi = 0.0;
  if (on) {
    const double* D = Dinv + ((long)s * t.nT + e) * 56;          // packed upper triangle: entry (a, b), a <= b, at a (21 - a) / 2 + b - a
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int a = i < j ? i : j, b = i < j ? j : i;
      zi += D[a * (21 - a) / 2 + b - a] * rs[el * 10 + j];
    }
    z[d] = zi;
  }
  red[tid] = ri * zi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if (tid == 0) prz[blk] = red[0];
  __syncthreads();
  red[tid] = ri * ri;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) prr[blk] = red[0];
  // restriction of the residual to the coarse space: partial sums phi_k . r of this workgroup's rows (fixed-order tree)
  for (int k = 0; k < nc; ++k) {
    __syncthreads();
    red[tid] = on ? ri * Phi[(e * 10 + i) * 4 + k] : 0.0;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (tid == 0) pr0[blk * 4 + k] = red[0];
  }
}

// ---- coarse level of the full-order preconditioner: M^-1 = blockdiag(A_ee)^-1 + R0^T (R0 A R0^T)^-1 R0 (additive, SPD), the rows
// of R0 = nc functions per subdomain (P1 in the subdomain-local coordinates by default: Phi [n][4], the same table everywhere).
// A1b [S][7][16]: the 4 x 4 blocks phi_{s,k}^T A phi_{t,l} for t = the subdomain itself (slot 3) and its six neighbours.
__global__ __launch_bounds__(256) void k3f_coarse_blocks(T3 t, const double* __restrict__ Phi, const double* __restrict__ Amu,
                                                         const double* __restrict__ Cmu, double* __restrict__ A1b) {
  __shared__ double red[16][256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int el = tid / 10, i = tid - el * 10;
  for (int t7 = 0; t7 < 7; ++t7) {
    double acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0;
    if (el < 25)
      for (int e = el; e < t.nT; e += 25) {
        const double* A = Amu + (((long)s * t.nT + e) * 5) * 100 + i * 10;
        const double* ph = Phi + (e * 10 + i) * 4;
        for (int slot = 0; slot < 5; ++slot) {
          int ee = e, tgt = 3;
          const double* L = A + slot * 100;
          if (slot > 0) {
            ee = t.nb_elem[e * 4 + slot - 1];
            if (ee < 0) {
              const int side = -(ee + 1);
              tgt = side_slot(side);
              if (t.nbr[s * 7 + tgt] < 0) continue;
              L = Cmu + (((long)s * 6 + side) * t.ncf + t.face_pos[e * 4 + slot - 1]) * 100 + i * 10;
              ee = t.nb_out[e * 4 + slot - 1];
            }
          }
          if (tgt != t7) continue;
          double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int j = 0; j < 10; ++j) {
            const double lj = L[j];
            const double* pj = Phi + (ee * 10 + j) * 4;
#pragma unroll
            for (int l = 0; l < 4; ++l) a[l] += lj * pj[l];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < 4; ++l) acc[k * 4 + l] += ph[k] * a[l];
        }
      }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) red[c][tid] = acc[c];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      for (int idx = tid; idx < 16 * w; idx += 256) {
        const int c = idx / w, k = idx - c * w;
        red[c][k] += red[c][k + w];
      }
      __syncthreads();
    }
    if (tid < 16) A1b[((long)s * 7 + t7) * 16 + tid] = red[tid][0];
  }
}

// dense coarse matrix [M][M], M = nc S, from the blocks (zero-filled before); identity for the inverse
__global__ __launch_bounds__(256) void k3f_coarse_dense(T3 t, int nc, const double* __restrict__ A1b, double* __restrict__ A1,
                                                        double* __restrict__ Id) {
  const long M = (long)nc * t.S;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < M) Id[idx * M + idx] = 1.0;
  if (idx >= (long)t.S * 7 * 16) return;
  const int kl = (int)(idx & 15), t7 = (int)((idx >> 4) % 7), s = (int)(idx / 112);
  const int k = kl >> 2, l = kl & 3;
  const int tt = t7 == 3 ? s : t.nbr[s * 7 + t7];
  if (tt < 0 || k >= nc || l >= nc) return;
  A1[((long)s * nc + k) * M + (long)tt * nc + l] = A1b[idx];
}

// r0 [M] from the workgroups' partial sums (fixed order)
__global__ __launch_bounds__(256) void k3f_coarse_r0(int S, int nbx, int nc, const double* __restrict__ pr0, double* __restrict__ r0) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= S * nc) return;
  const int s = m / nc, k = m - s * nc;
  double v = 0.0;
  for (int bx = 0; bx < nbx; ++bx) v += pr0[((long)s * nbx + bx) * 4 + k];
  r0[m] = v;
}

// y0 = A1inv r0, one wave per row; prc [M] = r0 . y0 contributions to r.z
__global__ __launch_bounds__(256) void k3f_coarse_apply(int M, const double* __restrict__ A1inv, const double* __restrict__ r0,
                                                        double* __restrict__ y0, double* __restrict__ prc) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const double* a = A1inv + (long)row * M;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int c = lane;
  for (; c + 192 < M; c += 256) {
    v0 += a[c] * r0[c];
    v1 += a[c + 64] * r0[c + 64];
    v2 += a[c + 128] * r0[c + 128];
    v3 += a[c + 192] * r0[c + 192];
  }
  for (; c < M; c += 64) v0 += a[c] * r0[c];
  double v = (v0 + v1) + (v2 + v3);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) {
    y0[row] = v;
    prc[row] = v * r0[row];
  }
}


template <typename T>
int upload(lrbms3_ctx* ctx, const T* host, long count, const T** dev) {
  void* p = nullptr;
  if (count <= 0) count = 1;
  HIP3(ctx, hipMalloc(&p, sizeof(T) * count));
  ctx->owned.push_back(p);
  if (host) HIP3(ctx, hipMemcpy(p, host, sizeof(T) * count, hipMemcpyHostToDevice));
  *dev = (const T*)p;
  return LRBMS_OK;
}

QV make_theta(int Q, const double* theta) {
  QV th{};
  for (int q = 0; q < Q && q < 8; ++q) th.v[q] = theta[q];
  return th;
}

}  // namespace

extern "C" {

int lrbms3_ctx_create(int device, lrbms3_ctx** out) {
  if (!out) return LRBMS_E_INVALID;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return LRBMS_E_HIP;
  if (hipSetDevice(device) != hipSuccess) return LRBMS_E_HIP;
  lrbms3_ctx* c = new lrbms3_ctx();
  c->device = device;
  for (int i = 0; i < 3; ++i)
    if ((c->aux[i] = lrbms_side_stream_acquire(device, i)) == nullptr ||
        hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess)
      return lrbms3_ctx_destroy(c), LRBMS_E_HIP;
  if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return lrbms3_ctx_destroy(c), LRBMS_E_HIP;
  *out = c;
  return LRBMS_OK;
}

int lrbms3_ctx_destroy(lrbms3_ctx* ctx) {
  if (!ctx) return LRBMS_E_INVALID;
  (void)hipSetDevice(ctx->device);
  for (void* p : ctx->owned) (void)hipFree(p);
  if (ctx->pg_part) (void)hipFree(ctx->pg_part);
  if (ctx->blas) (void)rocblas_destroy_handle((rocblas_handle)ctx->blas);
  if (ctx->fom_pc) (void)hipFree(ctx->fom_pc);
  for (int i = 0; i < 3; ++i) {
    if (ctx->aux[i]) lrbms_side_stream_release(ctx->device, i);
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  for (auto& k : ctx->ktimers) {
    (void)hipEventDestroy(k.e0);
    (void)hipEventDestroy(k.e1);
  }
  delete ctx;
  return LRBMS_OK;
}

const char* lrbms3_last_error(lrbms3_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lrbms3_ctx_set_option(lrbms3_ctx* ctx, int32_t option, int32_t value) {
  if (!ctx) return LRBMS_E_INVALID;
  int hi = 1;
  if (option == LRBMS3_OPT_KSPLIT) hi = 8;
  if (option == LRBMS3_OPT_WAVES) hi = 16;
  if (value < 0 || value > hi) return fail3(ctx, LRBMS_E_INVALID, "set_option: value out of range for this option");
  switch (option) {
    case LRBMS3_OPT_KSPLIT: ctx->opt_ksplit = value; break;
    case LRBMS3_OPT_SERIAL: ctx->opt_serial = value; break;
    case LRBMS3_OPT_WAVES: ctx->opt_waves = value; break;
    case LRBMS3_OPT_ESTIMATE_VALU: ctx->opt_estimate_valu = value; break;
    case LRBMS3_OPT_SOLVE_VALU: ctx->opt_solve_valu = value; break;
    case LRBMS3_OPT_FOM_COARSE: ctx->opt_fom_coarse = value; break;
    default: return fail3(ctx, LRBMS_E_INVALID, "set_option: unknown option");
  }
  return LRBMS_OK;
}

int lrbms3_mesh_upload(lrbms3_ctx* ctx, const lrbms3_mesh_desc* d, int32_t S, int32_t S_ext, const int32_t* nbr,
                       const int32_t* phys) {
  if (!ctx || !d || !nbr || !phys) return LRBMS_E_INVALID;
  if (ctx->has_mesh) return fail3(ctx, LRBMS_E_STATE, "mesh already uploaded (one template per context)");
  if (S <= 0 || S_ext < S || d->n_T <= 0 || d->ncf <= 0) return fail3(ctx, LRBMS_E_INVALID, "bad sizes");
  for (int s = 0; s < S; ++s) {
    if (nbr[s * 7 + 3] != s) return fail3(ctx, LRBMS_E_INVALID, "nbr[s][3] must be s");
    for (int k = 0; k < 7; ++k)
      if (nbr[s * 7 + k] < -1 || nbr[s * 7 + k] >= S_ext) return fail3(ctx, LRBMS_E_INVALID, "nbr entry out of range");
  }
  HIP3(ctx, hipSetDevice(ctx->device));
  T3& t = ctx->t;
  t.S = S; t.S_ext = S_ext;
  t.nT = d->n_T; t.n = 10 * d->n_T; t.nrt = d->n_rt; t.ncf = d->ncf; t.nbf = 6 * d->ncf; t.nvs = d->nvs;
  t.nnodes = d->n_nodes; t.nb = d->nb; t.nbel = d->nbel; t.nsel = d->nsel; t.nbd = d->nbd;
  t.nA = d->nA; t.nB = d->nB; t.nC = d->nC; t.nFs = d->nFs; t.nFf = d->nFf;
  t.o_fs = d->o_fs; t.o_ff = d->o_ff; t.o_c = d->o_c; t.lam_stride = d->lam_stride; t.hat_stride = d->hat_stride;
  t.f_stride = d->f_stride;
  t.volume = d->volume; t.kmin = d->kmin;
  const long nT = t.nT, n = t.n;
  int rc;
#define UP(field, count) if ((rc = upload(ctx, d->field, (long)(count), &t.field)) != LRBMS_OK) return rc
  UP(elem_type, nT); UP(up_face, nT * 4); UP(order, nT); UP(nb_elem, nT * 4); UP(nb_out, nT * 4); UP(face_pos, nT * 4); UP(tsign, nT * 4); UP(elem_rt, nT * 4);
  UP(rt_e0, t.nrt); UP(rt_f0, t.nrt); UP(rt_e1, t.nrt); UP(rt_f1, t.nrt);
  UP(side_elem, t.nbf); UP(side_face, t.nbf); UP(side_elem_out, t.nbf); UP(side_face_out, t.nbf);
  UP(dof_node, n); UP(node_ptr, t.nnodes + 1); UP(node_dofs, n); UP(node_mask, t.nnodes); UP(node_count, t.nnodes);
  UP(side_nodes, 6 * t.nvs); UP(sn_ptr, 6 * t.nvs + 1);
  UP(sn_dofs, d->sn_ptr[6 * t.nvs]);
  UP(dof_bslot, n); UP(bn_ptr, t.nb + 1); UP(bn_slots, t.nbd); UP(bnodes, t.nb); UP(bnode_sides, t.nb * 3); UP(bel_elem, t.nbel); UP(bel_bnode, t.nbel * 10); UP(sel_elem, t.nsel);
  UP(sel_sf, t.nsel * 4);
  UP(divc, 24);
  UP(TV, 6L * t.nA * 100); UP(TE, 6L * t.nB * 100); UP(TAA, 6L * t.nC * 100);
  UP(TFo, 24L * t.nFs * 100); UP(TFn, 24L * t.nFs * 100); UP(TFb, 24L * t.nFs * 100);
  UP(TPo, 24L * t.nFs * 100); UP(TPn, 24L * t.nFs * 100); UP(TPb, 24L * t.nFs * 100);
  UP(TC, 24L * t.nFf * 10); UP(TCb, 24L * t.nFf * 10);
  UP(TPH, 6L * t.nB * 10); UP(TM, 600); UP(TB, 6L * t.nC * 16); UP(TAB, 6L * t.nC * 40); UP(WB, t.nB); UP(WC, t.nC);
#undef UP
  if ((rc = upload(ctx, nbr, (long)S * 7, &t.nbr)) != LRBMS_OK) return rc;
  if ((rc = upload(ctx, phys, (long)S_ext, &t.phys)) != LRBMS_OK) return rc;
  {
    const double z[64] = {0.0};
    if ((rc = upload(ctx, z, 64L, &t.zeros)) != LRBMS_OK) return rc;
  }
  {      // concatenated table of the diagonal block: volume rule, then per face the inner-face and the Dirichlet table
    const long KD = t.nA + 8L * t.nFs;
    std::vector<double> tsd((size_t)6 * KD * 100);
    for (int ty = 0; ty < 6; ++ty) {
      double* dst = tsd.data() + (size_t)ty * KD * 100;
      memcpy(dst, d->TV + (size_t)ty * t.nA * 100, sizeof(double) * t.nA * 100);
      for (int f = 0; f < 4; ++f) {
        memcpy(dst + ((size_t)t.nA + (size_t)f * 2 * t.nFs) * 100, d->TFo + ((size_t)(ty * 4 + f) * t.nFs) * 100, sizeof(double) * t.nFs * 100);
        memcpy(dst + ((size_t)t.nA + (size_t)f * 2 * t.nFs + t.nFs) * 100, d->TFb + ((size_t)(ty * 4 + f) * t.nFs) * 100, sizeof(double) * t.nFs * 100);
      }
    }
    if ((rc = upload(ctx, tsd.data(), (long)tsd.size(), &t.TSD)) != LRBMS_OK) return rc;
    for (int ty = 0; ty < 6; ++ty) {      // the same layout with the penalty parts of the face tables: local energy product
      double* dst = tsd.data() + (size_t)ty * KD * 100;
      for (int f = 0; f < 4; ++f) {
        memcpy(dst + ((size_t)t.nA + (size_t)f * 2 * t.nFs) * 100, d->TPo + ((size_t)(ty * 4 + f) * t.nFs) * 100, sizeof(double) * t.nFs * 100);
        memcpy(dst + ((size_t)t.nA + (size_t)f * 2 * t.nFs + t.nFs) * 100, d->TPb + ((size_t)(ty * 4 + f) * t.nFs) * 100, sizeof(double) * t.nFs * 100);
      }
    }
    if ((rc = upload(ctx, tsd.data(), (long)tsd.size(), &t.TSP)) != LRBMS_OK) return rc;
  }
  ctx->nbr_host.assign(nbr, nbr + (long)S * 7);
  ctx->side_padding = false;
  for (int i = 0; i < t.nbf; ++i) ctx->side_padding = ctx->side_padding || d->side_elem[i] < 0;
  ctx->has_mesh = true;
  return LRBMS_OK;
}

int lrbms3_assemble_system(lrbms3_ctx* ctx, int32_t Q, const double* lam, double* A_diag, double* A_cpl, void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || !lam || !A_diag || !A_cpl) return fail3(ctx, LRBMS_E_INVALID, "assemble_system: bad argument");
  const T3& t = ctx->t;
  if (t.nT % 6) return fail3(ctx, LRBMS_E_INVALID, "assemble_system: template is not made of whole cubes");
  const dim3 grid(6 * ((t.nT / 6 + 63) / 64), t.S);
  // sides with fewer faces than the padded row length (unequal cubes per direction) leave positions no kernel writes: zero
  HIP3(ctx, hipMemsetAsync(A_cpl, 0, sizeof(double) * (size_t)Q * t.S * 6 * t.ncf * 100, (hipStream_t)stream));
  for (int q = 0; q < Q; ++q) {
    hipLaunchKernelGGL((k3_asm<6, 7>), grid, dim3(256), 0, (hipStream_t)stream, t, Q, q, 0, lam, (const double*)nullptr,
                       (const double*)nullptr, A_diag, (double*)nullptr);
    for (int f = 0; f < 4; ++f)
      hipLaunchKernelGGL((k3_asm<7, 7>), grid, dim3(256), 0, (hipStream_t)stream, t, Q, q, f, lam, (const double*)nullptr,
                         (const double*)nullptr, A_diag, A_cpl);
  }
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_assemble_rhs(lrbms3_ctx* ctx, const double* f_smp, const double* lhat, double* b, double* f2, double* ceps,
                        double* bdiv, void* stream) {
  REQUIRE3(ctx);
  if (!f_smp || !lhat || !b || !f2 || !ceps || !bdiv) return fail3(ctx, LRBMS_E_INVALID, "assemble_rhs: null argument");
  const T3& t = ctx->t;
  if (t.nT % 6) return fail3(ctx, LRBMS_E_INVALID, "assemble_rhs: template is not made of whole cubes");
  const dim3 grid(6 * ((t.nT / 6 + 63) / 64), t.S);
  hipLaunchKernelGGL((k3_asm<4, 1>), grid, dim3(256), 0, (hipStream_t)stream, t, 1, 0, 0, f_smp, f_smp, lhat, b, (double*)nullptr);
  hipLaunchKernelGGL((k3_asm<5, 1>), grid, dim3(256), 0, (hipStream_t)stream, t, 1, 0, 0, f_smp, f_smp, lhat, bdiv, (double*)nullptr);
  hipLaunchKernelGGL(k3_scalars, dim3(t.S), dim3(256), 0, (hipStream_t)stream, t, f_smp, lhat, f2, ceps);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_assemble_products(lrbms3_ctx* ctx, int32_t Q, const double* lam, const double* lbar, const double* lhat, double* ebar,
                             double* Aaa, double* Aab, double* Bbb, void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || !lam || !lbar || !lhat || !ebar || !Aaa || !Aab || !Bbb)
    return fail3(ctx, LRBMS_E_INVALID, "assemble_products: bad argument");
  const T3& t = ctx->t;
  if (t.nT % 6) return fail3(ctx, LRBMS_E_INVALID, "assemble_products: template is not made of whole cubes");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(6 * ((t.nT / 6 + 63) / 64), t.S);
  hipLaunchKernelGGL((k3_asm<0, 7>), grid, dim3(256), 0, st, t, Q, 0, 0, lam, lbar, lhat, ebar, (double*)nullptr);
  const long blk = (long)t.S * t.nT * 100;
  for (int q = 0; q < Q; ++q)
    for (int q2 = q; q2 < Q; ++q2)         // A_aa[q][q'] = A_aa[q'][q]: one contraction, two stores
      hipLaunchKernelGGL((k3_asm<1, 7>), grid, dim3(256), 0, st, t, Q, q, q2, lam, lbar, lhat, Aaa + ((long)q * Q + q2) * blk,
                         q2 != q ? Aaa + ((long)q2 * Q + q) * blk : (double*)nullptr);
  for (int q = 0; q < Q; ++q)
    hipLaunchKernelGGL((k3_asm<2, 3>), grid, dim3(256), 0, st, t, Q, q, q, lam, lbar, lhat, Aab + (long)q * t.S * t.nT * 40,
                       (double*)nullptr);
  hipLaunchKernelGGL((k3_asm<3, 1>), grid, dim3(256), 0, st, t, Q, 0, 0, lam, lbar, lhat, Bbb, (double*)nullptr);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_assemble_energy_product(lrbms3_ctx* ctx, int32_t Q, const double* theta_bar, const double* lam, double* P_diag,
                                   void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || !theta_bar || !lam || !P_diag) return fail3(ctx, LRBMS_E_INVALID, "assemble_energy_product: bad argument");
  const T3& t = ctx->t;
  if (t.nT % 6) return fail3(ctx, LRBMS_E_INVALID, "assemble_energy_product: template is not made of whole cubes");
  hipStream_t st = (hipStream_t)stream;
  if (!ctx->thbar) {
    HIP3(ctx, hipMalloc((void**)&ctx->thbar, sizeof(double) * 8));
    ctx->owned.push_back(ctx->thbar);
  }
  double* thb = ctx->thbar;
  HIP3(ctx, hipMemcpyAsync(thb, theta_bar, sizeof(double) * Q, hipMemcpyHostToDevice, st));
  const dim3 grid(6 * ((t.nT / 6 + 63) / 64), t.S);
  hipLaunchKernelGGL((k3_asm<8, 7>), grid, dim3(256), 0, st, t, Q, 0, 0, lam, (const double*)nullptr, (const double*)thb, P_diag,
                     (double*)nullptr);
  for (int f = 0; f < 4; ++f)
    hipLaunchKernelGGL((k3_asm<9, 7>), grid, dim3(256), 0, st, t, Q, 0, f, lam, (const double*)nullptr, (const double*)thb, P_diag,
                       (double*)nullptr);
  LAUNCH3(ctx);
  HIP3(ctx, hipStreamSynchronize(st));          // theta_bar is a host buffer of the caller
  return LRBMS_OK;
}

int lrbms3_energy_product_apply(lrbms3_ctx* ctx, int32_t M, const double* P_diag, const double* X, double* Y, void* stream) {
  REQUIRE3(ctx);
  if (M < 1 || !P_diag || !X || !Y) return fail3(ctx, LRBMS_E_INVALID, "energy_product_apply: bad argument");
  const T3& t = ctx->t;
  const double one = 1.0;
  hipLaunchKernelGGL(k3_fom_apply, dim3(t.nT, t.S), dim3(256), 0, (hipStream_t)stream, t, 1, M, make_theta(1, &one), P_diag,
                     (const double*)nullptr, X, Y);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_assemble_flux(lrbms3_ctx* ctx, int32_t Q, const double* lam, double* Cf, void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || !lam || !Cf) return fail3(ctx, LRBMS_E_INVALID, "assemble_flux: bad argument");
  const T3& t = ctx->t;
  hipLaunchKernelGGL(k3_assemble_flux, dim3(t.nT, t.S_ext, Q), dim3(64), 0, (hipStream_t)stream, t, lam, Cf);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int64_t lrbms3_work_size(lrbms3_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh) return -1;
  const T3& t = ctx->t;
  return (int64_t)t.S * t.nrt * Q * N + (int64_t)t.S * t.nnodes * N + (int64_t)t.S * t.nbd * N;
}

int lrbms3_project_estimate(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* A_diag, const double* A_cpl,
                            const double* b, const double* ebar, const double* Aaa, const double* Aab, const double* Bbb,
                            const double* bdiv, const double* Cf, double* work, double* B_sys, double* rhs_red, double* G_nc,
                            double* G_bb, double* G_rdd, double* G_ab, double* G_aa, double* r_fd, double* Rb, double* Yb,
                            double* Dp, double* Xab, double* As, double* Cn, void* stream) {
  return lrbms3_project_estimate_phase(ctx, 0, Q, N, V, A_diag, A_cpl, b, ebar, Aaa, Aab, Bbb, bdiv, Cf, work, B_sys, rhs_red, G_nc,
                                       G_bb, G_rdd, G_ab, G_aa, r_fd, Rb, Yb, Dp, Xab, As, Cn, stream);
}

int lrbms3_project_estimate_phase(lrbms3_ctx* ctx, int32_t phase, int32_t Q, int32_t N, const double* V, const double* A_diag,
                                  const double* A_cpl, const double* b, const double* ebar, const double* Aaa, const double* Aab,
                                  const double* Bbb, const double* bdiv, const double* Cf, double* work, double* B_sys,
                                  double* rhs_red, double* G_nc, double* G_bb, double* G_rdd, double* G_ab, double* G_aa,
                                  double* r_fd, double* Rb, double* Yb, double* Dp, double* Xab, double* As, double* Cn, void* stream) {
  REQUIRE3(ctx);
  if (phase < 0 || phase > 2) return fail3(ctx, LRBMS_E_INVALID, "project_estimate: phase must be 0, 1 or 2");
  const bool own = phase != 2, side = phase != 1;        // 1: everything that reads rank-local slabs only; 2: the rest
  if (Q < 1 || Q > 8 || N < 1 || N > 64 || Q * N > 64)
    return fail3(ctx, LRBMS_E_INVALID, "project_estimate: needs N <= 64 and Q N <= 64");
  if (!V || !A_diag || !A_cpl || !b || !ebar || !Aaa || !Aab || !Bbb || !bdiv || !Cf || !work || !B_sys || !rhs_red || !G_nc ||
      !G_bb || !G_rdd || !G_ab || !G_aa || !r_fd || !Rb || !Yb || !Dp || !Xab || !As || !Cn)
    return fail3(ctx, LRBMS_E_INVALID, "project_estimate: null argument");
  const T3& t = ctx->t;
  hipStream_t st = (hipStream_t)stream;
  ctx->ktime_n = ctx->ktime ? ctx->ktime_n : 0;
  double* Rs = work;
  double* Avg = work + (long)t.S * t.nrt * Q * N;
  double* Zb = Avg + (long)t.S * t.nnodes * N;
  // Three independent chains, each on its own stream so that the latency-bound preparation kernels run beside the MFMA kernels:
  //   caller's stream  B_sys (diagonal + coupling blocks), G_aa, rhs_red         -- need nothing but V and the element blocks
  //   aux 0            flux image  ->  G_ab, G_bb, G_rdd, r_fd, side-face factors
  //   aux 1            node averages  ->  G_nc, side-node factors
  // (while per-kernel timing is on, everything runs on the caller's stream: overlapping kernels would stretch each other's
  // event intervals)
  const bool serial = ctx->ktime || ctx->opt_serial != 0;      // LRBMS3_OPT_SERIAL: rocprofv3 kernel statistics of a serial pass
  hipStream_t sf = serial ? st : ctx->aux[0], sn = serial ? st : ctx->aux[1];
  HIP3(ctx, hipEventRecord(ctx->ev_fork, st));
  HIP3(ctx, hipStreamWaitEvent(sf, ctx->ev_fork, 0));
  HIP3(ctx, hipStreamWaitEvent(sn, ctx->ev_fork, 0));
  GA a{t, Q, N, V, A_diag, A_cpl, ebar, Aaa, Aab, Bbb, Rs, Avg, nullptr, Zb, G_rdd, r_fd, bdiv, b, rhs_red, Yb, Dp, Xab, 1, nullptr};
  if (own && ctx->side_padding) {      // padded side-face rows are written by no kernel: define them (the estimate multiplies them by 0)
    HIP3(ctx, hipMemsetAsync(Yb, 0, sizeof(double) * (size_t)t.S * t.nbf * Q * N, sf));
    HIP3(ctx, hipMemsetAsync(Dp, 0, sizeof(double) * (size_t)t.S * t.nbf * Q * N, sf));
    HIP3(ctx, hipMemsetAsync(Xab, 0, sizeof(double) * (size_t)Q * t.S * t.nbf * N, sf));
  }
  const int tn = (N + 15) / 16, tq = (Q * N + 15) / 16;
  const int nw_env = ctx->opt_waves;      // LRBMS3_OPT_WAVES
  const int nw = nw_env > 0 ? nw_env : 4;
  const int nw_s = nw_env > 0 ? nw_env : 8;      // kernels with one workgroup per subdomain only: more waves each
  int bad = 0;
  {
    const int r0 = own ? 0 : t.nrt, r1 = side ? t.nrt + t.nbf : t.nrt;      // own faces | side faces (the neighbours' share: halo)
    KScope3 k(ctx, own ? "k3_flux" : "k3_flux<side>", sf);
    hipLaunchKernelGGL(k3_flux, dim3(xcd_grid((r1 - r0 + 4 * FLUX_R * FLUX_LOOP - 1) / (4 * FLUX_R * FLUX_LOOP), t.S)), dim3(256), 0, sf, t, Q, N, r0, r1, V, Cf, Rs, Rb);
  }
  {
    const int r0 = own ? 0 : t.nnodes, r1 = side ? t.nnodes + 6 * t.nvs : t.nnodes;
    KScope3 k(ctx, own ? "k3_node_avg" : "k3_node_avg<side>", sn);
    hipLaunchKernelGGL(k3_node_avg, dim3(xcd_grid((r1 - r0 + 4 * NODE_LOOP - 1) / (4 * NODE_LOOP), t.S)), dim3(256), 0, sn, t, N, r0, r1, V, Avg, As);
  }
  const int npair = Q * (Q + 1) / 2;       // A_aa: pairs q <= q', the transposed blocks are written from the same accumulators
  // K-split for small per-rank subdomain counts (the 4 x 4 x 4 tile of an 8-GPU run has 64): one workgroup per (subdomain,
  // operator) would leave most of the 256 CUs idle, so the element range is dealt to ksplit workgroups and k3_pg_combine sums
  // their partial results in a fixed order.  Off (ksplit = 1, no extra launch) from ~400 workgroups per kernel on.
  const int ks_env = ctx->opt_ksplit;      // LRBMS3_OPT_KSPLIT
  // target workgroups per launch, measured on the 4 x 4 x 4 tile of an 8-GPU run (64 subdomains; LRBMS3_OPT_KSPLIT 2 / 4 / 6 / auto):
  // the kernels with ONE workgroup per subdomain and ~240 VGPRs (BB: 4 waves, NC: 8 waves) want one workgroup per CU (k3_pg<BB>
  // 104 us at 256 workgroups, 124 at 512; <NC> 44 vs 51), G_aa three per CU (70 us at 768, 85 at 576), the others two
  auto ksplit_of = [&](int nblocks, int target) {
    if (ks_env > 0) return ks_env < 8 ? ks_env : 8;
    if (nblocks >= 384) return 1;
    const int k = (target + nblocks - 1) / nblocks;
    return k < 1 ? 1 : (k < 8 ? k : 8);
  };
  const int QNl = Q * N;
  const int ks_sys = ksplit_of(Q * t.S, 512), ks_aaa = ksplit_of(npair * t.S, 768), ks_nc = ksplit_of(t.S, 256), ks_ab = ksplit_of(Q * t.S, 512),
            ks_bb = ksplit_of(t.S, 256), ks_cpl = ksplit_of(Q * t.S * 6, 512);
  const long need_sys = ks_sys > 1 ? (long)Q * t.S * ks_sys * pg_part_size<G_SYS>(N, QNl) : 0,
             need_aaa = ks_aaa > 1 ? (long)npair * t.S * ks_aaa * pg_part_size<G_AAA>(N, QNl) : 0,
             need_nc = ks_nc > 1 ? (long)t.S * ks_nc * pg_part_size<G_NC>(N, QNl) : 0,
             need_ab = ks_ab > 1 ? (long)Q * t.S * ks_ab * pg_part_size<G_AB>(N, QNl) : 0,
             need_bb = ks_bb > 1 ? (long)t.S * ks_bb * pg_part_size<G_BB>(N, QNl) : 0,
             need_cpl = ks_cpl > 1 ? (long)Q * t.S * 6 * ks_cpl * pg_part_size<G_CPL>(N, QNl) : 0;
  const long need = need_sys + need_aaa + need_nc + need_ab + need_bb + need_cpl;
  if (need > ctx->pg_part_cap) {
    HIP3(ctx, hipDeviceSynchronize());
    if (ctx->pg_part) (void)hipFree(ctx->pg_part);
    ctx->pg_part = nullptr;
    ctx->pg_part_cap = 0;
    HIP3(ctx, hipMalloc((void**)&ctx->pg_part, sizeof(double) * need));
    ctx->pg_part_cap = need;
  }
  double* part_sys = ctx->pg_part;
  double* part_aaa = part_sys + need_sys;
  double* part_nc = part_aaa + need_aaa;
  double* part_ab = part_nc + need_nc;
  double* part_bb = part_ab + need_ab;
  double* part_cpl = part_bb + need_bb;
  // (launch order measured: the MFMA-bound G_aa kernel first on the caller's stream, beside the latency-bound preparation
  // kernels of the other two chains, the HBM-bound system kernel after it: 2.39 -> 2.29 ms; a fourth stream for G_aa: slower)
  if (own) {
    KScope3 k(ctx, "k3_pg<AAA>", st);
    a.ksplit = ks_aaa;
    a.part = part_aaa;
    a.out = G_aa;
    bad |= dispatch_pg<G_AAA>(a, npair * t.S, tn, tn, nw, st);
  }
  if (own) {
    KScope3 k(ctx, "k3_pg<SYS>", st);
    a.ksplit = ks_sys;
    a.part = part_sys;
    a.out = B_sys;
    bad |= dispatch_pg<G_SYS>(a, Q * t.S, tn, tn, nw, st);
  }
  if (own) {
    KScope3 k(ctx, "k3_pg<AB>", sf);
    a.ksplit = ks_ab;
    a.part = part_ab;
    a.out = G_ab;
    bad |= dispatch_pg<G_AB>(a, Q * t.S, tn, tq, nw, sf);
  }
  if (own) {
    KScope3 k(ctx, "k3_pg<NC>", sn);
    a.ksplit = ks_nc;
    a.part = part_nc;
    a.out = G_nc;
    bad |= dispatch_pg<G_NC>(a, t.S, tn, tn, nw_s, sn);
  }
  if (own) {
    KScope3 k(ctx, "k3_pg<BB>", sf);
    a.ksplit = ks_bb;
    a.part = part_bb;
    a.out = G_bb;
    bad |= dispatch_pg<G_BB>(a, t.S, tq, tq, nw, sf);          // 4 waves: 291 us, 8 waves: 307 us (tools/nw_sweep.sh)
  }
  if (own) {
    KScope3 k(ctx, "k3_side_nc", sn);
    hipLaunchKernelGGL(k3_side_nc, dim3((t.nb + 3) / 4, t.S), dim3(256), 0, sn, t, N, Zb, Cn);
  }
  if (side) {
    KScope3 k(ctx, "k3_pg<CPL>", st);
    a.ksplit = ks_cpl;
    a.part = part_cpl;
    a.out = B_sys;
    bad |= dispatch_pg<G_CPL>(a, Q * t.S * 6, tn, tn, nw, st);
  }
  if (bad) return fail3(ctx, LRBMS_E_INVALID, "project_estimate: unsupported tile shape");
  HIP3(ctx, hipEventRecord(ctx->ev_join[0], sf));
  HIP3(ctx, hipEventRecord(ctx->ev_join[1], sn));
  HIP3(ctx, hipStreamWaitEvent(st, ctx->ev_join[0], 0));
  HIP3(ctx, hipStreamWaitEvent(st, ctx->ev_join[1], 0));
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_kernel_timing(lrbms3_ctx* ctx, int32_t enable) {
  if (!ctx) return LRBMS_E_INVALID;
  ctx->ktime = enable != 0;
  ctx->ktime_n = 0;
  return LRBMS_OK;
}

int lrbms3_kernel_timing_read(lrbms3_ctx* ctx, char* names, int64_t names_cap, double* ms, int32_t cap, int32_t* count) {
  if (!ctx || !names || !ms || !count) return LRBMS_E_INVALID;
  HIP3(ctx, hipDeviceSynchronize());
  std::string all;
  int n = 0;
  for (int i = 0; i < ctx->ktime_n && n < cap; ++i) {
    float f = 0.f;
    if (hipEventElapsedTime(&f, ctx->ktimers[i].e0, ctx->ktimers[i].e1) != hipSuccess) continue;
    ms[n++] = f;
    all += ctx->ktimers[i].name;
    all += "\n";
  }
  if ((int64_t)all.size() + 1 > names_cap) return fail3(ctx, LRBMS_E_INVALID, "kernel_timing_read: names buffer too small");
  memcpy(names, all.c_str(), all.size() + 1);
  *count = n;
  ctx->ktime_n = 0;
  return LRBMS_OK;
}

int lrbms3_reduced_estimate(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* u, const double* G_nc,
                            const double* G_bb, const double* G_rdd, const double* G_ab, const double* G_aa, const double* r_fd,
                            const double* Rb, const double* Yb, const double* Dp, const double* Xab, const double* As,
                            const double* Cn, const double* ebar, const double* Bbb, const double* bdiv, const double* f2,
                            const double* ceps, double hdiam, double* eta_loc, void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || N < 1 || Q * N > 64 || !theta || !u || !eta_loc)
    return fail3(ctx, LRBMS_E_INVALID, "reduced_estimate: bad argument");
  const T3& t = ctx->t;
  EA a{u, G_nc, G_bb, G_rdd, G_ab, G_aa, r_fd, Rb, Yb, Dp, Xab, As, Cn, ebar, Bbb, bdiv, f2, ceps, hdiam, eta_loc};
  const size_t lds = sizeof(double) * (7 * N + Q * N + t.nbf + t.nb + 256);
  if (lds > 64 * 1024) return fail3(ctx, LRBMS_E_INVALID, "reduced_estimate: template too large for the LDS");
  hipLaunchKernelGGL(k3_estimate, dim3(t.S), dim3(256), lds, (hipStream_t)stream, t, Q, N, make_theta(Q, theta), a);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_reduced_estimate_batch(lrbms3_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* u,
                                  const double* G_nc, const double* G_bb, const double* G_rdd, const double* G_ab, const double* G_aa,
                                  const double* r_fd, const double* Rb, const double* Yb, const double* Dp, const double* Xab,
                                  const double* As, const double* Cn, const double* ebar, const double* Bbb, const double* bdiv,
                                  const double* f2, const double* ceps, double hdiam, double* eta_loc, void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || N < 1 || Q * N > 64 || nmu < 1 || !theta || !u || !eta_loc)
    return fail3(ctx, LRBMS_E_INVALID, "reduced_estimate_batch: bad argument");
  const T3& t = ctx->t;
  EA a{u, G_nc, G_bb, G_rdd, G_ab, G_aa, r_fd, Rb, Yb, Dp, Xab, As, Cn, ebar, Bbb, bdiv, f2, ceps, hdiam, eta_loc};
  const bool est16_env = ctx->opt_estimate_valu == 0;      // LRBMS3_OPT_ESTIMATE_VALU
  const size_t lds16 = sizeof(double) * ((size_t)(7 * N + Q * N + t.nbf + t.nb) * EST16 + EST_NW * 6 * 16 + 4 * EST_NW * 16 + 3 * t.nvs + 1);
  if (est16_env && lds16 <= 160 * 1024 - 2048) {
    if (lds16 > 64 * 1024)
      HIP3(ctx, hipFuncSetAttribute((const void*)k3_estimate_batch16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
    for (int m0 = 0; m0 < nmu; m0 += EST16) {
      TB16 th{};
      for (int m = 0; m < EST16 && m0 + m < nmu; ++m)
        for (int q = 0; q < Q; ++q) th.v[m][q] = theta[(m0 + m) * Q + q];
      hipLaunchKernelGGL(k3_estimate_batch16, dim3(t.S), dim3(64 * EST_NW), lds16, (hipStream_t)stream, t, Q, N, nmu, m0, th, a);
    }
    LAUNCH3(ctx);
    return LRBMS_OK;
  }
  const size_t lds = sizeof(double) * ((size_t)(7 * N + Q * N + t.nbf + t.nb) * EST_MB + 256);
  if (lds > 64 * 1024) return fail3(ctx, LRBMS_E_INVALID, "reduced_estimate_batch: template too large for the LDS");
  for (int m0 = 0; m0 < nmu; m0 += EST_MB) {
    TB8 th{};
    for (int m = 0; m < EST_MB && m0 + m < nmu; ++m)
      for (int q = 0; q < Q; ++q) th.v[m][q] = theta[(m0 + m) * Q + q];
    hipLaunchKernelGGL(k3_estimate_batch, dim3(t.S), dim3(256), lds, (hipStream_t)stream, t, Q, N, nmu, m0, th, a);
  }
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int64_t lrbms3_reduced_solve_work_size(lrbms3_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh) return -1;
  const int64_t S = ctx->t.S;
  return S * 7 * N * N + S * N * N + 6 * S * N + 5 * S + 16;
}

int lrbms3_reduced_solve(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, const double* rhs_red,
                         double* work, double* u, double rtol, int32_t max_iter, double* info, void* stream) {
  REQUIRE3(ctx);
  const T3& t = ctx->t;
  if (t.S_ext != t.S) return fail3(ctx, LRBMS_E_INVALID, "reduced_solve: needs all subdomains on this rank");
  if (Q < 1 || Q > 8 || N < 1 || N > 64 || !theta || !B_sys || !rhs_red || !work || !u)
    return fail3(ctx, LRBMS_E_INVALID, "reduced_solve: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long S = t.S, per_q = S * 7 * N * N;
  double* Amu = work;
  double* Dinv = Amu + per_q;
  double* r = Dinv + S * N * N;
  double* z = r + S * N;
  double* p0 = z + S * N;
  double* p1 = p0 + S * N;
  double* Ap = p1 + S * N;
  double* prz0 = Ap + S * N;     // partial sums, ping-pong
  double* prz1 = prz0 + S;
  double* ppap = prz1 + S;
  double* prr = ppap + S;
  double* scal = prr + S;        // [0] rr, [1] bb
  hipLaunchKernelGGL(k3_combine, dim3((unsigned)((per_q + 255) / 256)), dim3(256), 0, st, per_q, Q, make_theta(Q, theta), B_sys, Amu);
  hipLaunchKernelGGL(k3_block_inverse, dim3(S), dim3(256), sizeof(double) * N * (2 * N + 1), st, N, Amu, Dinv);
  hipLaunchKernelGGL(k3_pcg_init, dim3(S), dim3(64), 0, st, N, rhs_red, Dinv, u, r, z, p0, prz0, prr);
  hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)S, prr, scal + 1);
  LAUNCH3(ctx);
  double bb = 0.0;
  HIP3(ctx, hipMemcpyAsync(&bb, scal + 1, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP3(ctx, hipStreamSynchronize(st));
  if (info) info[0] = 0, info[1] = 0;
  if (bb == 0.0) return LRBMS_OK;
  const size_t lds = sizeof(double) * (7 * N + 64);
  double *po = p0, *pn = p1, *rz_old = prz1, *rz_cur = prz0;
  int it = 0;
  double rel = 1.0;
  const int check = 8;
  while (it < max_iter) {
    for (int k = 0; k < check && it < max_iter; ++k, ++it) {
      hipLaunchKernelGGL(k3_pcg_matvec, dim3(S), dim3(64), lds, st, t, N, it == 0 ? 1 : 0, Amu, z, po, pn, Ap, rz_cur, rz_old, ppap);
      hipLaunchKernelGGL(k3_pcg_update, dim3(S), dim3(64), 0, st, (int)S, N, Dinv, pn, Ap, u, r, z, rz_cur, ppap, rz_old, prr);
      std::swap(po, pn);
      std::swap(rz_old, rz_cur);
    }
    hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)S, prr, scal);
    LAUNCH3(ctx);
    double rr = 0.0;
    HIP3(ctx, hipMemcpyAsync(&rr, scal, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP3(ctx, hipStreamSynchronize(st));
    rel = sqrt(rr / bb);
    if (!(rel == rel)) return fail3(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve: NaN residual");
    if (rel <= rtol) break;
  }
  if (info) info[0] = it, info[1] = rel;
  if (rel > rtol) return fail3(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve: not converged");
  return LRBMS_OK;
}

// doubles of work per group of <= 16 parameters of the batched reduced solve
static long reduced_batch_group_size(long S, int N) {
  return S * 7 * N * N + S * N * N + 5 * S * N * 16 + 4 * S * 16 + 5 * 16 + 16 + 3 * S * 16;      // (+ y0, prc [2] of the coarse level)
}

int64_t lrbms3_reduced_solve_batch_work_size(lrbms3_ctx* ctx, int32_t N, int32_t nmu) {
  if (!ctx || !ctx->has_mesh || nmu < 1) return -1;
  return (int64_t)((nmu + 15) / 16) * reduced_batch_group_size(ctx->t.S, N);
}

int lrbms3_reduced_solve_batch(lrbms3_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* B_sys,
                               const double* rhs_red, double* work, double* u, double rtol, int32_t max_iter, double* info,
                               void* stream) {
  REQUIRE3(ctx);
  const T3& t = ctx->t;
  if (t.S_ext != t.S) return fail3(ctx, LRBMS_E_INVALID, "reduced_solve_batch: needs all subdomains on this rank");
  if (Q < 1 || Q > 8 || N < 1 || N > 32 || nmu < 1 || nmu > 64 || !theta || !B_sys || !rhs_red || !work || !u)
    return fail3(ctx, LRBMS_E_INVALID, "reduced_solve_batch: needs N <= 32 and nmu <= 64");
  hipStream_t st = (hipStream_t)stream;
  const long S = t.S, per_q = S * 7 * N * N;
  // Up to four groups of <= 16 parameters, each an independent CG on its own stream (the caller's and the library's three side
  // streams): a group's kernels are 512 small workgroups that wait on memory most of the time, so two or three groups share the
  // chip at little cost to each other.  Launches are interleaved iteration by iteration; residuals are looked at together.
  const int ng = (nmu + 15) / 16;
  const long gsize = reduced_batch_group_size(S, N);
  struct Group {
    int nm, m0;
    TB th;
    double *Amu, *Dinv, *r, *z, *po, *pn, *Ap, *prz, *ppap, *prr, *scal, *y0, *prc;
    hipStream_t st;
    bool done;
    double rel;
    int it;
  } g[4];
  const double* A0inv = ctx->user_pc_N == N ? ctx->user_pc : nullptr;
  for (int k = 0; k < ng; ++k) {
    Group& G = g[k];
    G.m0 = 16 * k;
    G.nm = nmu - G.m0 < 16 ? nmu - G.m0 : 16;
    G.st = k == 0 ? st : ctx->aux[k - 1];
    G.done = false;
    G.rel = 0.0;
    G.it = 0;
    const long nv = S * N * 16;
    double* w = work + k * gsize;
    G.Amu = w;                        // blocks at the group-mean theta: only its diagonal blocks are used (preconditioner)
    G.Dinv = G.Amu + per_q;
    G.r = G.Dinv + S * N * N;
    G.z = G.r + nv;
    G.po = G.z + nv;
    G.pn = G.po + nv;
    G.Ap = G.pn + nv;
    G.prz = G.Ap + nv;                // [2][S][16]: r.z partials of the last two updates (beta needs both)
    G.ppap = G.prz + 2 * S * 16;
    G.prr = G.ppap + S * 16;
    G.scal = G.prr + S * 16;          // [5][16]: only the residual norms ([3], [4]) are reduced by a kernel of their own
    G.y0 = G.scal + 5 * 16 + 16;      // coarse level (lrbms3_reduced_precond_use): correction and its r.z contributions [2][S][16]
    G.prc = G.y0 + S * 16;
    G.th = TB{};
  }
  if (ng > 1) {
    HIP3(ctx, hipEventRecord(ctx->ev_fork, st));
    for (int k = 1; k < ng; ++k) HIP3(ctx, hipStreamWaitEvent(g[k].st, ctx->ev_fork, 0));
  }
  for (int k = 0; k < ng; ++k) {
    Group& G = g[k];
    QV mean{};
    for (int m = 0; m < G.nm; ++m)
      for (int q = 0; q < Q; ++q) {
        G.th.v[m][q] = theta[(G.m0 + m) * Q + q];
        mean.v[q] += theta[(G.m0 + m) * Q + q] / G.nm;
      }
    if (A0inv) {
      G.Dinv = const_cast<double*>(A0inv) + S * S;      // the prebuilt preconditioner carries its inverse diagonal blocks (read only)
    } else {
      hipLaunchKernelGGL(k3_combine, dim3((unsigned)((per_q + 255) / 256)), dim3(256), 0, G.st, per_q, Q, mean, B_sys, G.Amu);
      hipLaunchKernelGGL(k3_block_inverse, dim3(S), dim3(256), sizeof(double) * N * (2 * N + 1), G.st, N, G.Amu, G.Dinv);
    }
    HIP3(ctx, hipMemsetAsync(G.scal, 0, sizeof(double) * 80, G.st));
    hipLaunchKernelGGL(k3b_init, dim3(S), dim3(512), sizeof(double) * (N + 32 * 16), G.st, N, G.nm, nmu, rhs_red, G.Dinv, u + G.m0, G.r,
                       G.z, G.po, G.prz, G.prr);
    if (A0inv)
      hipLaunchKernelGGL(k3b_coarse_apply, dim3((unsigned)((S + 15) / 16)), dim3(1024), 0, G.st, (int)S, N, G.nm, A0inv, G.r, G.y0, G.prc);
    hipLaunchKernelGGL(k3b_reduce, dim3(1), dim3(256), 0, G.st, (int)S, G.nm, G.prr, G.scal + 64);
  }
  LAUNCH3(ctx);
  if (info) info[0] = 0, info[1] = 0;
  const int check = A0inv ? 12 : 8;    // iterations between two looks at the residuals (a host synchronisation each)
  const size_t lds_mv = sizeof(double) * (7 * N * 16 + 32 * 16), lds_up = sizeof(double) * (N * 16 + 32 * 16);
  const size_t lds_mm = sizeof(double) * (7 * 32 * 16 + 4 * (N <= 16 ? 1 : 2) * 256 + 32 * 16);
  const bool mfma_mv = ctx->opt_solve_valu == 0;      // LRBMS3_OPT_SOLVE_VALU: the VALU panel matvec
  int rc = LRBMS_OK;
  bool all_done = false;
  while (!all_done) {
    for (int c = 0; c < check; ++c)
      for (int k = 0; k < ng; ++k) {
        Group& G = g[k];
        if (G.done || G.it >= max_iter) continue;
        const int it = G.it;
        double* rz_cur = G.prz + (it & 1) * S * 16;          // written by the previous update (or k3b_init)
        double* rz_nxt = G.prz + ((it + 1) & 1) * S * 16;    // holds the r.z of the update before that until this update overwrites it
        double* rc_cur = G.prc + (it & 1) * S * 16;
        double* rc_nxt = G.prc + ((it + 1) & 1) * S * 16;
        const CoarseB cb{A0inv ? G.y0 : nullptr, rc_cur, rc_nxt};
        if (mfma_mv && N <= 16)
          hipLaunchKernelGGL(k3b_matvec_mfma<1>, dim3(S), dim3(256), lds_mm, G.st, t, Q, N, G.nm, it == 0 ? 1 : 0, G.th, B_sys, G.z, G.po,
                             G.pn, G.Ap, rz_cur, rz_nxt, G.ppap, cb);
        else if (mfma_mv)
          hipLaunchKernelGGL(k3b_matvec_mfma<2>, dim3(S), dim3(256), lds_mm, G.st, t, Q, N, G.nm, it == 0 ? 1 : 0, G.th, B_sys, G.z, G.po,
                             G.pn, G.Ap, rz_cur, rz_nxt, G.ppap, cb);
        else
          hipLaunchKernelGGL(k3b_matvec, dim3(S), dim3(512), lds_mv, G.st, t, Q, N, G.nm, it == 0 ? 1 : 0, G.th, B_sys, G.z, G.po, G.pn,
                             G.Ap, rz_cur, rz_nxt, G.ppap, cb);
        hipLaunchKernelGGL(k3b_update, dim3(S), dim3(512), lds_up, G.st, N, G.nm, nmu, G.Dinv, G.pn, G.Ap, u + G.m0, G.r, G.z, (int)S,
                           rz_cur, A0inv ? rc_cur : nullptr, G.ppap, rz_nxt, G.prr);
        if (A0inv)    // the coarse correction of the new residual: read by the next matvec (direction) and by the next update (r.z)
          hipLaunchKernelGGL(k3b_coarse_apply, dim3((unsigned)((S + 15) / 16)), dim3(1024), 0, G.st, (int)S, N, G.nm, A0inv, G.r, G.y0,
                             rc_nxt);
        std::swap(G.po, G.pn);
        ++G.it;
      }
    double rr[4][32];
    for (int k = 0; k < ng; ++k) {
      Group& G = g[k];
      if (G.done) continue;
      hipLaunchKernelGGL(k3b_reduce, dim3(1), dim3(256), 0, G.st, (int)S, G.nm, G.prr, G.scal + 48);
      HIP3(ctx, hipMemcpyAsync(rr[k], G.scal + 48, sizeof(double) * 32, hipMemcpyDeviceToHost, G.st));   // [3] residuals, [4] |b|^2
    }
    LAUNCH3(ctx);
    all_done = true;
    for (int k = 0; k < ng; ++k) {
      Group& G = g[k];
      if (G.done) continue;
      HIP3(ctx, hipStreamSynchronize(G.st));
      G.rel = 0.0;
      for (int m = 0; m < G.nm; ++m) {
        const double bbm = rr[k][16 + m], rm = bbm > 0.0 ? sqrt(rr[k][m] / bbm) : 0.0;
        if (!(rm == rm)) rc = LRBMS_E_NOT_CONVERGED;
        G.rel = rm > G.rel ? rm : G.rel;
      }
      if (G.rel <= rtol || G.it >= max_iter || rc != LRBMS_OK) G.done = true;
      if (!G.done) all_done = false;
    }
  }
  if (ng > 1)
    for (int k = 1; k < ng; ++k) {
      HIP3(ctx, hipEventRecord(ctx->ev_join[k - 1], g[k].st));
      HIP3(ctx, hipStreamWaitEvent(st, ctx->ev_join[k - 1], 0));
    }
  int it = 0;
  double rel = 0.0;
  for (int k = 0; k < ng; ++k) {
    it = g[k].it > it ? g[k].it : it;
    rel = g[k].rel > rel ? g[k].rel : rel;
  }
  if (info) info[0] = it, info[1] = rel;
  if (rc != LRBMS_OK) return fail3(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve_batch: NaN residual");
  if (rel > rtol) return fail3(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve_batch: not converged");
  return LRBMS_OK;
}

int64_t lrbms3_reduced_precond_size(lrbms3_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1) return -1;
  return (int64_t)ctx->t.S * ctx->t.S + (int64_t)ctx->t.S * N * N;          // coarse inverse | inverse diagonal blocks
}

int64_t lrbms3_reduced_precond_work_size(lrbms3_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1) return -1;
  const int64_t S = ctx->t.S;
  return std::max<int64_t>(2 * S * S + 2, S * 7 * N * N);
}

int lrbms3_reduced_precond_build(lrbms3_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, double* work,
                                 double* pc, void* stream) {
  REQUIRE3(ctx);
  const T3& t = ctx->t;
  if (t.S_ext != t.S) return fail3(ctx, LRBMS_E_INVALID, "reduced_precond_build: needs all subdomains on this rank");
  if (Q < 1 || Q > 8 || N < 1 || N > 32 || !theta || !B_sys || !work || !pc)
    return fail3(ctx, LRBMS_E_INVALID, "reduced_precond_build: bad argument (the batched solve it serves takes N <= 32)");
  hipStream_t st = (hipStream_t)stream;
  const long S = t.S;
  double* A0 = work;
  double* Id = A0 + S * S;
  rocblas_int* pinfo = (rocblas_int*)(Id + S * S);
  if (!ctx->blas) {
    rocblas_handle h = nullptr;
    if (rocblas_create_handle(&h) != rocblas_status_success) return fail3(ctx, LRBMS_E_HIP, "rocblas_create_handle failed");
    ctx->blas = h;
  }
  rocblas_handle h = (rocblas_handle)ctx->blas;
  if (rocblas_set_stream(h, st) != rocblas_status_success) return fail3(ctx, LRBMS_E_HIP, "rocblas_set_stream failed");
  HIP3(ctx, hipMemsetAsync(A0, 0, sizeof(double) * 2 * S * S, st));
  hipLaunchKernelGGL(k3r_coarse_fill, dim3((unsigned)((S * 7 + 255) / 256)), dim3(256), 0, st, t, Q, N, make_theta(Q, theta), B_sys, A0, Id);
  LAUNCH3(ctx);
  if (rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)S, A0, (rocblas_int)S, pinfo) != rocblas_status_success)
    return fail3(ctx, LRBMS_E_HIP, "rocsolver_dpotrf failed");
  rocblas_int hinfo = 0;
  HIP3(ctx, hipMemcpyAsync(&hinfo, pinfo, sizeof(rocblas_int), hipMemcpyDeviceToHost, st));
  HIP3(ctx, hipStreamSynchronize(st));
  if (hinfo != 0) return fail3(ctx, LRBMS_E_INVALID, "reduced_precond_build: the coarse matrix is not positive definite (first basis vectors)");
  if (rocsolver_dpotrs(h, rocblas_fill_lower, (rocblas_int)S, (rocblas_int)S, A0, (rocblas_int)S, Id, (rocblas_int)S) !=
      rocblas_status_success)
    return fail3(ctx, LRBMS_E_HIP, "rocsolver_dpotrs failed");
  HIP3(ctx, hipMemcpyAsync(pc, Id, sizeof(double) * S * S, hipMemcpyDeviceToDevice, st));
  // the inverse diagonal blocks at the same reference parameter (the dense work is done: its space holds the combined blocks)
  const long per_q = S * 7 * N * N;
  hipLaunchKernelGGL(k3_combine, dim3((unsigned)((per_q + 255) / 256)), dim3(256), 0, st, per_q, Q, make_theta(Q, theta), B_sys, work);
  hipLaunchKernelGGL(k3_block_inverse, dim3(S), dim3(256), sizeof(double) * N * (2 * N + 1), st, N, work, pc + S * S);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

int lrbms3_reduced_precond_use(lrbms3_ctx* ctx, int32_t N, const double* pc) {
  REQUIRE3(ctx);
  ctx->user_pc = pc;
  ctx->user_pc_N = pc ? N : 0;
  return LRBMS_OK;
}

int lrbms3_fom_coarse_space(lrbms3_ctx* ctx, int32_t nc, const double* Phi) {
  REQUIRE3(ctx);
  if (nc < 0 || nc > 4 || (nc > 0 && !Phi)) return fail3(ctx, LRBMS_E_INVALID, "fom_coarse_space: 0 <= nc <= 4 functions, Phi [n][nc]");
  if (!ctx->has_mesh) return fail3(ctx, LRBMS_E_STATE, "fom_coarse_space: upload the mesh first");
  const T3& t = ctx->t;
  ctx->fom_nc = 0;
  ctx->fom_pc_M = 0;                     // a kept coarse inverse belongs to the old space
  if (nc == 0) return LRBMS_OK;
  std::vector<double> padded((size_t)t.n * 4, 0.0);
  for (long d = 0; d < t.n; ++d)
    for (int k = 0; k < nc; ++k) padded[d * 4 + k] = Phi[d * nc + k];
  if (!ctx->fom_phi) {
    HIP3(ctx, hipMalloc((void**)&ctx->fom_phi, sizeof(double) * (size_t)t.n * 4));
    ctx->owned.push_back(ctx->fom_phi);
  }
  HIP3(ctx, hipMemcpy(ctx->fom_phi, padded.data(), sizeof(double) * padded.size(), hipMemcpyHostToDevice));
  ctx->fom_nc = nc;
  return LRBMS_OK;
}

int lrbms3_fom_precond_keep(lrbms3_ctx* ctx, int32_t keep) {
  REQUIRE3(ctx);
  ctx->fom_keep = keep != 0;
  if (!keep) {
    if (ctx->fom_pc) (void)hipFree(ctx->fom_pc);
    ctx->fom_pc = nullptr;
    ctx->fom_pc_M = 0;
  }
  return LRBMS_OK;
}

int64_t lrbms3_fom_solve_work_size(lrbms3_ctx* ctx) {
  if (!ctx || !ctx->has_mesh) return -1;
  const T3& t = ctx->t;
  const int64_t nblk = (int64_t)t.S * ((t.nT + FOM_EPB - 1) / FOM_EPB), M = std::min<int64_t>(4 * (int64_t)t.S, FOM_MAX_COARSE);
  return (int64_t)t.S * t.nT * 500 + (int64_t)t.S * 6 * t.ncf * 100 + (int64_t)t.S * t.nT * 100 + 4 * (int64_t)t.S * t.n + 3 * nblk + 16 +
         M + 4 * nblk + 2 * M + (int64_t)t.S * 112 + 2 * M * M + 2;       // coarse level: prc, pr0, r0, y0, blocks, A1, A1inv, info
}

int lrbms3_fom_solve(lrbms3_ctx* ctx, int32_t Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                     double* work, double* x, double rtol, int32_t max_iter, double* info, void* stream) {
  REQUIRE3(ctx);
  const T3& t = ctx->t;
  if (t.S_ext != t.S) return fail3(ctx, LRBMS_E_INVALID, "fom_solve: needs all subdomains on this rank");
  if (Q < 1 || Q > 8 || !theta || !A_diag || !A_cpl || !b || !work || !x) return fail3(ctx, LRBMS_E_INVALID, "fom_solve: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long S = t.S, nd = S * t.nT * 500, ncp = S * 6 * t.ncf * 100, total = S * t.n;
  const int nbx = (t.nT + FOM_EPB - 1) / FOM_EPB;
  const long nblk = S * nbx, Mmax = std::min<long>(4 * S, FOM_MAX_COARSE);
  double* Amu = work;
  double* Cmu = Amu + nd;
  double* Dinv = Cmu + ncp;
  double* r = Dinv + S * t.nT * 100;
  double* z = r + total;
  double* p = z + total;
  double* y = p + total;
  double* prz = y + total;
  double* prc = prz + nblk;            // coarse contributions to r.z: summed with prz by ONE reduction
  double* ppy = prc + Mmax;
  double* prr = ppy + nblk;
  double* scal = prr + nblk;
  double* pr0 = scal + 16;
  double* r0 = pr0 + 4 * nblk;
  double* y0 = r0 + Mmax;
  double* A1b = y0 + Mmax;
  double* A1 = A1b + S * 112;
  double* A1inv = A1 + Mmax * Mmax;
  rocblas_int* pinfo = (rocblas_int*)(A1inv + Mmax * Mmax);
  const QV th = make_theta(Q, theta);
  hipLaunchKernelGGL(k3_combine, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, nd, Q, th, A_diag, Amu);
  hipLaunchKernelGGL(k3_combine, dim3((unsigned)((ncp + 255) / 256)), dim3(256), 0, st, ncp, Q, th, A_cpl, Cmu);
  hipLaunchKernelGGL(k3f_block_inverse, dim3((t.nT + 63) / 64, S), dim3(64), 0, st, t, Amu, Dinv);
  // ---- coarse level (LRBMS3_OPT_FOM_COARSE 0 switches it off)
  const bool coarse_env = ctx->opt_fom_coarse != 0;
  int nc = coarse_env ? ctx->fom_nc : 0;
  if (nc > 1 && (long)nc * S > FOM_MAX_COARSE) nc = 1;       // the dense coarse inverse is capped: constants instead of P1, then none
  if ((long)nc * S > FOM_MAX_COARSE) nc = 0;
  const double* Phi = ctx->fom_phi;
  const long M = (long)nc * S;
  const bool kept = nc > 0 && ctx->fom_keep && ctx->fom_pc && ctx->fom_pc_M == M && ctx->fom_pc_nc == nc;
  if (kept) A1inv = ctx->fom_pc;          // built by an earlier solve (another parameter: any SPD preconditioner is admissible)
  if (nc > 0 && !kept) {
    if (!ctx->blas) {
      rocblas_handle h = nullptr;
      if (rocblas_create_handle(&h) != rocblas_status_success) return fail3(ctx, LRBMS_E_HIP, "rocblas_create_handle failed");
      ctx->blas = h;
    }
    rocblas_handle h = (rocblas_handle)ctx->blas;
    if (rocblas_set_stream(h, st) != rocblas_status_success) return fail3(ctx, LRBMS_E_HIP, "rocblas_set_stream failed");
    hipLaunchKernelGGL(k3f_coarse_blocks, dim3(S), dim3(256), 0, st, t, Phi, Amu, Cmu, A1b);
    HIP3(ctx, hipMemsetAsync(A1, 0, sizeof(double) * 2 * M * M, st));             // A1 and the identity behind it (A1inv at M * M when nc = 4)
    double* Id = A1 + M * M;
    hipLaunchKernelGGL(k3f_coarse_dense, dim3((unsigned)((std::max(M, S * 112) + 255) / 256)), dim3(256), 0, st, t, nc, A1b, A1, Id);
    LAUNCH3(ctx);
    if (rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)M, A1, (rocblas_int)M, pinfo) != rocblas_status_success)
      return fail3(ctx, LRBMS_E_HIP, "rocsolver_dpotrf failed");
    rocblas_int hinfo = 0;
    HIP3(ctx, hipMemcpyAsync(&hinfo, pinfo, sizeof(rocblas_int), hipMemcpyDeviceToHost, st));
    HIP3(ctx, hipStreamSynchronize(st));
    if (hinfo != 0) nc = 0;                                   // not positive definite (dependent functions): block-Jacobi alone
    else if (rocsolver_dpotrs(h, rocblas_fill_lower, (rocblas_int)M, (rocblas_int)M, A1, (rocblas_int)M, Id, (rocblas_int)M) !=
             rocblas_status_success)
      return fail3(ctx, LRBMS_E_HIP, "rocsolver_dpotrs failed");
    A1inv = Id;
    if (nc > 0 && ctx->fom_keep) {
      if (ctx->fom_pc && ctx->fom_pc_M != M) {
        HIP3(ctx, hipFree(ctx->fom_pc));
        ctx->fom_pc = nullptr;
      }
      ctx->fom_pc_M = 0;
      if (!ctx->fom_pc) HIP3(ctx, hipMalloc((void**)&ctx->fom_pc, sizeof(double) * M * M));
      HIP3(ctx, hipMemcpyAsync(ctx->fom_pc, Id, sizeof(double) * M * M, hipMemcpyDeviceToDevice, st));
      ctx->fom_pc_M = M;
      ctx->fom_pc_nc = nc;
    }
  }
  const long nrz = nblk + (nc > 0 ? M : 0);
  auto coarse = [&]() {
    if (nc == 0) return;
    hipLaunchKernelGGL(k3f_coarse_r0, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, (int)S, nbx, nc, pr0, r0);
    hipLaunchKernelGGL(k3f_coarse_apply, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (int)M, A1inv, r0, y0, prc);
  };
  const dim3 grid(nbx, S);
  hipLaunchKernelGGL(k3f_update, grid, dim3(256), 0, st, t, 1, 0, scal, Dinv, p, y, b, x, r, z, prz, prr, nc, Phi, pr0);
  coarse();
  hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)nrz, prz, scal + 0);
  hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)nblk, prr, scal + 4);
  LAUNCH3(ctx);
  double bb = 0.0;
  HIP3(ctx, hipMemcpyAsync(&bb, scal + 4, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP3(ctx, hipStreamSynchronize(st));
  if (info) info[0] = 0, info[1] = 0;
  if (bb == 0.0) return LRBMS_OK;
  int it = 0;
  double rel = 1.0;
  const int check = 16;
  while (it < max_iter) {
    for (int k = 0; k < check && it < max_iter; ++k, ++it) {
      const int cur = it & 1;                       // slot of the current r.z (the update of this iteration writes the other one)
      hipLaunchKernelGGL(k3f_dir, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, it == 0 ? 1 : 0, cur, scal, z, p, t.n, nc,
                         Phi, y0);
      hipLaunchKernelGGL(k3f_matvec, dim3(xcd_grid(nbx, (int)S)), dim3(256), 0, st, t, nbx, Amu, Cmu, p, y, ppy);
      hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)nblk, ppy, scal + 2);
      hipLaunchKernelGGL(k3f_update, grid, dim3(256), 0, st, t, 0, cur, scal, Dinv, p, y, b, x, r, z, prz, prr, nc, Phi, pr0);
      coarse();
      hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)nrz, prz, scal + (cur ^ 1));
    }
    hipLaunchKernelGGL(k3_reduce1, dim3(1), dim3(256), 0, st, (int)nblk, prr, scal + 3);
    LAUNCH3(ctx);
    double rr = 0.0;
    HIP3(ctx, hipMemcpyAsync(&rr, scal + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP3(ctx, hipStreamSynchronize(st));
    rel = sqrt(rr / bb);
    if (!(rel == rel)) return fail3(ctx, LRBMS_E_NOT_CONVERGED, "fom_solve: NaN residual");
    if (rel <= rtol) break;
  }
  if (info) info[0] = it, info[1] = rel;
  if (rel > rtol) return fail3(ctx, LRBMS_E_NOT_CONVERGED, "fom_solve: not converged");
  return LRBMS_OK;
}

int lrbms3_fom_apply(lrbms3_ctx* ctx, int32_t Q, int32_t M, const double* theta, const double* A_diag, const double* A_cpl,
                     const double* x, double* y, void* stream) {
  REQUIRE3(ctx);
  if (Q < 1 || Q > 8 || M < 1 || !theta || !A_diag || !A_cpl || !x || !y) return fail3(ctx, LRBMS_E_INVALID, "fom_apply: bad argument");
  const T3& t = ctx->t;
  hipLaunchKernelGGL(k3_fom_apply, dim3(t.nT, t.S), dim3(256), 0, (hipStream_t)stream, t, Q, M, make_theta(Q, theta), A_diag, A_cpl, x, y);
  LAUNCH3(ctx);
  return LRBMS_OK;
}

}  // extern "C"
