// Fused project + estimate-offline pass ("v2"): SURVEY.md section 8a rows K7 + K8 + P1 + P2 in eight launches that
// never materialise the padded image bases Wt / Rt or any "operator x basis" intermediate in HBM.
//
// Why this shape.  Every operator of the path is a sum over fine elements T of a 3x3 (or 3x3-gathered) local
// block, so every projected operator is   G = sum_T  X_T^T  L_T  Y_T   with X_T, Y_T the 3 local rows of a basis-like
// operand.  Two facts make most of the canonical GEMM count of SURVEY 8(d) structurally zero:
//   * the image of a NEIGHBOUR's basis restricted to the target subdomain lives only on the elements touching the
//     shared side (Oswald: elements with a vertex on the side; flux: the 2k side faces), so all blocks of the big
//     Gram matrices that involve a neighbour slot are rank-<=3*ntouch / rank-ncf updates ("thin" kernels, VALU);
//   * blocks between two different neighbour slots are zero (or corner-element-only) and are only zero-filled.
// What remains dense is the self x self part: V^T{A_q V, P V, M V, c K V, A_ab R}, W^T E W, R^T B R, D^T |T| D,
// all with K = n (or n_T) -- these run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), with the "L_T Y_T"
// operand built on the fly from L2-resident rows into LDS by producer waves while consumer waves issue the MFMAs.
//
// Staging is wave-uniform: producer wave w of a workgroup builds the rows of element c0 + w.  What depends only on the
// element (adjacency, areas, rhs) comes through the scalar cache (constant-address-space loads); the element's blocks
// reach the math without any broadcast: the four-block applies are a small MFMA whose operands are loaded in lane
// layout, the few remaining entries sit one per lane and are read with v_readlane.  All vector loads of the producers
// are asm-managed prefetch sets (gload_f64 / s_waitcnt vmcnt(n)), see k_f1.
//
// Launches (grid = one workgroup per subdomain unless noted); "A" = reads only the rank's own basis slabs (phase 1 of a
// sharded pass, overlapped with the halo exchange), "B" = needs the neighbours' rows (phase 2):
//   k_flux_compact   A(+B)  R_self [S][n_rt][QN], R_side [S][4][ncf][QN]                 (HBM-bound, small)
//   k_vertex_avg     A(+B)  Oswald vertex averages Avg_self [S][nv][N], Avg_side [S][4][nvs][N]
//   k_flux_side, k_vertex_side   B   R_side / Avg_side alone (phase 2 of a sharded pass)
//   k_f1<NTX,7,Q>    A   X = V:  B_sys diag, E_red, M_red, G_aa, G_ab[:, self], rhs_red  (MFMA; grid.z = 2: K-split)
//   k_f2<NR>         A   X = R~: G_bb[self,self], G_rdd[self,self], r_fd[self]           (MFMA, symmetric tiles)
//   k_f3<NTX>        A   X = W_self: G_nc[self,self]                                     (MFMA, symmetric tiles)
//   k_thin_nc        B   grid (4 sides, S): block-row `a` and block [self,a] of G_nc     (MFMA + VALU, latency-bound)
//   k_thin_rt        B   grid (4 sides, S): blocks [a,self], [a,a] of G_bb, G_rdd; G_ab[:, a], r_fd[a]  (write-bound)
//   k_coupling       B   grid (4 sides, S): off-diagonal blocks of B_sys                 (MFMA, small)
// Below 192 subdomains per rank the independent kernels are forked over the library's streams.  Every output element is
// written exactly once (zeros included); all reductions have a fixed order (the 2-way K-split meets by atomic add, which
// is order-independent for two contributions).
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "lrbms_dev.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
// One or two adjacent basis columns per lane (W = 2 whenever N is even).  A CU moves ~50 GB/s with 8-byte and ~70 GB/s with 16-byte
// lane loads (MI355X_MICROARCH.md, indexed rows), and the thin kernels are bound by exactly that rate (tools/thin_trace.py).
template <int W> struct VecT { using T = double; };
template <> struct VecT<2> { using T = d2; };
template <int W> __device__ inline typename VecT<W>::T ldv(const double* p) { return *reinterpret_cast<const typename VecT<W>::T*>(p); }
template <int W> __device__ inline void stv(double* p, typename VecT<W>::T v) { *reinterpret_cast<typename VecT<W>::T*>(p) = v; }
template <int W> __device__ inline typename VecT<W>::T zerov() { return typename VecT<W>::T(0.0); }
using W1 = std::integral_constant<int, 1>;
using W2 = std::integral_constant<int, 2>;

// Experiment switches (tools/build_variant.sh builds A/B variants with them) change what the kernels compute or skip.
// A product build must define none of them: a stray -D would silently produce wrong results.
#if !defined(LRBMS_EXPERIMENT_BUILD) &&                                                                               \
    (defined(F1_NO_STAGE) || defined(F1_NO_APPLY) || defined(F1_NO_VALU_STAGE) || defined(F1_NO_MFMA) ||             \
     defined(F1_LDS_FILL) || defined(F1_PRODUCER_PRIO) || defined(F1_SPLIT_SIMD) || defined(F1_PF) ||                \
     defined(F2_NO_STAGE) || defined(F2_NO_MFMA) || defined(F1_TRACE) || defined(F1V_NO_STORE) || defined(F1V_NO_MIRROR) || defined(F3_EW_X) || defined(PREP_TRACE) || defined(THIN_TRACE))
#error "experiment switch defined in a product build of fused.hip (use tools/build_variant.sh, which sets LRBMS_EXPERIMENT_BUILD)"
#endif
#ifndef F1_SPLIT_SIMD
#define F1_SPLIT_SIMD 0
#endif
#ifndef F1_PF
#define F1_PF 1     // prefetch distance of the k_f1 producers in chunks (1 or 2; measured at config 3: 550 us vs 566 us)
#endif
#ifdef F1_TRACE   // experiment build: cycle stamps of producer wave 0 / consumer wave 4 of workgroup 0 (tools/f1_trace.py)
__device__ unsigned long long g_f1_trace[2][64][8];
#define F1_STAMP(role, c, k)                                                                                   \
  do {                                                                                                         \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0 && (c) < 64)        \
      g_f1_trace[role][c][k] = __builtin_amdgcn_s_memtime();                                                  \
  } while (0)
extern "C" int lrbms_debug_f1_trace(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_f1_trace), sizeof(g_f1_trace));
}
#else
#define F1_STAMP(role, c, k) do {} while (0)
#endif
#ifdef THIN_TRACE   // experiment build: cycle stamps of thread 0 of the three workgroups (side 1, subdomain 500) of k_thin3
__device__ unsigned long long g_thin_trace[3][8];
#define THIN_STAMP(z, k)                                                                                       \
  do {                                                                                                         \
    if (side == 1 && s == 500 && threadIdx.x == 0) g_thin_trace[z][k] = __builtin_amdgcn_s_memtime();          \
  } while (0)
extern "C" int lrbms_debug_thin_trace(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_thin_trace), sizeof(g_thin_trace));
}
#else
#define THIN_STAMP(z, k) do {} while (0)
#endif
#ifdef PREP_TRACE   // experiment build: cycle stamps of wave 0 (own rows) and the last wave (neighbours' shares) of workgroup 5 of k_prep_lds
__device__ unsigned long long g_prep_trace[2][16];
#define PREP_STAMP(k)                                                                                          \
  do {                                                                                                         \
    if (blockIdx.x == 5 && prep_trace_on && (threadIdx.x == 0 || threadIdx.x == PREP_LDS_THREADS - 64))        \
      g_prep_trace[threadIdx.x == 0 ? 0 : 1][k] = __builtin_amdgcn_s_memtime();                               \
  } while (0)
extern "C" int lrbms_debug_prep_trace(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prep_trace), sizeof(g_prep_trace));
}
#else
#define PREP_STAMP(k) do {} while (0)
#endif
constexpr int EC = 4;               // elements per K-chunk (12 DG rows = 3 MFMA k-steps) = staging waves
constexpr int F1_NTY = 7;           // Y column tiles per wave in k_f1
constexpr int F1_MAXG = 12;

__host__ __device__ constexpr int padded_ld(int tiles) { return (tiles * 16) % 32 == 16 ? tiles * 16 : tiles * 16 + 16; }

__device__ inline int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Wave-uniform data through the scalar unit: a pointer cast to the constant address space makes every load with a
// uniform address an s_load (scalar cache, SGPR result usable directly as the scalar operand of v_fma_f64).
typedef const __attribute__((address_space(4))) int* cint_p;
typedef const __attribute__((address_space(4))) double* cdbl_p;

// A global load the compiler does not track.  The producers of k_f1 prefetch the rows of the NEXT chunk while they stage
// the current one; that only works if the wait before the staging is `s_waitcnt vmcnt(<loads just issued>)`.  With
// ordinary loads hipcc (ROCm 7.2) emits vmcnt(0) there -- its wait-count bookkeeping gives up across the unrolled
// ping-pong loop -- which also waits for the loads just issued, i.e. exposes one full memory latency per chunk.  So the
// prefetch loads are inline asm (invisible to that pass) and are completed by an explicit wait that names every
// destination register as an in/out operand (so no use can be scheduled above it).
__device__ inline double gload_f64(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline double gload_s64(const double* base, unsigned off) {      // base: wave-uniform (SGPR pair); off: bytes
  double v;
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
  return v;
}
__device__ inline d2 gload_s128(const double* base, unsigned off) {
  d2 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
  return v;
}

// Broadcast lane `l` of a per-lane double to the whole wave as an SGPR pair (two v_readlane_b32, no LDS round trip).
__device__ inline double bcast_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it waits for every
// global load in flight -- which would serialise the register prefetch of the next chunk behind each barrier.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline void stiffness3(const Tmpl& t, int e, double K[9]) {
  for (int i = 0; i < 3; ++i) {
    const double gx = t.grad[(e * 3 + i) * 2], gy = t.grad[(e * 3 + i) * 2 + 1];
    for (int j = 0; j < 3; ++j) {
      const double hx = t.grad[(e * 3 + j) * 2], hy = t.grad[(e * 3 + j) * 2 + 1];
      K[i * 3 + j] = gx * (t.kappa[0] * hx + t.kappa[1] * hy) + gy * (t.kappa[2] * hx + t.kappa[3] * hy);
    }
  }
}

// Oswald data of lattice vertex v of subdomain s: inverse patch size (0 on the physical boundary) and, per side,
// the matching lattice vertex of the neighbour (or -1).
struct OsInfo {
  double inv;
  int vside[4];
  int pos[4];   // position of the vertex along the side
  // LRBMS_OPT_OSWALD_VERTEX_PATCH: at a cross point (a corner vertex whose two side neighbours exist) the matching corner vertex
  // of the DIAGONAL subdomain, the corner id (0 SW, 1 SE, 2 NW, 3 NE) and the two sides that meet there; -1 otherwise
  int vdiag, corner, sda, sdb;
};

template <typename IP>
__device__ inline OsInfo oswald_vertex(const Tmpl& t, const int* nbr_s, int v, IP vdof_ptr) {      // vdof_ptr: t.vdof_ptr or a copy of it (LDS)
  OsInfo o;
  const int lx = v % t.nvx, ly = v / t.nvx;
  o.vside[0] = (ly == 0) ? lx + t.nvx * (t.nvy - 1) : -1;
  o.vside[1] = (lx == 0) ? (t.nvx - 1) + t.nvx * ly : -1;
  o.vside[2] = (lx == t.nvx - 1) ? t.nvx * ly : -1;
  o.vside[3] = (ly == t.nvy - 1) ? lx : -1;
  o.pos[0] = o.pos[3] = lx;
  o.pos[1] = o.pos[2] = ly;
  int cnt = vdof_ptr[v + 1] - vdof_ptr[v];
  bool dirichlet = false;
  for (int sd = 0; sd < 4; ++sd) {
    if (o.vside[sd] < 0) continue;
    if (nbr_s[side_to_slot(sd)] < 0 || t.opt_oswald_subdomain)
      dirichlet = true;
    else
      cnt += vdof_ptr[o.vside[sd] + 1] - vdof_ptr[o.vside[sd]];
  }
  o.vdiag = o.corner = o.sda = o.sdb = -1;
  const bool cx = lx == 0 || lx == t.nvx - 1, cy = ly == 0 || ly == t.nvy - 1;
  if (t.opt_oswald_vertex && cx && cy && !dirichlet) {
    // a corner vertex whose two sides both have a neighbour: on the Cartesian subdomain grid the diagonal subdomain exists, and
    // its elements at the opposite corner belong to the patch
    o.corner = (ly == 0 ? 0 : 2) + (lx == 0 ? 0 : 1);
    o.sda = ly == 0 ? 0 : 3;
    o.sdb = lx == 0 ? 1 : 2;
    o.vdiag = (lx == 0 ? t.nvx - 1 : 0) + t.nvx * (ly == 0 ? t.nvy - 1 : 0);
    cnt += vdof_ptr[o.vdiag + 1] - vdof_ptr[o.vdiag];
  }
  o.inv = dirichlet ? 0.0 : 1.0 / (double)cnt;
  return o;
}
__device__ inline OsInfo oswald_vertex(const Tmpl& t, const int* nbr_s, int v) { return oswald_vertex(t, nbr_s, v, t.vdof_ptr); }

__device__ inline int face_sign_at(const Tmpl& t, const int* nbr_s, int e, int f) {
  const int nb = t.nb_elem[e * 3 + f];
  if (nb < 0 && nbr_s[side_to_slot(-1 - nb)] < 0) return 1;   // domain boundary: outward
  return t.face_sign[e * 3 + f];
}

__device__ inline int nvs_of(const Tmpl& t) { return t.nvx > t.nvy ? t.nvx : t.nvy; }

// ---------------------------------------------------------------------------------------------------------
// compact flux reconstruction.  write_side = 0: only R_self (needs no neighbour data: the halo-independent phase of a
// sharded pass); write_side = 1: R_self and R_side in one sweep.
// (`by` of `gy` workgroups share subdomain `s`: the bodies below are launched on their own for large subdomain counts and
// merged into one launch -- k_prep, k_prep_side, k_thin -- for small ones, where the pass is bound by launches, not work)
template <int NTHR = 256, typename VP = const double*>
__device__ __forceinline__ void flux_compact_body(const Tmpl& t, int S, const int* __restrict__ nbr, int Q, int N,
                                                  const double* __restrict__ F, const double* __restrict__ V,
                                                  double* __restrict__ Rself, double* __restrict__ Rside, int write_side,
                                                  int s, int by, int gy, VP Vown) {
  // Vown: the basis rows of subdomain s itself, [n][N] (V + s n N, or the workgroup's copy of that slab in LDS: k_prep_lds)
  const int QN = Q * N;
  for (int it = by * NTHR + threadIdx.x; it < t.nrt * N; it += gy * NTHR) {   // 32-bit index arithmetic only
    const int r = it / N, j = it - r * N;
    const int e0 = t.rt_e0[r], e1 = t.rt_e1[r], side = t.rt_side[r];
    const int s2 = (side >= 0 && write_side) ? nbr[s * 5 + side_to_slot(side)] : -1;
    double v0[3], v1[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) v0[i] = Vown[(3 * e0 + i) * N + j];
    if (side < 0)
      for (int i = 0; i < 3; ++i) v1[i] = Vown[(3 * e1 + i) * N + j];
    else if (s2 >= 0)
      for (int i = 0; i < 3; ++i) v1[i] = V[((long)s2 * t.n + 3 * e1 + i) * N + j];
    for (int q = 0; q < Q; ++q) {
      const double* f = F + (((long)q * S + s) * t.nrt + r) * 6;
      const double self = f[0] * v0[0] + f[1] * v0[1] + f[2] * v0[2];
      const double other = f[3] * v1[0] + f[4] * v1[1] + f[5] * v1[2];
      Rself[((long)s * t.nrt + r) * QN + q * N + j] = side < 0 ? self + other : self;
      if (side >= 0 && write_side)
        Rside[(((long)s * 4 + side) * t.ncf + t.elem_side_pos[e0 * 3 + t.rt_f0[r]]) * QN + q * N + j] = s2 >= 0 ? other : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void k_flux_compact(Tmpl t, int S, const int* __restrict__ nbr, int Q, int N,
                                                      const double* __restrict__ F, const double* __restrict__ V,
                                                      double* __restrict__ Rself, double* __restrict__ Rside, int write_side) {
  const int s = subdomain_of(t, blockIdx.x);
  flux_compact_body(t, S, nbr, Q, N, F, V, Rself, Rside, write_side, s, blockIdx.y, gridDim.y,
                    V + (long)s * t.n * N);   // grid (S, chunks of n_rt * N)
}

// R_side alone (the halo-dependent phase of a sharded pass): one item per (subdomain, side, side face, column).
__device__ __forceinline__ void flux_side_body(const Tmpl& t, int S, const int* __restrict__ nbr, int Q, int N,
                                               const double* __restrict__ F, const double* __restrict__ V,
                                               double* __restrict__ Rside, int bx, int gx) {
  const long total = (long)(t.sub_list ? t.sub_count : S) * 4 * t.ncf * N;
  const int QN = Q * N;
  for (long idx = (long)bx * 256 + threadIdx.x; idx < total; idx += (long)gx * 256) {
    const int j = (int)(idx % N);
    long rem = idx / N;
    const int pos = (int)(rem % t.ncf);
    rem /= t.ncf;
    const int side = (int)(rem % 4), s = subdomain_of(t, (int)(rem / 4));
    if (pos >= t.side_count[side]) continue;
    const int e0 = t.side_elem[side * t.ncf + pos], e1 = t.side_elem_out[side * t.ncf + pos];
    int f0 = 0;
    for (int f = 0; f < 3; ++f)
      if (t.nb_elem[e0 * 3 + f] == -(1 + side)) f0 = f;
    const int r = t.elem_rt[e0 * 3 + f0];
    const int s2 = nbr[s * 5 + side_to_slot(side)];
    double v1[3] = {0, 0, 0};
    if (s2 >= 0)
      for (int i = 0; i < 3; ++i) v1[i] = V[((long)s2 * t.n + 3 * e1 + i) * N + j];
    for (int q = 0; q < Q; ++q) {
      const double* f = F + (((long)q * S + s) * t.nrt + r) * 6;
      const double other = f[3] * v1[0] + f[4] * v1[1] + f[5] * v1[2];
      Rside[(((long)s * 4 + side) * t.ncf + pos) * QN + q * N + j] = s2 >= 0 ? other : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void k_flux_side(Tmpl t, int S, const int* __restrict__ nbr, int Q, int N,
                                                   const double* __restrict__ F, const double* __restrict__ V,
                                                   double* __restrict__ Rside) {
  flux_side_body(t, S, nbr, Q, N, F, V, Rside, blockIdx.x, gridDim.x);
}

// Oswald vertex averages: Avg_self[s][v][j] = inv(v) sum_{star_s(v)} V_s ;  Avg_side[s][sd][pos][j] = inv(v) sum over
// the star of the matching vertex in the neighbour across side sd (0 if there is none).
template <int NTHR = 256, typename VP = const double*>
__device__ __forceinline__ void vertex_avg_body(const Tmpl& t, int S, const int* __restrict__ nbr, int N,
                                                const double* __restrict__ V, double* __restrict__ AvgSelf,
                                                double* __restrict__ AvgSide, int write_side, int s, int by, int gy, VP Vown) {
  const int nvs = nvs_of(t);
  for (int it = by * NTHR + threadIdx.x; it < t.nv * N; it += gy * NTHR) {
    const int v = it / N, j = it - v * N;
    const OsInfo o = oswald_vertex(t, nbr + s * 5, v);
    // the (at most 8) values at the vertex are loaded together and summed in the same order (an `acc += V[...]` loop
    // waits one memory round trip per value)
    double acc = 0.0;
    const int p0 = t.vdof_ptr[v], p1 = t.vdof_ptr[v + 1];
    for (int pb = p0; pb < p1; pb += 8) {
      double val[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) val[k] = pb + k < p1 ? Vown[t.vdof_idx[pb + k] * N + j] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (pb + k < p1) acc += val[k];
    }
    AvgSelf[((long)s * t.nv + v) * N + j] = o.inv * acc;
    if (!write_side) continue;                     // Avg_self needs no neighbour data (only the patch sizes)
    for (int sd = 0; sd < 4; ++sd) {
      if (o.vside[sd] < 0) continue;
      const int s2 = nbr[s * 5 + side_to_slot(sd)];
      double a2 = 0.0;
      if (s2 >= 0 && o.inv != 0.0) {
        const int v2 = o.vside[sd];
        const int q0 = t.vdof_ptr[v2], q1 = t.vdof_ptr[v2 + 1];
        for (int pb = q0; pb < q1; pb += 8) {
          double val[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) val[k] = pb + k < q1 ? V[((long)s2 * t.n + t.vdof_idx[pb + k]) * N + j] : 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (pb + k < q1) a2 += val[k];
        }
        a2 *= o.inv;
      }
      AvgSide[(((long)s * 4 + sd) * nvs + o.pos[sd]) * N + j] = a2;
    }
    if (t.opt_oswald_vertex) {
      // the diagonal subdomain's share at the four corner vertices: Avg_corner [S][4][N], behind Avg_side in the work buffer
      const int lx = v % t.nvx, ly = v / t.nvx;
      if ((lx == 0 || lx == t.nvx - 1) && (ly == 0 || ly == t.nvy - 1)) {
        double a3 = 0.0;
        if (o.vdiag >= 0) {
          const int sdg = t.nbr_diag[s * 4 + o.corner];                          // exists (checked by the launcher for sharded grids)
          const int q0 = t.vdof_ptr[o.vdiag], q1 = t.vdof_ptr[o.vdiag + 1];
          for (int pb = q0; pb < q1; ++pb) a3 += V[((long)sdg * t.n + t.vdof_idx[pb]) * N + j];
          a3 *= o.inv;
        }
        const int corner = (ly == 0 ? 0 : 2) + (lx == 0 ? 0 : 1);
        (AvgSide + (long)S * 4 * nvs * N)[((long)s * 4 + corner) * N + j] = a3;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_vertex_avg(Tmpl t, int S, const int* __restrict__ nbr, int N,
                                                    const double* __restrict__ V, double* __restrict__ AvgSelf,
                                                    double* __restrict__ AvgSide, int write_side) {
  const int s = subdomain_of(t, blockIdx.x);
  vertex_avg_body(t, S, nbr, N, V, AvgSelf, AvgSide, write_side, s, blockIdx.y, gridDim.y,
                  V + (long)s * t.n * N);   // grid (S, chunks of n_v * N)
}

// Both preparation sweeps from ONE copy of the subdomain's basis slab in LDS (round 3).  The streaming sweeps above are bound by
// memory latency, not bandwidth: every item waits for a chain of dependent loads (row tables -> coefficients / basis rows), and
// every basis row is fetched ~3 times by the flux sweep and ~twice by the vertex averages through L1 / L2.  Here one workgroup
// per subdomain issues ALL its global loads at once -- the n x N slab (fully coalesced 16-byte loads), the subdomain's flux
// coefficients, the row and vertex tables -- parks them in LDS, and the sweeps then read LDS only and write their results with
// 16-byte stores (two basis columns per item).  With write_side the last eight waves compute the neighbours' shares (R_side,
// Avg_side, Avg_corner: the only items that still wait for global loads) while the other eight do the own rows.
// With the fold (GncArgs) the same workgroup goes on to G_nc[self, self] -- the whole of k_f3, which would read the slab and the
// averages again -- from the slab and the averages it has in LDS.  164 us (two sweeps) -> 82 us; + G_nc: 120 us against 82 + 57.
// Where a workgroup's time goes at config 3 (tools/prep_trace.py, cycles): loads 22 k (all 256 workgroups of a round fetch their
// 150 KB at the same time: HBM-bound), flux rows 11 k, averages 7 k, Z rows 5 k, MFMA 9 k, reduction 3 k.
// (Measured and dropped: PERSISTENT workgroups, one per CU, that request the next subdomain's slab into registers in front of the
// MFMA phase -- the trace shows the load phase shrink from 22 k to 10 k cycles, but with 1 024 threads the kernel has 128 VGPRs:
// half a slab in registers is all that fits without spilling the prefetched values -- a spill waits for the load -- and the
// kernel went from 120 to 117 us; the whole slab in registers, spilled: 169 us.)
// Needs even N and prep_lds_bytes() of LDS (156 KB at config 3 with the fold); the launcher falls back to the sweeps above otherwise.
constexpr int PREP_LDS_THREADS = 1024, PREP_SIDE_THREADS = 512;
// LDS layout (bytes): Vl [n][N] (at least PREP_GNC_PART with the G_nc fold) | vinfo [nv] | Lg [nT][6] (fold) | vptr [nv + 1], vidx [n],
// dvt [n] (ints, padded to 16) | region B = max(Fl [Q][nrt][6] + rinfo [nrt] int4 + srow [4 ncf],  Al [nv][N] with the fold: it
// overlays Fl / rinfo / srow once the flux rows are done)
constexpr size_t PREP_GNC_PART = sizeof(double) * 16 * 3 * 256;      // partial tiles of the G_nc fold: 16 waves x 3 tiles
__host__ __device__ inline size_t prep_lds_tab_bytes(const Tmpl& t, bool gnc) {      // vinfo [nv] | Lg [nT][6] (fold only) | the ints
  return 16 * (size_t)t.nv + (gnc ? sizeof(double) * 6 * (size_t)t.nT : 0) + ((sizeof(int) * ((size_t)t.nv + 1 + 2 * t.n) + 15) & ~(size_t)15);
}
__host__ __device__ inline size_t prep_lds_slab_bytes(const Tmpl& t, int N, bool gnc) {
  const size_t slab = sizeof(double) * (size_t)t.n * N;
  return gnc && slab < PREP_GNC_PART ? PREP_GNC_PART : slab;
}
static size_t prep_lds_bytes(const Tmpl& t, int Q, int N, bool gnc) {
  const size_t flux = sizeof(double) * (size_t)Q * t.nrt * 6 + sizeof(int) * (4 * (size_t)t.nrt + 4 * t.ncf);
  const size_t avg = gnc ? sizeof(double) * (size_t)t.nv * N : 0;
  return prep_lds_slab_bytes(t, N, gnc) + prep_lds_tab_bytes(t, gnc) + (flux > avg ? flux : avg);
}
// persistent form (a workgroup takes several subdomains): the row tables rinfo / srow stay valid across them, BEHIND region B =
// max(Fl, Al) instead of inside it
__host__ __device__ inline size_t prep_lds_region_b_bytes(const Tmpl& t, int Q, int N, bool gnc) {
  const size_t fl = sizeof(double) * (size_t)Q * t.nrt * 6, avg = gnc ? sizeof(double) * (size_t)t.nv * N : 0;
  return ((fl > avg ? fl : avg) + 15) & ~(size_t)15;
}
static size_t prep_lds_bytes_persistent(const Tmpl& t, int Q, int N, bool gnc) {
  return prep_lds_slab_bytes(t, N, gnc) + prep_lds_tab_bytes(t, gnc) + prep_lds_region_b_bytes(t, Q, N, gnc) +
         sizeof(int) * (4 * (size_t)t.nrt + 4 * t.ncf);
}

// G_nc[self, self] folded into k_prep_lds (below): the tiles a wave owns.  Even waves: the first row of tiles, odd waves: the rest.
template <int NTX>
__device__ inline void gnc_tile(int half, int k, int& ti, int& tj) {      // k-th tile (row, column) of a wave of that parity
  if (NTX == 1) {
    ti = tj = 0;
  } else if (NTX == 2) {
    ti = half ? 1 : 0;
    tj = half ? 1 : k;
  } else {
    ti = half ? (k < 2 ? 1 : 2) : 0;
    tj = half ? (k == 0 ? 1 : 2) : k;
  }
}
template <int NTX>
__host__ __device__ constexpr int gnc_count(int half) { return NTX == 1 ? (half ? 0 : 1) : NTX == 2 ? (half ? 1 : 2) : 3; }

struct GncArgs {           // G_nc != nullptr: the fold runs (the launcher then skips k_f3)
  const double* ebar;
  double* G_nc;
  long gsub;
  int gld, goff;
};

// NTHR: 1 024 threads (everything), or 256 for the slab-less launch of the halo-dependent phase (write_side == 2 only: small workgroups,
// several per CU, beside the dense kernels)
template <int NTX, int NTHR = PREP_LDS_THREADS>
__global__ __launch_bounds__(NTHR) void k_prep_lds(Tmpl t, int S, const int* __restrict__ nbr, int Q, int N,
                                                               const double* __restrict__ F, const double* __restrict__ V,
                                                               double* __restrict__ Rself, double* __restrict__ Rside,
                                                               double* __restrict__ AvgSelf, double* __restrict__ AvgSide,
                                                               int write_side, GncArgs ga, int persistent) {
  extern __shared__ double Vl[];
  const int tid = threadIdx.x, N2 = N / 2, QN = Q * N, nvs = nvs_of(t);
  // persistent != 0 (the launcher: more subdomains than CUs, the slab in one round of loads, room for the row tables behind region B):
  // gridDim.x < count workgroups, workgroup b takes the subdomains b, b + gridDim.x, ... of the list, and while it works on one it
  // holds the NEXT one's slab and flux coefficients in registers (requested right behind the barrier that opens the LDS copy to
  // readers): the 22 k cycles in which all workgroups of a round used to wait for their 150 KB with nothing to overlap (of 60 k per
  // subdomain) are hidden behind the 38 k of LDS work.  The template's tables stay in LDS across the subdomains of a workgroup.
  const int count = t.sub_list ? t.sub_count : S, G = gridDim.x;
  int idx = blockIdx.x;
#ifdef PREP_TRACE
  bool prep_trace_on = true;                     // (persistent form: the stamps of the workgroup's THIRD subdomain, steady state)
#endif
  int s = subdomain_of(t, idx);
  // gridDim.y == 2 (ranks with at most half as many subdomains as the chip has CUs): the work of a subdomain is dealt to TWO workgroups
  // that each load the slab -- part 0 the flux image, part 1 the vertex averages and G_nc -- instead of leaving half of the CUs idle
  const bool do_flux = gridDim.y == 1 || blockIdx.y == 0, do_avg = gridDim.y == 1 || blockIdx.y == 1;      // workgroup-uniform
  // write_side: 0 the own rows only (halo-independent phase of a sharded pass), 1 own rows and the neighbours' shares, 2 the
  // neighbours' shares only (the halo-dependent phase: no slab, every thread takes side items -- the same code, hence the same bits,
  // as the whole pass)
  const bool side_only = write_side == 2;
  const bool gnc = NTHR == 1024 && ga.G_nc != nullptr && do_avg && !side_only;      // (the fold deals its tiles to 16 waves)
  PREP_STAMP(0);
  struct VInfo { double inv; int p0, cnt; };     // per lattice vertex: inverse patch size (0 on a Dirichlet vertex), its DoF list
  VInfo* vinfo = reinterpret_cast<VInfo*>(reinterpret_cast<char*>(Vl) + (write_side == 2 ? 0 : prep_lds_slab_bytes(t, N, gnc)));      // (side only: no slab)
  double* Lg = reinterpret_cast<double*>(vinfo + t.nv);                                                       // [nT][2][3]: L^T G_T (fold only)
  int* vptr = reinterpret_cast<int*>(Lg + (gnc ? 6 * t.nT : 0));
  int* vidx = vptr + t.nv + 1;
  int* dvt = vidx + t.n;
  double* Fl = reinterpret_cast<double*>(reinterpret_cast<char*>(vinfo) + prep_lds_tab_bytes(t, gnc));
  double* Al = Fl;                               // with the fold: the vertex averages, once the flux rows are done
  // the row tables: behind the coefficients (the averages overwrite them, a workgroup with one subdomain) or behind region B (persistent)
  int4* rinfo = reinterpret_cast<int4*>(persistent ? reinterpret_cast<char*>(Fl) + prep_lds_region_b_bytes(t, Q, N, gnc)
                                                   : reinterpret_cast<char*>(Fl + Q * t.nrt * 6));
  int* srow = reinterpret_cast<int*>(rinfo + t.nrt);
  // ---- every global load of the own-rows work, issued together: the row tables first (a chain of two dependent loads whose
  // second step is requested while the slab is in flight), then the slab, the coefficients and the vertex tables
  int nb0 = nbr[s * 5], nb1 = nbr[s * 5 + 1], nb3 = nbr[s * 5 + 3], nb4 = nbr[s * 5 + 4];      // wave-uniform: scalar loads
  auto nbr_slot = [&](int slot) { return slot == 0 ? nb0 : slot == 1 ? nb1 : slot == 3 ? nb3 : nb4; };
  const int sc0 = t.side_count[0], sc1 = t.side_count[1], sc2 = t.side_count[2], sc3 = t.side_count[3];
  constexpr int U = 8, FU = 2;                   // 16-byte pieces per thread of the slab / of the flux coefficients (persistent form)
  static_assert(NTHR != 1024 || NTHR * U * 2 >= 384 * 40, "config 3: the slab in one round of loads");
  const int total2 = t.n * N2;
  const int f2 = t.nrt * 3;                      // 16-byte pieces of one component's coefficient rows
  {
    const d2* src = reinterpret_cast<const d2*>(V + (long)s * t.n * N);
    d2* dst = reinterpret_cast<d2*>(Vl);
    int e0 = 0, e1 = 0, side = -1, f0 = 0;
    if (do_flux && tid < t.nrt) e0 = t.rt_e0[tid], e1 = t.rt_e1[tid], side = t.rt_side[tid], f0 = t.rt_f0[tid];
    for (int base = 0; base < (side_only ? 1 : total2); base += U * NTHR) {
      d2 tmp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * NTHR + tid;
        tmp[u] = side_only ? (d2){0.0, 0.0} : src[i < total2 ? i : total2 - 1];
      }
      if (base == 0) {
        for (int i = tid; i <= t.nv; i += NTHR) vptr[i] = t.vdof_ptr[i];
        for (int i = tid; i < t.n; i += NTHR) vidx[i] = t.vdof_idx[i];
        for (int v = tid; v < t.nv; v += NTHR) {          // (integer divisions, a chain of table reads and an fp64 division:
          const OsInfo o = oswald_vertex(t, nbr + s * 5, v);          //  once per vertex, not once per item)
          const int p0 = t.vdof_ptr[v];
          vinfo[v] = VInfo{o.inv, p0, t.vdof_ptr[v + 1] - p0};
        }
        if (gnc) {
          for (int i = tid; i < t.n; i += NTHR) dvt[i] = t.dof_vertex[i];
          const double ksym = 0.5 * (t.kappa[1] + t.kappa[2]);
          const double l00 = sqrt(t.kappa[0]), l10 = ksym / l00, l11 = sqrt(t.kappa[3] - l10 * l10);
          for (int i = tid; i < 3 * t.nT; i += NTHR) {      // (element, shape function): rows of L^T applied to its gradient
            const int T = i / 3, k = i - 3 * T;
            const double gx = t.grad[i * 2], gy = t.grad[i * 2 + 1];
            Lg[T * 6 + k] = l00 * gx + l10 * gy;
            Lg[T * 6 + 3 + k] = l11 * gy;
          }
        }
        for (int i = tid; do_flux && i < Q * f2; i += NTHR) {
          const int q = i / f2, k = i - q * f2;
          reinterpret_cast<d2*>(Fl)[i] = reinterpret_cast<const d2*>(F + ((long)q * S + s) * t.nrt * 6)[k];
        }
        for (int r = tid; do_flux && r < t.nrt; r += NTHR) {
          if (r != tid) e0 = t.rt_e0[r], e1 = t.rt_e1[r], side = t.rt_side[r], f0 = t.rt_f0[r];      // (templates with n_rt > 1024)
          const int pos = side >= 0 ? t.elem_side_pos[e0 * 3 + f0] : 0;
          rinfo[r] = make_int4(e0, e1, side, pos);
          if (side >= 0) srow[side * t.ncf + pos] = r;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * NTHR + tid;
        if (i < total2 && !side_only) dst[i] = tmp[u];
      }
    }
  }
  PREP_STAMP(1);
  const d2* Vl2 = reinterpret_cast<const d2*>(Vl);
  // (half of the threads for the neighbours' shares: with 256 / 128 of the 1 024 the kernel takes 121 / 136 us instead of 111 --
  // their items are chains of dependent loads, the longer chain of the workgroup)
  const int nmain = side_only ? 0 : (write_side ? NTHR - NTHR / 2 : NTHR);
  const int ts = tid - nmain, nst = NTHR - nmain;      // side threads: index, count
  // nmain is a multiple of the wave size: a wave is either one of the own rows (MAIN) or one of the neighbours' shares, and each kind
  // runs its OWN copy of the loop over the workgroup's subdomains (same barriers, in the same order).  With both kinds in one body --
  // under an exec mask or behind two scalar branches -- hipcc's wait-count pass carries the loads still pending at the exits of the
  // neighbours' code into the own rows' code, finds their registers reused there and drains vmcnt(0) in front of the flux loop: the
  // prefetch would be waited for as soon as it is requested.
  const bool mainw = uniform(tid >> 6) * 64 < nmain;
  const d2* V2 = reinterpret_cast<const d2*>(V);
  d2 pre[U], fpre[FU];                                 // persistent form: the next subdomain's slab and coefficients (written by
                                                       // request_next before they are read; no initialisation: it would keep 40
                                                       // registers alive through the first load phase)
  auto request_next = [&](int sn) {                    // clamped addresses, every load unconditional
    // (the thread index through an opaque move: hipcc otherwise hoists the sixteen clamped offsets of these loads and of the LDS
    // stores at the end of the loop out of the loop, keeps them alive through every phase and spills -- and a spill reload in the
    // middle of a phase is a vector-memory load whose wait also waits for the prefetch)
    // asm loads (gload_s128: wave-uniform base + 32-bit byte offset), completed by the asm vmcnt(0) in front of the LDS stores at the
    // end of the loop: loads hipcc can see make its wait-count pass drain vmcnt(0) in the preheader of the flux and average loops
    // (loops with stores and no loads: SIInsertWaitcnts flushes in front of them whenever it believes a pending register is used
    // inside), i.e. right behind the request.  pylrbms_amd/_isa_check.py walks the emitted code: no compiler instruction may touch
    // these registers while the loads are in flight, and the kernel may not spill.
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const double* src = V + (long)sn * t.n * N;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = u * NTHR + tl;
      pre[u] = gload_s128(src, 16u * (unsigned)(i < total2 ? i : total2 - 1));
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      int i = u * NTHR + tl;
      i = i < Q * f2 ? i : Q * f2 - 1;
      const int q = i / f2, k = i - q * f2;
      fpre[u] = gload_s128(F, (unsigned)(((long)q * S + sn) * t.nrt * 48 + 16 * k));      // (the launcher: Q S n_rt 48 < 2^31)
    }
  };
  auto subdomains = [&](auto role) {
  constexpr bool MAIN = decltype(role)::value;
  // (a wait hipcc can see: whatever its wait-count pass still carries as pending from the load phase above would otherwise reach the
  // preheaders of the flux and average loops -- loops with stores and no loads -- and be flushed THERE with vmcnt(0), behind the request)
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), gfx9 encoding
  for (;;) {
#ifdef PREP_TRACE
  prep_trace_on = !persistent || idx == (int)blockIdx.x + 2 * G;
#endif
  PREP_STAMP(13);
  __syncthreads();
  PREP_STAMP(2);
  const int idx_n = idx + G;
  const bool more = persistent && idx_n < count;       // workgroup-uniform
  const int s_n = more ? subdomain_of(t, idx_n) : s;
  // Where the request goes (each placement measured, k_prep_lds at config 3: 127 us with one subdomain per workgroup; 111 us with the
  // request here, right behind the barrier; 108.5 us as below): hipcc protects the operand registers of the stores still in flight with
  // vmcnt waits of its own (gfx950 has no separate store counter) and flushes vmcnt(0) in the preheader of every loop that overwrites
  // such a register -- the flux loop, the average loop, the Z rows, the MFMA loop -- and any of those waits also waits for the
  // prefetch.  So the waves of the own rows request INSIDE the first round of the flux loop (behind its flush; the next flush, in
  // front of the averages, comes 11 k cycles later), the waves of the neighbours' shares -- whose items wait for global loads of
  // their own all along -- once those are done.
  if (!do_flux) {
  } else if (MAIN) {
    // ---- R_self: one item per (RT0 row, pair of columns)
    d2* R2 = reinterpret_cast<d2*>(Rself + (long)s * t.nrt * QN);
    for (int it = tid;; it += nmain) {           // (ONE request site, reached by every thread -- also one without a flux item:
      if (more && it == tid) request_next(s_n);  //  with several sites hipcc may give the set different registers and move them)
      if (it >= t.nrt * N2) break;
      const int r = it / N2, j2 = it - r * N2;
      const int4 ri = rinfo[r];
      d2 v0[3], v1[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        v0[i] = Vl2[(3 * ri.x + i) * N2 + j2];
        v1[i] = ri.z < 0 ? Vl2[(3 * ri.y + i) * N2 + j2] : (d2){0.0, 0.0};
      }
      for (int q = 0; q < Q; ++q) {
        const double* f = Fl + (q * t.nrt + r) * 6;
        d2 o;
        {
          const double self = f[0] * v0[0].x + f[1] * v0[1].x + f[2] * v0[2].x;
          const double other = f[3] * v1[0].x + f[4] * v1[1].x + f[5] * v1[2].x;
          o.x = ri.z < 0 ? self + other : self;
        }
        {
          const double self = f[0] * v0[0].y + f[1] * v0[1].y + f[2] * v0[2].y;
          const double other = f[3] * v1[0].y + f[4] * v1[1].y + f[5] * v1[2].y;
          o.y = ri.z < 0 ? self + other : self;
        }
        R2[(r * QN + q * N) / 2 + j2] = o;
      }
    }
  } else {
    // ---- the neighbours' share of the flux image (write_side only): the last NTHR / 2 threads.  Every item of theirs waits
    // for global loads (the neighbours' rows)
    // (two items per round, the neighbour's rows of both requested before the first is used: these waves are the critical path of
    // the workgroup -- every item of theirs is a chain LDS tables -> global rows -- and a round trip costs the same for six loads
    // as for three; every load unconditional, from a clamped address)
    d2* Rs2 = reinterpret_cast<d2*>(Rside + (long)s * 4 * t.ncf * QN);
    const int nitems = 4 * t.ncf * N2;
    auto flux_item = [&](int it, int& sp, int& j2, int& r, int& s2) -> bool {      // false: a side with fewer than ncf faces
      sp = it / N2, j2 = it - sp * N2;
      const int sd = sp / t.ncf;
      const bool ok = sp - sd * t.ncf < (sd == 0 ? sc0 : sd == 1 ? sc1 : sd == 2 ? sc2 : sc3);
      r = ok ? srow[sp] : 0;
      s2 = nbr_slot(side_to_slot(sd));
      return ok;
    };
    for (int it0 = ts; it0 < nitems; it0 += 2 * nst) {
      d2 v1[2][3];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int sp, j2, r, s2;
        const bool use = flux_item(it0 + b * nst < nitems ? it0 + b * nst : it0, sp, j2, r, s2) && it0 + b * nst < nitems && s2 >= 0;
        const long row0 = (long)(use ? s2 : s) * t.n + 3 * (use ? rinfo[r].y : 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) v1[b][i] = V2[(row0 + i) * N2 + j2];
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (it0 + b * nst >= nitems) continue;
        int sp, j2, r, s2;
        if (!flux_item(it0 + b * nst, sp, j2, r, s2)) continue;
        for (int q = 0; q < Q; ++q) {
          const double* f = Fl + (q * t.nrt + r) * 6;
          const double ox = f[3] * v1[b][0].x + f[4] * v1[b][1].x + f[5] * v1[b][2].x;
          const double oy = f[3] * v1[b][0].y + f[4] * v1[b][1].y + f[5] * v1[b][2].y;
          Rs2[(sp * QN + q * N) / 2 + j2] = s2 >= 0 ? (d2){ox, oy} : (d2){0.0, 0.0};
        }
      }
    }
  }
  PREP_STAMP(3);
  // (the barriers of the fold order LDS traffic only: __syncthreads() would also wait for the global stores of the rows just written)
  if (do_avg) {
  if (gnc) lds_barrier();                        // Fl (and, in a workgroup with one subdomain, rinfo, srow) are dead: the averages may overwrite them
  PREP_STAMP(4);
  if (MAIN) {
    // ---- Avg_self: one item per (lattice vertex, pair of columns); the values at a vertex summed in the order of its DoF list
    d2* A2 = reinterpret_cast<d2*>(AvgSelf + (long)s * t.nv * N);
    for (int it = tid; it < t.nv * N2; it += nmain) {
      const int v = it / N2, j2 = it - v * N2;
      const VInfo o = vinfo[v];
      d2 acc = {0.0, 0.0};
      const int p0 = o.p0, p1 = o.p0 + o.cnt;
      for (int pb = p0; pb < p1; pb += 8) {
        d2 val[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) val[k] = Vl2[vidx[pb + k < p1 ? pb + k : p1 - 1] * N2 + j2];      // unconditional reads (clamped index):
#pragma unroll                                                                                    // no branch between the LDS round trips
        for (int k = 0; k < 8; ++k) {
          acc.x += pb + k < p1 ? val[k].x : 0.0;
          acc.y += pb + k < p1 ? val[k].y : 0.0;
        }
      }
      const d2 av = {o.inv * acc.x, o.inv * acc.y};
      A2[v * N2 + j2] = av;
      if (gnc) reinterpret_cast<d2*>(Al)[v * N2 + j2] = av;
    }
  } else {
    // ---- the neighbours' shares of the vertex averages, beside the own ones
    // (two items per round as for the flux rows: the first four DoFs of both items' vertices requested together -- a side vertex of
    // the neighbour has three; the sum runs in the order of the DoF list, item by item: the same additions as one item at a time)
    d2* As2 = reinterpret_cast<d2*>(AvgSide + (long)s * 4 * nvs * N);
    const int nitems = 4 * nvs * N2;
    // (everything about an item but the loaded rows is recomputed from its index in the second half: integers and LDS reads are
    // cheap, registers are not -- 128 per wave, and a spill reload would be a vector-memory load in front of the prefetch)
    auto avg_item = [&](int it, int& sp, int& j2, int& q0, int& q1, int& s2, double& inv) -> bool {      // false: no such side vertex
      sp = it / N2, j2 = it - sp * N2;
      const int sd = sp / nvs, pos = sp - sd * nvs;
      const bool ok = pos < ((sd == 0 || sd == 3) ? t.nvx : t.nvy);
      const int posc = ok ? pos : 0;
      const int v = sd == 0 ? posc : sd == 1 ? posc * t.nvx : sd == 2 ? posc * t.nvx + t.nvx - 1 : (t.nvy - 1) * t.nvx + posc;
      inv = vinfo[v].inv;
      s2 = nbr_slot(side_to_slot(sd));
      // the matching lattice vertex of the neighbour across side sd (oswald_vertex: vside)
      const int v2 = sd == 0 ? posc + t.nvx * (t.nvy - 1) : sd == 1 ? (t.nvx - 1) + t.nvx * posc : sd == 2 ? t.nvx * posc : posc;
      q0 = vinfo[v2].p0, q1 = q0 + vinfo[v2].cnt;
      return ok;
    };
    for (int it0 = ts; it0 < nitems; it0 += 2 * nst) {
      d2 val[2][4];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int sp, j2, q0, q1, s2;
        double inv;
        const bool ok = avg_item(it0 + b * nst < nitems ? it0 + b * nst : it0, sp, j2, q0, q1, s2, inv) && it0 + b * nst < nitems;
        const long slab = (long)(ok && s2 >= 0 && inv != 0.0 ? s2 : s) * t.n;
#pragma unroll
        for (int k = 0; k < 4; ++k) val[b][k] = V2[(slab + vidx[q0 + k < q1 ? q0 + k : q1 - 1]) * N2 + j2];
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (it0 + b * nst >= nitems) continue;
        int sp, j2, q0, q1, s2;
        double inv;
        if (!avg_item(it0 + b * nst, sp, j2, q0, q1, s2, inv)) continue;
        d2 a2 = {0.0, 0.0};
        if (s2 >= 0 && inv != 0.0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            a2.x += q0 + k < q1 ? val[b][k].x : 0.0;
            a2.y += q0 + k < q1 ? val[b][k].y : 0.0;
          }
          for (int pb = q0 + 4; pb < q1; pb += 4) {      // (a vertex with more than four DoFs in its patch)
            d2 more4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) more4[k] = V2[((long)s2 * t.n + vidx[pb + k < q1 ? pb + k : q1 - 1]) * N2 + j2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              a2.x += pb + k < q1 ? more4[k].x : 0.0;
              a2.y += pb + k < q1 ? more4[k].y : 0.0;
            }
          }
          a2.x *= inv;
          a2.y *= inv;
        }
        As2[sp * N2 + j2] = a2;
      }
    }
    if (t.opt_oswald_vertex) {                   // the diagonal subdomains' shares at the four corners: Avg_corner [S][4][N]
      d2* Ac2 = reinterpret_cast<d2*>(AvgSide + (long)S * 4 * nvs * N + (long)s * 4 * N);
      for (int it = ts; it < 4 * N2; it += nst) {
        const int corner = it / N2, j2 = it - corner * N2;
        const int v = ((corner & 1) ? t.nvx - 1 : 0) + t.nvx * ((corner & 2) ? t.nvy - 1 : 0);
        const OsInfo o = oswald_vertex(t, nbr + s * 5, v);
        d2 a3 = {0.0, 0.0};
        if (o.vdiag >= 0) {
          const int sdg = t.nbr_diag[s * 4 + corner];
          for (int pb = vptr[o.vdiag]; pb < vptr[o.vdiag + 1]; ++pb) {
            const d2 x = V2[((long)sdg * t.n + vidx[pb]) * N2 + j2];
            a3.x += x.x;
            a3.y += x.y;
          }
          a3.x *= o.inv;
          a3.y *= o.inv;
        }
        Ac2[corner * N2 + j2] = a3;
      }
    }
    }
  PREP_STAMP(5);
  if (gnc) {
  // ---- G_nc[self, self] = W^T E W, W = V - P Avg (the Oswald interpolation error of the own basis), E_T = ebar_T K_T.
  // K_T = G_T^T kappa G_T has rank 2 (G_T: the gradients of the three P1 shape functions): with kappa = L L^T,
  //   G_nc = sum_T ebar_T Z_T^T Z_T,   Z_T = L^T G_T W_T   (2 rows per element instead of 3),
  // and Z_T overwrites the first two of the element's three rows of the slab IN PLACE, so the product needs no staging: both MFMA
  // operands of a k-step are plain LDS rows (the B operand scaled by ebar_T).
  lds_barrier();                                 // Al complete
  PREP_STAMP(6);
  {
    d2* Vw = reinterpret_cast<d2*>(Vl);
    const d2* Al2 = reinterpret_cast<const d2*>(Al);
    for (int it = tid; it < t.nT * N2; it += NTHR) {
      const int T = it / N2, j2 = it - T * N2;
      d2 w[3];
      double g0[3], g1[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const d2 v = Vw[(3 * T + i) * N2 + j2], av = Al2[dvt[3 * T + i] * N2 + j2];
        w[i] = (d2){v.x - av.x, v.y - av.y};
        g0[i] = Lg[T * 6 + i];
        g1[i] = Lg[T * 6 + 3 + i];
      }
      Vw[(3 * T) * N2 + j2] = (d2){g0[0] * w[0].x + g0[1] * w[1].x + g0[2] * w[2].x, g0[0] * w[0].y + g0[1] * w[1].y + g0[2] * w[2].y};
      Vw[(3 * T + 1) * N2 + j2] = (d2){g1[0] * w[0].x + g1[1] * w[1].x + g1[2] * w[2].x, g1[0] * w[0].y + g1[1] * w[1].y + g1[2] * w[2].y};
      if (j2 == 0) Vl[(3 * T + 2) * N] = ga.ebar[(long)s * t.nT + T];      // the element's third row is free now: its first entry carries ebar_T
    }
  }
  PREP_STAMP(7);
  lds_barrier();  
  PREP_STAMP(8);
  {
    const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wave = uniform(tid >> 6), half = wave & 1, kp = wave >> 1;
    const int nsteps = t.nT / 2, per = (nsteps + 7) / 8;
    const int st0 = kp * per, st1 = st0 + per < nsteps ? st0 + per : nsteps;
    d4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ntile = half ? gnc_count<NTX>(1) : gnc_count<NTX>(0);
    for (int st = st0; st < st1; ++st) {
      const int T = 2 * st + (lk >> 1), row = 3 * T + (lk & 1);
      const double eb = Vl[(3 * T + 2) * N];
      double x[NTX];
#pragma unroll
      for (int c = 0; c < NTX; ++c) x[c] = Vl[row * N + 16 * c + li];      // columns >= N: finite garbage in entries that are not stored
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < gnc_count<NTX>(0)) {
          int ti, tj;
          gnc_tile<NTX>(half, k, ti, tj);
          double av = x[0], bv = x[0];
#pragma unroll
          for (int c = 1; c < NTX; ++c) {
            av = ti == c ? x[c] : av;
            bv = tj == c ? x[c] : bv;
          }
          if (k < ntile) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, eb * bv, acc[k], 0, 0, 0);
        }
      // the waves of the neighbours' shares: behind the LAST vmcnt flush of the subdomain (hipcc puts one in front of this loop for
      // the stores of the averages; requested before it, the prefetch stalled these waves 5.5 k cycles in front of their MFMAs)
      if (more && !MAIN && st == st0) request_next(s_n);
    }
    PREP_STAMP(9);
    lds_barrier();                               // every wave is done with the Z rows: the slab region takes the partial tiles
    PREP_STAMP(10);
    double* part = Vl + (long)wave * 3 * 256;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(k * 4 + r) * 64 + lane] = acc[k][r];
    lds_barrier();  
    PREP_STAMP(11);
    // fixed-order sum over the eight K parts; one item per (tile, accumulator register, lane)
    double* g = ga.G_nc + (long)s * ga.gsub + ga.goff;
    constexpr int NTRI = NTX * (NTX + 1) / 2;
    for (int it = tid; it < NTRI * 256; it += NTHR) {
      const int tile = it >> 8, r = (it >> 6) & 3, ln = it & 63;
      const int hf = tile < gnc_count<NTX>(0) ? 0 : 1, k = hf ? tile - gnc_count<NTX>(0) : tile;
      double sum = 0.0;
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) sum += Vl[(long)(2 * p8 + hf) * 3 * 256 + (k * 4 + r) * 64 + ln];
      int ti, tj;
      gnc_tile<NTX>(hf, k, ti, tj);
      const int row = 16 * ti + (ln >> 4) + 4 * r, col = 16 * tj + (ln & 15);
      if (row < N && col < N) {
        g[(long)row * ga.gld + col] = sum;
        if (ti != tj) g[(long)col * ga.gld + row] = sum;      // mirror: exactly symmetric across the off-diagonal tiles (inside a diagonal tile both halves are sums of their own: symmetric to rounding)
      }
    }
  }
  PREP_STAMP(12);
  }      // gnc
  }      // do_avg
  if (!more) break;
  PREP_STAMP(14);
  // ---- the next subdomain of this workgroup: its slab and coefficients are in registers (or on their way), the template's tables in LDS
  __syncthreads();                               // every reader of the LDS copy is done (and every prefetch has landed)
  idx = idx_n;
  s = s_n;
  nb0 = nbr[s * 5], nb1 = nbr[s * 5 + 1], nb3 = nbr[s * 5 + 3], nb4 = nbr[s * 5 + 4];
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prefetch has landed
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // (the same wait where hipcc can see it: its wait-count pass then knows that the
                                                          //  stores of this subdomain are done as well and carries nothing around the loop)
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(pre[u]));
#pragma unroll
    for (int u = 0; u < FU; ++u) asm volatile("" : "+v"(fpre[u]));
    int tl = tid;
    asm volatile("" : "+v"(tl));
    d2* dst = reinterpret_cast<d2*>(Vl);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = u * NTHR + tl;
      if (i < total2) dst[i] = pre[u];
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int i = u * NTHR + tl;
      if (i < Q * f2) reinterpret_cast<d2*>(Fl)[i] = fpre[u];
    }
    for (int v = tid; v < t.nv; v += NTHR) {     // the same arithmetic on the LDS copy of the DoF-list offsets: the same bits
      const OsInfo o = oswald_vertex(t, nbr + s * 5, v, vptr);
      const int p0 = vptr[v];
      vinfo[v] = VInfo{o.inv, p0, vptr[v + 1] - p0};
    }
  }
  }      // subdomains of this workgroup
  };
  if (mainw) subdomains(std::true_type{});
  else subdomains(std::false_type{});
}

// Avg_side alone (the halo-dependent phase of a sharded pass): one item per (subdomain, side, side vertex, column).
__device__ __forceinline__ void vertex_side_body(const Tmpl& t, int S, const int* __restrict__ nbr, int N,
                                                 const double* __restrict__ V, double* __restrict__ AvgSide, int bx, int gx) {
  const int nvs = nvs_of(t);
  const long total = (long)(t.sub_list ? t.sub_count : S) * 4 * nvs * N;
  for (long idx = (long)bx * 256 + threadIdx.x; idx < total; idx += (long)gx * 256) {
    const int j = (int)(idx % N);
    long rem = idx / N;
    const int pos = (int)(rem % nvs);
    rem /= nvs;
    const int sd = (int)(rem % 4), s = subdomain_of(t, (int)(rem / 4));
    if (pos >= ((sd == 0 || sd == 3) ? t.nvx : t.nvy)) continue;
    const int v = sd == 0 ? pos : sd == 1 ? pos * t.nvx : sd == 2 ? pos * t.nvx + t.nvx - 1 : (t.nvy - 1) * t.nvx + pos;
    const OsInfo o = oswald_vertex(t, nbr + s * 5, v);
    const int s2 = nbr[s * 5 + side_to_slot(sd)];
    double a2 = 0.0;
    if (s2 >= 0 && o.inv != 0.0) {
      const int v2 = o.vside[sd];
      for (int p = t.vdof_ptr[v2]; p < t.vdof_ptr[v2 + 1]; ++p) a2 += V[((long)s2 * t.n + t.vdof_idx[p]) * N + j];
      a2 *= o.inv;
    }
    AvgSide[(((long)s * 4 + sd) * nvs + pos) * N + j] = a2;
  }
  if (t.opt_oswald_vertex) {      // the diagonal subdomains' shares at the four corners: Avg_corner [S][4][N] behind Avg_side
    const long totc = (long)(t.sub_list ? t.sub_count : S) * 4 * N;
    for (long idx = (long)bx * 256 + threadIdx.x; idx < totc; idx += (long)gx * 256) {
      const int j = (int)(idx % N), corner = (int)((idx / N) % 4), s = subdomain_of(t, (int)(idx / N / 4));
      const int v = ((corner & 1) ? t.nvx - 1 : 0) + t.nvx * ((corner & 2) ? t.nvy - 1 : 0);
      const OsInfo o = oswald_vertex(t, nbr + s * 5, v);
      double a3 = 0.0;
      if (o.vdiag >= 0) {
        const int sdg = t.nbr_diag[s * 4 + corner];
        for (int p = t.vdof_ptr[o.vdiag]; p < t.vdof_ptr[o.vdiag + 1]; ++p) a3 += V[((long)sdg * t.n + t.vdof_idx[p]) * N + j];
        a3 *= o.inv;
      }
      (AvgSide + (long)S * 4 * nvs * N)[((long)s * 4 + corner) * N + j] = a3;
    }
  }
}

__global__ __launch_bounds__(256) void k_vertex_side(Tmpl t, int S, const int* __restrict__ nbr, int N,
                                                     const double* __restrict__ V, double* __restrict__ AvgSide) {
  vertex_side_body(t, S, nbr, N, V, AvgSide, blockIdx.x, gridDim.x);
}

// Oswald interpolation error rows of one element, one column: slot 2 = own basis, other slots = neighbour images
__device__ inline void oswald_rows(const Tmpl& t, int s, int T, int slot, int N, int j, const double* __restrict__ V,
                                   const double* __restrict__ AvgSelf, const double* __restrict__ AvgSide, double w[3]) {
  const int nvs = nvs_of(t);
  for (int i = 0; i < 3; ++i) {
    const int r = 3 * T + i, v = t.dof_vertex[r];
    if (slot == 2) {
      w[i] = V[((long)s * t.n + r) * N + j] - AvgSelf[((long)s * t.nv + v) * N + j];
    } else {
      const int sd = slot_to_side(slot);
      const int lx = v % t.nvx, ly = v / t.nvx;
      const bool on = sd == 0 ? ly == 0 : sd == 1 ? lx == 0 : sd == 2 ? lx == t.nvx - 1 : ly == t.nvy - 1;
      const int pos = (sd == 0 || sd == 3) ? lx : ly;
      w[i] = on ? -AvgSide[(((long)s * 4 + sd) * nvs + pos) * N + j] : 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// F1: X = V.  Column groups of Y (N columns each), destination = base + s * sstride + row * ld + col
enum { G_SYS = 0, G_ENERGY = 1, G_MASS = 2, G_AA = 3, G_AB = 4 };
struct Grp {
  int kind, q, q2, ld;
  double* dst;
  double* dst_t;   // optional transposed copy (symmetric pair of G_aa), same strides
  long sstride;
  // k_f1v only: LDS column of local column c of this group = cb[c / 16] + c % 16; sym: the group's operator is symmetric
  // (only the entries row <= column are taken from the tiles, the others are their mirror images)
  int cb[4];
  int sym;
};
struct GrpTable {
  Grp g[F1_MAXG];
  int n;
};
struct F1Args {
  const double *V, *A_diag, *P_diag, *caa, *Aab, *Rself, *b;
  double* rhs_red;   // may be null (written only by the launch that carries it)
  int Q, N, S;
  // K-split of k_f1u (gridDim.z > 1): partial tiles [S][gridDim.z][f1u_part_size(NTX)] and one arrival counter per
  // subdomain (zero between launches: the workgroup that arrives last resets it)
  double* part;
  int* ticket;
};

// Producer / consumer workgroup of 8 waves (one workgroup per CU):
//   waves 0-3 (producers): wave w builds the X / Y rows of element 4 c + w of chunk c in LDS buffer c & 1, with the
//     global loads of chunk c + 1 (basis rows of the element and its three neighbours, flux rows, element blocks)
//     already in flight; VALU + memory only.
//   waves 4-7 (consumers): MFMA over chunk c from buffer c & 1 while the producers fill the other buffer; each owns
//     NTY column tiles x NTX row tiles of accumulators.
//   One s_barrier per chunk.  Waves w and w + 4 share a SIMD, so every SIMD holds one producer (VALU pipe) and one
//   consumer (matrix pipe), which execute concurrently.
template <int NTX, int NTY, int QP>
__global__ __launch_bounds__(512, 2) void k_f1(Tmpl t, F1Args a, GrpTable gt0, GrpTable gt1, GrpTable gt2) {
  constexpr int LDX = padded_ld(NTX);
  constexpr int LDY = 4 * NTY * 16 + 16;
  constexpr int PRE = 4;                       // per-lane prefetch registers for the element blocks (ESTR <= 256)
  extern __shared__ int idx[];                 // template adjacency cached once: nb_elem [nT][3], elem_rt [nT][3]
  __shared__ double Xs[2][3 * EC * LDX];
  __shared__ double Ys[2][3 * EC * LDY];
  __shared__ double Eb[EC * 256];              // per producer wave: A_q blocks, P block, A_ab^q blocks, c^{qq'} of its element
  __shared__ double red[EC * 64];
  __shared__ Grp grp[F1_MAXG];
  __shared__ int grp_n;
  const int s = subdomain_of(t, blockIdx.x), tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
#if F1_SPLIT_SIMD   // experiment: producers on SIMDs 0 / 2, consumers on SIMDs 1 / 3 (hardware wave w runs on SIMD w % 4)
  const int hw_wave = uniform(tid >> 6);
  const int wave = (hw_wave & 1) ? EC + (hw_wave >> 1) : (hw_wave >> 1);
#else
  const int wave = uniform(tid >> 6);
#endif
  const int N = a.N, Q = a.Q, S = a.S, QN = Q * N;
  const int oP = 36 * Q, oAb = oP + 36, oC = oAb + 9 * Q, ESTR = oC + Q * Q;    // <= 256 for Q <= 4
  // select this slice's group table with static indices only (a runtime index into a kernel-argument array would
  // make the compiler copy the tables to scratch)
#pragma unroll
  for (int g = 0; g < F1_MAXG; ++g)
    if (tid == g) grp[g] = blockIdx.y == 0 ? gt0.g[g] : blockIdx.y == 1 ? gt1.g[g] : gt2.g[g];
  if (tid == 0) grp_n = blockIdx.y == 0 ? gt0.n : blockIdx.y == 1 ? gt1.n : gt2.n;
  int* nbl = idx;
  int* rtl = idx + 3 * t.nT;
  double* Kl = reinterpret_cast<double*>(idx + 6 * t.nT);   // template stiffness K_T [nT][9] (same for all subdomains)
  if (QP == 0) {   // the straight-line producer reads the adjacency through the scalar cache instead
    for (int i = tid; i < 3 * t.nT; i += 512) {
      nbl[i] = t.nb_elem[i];
      rtl[i] = t.elem_rt[i];
    }
  }
  for (int i = tid; i < 9 * t.nT; i += 512) Kl[i] = t.stiff[i];
#ifndef F1_LDS_FILL
#define F1_LDS_FILL 0.0   // timing experiments: non-zero operands for the consumers when the producers do not stage
#endif
  for (int i = tid; i < 2 * 3 * EC * LDX; i += 512) (&Xs[0][0])[i] = F1_LDS_FILL * (1 + (i % 7));
  for (int i = tid; i < 2 * 3 * EC * LDY; i += 512) (&Ys[0][0])[i] = F1_LDS_FILL * (1 + (i % 5));
  __syncthreads();
  const int ng = uniform(grp_n);
  const int ncols = ng * N;
  // K-split (grid.z = 2, only when a rank has so few subdomains that half the CUs would idle): each half of the
  // elements is a workgroup of its own and the two partial results meet in zeroed outputs by atomic add -- with
  // exactly two contributions, 0 + a + b is the same double in either order.
  const int ksplit = gridDim.z;
  const int nchunks = t.nT / EC / ksplit;
  const int T0 = blockIdx.z * nchunks * EC;        // first element of this workgroup's share

  if (wave < EC) {
    // ================================================= producers
#ifdef F1_PRODUCER_PRIO
    __builtin_amdgcn_s_setprio(F1_PRODUCER_PRIO);
#endif
   if constexpr (QP > 0) {
    // ---- straight-line producer (compile-time Q, all groups in this slice; canonical group order
    // [SYS q][ENERGY][MASS][AA q<=q'][AB q q']).  Per element:
    //  * the QP + 1 four-block applies (A_q V, P V: 108 of the ~170 multiply-adds of an element) are ONE small MFMA
    //    product: rows = the 3 (QP + 1) stacked output rows, K = 12 = (block, local column), columns = basis columns.
    //    Both operands come straight from global memory in the MFMA lane layout, so the element blocks need no
    //    broadcast at all (on the VALU they cost two v_readlane per entry, or an LDS round trip that saturated the
    //    LDS pipe), and VALU / f64-MFMA issue serialise per SIMD on this chip, so 9 MFMAs replace ~330 VALU issues;
    //  * everything else (X rows, rhs, mass, c^{qq'} K V, A_ab R) stays on the VALU with lanes = basis columns; the
    //    A_ab^q / c^{qq'} entries sit one per lane in a prefetch register and are broadcast with v_readlane;
    //  * all loads of an element are a prefetch set issued one chunk ahead as asm loads and completed by an explicit
    //    s_waitcnt vmcnt(NLOADS) (see gload_f64).
    constexpr int R = 3 * (QP + 1);
    constexpr int NLOADS = 3 + 3 * NTX + 3 * QP + 1;
    struct Set {
      double A[3], B[3][NTX], rv[3][QP], ab;   // the element's own rows are B[0][.] of the lanes kq < 3
    };
    const bool do_rhs = a.rhs_red != nullptr && blockIdx.y == 0;
    const double* Vs = a.V + (long)s * t.n * N;
    const double* Rs = a.Rself + (long)s * t.nrt * QN;
    const int j = lane;
    const bool colj = j < N;
    const int jc = colj ? j : N - 1;   // idle lanes load (and never use) the last column: no exec masking around loads
    const int r16 = lane & 15, kq = lane >> 4;
    const double* asrc[3];
    int bbk[3], ck[3], colB[NTX], doff[3][NTX];
    {
      const int g = r16 / 3 < QP + 1 ? r16 / 3 : QP, i = r16 % 3;   // rows >= R repeat a valid row; they are never stored
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int k = 4 * ks + kq, bb = k / 3, c = k - 3 * bb;
        asrc[ks] = (g < QP ? a.A_diag + ((long)g * S + s) * t.nT * 36 : a.P_diag + (long)s * t.nT * 36) + bb * 9 + i * 3 + c;
        bbk[ks] = bb;
        ck[ks] = c;
      }
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) colB[ct] = 16 * ct + r16 < N ? 16 * ct + r16 : N - 1;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {           // LDS offset of output (row kq + 4 rr, column 16 ct + r16): group r / 3,
        const int r = kq + 4 * rr;               // local row r % 3 -- or a dump slot in the row padding (columns >= 448 are
#pragma unroll                                   // never read), so that the stores need no exec masking
        for (int ct = 0; ct < NTX; ++ct)
          doff[rr][ct] = (r < R && 16 * ct + r16 < N) ? (3 * wave + r % 3) * LDY + (r / 3) * N + 16 * ct + r16
                                                       : (3 * wave) * LDY + 4 * NTY * 16 + r16;
      }
    }
    const double* absrc = a.Aab + (long)s * t.nT * 9;   // lanes beyond the record re-read its first entry
    int abstr = 9;
    if (lane < 9 * QP) {
      absrc = a.Aab + ((long)(lane / 9) * S + s) * t.nT * 9 + lane % 9;
    } else if (lane < 9 * QP + QP * QP) {
      absrc = a.caa + ((long)(lane - 9 * QP) * S + s) * t.nT;
      abstr = 1;
    }
    const cint_p nbc = (cint_p)t.nb_elem;     // template adjacency of the (wave-uniform) element through the scalar cache
    const cint_p rtc = (cint_p)t.elem_rt;
    auto load_set = [&](int T, Set& x) {
      // A face without an in-subdomain neighbour has an all-zero block in A_diag / P_diag (its coupling lives in
      // A_cpl), so its rows may be any finite values: the element's own rows, which keeps every load unconditional.
      int nbT[4], rtT[3];
      nbT[0] = T;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const int nb = nbc[T * 3 + f];
        nbT[1 + f] = nb >= 0 ? nb : T;
        rtT[f] = rtc[T * 3 + f];
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) x.A[ks] = gload_f64(asrc[ks] + (long)T * 36);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int e = bbk[ks] == 0 ? nbT[0] : bbk[ks] == 1 ? nbT[1] : bbk[ks] == 2 ? nbT[2] : nbT[3];
        const double* rowp = Vs + (long)(3 * e + ck[ks]) * N;
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) x.B[ks][ct] = gload_f64(rowp + colB[ct]);
      }
#pragma unroll
      for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int q2 = 0; q2 < QP; ++q2) x.rv[f][q2] = gload_f64(Rs + (long)rtT[f] * QN + q2 * N + jc);
      x.ab = gload_f64(absrc + (long)T * abstr);
    };
    auto tie = [](double& v) { asm volatile("" : "+v"(v)); };
    auto tie_set = [&](Set& x) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        tie(x.A[ks]);
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) tie(x.B[ks][ct]);
#pragma unroll
        for (int q2 = 0; q2 < QP; ++q2) tie(x.rv[ks][q2]);
      }
      tie(x.ab);
    };
    auto wait_set = [&](Set& x) {   // everything older than the F1_PF younger prefetch sets is complete
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F1_PF * NLOADS));
      tie_set(x);
    };
    // Prefetch distance F1_PF chunks (F1_PF + 1 register sets in rotation).  Measured per chunk at config 3: loads +
    // barriers alone 0.87 us, consumer MFMAs alone 1.36 us, both together 2.2 us -- on this chip the load stream, the
    // producers' VALU work and the consumers' f64 MFMAs of one SIMD add up instead of overlapping, with distance 1 or
    // 2 alike, so the kernel time is about base + loads + staging + MFMA and each term has to be cut on its own.
    Set s0, s1, s2;
    load_set(T0 + wave, s0);
    if (F1_PF == 2) load_set(T0 + EC + wave, s1);  // every share has at least two chunks
    double rhs_part = 0.0;
    auto step = [&](int c, Set& cur, Set& nxt) {
      const int T = T0 + c * EC + wave;            // wave-uniform element
      double* Xb = &Xs[c & 1][0];
      double* Yb = &Ys[c & 1][0];
      if (wave == 0) F1_STAMP(0, c, 0);
      load_set(c + F1_PF < nchunks ? T + F1_PF * EC : T, nxt);   // unconditional (tail: this element again): the wait counts loads
      if (wave == 0) F1_STAMP(0, c, 1);
      wait_set(cur);
      if (wave == 0) F1_STAMP(0, c, 2);
#ifndef F1_NO_STAGE   // timing experiments (tools/build_variant.sh): producers only load and synchronise
#ifndef F1_NO_APPLY
      // ---- the stacked four-block applies on the matrix pipe
      d4 D[NTX];
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) D[ct] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A[ks], cur.B[ks][ct], D[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
          Yb[doff[rr][ct]] = D[ct][rr];
      if (wave == 0) F1_STAMP(0, c, 3);
#endif
      // ---- X rows: the own rows sit in the B operand of k-step 0 (lanes kq < 3 hold row kq); the padding columns
      // N .. 16 NTX - 1 get the clamped last column: they only reach output rows >= N, which are never stored
      if (kq < 3) {
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Xb[(3 * wave + kq) * LDX + 16 * ct + r16] = cur.B[0][ct];
      }
      // ---- the rest on the VALU, lanes = basis columns (own rows re-read from LDS in that layout)
#ifdef F1_NO_VALU_STAGE
      if (false) {
#else
      if (colj) {
#endif
        double v0[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v0[i] = Xb[(3 * wave + i) * LDX + j];
        if (do_rhs) {   // b_T through the scalar unit (lgkmcnt)
          const cdbl_p be = (cdbl_p)(a.b + (long)s * t.n + 3 * T);
          rhs_part += be[0] * v0[0] + be[1] * v0[1] + be[2] * v0[2];
        }
        double kv[3];
        const double* K = Kl + T * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          kv[i] = __builtin_fma(K[i * 3 + 2], v0[2], __builtin_fma(K[i * 3 + 1], v0[1], K[i * 3] * v0[0]));
        int g = QP + 1;
        auto put = [&](const double (&y)[3]) {
#pragma unroll
          for (int i = 0; i < 3; ++i) Yb[(3 * wave + i) * LDY + g * N + j] = y[i];
          ++g;
        };
        {
          const double m = ((cdbl_p)t.area)[T] * (1.0 / 12.0), sum = v0[0] + v0[1] + v0[2];
          const double y[3] = {m * (sum + v0[0]), m * (sum + v0[1]), m * (sum + v0[2])};
          put(y);
        }
#pragma unroll
        for (int q = 0; q < QP; ++q)
#pragma unroll
          for (int q2 = q; q2 < QP; ++q2) {
            const double cc = bcast_d(cur.ab, 9 * QP + q * QP + q2);
            const double y[3] = {cc * kv[0], cc * kv[1], cc * kv[2]};
            put(y);
          }
#pragma unroll
        for (int q = 0; q < QP; ++q) {
          double A[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) A[i] = bcast_d(cur.ab, 9 * q + i);
#pragma unroll
          for (int q2 = 0; q2 < QP; ++q2) {
            double y[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
              y[i] = __builtin_fma(A[i * 3 + 2], cur.rv[2][q2], __builtin_fma(A[i * 3 + 1], cur.rv[1][q2], A[i * 3] * cur.rv[0][q2]));
            put(y);
          }
        }
      }
#endif
      if (wave == 0) F1_STAMP(0, c, 4);
      lds_barrier();                               // barrier c: buffer c & 1 is complete (global loads stay in flight)
      if (wave == 0) F1_STAMP(0, c, 5);
    };
    int c = 0;
    if (F1_PF == 2) {
      for (; c + 3 <= nchunks; c += 3) {           // static register-set rotation: (cur, target of the new loads)
        step(c, s0, s2);
        step(c + 1, s1, s0);
        step(c + 2, s2, s1);
      }
      if (c < nchunks) step(c++, s0, s2);
      if (c < nchunks) step(c++, s1, s0);
    } else {
      for (; c < nchunks; c += 2) {                // nT is a multiple of 8, so nchunks is even
        step(c, s0, s1);
        step(c + 1, s1, s0);
      }
    }
    lds_barrier();                                 // final barrier (matches the consumers' count)
    // The sets loaded by the tail steps are never consumed.  They must still count as live until their loads have
    // landed: a destination the compiler considers dead is handed to other values while the load is in flight, and
    // the late write then corrupts them (seen as a GPU abort with prefetch distance 2).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_set(s0);
    tie_set(s1);
    if (F1_PF == 2) tie_set(s2);
    if (do_rhs) red[wave * 64 + lane] = rhs_part;
   } else {
    // ---- generic producer (runtime Q, or the groups do not fit in one slice): group table, the element record goes
    // through a per-wave LDS record, compiler-managed loads (the slow path: small templates, N > 40, Q > 2).
    const bool do_rhs = a.rhs_red != nullptr && blockIdx.y == 0;
    // element-record fetch of this wave's element: lane item o = lane + 64 k; (pointer, per-element stride) fixed
    const double* fsrc[PRE];
    int fstr[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int o = lane + 64 * k;
      fsrc[k] = a.A_diag + (long)s * t.nT * 36;   // lanes beyond the record re-read its first entry
      fstr[k] = 36;
      if (o < oP) {
        const int q = o / 36;
        fsrc[k] = a.A_diag + ((long)q * S + s) * t.nT * 36 + (o - 36 * q);
      } else if (o < oAb) {
        fsrc[k] = a.P_diag + (long)s * t.nT * 36 + (o - oP);
      } else if (o < oC) {
        const int q = (o - oAb) / 9;
        fsrc[k] = a.Aab + ((long)q * S + s) * t.nT * 9 + (o - oAb - 9 * q);
        fstr[k] = 9;
      } else if (o < ESTR) {
        fsrc[k] = a.caa + ((long)(o - oC) * S + s) * t.nT;
        fstr[k] = 1;
      }
    }
    double* Ee = Eb + wave * 256;
    const double* Vs = a.V + (long)s * t.n * N;
    const double* Rs = a.Rself + (long)s * t.nrt * QN;
    const int j = lane;
    const bool colj = j < N;
    const int jc = colj ? j : N - 1;
    double pre[PRE], npre[PRE], vb[4][3], nvb[4][3];
    auto fetch = [&](int T, double (&pr)[PRE], double (&vbx)[4][3]) {
#pragma unroll
      for (int k = 0; k < PRE; ++k) pr[k] = fsrc[k][(long)T * fstr[k]];
#pragma unroll
      for (int i = 0; i < 3; ++i) vbx[0][i] = Vs[(long)(3 * T + i) * N + jc];
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const int nb = nbl[T * 3 + f];
        const int e = nb >= 0 ? nb : T;            // no in-subdomain neighbour: the block is zero, any finite rows do
#pragma unroll
        for (int i = 0; i < 3; ++i) vbx[1 + f][i] = Vs[(long)(3 * e + i) * N + jc];
      }
    };
    fetch(T0 + wave, pre, vb);
    double rhs_part = 0.0;
    auto step = [&](int c, const double (&pre)[PRE], const double (&vb)[4][3], double (&npre)[PRE], double (&nvb)[4][3]) {
      const int T = T0 + c * EC + wave;            // wave-uniform element
      double* Xb = &Xs[c & 1][0];
      double* Yb = &Ys[c & 1][0];
#pragma unroll
      for (int k = 0; k < PRE; ++k)                // this wave's element record -> its private LDS copy
        if (lane + 64 * k < ESTR) Ee[lane + 64 * k] = pre[k];
      fetch(c + 1 < nchunks ? T + EC : T, npre, nvb);
      if (colj) {
#pragma unroll
        for (int i = 0; i < 3; ++i) Xb[(3 * wave + i) * LDX + j] = vb[0][i];
        if (do_rhs) {
          const double* be = a.b + (long)s * t.n + 3 * T;
          rhs_part += be[0] * vb[0][0] + be[1] * vb[0][1] + be[2] * vb[0][2];
        }
        double kv[3];
        const double* K = Kl + T * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i) kv[i] = K[i * 3] * vb[0][0] + K[i * 3 + 1] * vb[0][1] + K[i * 3 + 2] * vb[0][2];
        for (int g = 0; g < ng; ++g) {
          const int kind = uniform(grp[g].kind), q = uniform(grp[g].q), q2 = uniform(grp[g].q2);
          double y[3] = {0, 0, 0};
          if (kind == G_SYS || kind == G_ENERGY) {
            const double* blk = Ee + (kind == G_SYS ? 36 * q : oP);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
              for (int i = 0; i < 3; ++i)
                y[i] += blk[bb * 9 + i * 3] * vb[bb][0] + blk[bb * 9 + i * 3 + 1] * vb[bb][1] + blk[bb * 9 + i * 3 + 2] * vb[bb][2];
          } else if (kind == G_MASS) {
            const double m = t.area[T] / 12.0, sum = vb[0][0] + vb[0][1] + vb[0][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) y[i] = m * (sum + vb[0][i]);
          } else if (kind == G_AA) {
            const double cc = Ee[oC + q * Q + q2];
#pragma unroll
            for (int i = 0; i < 3; ++i) y[i] = cc * kv[i];
          } else {  // G_AB: A_ab^q restricted to the self part of the flux image
            const double* A = Ee + oAb + 9 * q;
            double r3[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) r3[f] = Rs[(long)rtl[T * 3 + f] * QN + q2 * N + j];
#pragma unroll
            for (int i = 0; i < 3; ++i) y[i] = A[i * 3] * r3[0] + A[i * 3 + 1] * r3[1] + A[i * 3 + 2] * r3[2];
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) Yb[(3 * wave + i) * LDY + g * N + j] = y[i];
        }
      }
      lds_barrier();                               // barrier c: buffer c & 1 is complete
    };
    for (int c = 0; c < nchunks; c += 2) {         // nT is a multiple of 8, so nchunks is even
      step(c, pre, vb, npre, nvb);
      step(c + 1, npre, nvb, pre, vb);
    }
    lds_barrier();                                 // final barrier (matches the consumers' count)
    if (do_rhs) red[wave * 64 + lane] = rhs_part;
   }
  } else {
    // ================================================= consumers
    const int cw = wave - EC;
    d4 acc[NTX][NTY];
#pragma unroll
    for (int i = 0; i < NTX; ++i)
#pragma unroll
      for (int jj = 0; jj < NTY; ++jj) acc[i][jj] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nchunks; ++c) {
      if (cw == 0) F1_STAMP(1, c, 0);
      lds_barrier();                               // barrier c
      if (cw == 0) F1_STAMP(1, c, 1);
      const double* Xb = &Xs[c & 1][0];
      const double* Yb = &Ys[c & 1][0];
#ifndef F1_NO_MFMA    // timing experiments: consumers only synchronise
#pragma unroll
      for (int kk = 0; kk < 3 * EC; kk += 4) {
        double av[NTX];
#pragma unroll
        for (int i = 0; i < NTX; ++i) av[i] = Xb[(kk + lk) * LDX + i * 16 + li];
#pragma unroll
        for (int jt = 0; jt < NTY; ++jt) {         // tiles beyond ncols multiply zero columns (harmless, branch-free)
          const double bv = Yb[(kk + lk) * LDY + (cw * NTY + jt) * 16 + li];
#pragma unroll
          for (int i = 0; i < NTX; ++i) acc[i][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv, acc[i][jt], 0, 0, 0);
        }
      }
#endif
      if (cw == 0) F1_STAMP(1, c, 2);
    }
    lds_barrier();                                 // final barrier
    // ---- epilogue: scatter the tiles to their destination arrays (static accumulator indices only)
#pragma unroll
    for (int jt = 0; jt < NTY; ++jt) {
      const int col = (cw * NTY + jt) * 16 + li;
      const bool live = col < ncols;
      const int g = live ? col / N : 0, jj = col - g * N;
      const int ld = grp[g].ld;
      double* dst = grp[g].dst + (long)s * grp[g].sstride + jj;
      double* dst_t = grp[g].dst_t ? grp[g].dst_t + (long)s * grp[g].sstride + (long)jj * ld : nullptr;
#pragma unroll
      for (int i = 0; i < NTX; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i * 16 + lk + 4 * r;
          const double val = acc[i][jt][r];
          if (live && row < N) {
            if (ksplit == 1) {
              dst[(long)row * ld] = val;
              if (dst_t) dst_t[row] = val;
            } else {
              unsafeAtomicAdd(dst + (long)row * ld, val);
              if (dst_t) unsafeAtomicAdd(dst_t + row, val);
            }
          }
        }
      }
    }
  }
  if (a.rhs_red != nullptr && blockIdx.y == 0) {   // fixed-order sum over the EC producer waves
    __syncthreads();
    if (tid < N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      if (ksplit == 1)
        a.rhs_red[(long)s * N + tid] = sum;
      else
        unsafeAtomicAdd(a.rhs_red + (long)s * N + tid, sum);
    }
  }
}

// Write-through (`sc1`) stores and L1-bypassing (`sc1`) loads for data one workgroup hands to another inside a launch
// (K-split of k_f1u).
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_sc1_b128(double* p, double x, double y) {
  const d2v v = {x, y};
  // s_nop 1 inside the string: the store reads its four data registers after issue, and nothing outside the string
  // keeps the compiler's next VALU instruction from overwriting them
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sc1_b64(double* p, double x) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1_b64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// ---------------------------------------------------------------------------------------------------------
// F1, unified-role form (compile-time Q, every column group in one slice: the configurations of BASELINE.json).
//
// What the producer / consumer form above runs into on gfx950 (tools/ubench/overlap64.hip, occ64.hip, tools/f1_trace.py):
// while a wave streams v_mfma_f64_16x16x4 (68-70 cycles each, no pipelining between them), every other wave of the same
// SIMD is starved -- f64 FMAs, integer VALU, LDS and global-memory instructions alike; a second workgroup per CU does not
// fill the gaps either.  So a SIMD's time is (MFMA issue) + (everything else), never the maximum of the two, and in the
// producer / consumer form the "everything else" of a chunk was ONE latency-bound producer wave per SIMD (3 850 of the
// 8 250 cycles per chunk) while the consumer wave sat at the barrier.
//
// Here all eight waves alternate between the two kinds of work, chunk by chunk:
//   stage(c):  waves 0-3 (role A, element 4 c + w): the stacked four-block applies A_q V, P V on the matrix pipe, the X
//              rows, b . v, the mass group, K_T v and the c^{qq'} K V groups; waves 4-7 (role B, element 4 c + w - 4):
//              the A_ab R groups (the v_readlane-heavy part).  (The c^{qq'} K V groups were role B's at first: its stage
//              was then the longer one by ~1 350 cycles per chunk, which role A spent at the barrier -- 444 -> 419 us
//              with them moved and the odd column tile given to role B instead.)  Two waves per SIMD share the staging of one element, so
//              their latencies overlap; every wave's global loads are a prefetch set issued one chunk ahead (asm loads +
//              hand-counted s_waitcnt vmcnt(n), see gload_f64) that lands during the MFMA burst in between.
//   barrier c
//   mfma(c):   all eight waves; wave w owns 3 (w < 4) or 4 (w >= 4) of its SIMD's 7 column tiles x NTX row tiles.
// One barrier per chunk: a wave reaches stage(c + 1) (writes buffer (c + 1) & 1) only after its own mfma(c), and every
// wave finished mfma(c - 1) -- the last reader of that buffer -- before it arrived at barrier c.
template <int NTX, int QP, int ROLE>
__device__ __forceinline__ void f1u_body(const Tmpl& t, const F1Args& a, double* __restrict__ Xs, double* __restrict__ Ys,
                                         const double* __restrict__ Kl, double* __restrict__ red, int* __restrict__ flag,
                                         const Grp* grp, int ng) {
  constexpr int LDX = padded_ld(NTX);
  constexpr int NTYS = F1_NTY;                         // column tiles per SIMD (waves w and w + 4)
  constexpr int LDY = 4 * NTYS * 16 + 16;
  constexpr int NT = ROLE == 0 ? NTYS / 2 : (NTYS + 1) / 2;   // role A stages more (below), so role B takes the odd tile
  const int s = subdomain_of(t, blockIdx.x), tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6), e = wave & 3;    // e: element of the chunk this wave stages, SIMD it runs on
  // column tiles are dealt round-robin over the SIMDs (tile 4 k + e belongs to SIMD e; role A takes the even k, role B the
  // odd ones), so that basis sizes whose groups fill only part of the 4 NTYS tiles still load every SIMD evenly; tiles
  // beyond the last column are skipped (wave-uniform branch)
  auto tile_of = [&](int jt) { return 4 * (2 * jt + (ROLE == 0 ? 1 : 0)) + e; };
  const int N = a.N, S = a.S, QN = QP * N;
  const int ncols = ng * N;
  const int ksplit = gridDim.z;
  const int nchunks = t.nT / EC / ksplit;
  const int T0 = blockIdx.z * nchunks * EC;
  const double* Vs = a.V + (long)s * t.n * N;
  const int j = lane;
  const bool colj = j < N;
  const int jc = colj ? j : N - 1;                     // idle lanes load (and never use) the last column: no exec masking
  const bool do_rhs = ROLE == 0 && a.rhs_red != nullptr;
  double rhs_part = 0.0;

  d4 acc[NTX][NT];
#pragma unroll
  for (int i = 0; i < NTX; ++i)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) acc[i][jt] = (d4){0.0, 0.0, 0.0, 0.0};

  // (Measured and dropped: skipping the 14 of 84 tiles that lie strictly below the diagonal of a symmetric group and
  // storing them from the lanes that hold the mirror entries -- the wave-uniform branch around every MFMA breaks up the
  // compiler's operand schedule of this loop: 465 us against 443 us at config 3.  Second form: one liveness bit mask per
  // wave, tested once per TILE and phase, all LDS operands of the phase requested up front, the mirror entries written by
  // the epilogue -- correct, 14 % fewer MFMAs, and slower again: 464 us against 419 us.)
  auto mfma_phase = [&](int c) {
    const double* Xb = Xs + (c & 1) * 3 * EC * LDX;
    const double* Yb = Ys + (c & 1) * 3 * EC * LDY;
#pragma unroll
    for (int kk = 0; kk < 3 * EC; kk += 4) {
      double av[NTX];
#pragma unroll
      for (int i = 0; i < NTX; ++i) av[i] = Xb[(kk + lk) * LDX + i * 16 + li];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if (tile_of(jt) * 16 >= ncols) continue;       // wave-uniform
        const double bv = Yb[(kk + lk) * LDY + tile_of(jt) * 16 + li];
#pragma unroll
        for (int i = 0; i < NTX; ++i) acc[i][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv, acc[i][jt], 0, 0, 0);
      }
    }
  };
  auto tie = [](double& v) { asm volatile("" : "+v"(v)); };

  if constexpr (ROLE == 0) {
    // ------------------------------------------------------------- role A: applies on the matrix pipe, X rows, rhs, mass
    constexpr int R = 3 * (QP + 1);
    constexpr int NLOADS = 3 + 3 * NTX + 3 + 1;
    static_assert(NLOADS <= 63, "vmcnt is a 6-bit counter");
    struct Set {
      double A[3], B[3][NTX], v0[3], c;   // c: the element's c^{qq'} (one entry per lane q QP + q', broadcast with v_readlane)
    };
    const int r16 = li, kq = lk;
    const double* asrc[3];
    int bbk[3], ck[3], colB[NTX], doff[3][NTX];
    {
      const int g = r16 / 3 < QP + 1 ? r16 / 3 : QP, i = r16 % 3;   // rows >= R repeat a valid row; they are never stored
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int k = 4 * ks + kq, bb = k / 3, cc = k - 3 * bb;
        asrc[ks] = (g < QP ? a.A_diag + ((long)g * S + s) * t.nT * 36 : a.P_diag + (long)s * t.nT * 36) + bb * 9 + i * 3 + cc;
        bbk[ks] = bb;
        ck[ks] = cc;
      }
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) colB[ct] = 16 * ct + r16 < N ? 16 * ct + r16 : N - 1;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {           // LDS offset of output (row kq + 4 rr, column 16 ct + r16): group r / 3, local
        const int r = kq + 4 * rr;               // row r % 3 -- or a dump slot in the row padding (never read)
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct)
          doff[rr][ct] = (r < R && 16 * ct + r16 < N) ? (3 * e + r % 3) * LDY + (r / 3) * N + 16 * ct + r16
                                                       : (3 * e) * LDY + 4 * NTYS * 16 + r16;
      }
    }
    const double* csrc = a.caa + ((long)(lane < QP * QP ? lane : 0) * S + s) * t.nT;
    const cint_p nbc = (cint_p)t.nb_elem;     // template adjacency of the (wave-uniform) element through the scalar cache
    // Wave-uniform data of an element (its three in-subdomain neighbours, b_T, |T|) comes through the scalar cache.  A
    // scalar load issued where its value is needed exposes its whole latency (the wave has nothing else to issue), so
    // these are fetched one stage ahead as well: `Sc` of the element staged NEXT is requested at the start of a stage
    // and first read in the following one, after an MFMA phase.
    struct Sc {
      int nb[3];
      double be[3], area;
    };
    auto load_sc = [&](int T, Sc& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) x.nb[f] = nbc[T * 3 + f];
      const cdbl_p be = (cdbl_p)(a.b + (long)s * t.n + 3 * T);
#pragma unroll
      for (int i = 0; i < 3; ++i) x.be[i] = be[i];
      x.area = ((cdbl_p)t.area)[T];
    };
    auto load_set = [&](int T, const Sc& sc, Set& x) {
      // A face without an in-subdomain neighbour has an all-zero block in A_diag / P_diag (its coupling lives in A_cpl),
      // so its rows may be any finite values: the element's own rows, which keeps every load unconditional.
      int nbT[4];
      nbT[0] = T;
#pragma unroll
      for (int f = 0; f < 3; ++f) nbT[1 + f] = sc.nb[f] >= 0 ? sc.nb[f] : T;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) x.A[ks] = gload_f64(asrc[ks] + (long)T * 36);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int el = bbk[ks] == 0 ? nbT[0] : bbk[ks] == 1 ? nbT[1] : bbk[ks] == 2 ? nbT[2] : nbT[3];
        const double* rowp = Vs + (long)(3 * el + ck[ks]) * N;
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) x.B[ks][ct] = gload_f64(rowp + colB[ct]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) x.v0[i] = gload_f64(Vs + (long)(3 * T + i) * N + jc);
      x.c = gload_f64(csrc + T);
    };
    auto tie_set = [&](Set& x) {
      tie(x.c);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        tie(x.A[ks]);
        tie(x.v0[ks]);
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) tie(x.B[ks][ct]);
      }
    };
    // stage c: `cur` / `sc_cur` belong to element T (chunk c), `nxt` / `sc_nxt` to the element of chunk c + 1 (requested
    // in stage c - 1), `sc_far` receives the scalars of chunk c + 2
    auto stage = [&](int c, Set& cur, Set& nxt, const Sc& sc_cur, const Sc& sc_nxt, Sc& sc_far) {
      const int T = T0 + c * EC + e;               // wave-uniform element
      double* Xb = Xs + (c & 1) * 3 * EC * LDX;
      double* Yb = Ys + (c & 1) * 3 * EC * LDY;
      if (e == 0) F1_STAMP(0, c, 0);
      load_sc(c + 2 < nchunks ? T + 2 * EC : T, sc_far);
      load_set(c + 1 < nchunks ? T + EC : T, sc_nxt, nxt);   // unconditional (tail: this element again): the wait counts loads
      if (e == 0) F1_STAMP(0, c, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS));
      tie_set(cur);
      if (e == 0) F1_STAMP(0, c, 2);
      d4 D[NTX];
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) D[ct] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A[ks], cur.B[ks][ct], D[ct], 0, 0, 0);
      // X rows: the own rows sit in the B operand of k-step 0 (lanes kq < 3 hold row kq); the padding columns N .. 16 NTX - 1
      // get the clamped last column: they only reach output rows >= N, which are never stored
      if (kq < 3) {
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Xb[(3 * e + kq) * LDX + 16 * ct + r16] = cur.B[0][ct];
      }
      if (colj) {
        if (do_rhs) rhs_part += sc_cur.be[0] * cur.v0[0] + sc_cur.be[1] * cur.v0[1] + sc_cur.be[2] * cur.v0[2];
        const double m = sc_cur.area * (1.0 / 12.0), sum = cur.v0[0] + cur.v0[1] + cur.v0[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) Yb[(3 * e + i) * LDY + (QP + 1) * N + j] = m * (sum + cur.v0[i]);
        // K_T v and the c^{qq'} K V groups (moved here from role B, whose stage was the longer one: role A waited ~1 350
        // cycles per chunk at the barrier)
        double kv[3];
        const double* K = Kl + (T - T0) * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          kv[i] = __builtin_fma(K[i * 3 + 2], cur.v0[2], __builtin_fma(K[i * 3 + 1], cur.v0[1], K[i * 3] * cur.v0[0]));
        int g = QP + 2;
#pragma unroll
        for (int q = 0; q < QP; ++q)
#pragma unroll
          for (int q2 = q; q2 < QP; ++q2) {
            const double cc = bcast_d(cur.c, q * QP + q2);
#pragma unroll
            for (int i = 0; i < 3; ++i) Yb[(3 * e + i) * LDY + g * N + j] = cc * kv[i];
            ++g;
          }
      }
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) Yb[doff[rr][ct]] = D[ct][rr];
      if (e == 0) F1_STAMP(0, c, 3);
    };
    Set s0, s1;
    Sc c0, c1, c2;
    load_sc(T0 + e, c0);
    load_sc(nchunks > 1 ? T0 + EC + e : T0 + e, c1);
    load_set(T0 + e, c0, s0);
    // nT is a multiple of 8, so nchunks is even; the scalar sets rotate with period 3, hence six stages per round
    auto round = [&](int c, Set& sa, Set& sb, Sc& x0, Sc& x1, Sc& x2) {
      stage(c, sa, sb, x0, x1, x2);
      lds_barrier();
      if (e == 0) F1_STAMP(0, c, 4);
      mfma_phase(c);
      if (e == 0) F1_STAMP(0, c, 5);
    };
    int c = 0;
    for (; c + 6 <= nchunks; c += 6) {
      round(c, s0, s1, c0, c1, c2);
      round(c + 1, s1, s0, c1, c2, c0);
      round(c + 2, s0, s1, c2, c0, c1);
      round(c + 3, s1, s0, c0, c1, c2);
      round(c + 4, s0, s1, c1, c2, c0);
      round(c + 5, s1, s0, c2, c0, c1);
    }
    if (c < nchunks) { round(c, s0, s1, c0, c1, c2); ++c; }
    if (c < nchunks) { round(c, s1, s0, c1, c2, c0); ++c; }
    if (c < nchunks) { round(c, s0, s1, c2, c0, c1); ++c; }
    if (c < nchunks) { round(c, s1, s0, c0, c1, c2); ++c; }
    // the set loaded by the tail step is never consumed; it must count as live until its loads have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_set(s0);
    tie_set(s1);
    if (do_rhs) red[e * 64 + lane] = rhs_part;
  } else {
    // ------------------------------------------------------------- role B: K_T v, c^{qq'} K V and A_ab R groups
    constexpr int NLOADS = 3 * QP + 1;
    struct Set {
      double rv[3][QP], ab;
    };
    const double* Rs = a.Rself + (long)s * t.nrt * QN;
    // the element's A_ab^q blocks and c^{qq'}: one entry per lane in a prefetch register, broadcast with v_readlane
    // (through the scalar cache into SGPRs instead they cost 44 SGPRs held across the MFMA phase: the kernel then spills
    // SGPRs and runs 10 % slower -- measured)
    const double* absrc = a.Aab + (long)s * t.nT * 9;   // lanes beyond the record re-read its first entry
    constexpr int abstr = 9;
    if (lane < 9 * QP) absrc = a.Aab + ((long)(lane / 9) * S + s) * t.nT * 9 + lane % 9;
    const cint_p rtc = (cint_p)t.elem_rt;
    struct Sc {       // RT0 rows of the element's three faces, fetched through the scalar cache one stage ahead (see role A)
      int rt[3];
    };
    auto load_sc = [&](int T, Sc& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) x.rt[f] = rtc[T * 3 + f];
    };
    auto load_set = [&](int T, const Sc& sc, Set& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int q2 = 0; q2 < QP; ++q2) x.rv[f][q2] = gload_f64(Rs + (long)sc.rt[f] * QN + q2 * N + jc);
      x.ab = gload_f64(absrc + (long)T * abstr);
    };
    auto tie_set = [&](Set& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) {
#pragma unroll
        for (int q2 = 0; q2 < QP; ++q2) tie(x.rv[f][q2]);
      }
      tie(x.ab);
    };
    auto stage = [&](int c, Set& cur, Set& nxt, const Sc& sc_nxt, Sc& sc_far) {
      const int T = T0 + c * EC + e;
      double* Yb = Ys + (c & 1) * 3 * EC * LDY;
      if (e == 0) F1_STAMP(1, c, 0);
      load_sc(c + 2 < nchunks ? T + 2 * EC : T, sc_far);
      load_set(c + 1 < nchunks ? T + EC : T, sc_nxt, nxt);
      if (e == 0) F1_STAMP(1, c, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS));
      tie_set(cur);
      if (e == 0) F1_STAMP(1, c, 2);
      if (colj) {
        int g = QP + 2 + QP * (QP + 1) / 2;
        auto put = [&](const double (&y)[3]) {
#pragma unroll
          for (int i = 0; i < 3; ++i) Yb[(3 * e + i) * LDY + g * N + j] = y[i];
          ++g;
        };
#pragma unroll
        for (int q = 0; q < QP; ++q) {
          double A[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) A[i] = bcast_d(cur.ab, 9 * q + i);
#pragma unroll
          for (int q2 = 0; q2 < QP; ++q2) {
            double y[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
              y[i] = __builtin_fma(A[i * 3 + 2], cur.rv[2][q2], __builtin_fma(A[i * 3 + 1], cur.rv[1][q2], A[i * 3] * cur.rv[0][q2]));
            put(y);
          }
        }
      }
      if (e == 0) F1_STAMP(1, c, 3);
    };
    Set s0, s1;
    Sc c0, c1;
    load_sc(T0 + e, c0);
    load_set(T0 + e, c0, s0);
    load_sc(nchunks > 1 ? T0 + EC + e : T0 + e, c0);       // scalars of chunk 1 (used by stage 0), then alternating
    auto round = [&](int c, Set& sa, Set& sb, Sc& x0, Sc& x1) {
      stage(c, sa, sb, x0, x1);
      lds_barrier();
      if (e == 0) F1_STAMP(1, c, 4);
      mfma_phase(c);
      if (e == 0) F1_STAMP(1, c, 5);
    };
    for (int c = 0; c < nchunks; c += 2) {                   // nT is a multiple of 8, so nchunks is even
      round(c, s0, s1, c0, c1);
      round(c + 1, s1, s0, c1, c0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_set(s0);
    tie_set(s1);
  }
  // ---- K-split: every workgroup of the subdomain leaves its partial tiles in `part`; the one whose arrival is counted
  // last sums them in the fixed order z = 0 .. ksplit - 1 (its own included, re-read from memory), so the result does not
  // depend on the arrival order, and goes on to the epilogue.  No workgroup waits for another one.
  // Visibility across CUs / XCDs (per-CU L1s are never refreshed, per-XCD L2s are not coherent): the partials are stored
  // write-through (16-byte `sc1` stores), every storing wave drains its stores (s_waitcnt vmcnt(0)) before the
  // workgroup barrier, one lane then adds to the subdomain's counter (agent-scope atomic), and the last arriver reads
  // every partial with `sc1` loads only -- no cache-wide release / acquire, which at eight XCDs costs more than the
  // split saves (measured: __threadfence() on both sides made the split launch slower than the unsplit one).
  if (ksplit > 1) {
    constexpr int PW = NTX * 4 * 4 * 64;               // doubles per wave: [i][jt < 4][r / 2][lane][r % 2]
    const long wg = 8L * PW + 64;                      // + the partial rhs_red
    double* mine = a.part + ((long)s * ksplit + blockIdx.z) * wg;
    double* pw = mine + (long)wave * PW + 2 * lane;
#pragma unroll
    for (int i = 0; i < NTX; ++i)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int h = 0; h < 2; ++h) store_sc1_b128(pw + ((i * 4 + jt) * 2 + h) * 128, acc[i][jt][2 * h], acc[i][jt][2 * h + 1]);
    __syncthreads();                                   // red[] of the role-A waves is complete
    if (a.rhs_red != nullptr && tid < N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      store_sc1_b64(mine + 8L * PW + tid, sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier in front of the arrival
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.ticket + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != ksplit - 1) return;                   // workgroup-uniform
    if (tid == 0) __hip_atomic_store(a.ticket + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    const double* all = a.part + (long)s * ksplit * wg;
#pragma unroll
    for (int i = 0; i < NTX; ++i)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) acc[i][jt] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int z = 0; z < ksplit; ++z) {
      const double* pz = all + z * wg + (long)wave * PW + 2 * lane;
#pragma unroll
      for (int i = 0; i < NTX; ++i)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][jt][r] += load_sc1_b64(pz + ((i * 4 + jt) * 2 + r / 2) * 128 + r % 2);
    }
    if (a.rhs_red != nullptr && tid < N) {
      double sum = 0.0;
      for (int z = 0; z < ksplit; ++z) sum += load_sc1_b64(all + z * wg + 8L * PW + tid);
      a.rhs_red[(long)s * N + tid] = sum;
    }
  }
  // ---- epilogue: scatter the tiles to their destination arrays (static accumulator indices only)
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int col = tile_of(jt) * 16 + li;
    const bool live = col < ncols;
    const int g = live ? col / N : 0, jj = col - g * N;
    const int ld = grp[g].ld;
    double* dst = grp[g].dst + (long)s * grp[g].sstride + jj;
    double* dst_t = grp[g].dst_t ? grp[g].dst_t + (long)s * grp[g].sstride + (long)jj * ld : nullptr;
#pragma unroll
    for (int i = 0; i < NTX; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i * 16 + lk + 4 * r;
        const double val = acc[i][jt][r];
        if (live && row < N) {
          dst[(long)row * ld] = val;
          if (dst_t) dst_t[row] = val;
        }
      }
    }
  }
}

// doubles of K-split scratch per workgroup (see the K-split block of f1u_body)
constexpr long f1u_part_size(int ntx) { return 8L * ntx * 4 * 4 * 64 + 64; }

template <int NTX, int QP>
__global__ __launch_bounds__(512, 2) void k_f1u(Tmpl t, F1Args a, GrpTable gt) {
  constexpr int LDX = padded_ld(NTX);
  constexpr int LDY = 4 * F1_NTY * 16 + 16;
  extern __shared__ double Kl[];               // template stiffness K_T [nT / gridDim.z][9] of this workgroup's element range
  __shared__ double Xs[2 * 3 * EC * LDX];
  __shared__ double Ys[2 * 3 * EC * LDY];
  __shared__ double red[EC * 64];
  __shared__ Grp grp[F1_MAXG];
  __shared__ int flag;
  const int tid = threadIdx.x;
#pragma unroll
  for (int g = 0; g < F1_MAXG; ++g)
    if (tid == g) grp[g] = gt.g[g];
  {
    const int nel = t.nT / (int)gridDim.z;
    const double* src = t.stiff + 9L * nel * blockIdx.z;
    for (int i = tid; i < 9 * nel; i += 512) Kl[i] = src[i];
  }
  for (int i = tid; i < 2 * 3 * EC * LDX; i += 512) Xs[i] = 0.0;
  for (int i = tid; i < 2 * 3 * EC * LDY; i += 512) Ys[i] = 0.0;
  __syncthreads();
  const int ng = gt.n;
  if (uniform(tid >> 6) < EC)
    f1u_body<NTX, QP, 0>(t, a, Xs, Ys, Kl, red, &flag, grp, ng);
  else
    f1u_body<NTX, QP, 1>(t, a, Xs, Ys, Kl, red, &flag, grp, ng);
  if (gridDim.z == 1 && a.rhs_red != nullptr) {   // fixed-order sum over the EC role-A waves (K-split: done in f1u_body)
    __syncthreads();
    if (tid < a.N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      a.rhs_red[(long)subdomain_of(t, blockIdx.x) * a.N + tid] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// F1, lean form (k_f1v; round 3).  Same workgroup shape, LDS layout, tile ownership, K-split and epilogue as k_f1u; what changed
// is the number of non-MFMA instructions of a chunk.  On gfx950 a SIMD's time is (f64 MFMA issue) + (every other instruction of
// its two waves) (tools/ubench/overlap64.hip), and tools/ubench/interleave64.hip shows that the vector-memory loads themselves
// are cheap (~4 cycles per instruction and CU when all eight waves issue them in one burst; one load BETWEEN two MFMAs costs a
// whole MFMA slot): the ~3 050 cycles a chunk of k_f1u spends outside the matrix pipe are instruction count -- 64-bit address
// arithmetic, v_readlane broadcasts, f64 FMAs on 40 of 64 lanes, a wave-uniform liveness test in front of every MFMA.  Hence:
//   * the mass, stiffness and right-hand-side rows ride on the apply MFMA of role A: its A operand has 16 rows, of which the
//     stacked blocks A_q, P use 3 (Q + 1); rows 3 (Q + 1) .. hold M_T = |T|/12 (I + J), K_T and b_T (self block only, the other
//     k read zeros), so M V, K V and b . V leave the matrix pipe with the system rows -- no v0 loads, no FMAs; the c^{qq'} K V
//     groups are K V times a scalar that comes through the scalar cache (one v_mul per tile);
//   * every load of the neighbours' rows has a wave-uniform base (SGPR pair) and a 32-bit lane offset: el * 24 N + lane constant;
//     a lane owns ADJACENT basis columns (tile 2p: column 32 p + 2 l, tile 2p + 1: column 32 p + 2 l + 1), so one 16-byte load
//     feeds two column tiles (9 -> 6 loads per element at N = 40; the permutation is undone by the LDS addresses, which are
//     lane constants anyway);
//   * role B reads a flux row of both components with ONE 16-byte load (lanes 0-31: component 0, lanes 32-63: component 1, two
//     adjacent columns each) and writes its Y rows as 16-byte LDS stores: 3 + 1 loads and 6 stores per element instead of 7 and 12;
//   * the number of column tiles per SIMD is a template parameter chosen on the host from Q N: no liveness test in the MFMA loop
//     (at most three tiles beyond the last column multiply zeros).
// Needs even N (adjacent-column pairs).  Everything else (odd N, N > 48 ...) runs k_f1u / k_f1.

constexpr int F1V_SLOTS = 16;      // most accumulator tiles one wave of k_f1v owns
constexpr long f1v_part_size(int) { return 8L * F1V_SLOTS * 4 * 64 + 64; }

// Column tiles of k_f1v ("levels": level l = the four column tiles 4 l + e of the SIMDs e = 0 .. 3).  The launcher orders the
// columns of Y so that the tiles come in ascending order of the number of row tiles they need (see f1v_layout): a symmetric
// group is cut into blocks of 16 columns, block j needs the row tiles 0 .. j only (the entries below the diagonal are mirror
// images), the short last blocks of all symmetric groups are packed into shared tiles.  L1 / L2 / L3 levels need 1 / 2 / 3 row
// tiles -- compile-time, the same for every SIMD: no liveness test in the MFMA loop and no MFMA on a dead tile
// (N = 40, Q = 2: 17 instead of 21 MFMAs per SIMD and k-step).
template <int L1, int L2, int L3>
struct F1vLevels {
  static constexpr int NL = L1 + L2 + L3;
  static constexpr int cls(int l) { return l < L1 ? 1 : l < L1 + L2 ? 2 : 3; }
  // role A (whose staging holds more registers) takes the first LA levels: the cheap ones
  static constexpr int LA = NL >= 6 ? 2 : NL >= 3 ? 1 : 0;
  static constexpr int first(int role) { return role == 0 ? 0 : LA; }
  static constexpr int count(int role) { return role == 0 ? LA : NL - LA; }
  static constexpr int slots_before(int role, int jt) {      // accumulator slot of (owned level jt, row tile 0)
    int n = 0;
    for (int k = 0; k < jt; ++k) n += cls(first(role) + k);
    return n;
  }
  static constexpr int slots(int role) { return slots_before(role, count(role)); }
};

template <int NTX, int QP, int L1, int L2, int L3, int ROLE>
__device__ __forceinline__ void f1v_body(const Tmpl& t, const F1Args& a, double* __restrict__ Xs, double* __restrict__ Ys,
                                         double* __restrict__ red, int* __restrict__ flag, const int* __restrict__ colmap,
                                         const Grp* grp) {
  static_assert(QP == 1 || QP == 2, "rows of the stacked apply: 3 (Q + 3) + 1 <= 16");
  using LV = F1vLevels<L1, L2, L3>;
  constexpr int NTYS = LV::NL;
  static_assert((L3 == 0 || NTX >= 3) && (L2 == 0 || NTX >= 2), "a level cannot need more row tiles than there are");
  constexpr int LDX = padded_ld(NTX);
  constexpr int LDY = 4 * NTYS * 16 + 16;
  constexpr int NT = LV::count(ROLE), LV0 = LV::first(ROLE);
  constexpr int NS = LV::slots(ROLE) > 0 ? LV::slots(ROLE) : 1;
  static_assert(LV::slots(ROLE) <= F1V_SLOTS, "K-split partial layout");
  constexpr int NP = QP * (QP + 1) / 2;                  // pairs q <= q' of the c^{qq'} K V groups
  constexpr int NPAIR = NTX / 2, NSING = NTX % 2;        // 16-byte loads (two column tiles each), one 8-byte load for an odd last tile
  const int s = subdomain_of(t, blockIdx.x), tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6), e = wave & 3;
  const int N = a.N, S = a.S, QN = QP * N;
  const int ksplit = gridDim.z;
  const int nchunks = t.nT / EC / ksplit;
  const int T0 = blockIdx.z * nchunks * EC;
  // LDS column of local column c of group g
  auto pos = [&](int g, int c) { return grp[g].cb[c >> 4] + (c & 15); };
  // the basis column a lane holds in column tile ct of an apply operand: tile 2p: 32 p + 2 li, tile 2p + 1: 32 p + 2 li + 1 (one
  // 16-byte load feeds both), an odd last tile: 16 ct + li; beyond N: a duplicate of a column < N (clamped loads)
  auto colc = [&](int ct) {
    return ct < 2 * NPAIR ? (32 * (ct >> 1) + 2 * li + 1 < N ? 32 * (ct >> 1) + 2 * li : N - 2) + (ct & 1) : (16 * ct + li < N ? 16 * ct + li : N - 1);
  };
  const double* Vs = (const double*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(a.V + (long)s * t.n * N) >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)(a.V + (long)s * t.n * N)));

  d4 acc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};

  auto mfma_phase = [&](int c) {
    const double* Xb = Xs + (c & 1) * 3 * EC * LDX;
    const double* Yb = Ys + (c & 1) * 3 * EC * LDY;
#pragma unroll
    for (int kk = 0; kk < 3 * EC; kk += 4) {
      double av[NTX];
#pragma unroll
      for (int i = 0; i < NTX; ++i) av[i] = Xb[(kk + lk) * LDX + i * 16 + li];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const double bv = Yb[(kk + lk) * LDY + (4 * (LV0 + jt) + e) * 16 + li];
#pragma unroll
        for (int i = 0; i < LV::cls(LV0 + jt); ++i)
          acc[LV::slots_before(ROLE, jt) + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv, acc[LV::slots_before(ROLE, jt) + i], 0, 0, 0);
      }
    }
  };
  auto tie = [](double& v) { asm volatile("" : "+v"(v)); };
  auto tie2 = [](d2& v) { asm volatile("" : "+v"(v)); };
  double rhs_part[NTX];
#pragma unroll
  for (int ct = 0; ct < NTX; ++ct) rhs_part[ct] = 0.0;
  constexpr int R_GRP = 3 * (QP + 2);                   // rows 0 .. R_GRP - 1: A_q V, P V, M V  (column groups 0 .. Q + 1)
  constexpr int RRK = R_GRP / 4;                        // accumulator register that holds the three K V rows
  constexpr int R_B = 3 * (QP + 3), RRB = R_B / 4, KQB = R_B % 4;      // row of b . V
  static_assert((R_GRP + 2) / 4 == RRK && R_B < 16, "K rows in one accumulator register");

  if constexpr (ROLE == 0) {
    // ------------------------------------------------------------- role A: the stacked apply, X rows
    struct Set {
      double A[3];
      d2 Bp[3][NPAIR > 0 ? NPAIR : 1];
      double Bs[3];
    };
    const int r16 = li, kq = lk;
    const double* ap[3];                               // per-lane source of A[r16][4 ks + kq] for the wave's first element
    unsigned ainc = 0;                                 // its advance per chunk (bytes): the same for the three k-steps of a lane (lanes that
                                                       // must read zeros walk through a zero table with the stride of their row)
    int bbk[3];
    unsigned lcp[3][NPAIR > 0 ? NPAIR : 1], lcs[3];    // lane constants of the B operand offsets (bytes)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int k = 4 * ks + kq, bb = k / 3, cc = k - 3 * bb;
      bbk[ks] = bb;
      const double* src = t.zero64;
      unsigned stride = r16 < R_B ? 9 : r16 == R_B ? 3 : 0;
      if (r16 < 3 * (QP + 1)) {
        const int g = r16 / 3, i = r16 % 3;
        src = (g < QP ? a.A_diag + ((long)g * S + s) * t.nT * 36 : a.P_diag + (long)s * t.nT * 36) + bb * 9 + i * 3 + cc;
        stride = 36;
      } else if (bb == 0 && r16 < R_GRP) {
        src = t.mass9 + (r16 - 3 * (QP + 1)) * 3 + cc;
        stride = 9;
      } else if (bb == 0 && r16 < R_B) {
        src = t.stiff + (r16 - R_GRP) * 3 + cc;
        stride = 9;
      } else if (bb == 0 && r16 == R_B) {
        src = a.b + (long)s * t.n + cc;
        stride = 3;
      }
      ap[ks] = src + (long)(T0 + e) * stride;
      ainc = 8u * EC * stride;
#pragma unroll
      for (int pp = 0; pp < NPAIR; ++pp) {
        const int col = 32 * pp + 2 * li + 1 < N ? 32 * pp + 2 * li : N - 2;
        lcp[ks][pp] = 8u * (unsigned)(cc * N + col);
      }
      const int cs = 16 * (NTX - 1) + li < N ? 16 * (NTX - 1) + li : N - 1;
      lcs[ks] = 8u * (unsigned)(cc * N + cs);
    }
    const unsigned rs3 = 24u * (unsigned)N;              // bytes of the three rows of an element
    // LDS offsets (doubles) of the apply's outputs: lane (kq, r16) holds row kq + 4 rr, basis column colc(ct) of tile ct.
    // A lane whose natural column lies beyond N holds a duplicate of a column < N (the clamped loads above), hence the SAME
    // apply results as the lane that owns that column: it stores them to the same LDS address (a benign same-value write), and
    // the X / Y columns >= N keep their initial zeros.  Only lanes without a row of the kind being stored need a dump slot.
    int xoff[NTX], yoff[4][NTX], koff[NP][NTX];
    const int dump = (3 * e) * LDY + 4 * NTYS * 16 + r16;
    const int rk = kq + 4 * RRK - R_GRP;                 // 0 .. 2 on the lanes that hold a K V row
    const bool krow = rk >= 0 && rk < 3;
#pragma unroll
    for (int ct = 0; ct < NTX; ++ct) {
      const int col = colc(ct);
      xoff[ct] = (3 * e + (kq < 3 ? kq : 0)) * LDX + col;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = kq + 4 * rr;
        yoff[rr][ct] = r < R_GRP ? (3 * e + r % 3) * LDY + pos(r / 3, col) : dump;
      }
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) koff[pr][ct] = (3 * e + (krow ? rk : 0)) * LDY + pos(QP + 2 + pr, col);
    }
    const cint_p nbc = (cint_p)t.nb_elem;
    struct Sc {
      int nb[3];
      double cc[NP];
    };
    auto load_sc = [&](int T, Sc& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) x.nb[f] = nbc[T * 3 + f];
      int pr = 0;
#pragma unroll
      for (int q = 0; q < QP; ++q)
#pragma unroll
        for (int q2 = q; q2 < QP; ++q2) x.cc[pr++] = ((cdbl_p)(a.caa + ((long)(q * QP + q2) * S + s) * t.nT))[T];
    };
    auto load_set = [&](int T, const Sc& sc, Set& x) {
      // a face without an in-subdomain neighbour has an all-zero block: its rows may be any finite values (the element's own)
      unsigned eo[4];
      eo[0] = (unsigned)T * rs3;
#pragma unroll
      for (int f = 0; f < 3; ++f) eo[1 + f] = (unsigned)(sc.nb[f] >= 0 ? sc.nb[f] : T) * rs3;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) x.A[ks] = gload_f64(ap[ks]);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const unsigned eoff = bbk[ks] == 0 ? eo[0] : bbk[ks] == 1 ? eo[1] : bbk[ks] == 2 ? eo[2] : eo[3];
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) x.Bp[ks][pp] = gload_s128(Vs, eoff + lcp[ks][pp]);
        if (NSING) x.Bs[ks] = gload_s64(Vs, eoff + lcs[ks]);
      }
    };
    auto tie_set = [&](Set& x) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        tie(x.A[ks]);
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) tie2(x.Bp[ks][pp]);
        if (NSING) tie(x.Bs[ks]);
      }
    };
    auto bop = [&](const Set& x, int ks, int ct) { return ct < 2 * NPAIR ? x.Bp[ks][ct >> 1][ct & 1] : x.Bs[ks]; };
    // ONE register set: stage c waits for it, consumes it (apply MFMAs, stores) and then requests the rows of chunk c + 1 into
    // the same registers -- they land during the MFMA phase in between (interleave64: a burst of loads is cheap, and the second
    // set of the ping-pong form cost 24 VGPRs that this role does not have).  The element's scalars (neighbours for the address
    // arithmetic of the loads, c^{qq'}) are requested at the START of the previous stage: sc_nxt is read at its end and in the next.
    auto stage = [&](int c, Set& cur, const Sc& sc_cur, Sc& sc_nxt) {
      const int T = T0 + c * EC + e;               // wave-uniform element
      const bool more = c + 1 < nchunks;
      double* Xb = Xs + (c & 1) * 3 * EC * LDX;
      double* Yb = Ys + (c & 1) * 3 * EC * LDY;
      if (e == 0) F1_STAMP(0, c, 0);
      load_sc(more ? T + EC : T, sc_nxt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tie_set(cur);
      if (e == 0) F1_STAMP(0, c, 1);
      d4 D[NTX];
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) D[ct] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A[ks], bop(cur, ks, ct), D[ct], 0, 0, 0);
      // X rows: the own rows sit in the B operand of k-step 0 (lanes kq < 3 hold row kq)
      if (kq < 3) {
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Xb[xoff[ct]] = bop(cur, 0, ct);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (4 * rr < R_GRP) {                      // compile time: this register holds rows of the column groups 0 .. Q + 1
#pragma unroll
          for (int ct = 0; ct < NTX; ++ct) Yb[yoff[rr][ct]] = D[ct][rr];
        }
      }
      if (krow) {                                  // lane constant: one exec region for all c^{qq'} K V stores
#pragma unroll
        for (int pr = 0; pr < NP; ++pr)
#pragma unroll
          for (int ct = 0; ct < NTX; ++ct) Yb[koff[pr][ct]] = sc_cur.cc[pr] * D[ct][RRK];
      }
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) rhs_part[ct] += D[ct][RRB];      // meaningful on the lanes kq == KQB
      if (e == 0) F1_STAMP(0, c, 2);
      if (more) {                                  // wave-uniform; the last stage requests nothing
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) ap[ks] = (const double*)((const char*)ap[ks] + ainc);
        load_set(T + EC, sc_nxt, cur);
      }
      if (e == 0) F1_STAMP(0, c, 3);
    };
    Set s0;
    Sc c0, c1;
    load_sc(T0 + e, c0);
    load_set(T0 + e, c0, s0);
    auto round = [&](int c, Sc& x0, Sc& x1) {
      stage(c, s0, x0, x1);
      lds_barrier();
      if (e == 0) F1_STAMP(0, c, 4);
      mfma_phase(c);
      if (e == 0) F1_STAMP(0, c, 5);
    };
    for (int c = 0; c < nchunks; c += 2) {                   // nT is a multiple of 8, so nchunks is even
      round(c, c0, c1);
      round(c + 1, c1, c0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_set(s0);
    if (a.rhs_red != nullptr && kq == KQB) {
      // every basis column once: the lane that owns it naturally (lanes beyond N hold duplicates)
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) {
        const int nat = ct < 2 * NPAIR ? 32 * (ct >> 1) + 2 * li + (ct & 1) : 16 * ct + li;
        if (nat < N) red[e * 64 + nat] = rhs_part[ct];
      }
    }
  } else {
    // ------------------------------------------------------------- role B: the A_ab R groups, on the matrix pipe as well
    // Y^{q,q2}[i][col] = sum_f A_ab^q[i][f] R[rt_f][q2 N + col]: one MFMA per (q2, column tile) with the 3 Q rows (q, i) of the
    // element's A_ab blocks as the A operand (k = face f, the fourth k reads zeros) and the three flux rows as the B operand
    // (lane: column, k: face).  (The VALU form of k_f1u -- 18 broadcasts = 36 v_readlane, 36 f64 FMAs per element -- takes as
    // long as the Q NTX MFMAs, tools/f1_trace.py; this form needs fewer registers.)
    struct Set {
      double A;
      d2 Bp[QP][NPAIR > 0 ? NPAIR : 1];
      double Bs[QP];
    };
    const double* Rs = (const double*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(a.Rself + (long)s * t.nrt * QN) >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)(a.Rself + (long)s * t.nrt * QN)));
    const int r16 = li, kq = lk;
    const double* ap = t.zero64;                        // lanes without an entry walk through the zero table with the same stride
    if (r16 < 3 * QP && kq < 3) ap = a.Aab + ((long)(r16 / 3) * S + s) * t.nT * 9 + (r16 % 3) * 3 + kq;
    ap += (long)(T0 + e) * 9;
    unsigned lcp[QP][NPAIR > 0 ? NPAIR : 1], lcs[QP];  // lane constants of the B operand offsets (bytes)
    int yo0[QP][NTX], yo1[QP][NTX];                    // LDS offsets (doubles) of accumulator registers 0 / 1 (rows kq, 4 + kq)
    const bool v0 = kq < 3 * QP, v1 = 4 + kq < 3 * QP;
#pragma unroll
    for (int q2 = 0; q2 < QP; ++q2) {
#pragma unroll
      for (int pp = 0; pp < NPAIR; ++pp) lcp[q2][pp] = 8u * (unsigned)(q2 * N + (32 * pp + 2 * li + 1 < N ? 32 * pp + 2 * li : N - 2));
      lcs[q2] = 8u * (unsigned)(q2 * N + (16 * (NTX - 1) + li < N ? 16 * (NTX - 1) + li : N - 1));
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) {
        const int col = colc(ct);
        const int ra = v0 ? kq : 0, rb = v1 ? 4 + kq : 0;
        yo0[q2][ct] = (3 * e + ra % 3) * LDY + pos(QP + 2 + NP + (ra / 3) * QP + q2, col);
        yo1[q2][ct] = (3 * e + rb % 3) * LDY + pos(QP + 2 + NP + (rb / 3) * QP + q2, col);
      }
    }
    const unsigned rsr = 8u * (unsigned)QN;
    const cint_p rtc = (cint_p)t.elem_rt;
    struct Sc {
      int rt[3];
    };
    auto load_sc = [&](int T, Sc& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) x.rt[f] = rtc[T * 3 + f];
    };
    auto load_set = [&](const Sc& sc, Set& x) {
      const unsigned ro = (unsigned)(kq == 1 ? sc.rt[1] : kq == 2 ? sc.rt[2] : sc.rt[0]) * rsr;      // k = 3 meets zeros of A: any finite row
      x.A = gload_f64(ap);
#pragma unroll
      for (int q2 = 0; q2 < QP; ++q2) {
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) x.Bp[q2][pp] = gload_s128(Rs, ro + lcp[q2][pp]);
        if (NSING) x.Bs[q2] = gload_s64(Rs, ro + lcs[q2]);
      }
    };
    auto tie_set = [&](Set& x) {
      tie(x.A);
#pragma unroll
      for (int q2 = 0; q2 < QP; ++q2) {
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) tie2(x.Bp[q2][pp]);
        if (NSING) tie(x.Bs[q2]);
      }
    };
    auto bop = [&](const Set& x, int q2, int ct) { return ct < 2 * NPAIR ? x.Bp[q2][ct >> 1][ct & 1] : x.Bs[q2]; };
    auto stage = [&](int c, Set& cur, Sc& sc_nxt) {
      const int T = T0 + c * EC + e;
      const bool more = c + 1 < nchunks;
      double* Yb = Ys + (c & 1) * 3 * EC * LDY;
      if (e == 0) F1_STAMP(1, c, 0);
      load_sc(more ? T + EC : T, sc_nxt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tie_set(cur);
      if (e == 0) F1_STAMP(1, c, 1);
#pragma unroll
      for (int q2 = 0; q2 < QP; ++q2) {
        d4 D[NTX];
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct)
          D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A, bop(cur, q2, ct), (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        if (v0) {
#pragma unroll
          for (int ct = 0; ct < NTX; ++ct) Yb[yo0[q2][ct]] = D[ct][0];
        }
        if (3 * QP > 4 && v1) {
#pragma unroll
          for (int ct = 0; ct < NTX; ++ct) Yb[yo1[q2][ct]] = D[ct][1];
        }
      }
      if (e == 0) F1_STAMP(1, c, 2);
      if (more) {
        ap += 9 * EC;
        load_set(sc_nxt, cur);
      }
      if (e == 0) F1_STAMP(1, c, 3);
    };
    Set s0;
    Sc c0;
    load_sc(T0 + e, c0);
    load_set(c0, s0);
    for (int c = 0; c < nchunks; ++c) {
      stage(c, s0, c0);
      lds_barrier();
      if (e == 0) F1_STAMP(1, c, 4);
      mfma_phase(c);
      if (e == 0) F1_STAMP(1, c, 5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_set(s0);
  }
  // ---- K-split (see f1u_body: partial tiles written through, the last arriver sums them in the fixed order of the parts)
  if (ksplit > 1) {
    constexpr int PW = F1V_SLOTS * 4 * 64;             // doubles per wave: [slot][r / 2][lane][r % 2]
    const long wg = 8L * PW + 64;                      // + the partial rhs_red
    double* mine = a.part + ((long)s * ksplit + blockIdx.z) * wg;
    double* pw = mine + (long)wave * PW + 2 * lane;
#pragma unroll
    for (int i = 0; i < LV::slots(ROLE); ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) store_sc1_b128(pw + (i * 2 + h) * 128, acc[i][2 * h], acc[i][2 * h + 1]);
    __syncthreads();                                   // red[] of the role-A waves is complete
    if (a.rhs_red != nullptr && tid < N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      store_sc1_b64(mine + 8L * PW + tid, sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier in front of the arrival
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.ticket + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != ksplit - 1) return;                   // workgroup-uniform
    if (tid == 0) __hip_atomic_store(a.ticket + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    const double* all = a.part + (long)s * ksplit * wg;
#pragma unroll
    for (int i = 0; i < LV::slots(ROLE); ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int z = 0; z < ksplit; ++z) {
      const double* pz = all + z * wg + (long)wave * PW + 2 * lane;
#pragma unroll
      for (int i = 0; i < LV::slots(ROLE); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] += load_sc1_b64(pz + (i * 2 + r / 2) * 128 + r % 2);
    }
    if (a.rhs_red != nullptr && tid < N) {
      double sum = 0.0;
      for (int z = 0; z < ksplit; ++z) sum += load_sc1_b64(all + z * wg + 8L * PW + tid);
      a.rhs_red[(long)s * N + tid] = sum;
    }
  }
  // ---- epilogue: scatter the tiles to their destination arrays (static accumulator indices only).  colmap: LDS column ->
  // (group << 8 | local column) or -1.  Symmetric groups: a tile delivers the entries row <= column and their mirror images.
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int m = colmap[(4 * (LV0 + jt) + e) * 16 + li];
    const bool live = m >= 0;
    const int g = live ? m >> 8 : 0, jj = m & 255;
    const int ld = grp[g].ld;
    const bool sym = grp[g].sym != 0;
    double* base = grp[g].dst + (long)s * grp[g].sstride;
    double* base_t = grp[g].dst_t ? grp[g].dst_t + (long)s * grp[g].sstride : nullptr;
#pragma unroll
    for (int i = 0; i < LV::cls(LV0 + jt); ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i * 16 + lk + 4 * r;
        const double val = acc[LV::slots_before(ROLE, jt) + i][r];
        if (live && row < N && (!sym || row <= jj)) {
#ifndef F1V_NO_STORE
          base[(long)row * ld + jj] = val;
#ifndef F1V_NO_MIRROR
          if (sym && row < jj) base[(long)jj * ld + row] = val;
          if (base_t) base_t[(long)jj * ld + row] = val;
#endif
#else
          if (val == 1.2345e300) base[(long)row * ld + jj] = val;
#endif
        }
      }
    }
  }
}

template <int NTX, int QP, int L1, int L2, int L3>
__global__ __launch_bounds__(512, 2) void k_f1v(Tmpl t, F1Args a, GrpTable gt) {
  constexpr int NTYS = L1 + L2 + L3;
  constexpr int LDX = padded_ld(NTX);
  constexpr int LDY = 4 * NTYS * 16 + 16;
  __shared__ double Xs[2 * 3 * EC * LDX];
  __shared__ double Ys[2 * 3 * EC * LDY];
  __shared__ double red[EC * 64];
  __shared__ int colmap[4 * NTYS * 16];
  __shared__ Grp grp[F1_MAXG];
  __shared__ int flag;
  const int tid = threadIdx.x;
#pragma unroll
  for (int g = 0; g < F1_MAXG; ++g)
    if (tid == g) grp[g] = gt.g[g];
  for (int i = tid; i < 2 * 3 * EC * LDX; i += 512) Xs[i] = 0.0;
  for (int i = tid; i < 2 * 3 * EC * LDY; i += 512) Ys[i] = 0.0;
  for (int i = tid; i < 4 * NTYS * 16; i += 512) colmap[i] = -1;
  if (tid < EC * 64) red[tid] = 0.0;
  __syncthreads();
  for (int i = tid; i < gt.n * a.N; i += 512) {
    const int g = i / a.N, c = i - g * a.N;
    colmap[grp[g].cb[c >> 4] + (c & 15)] = (g << 8) | c;
  }
  __syncthreads();
  if (uniform(tid >> 6) < EC)
    f1v_body<NTX, QP, L1, L2, L3, 0>(t, a, Xs, Ys, red, &flag, colmap, grp);
  else
    f1v_body<NTX, QP, L1, L2, L3, 1>(t, a, Xs, Ys, red, &flag, colmap, grp);
  if (gridDim.z == 1 && a.rhs_red != nullptr) {   // fixed-order sum over the EC role-A waves (K-split: done in f1v_body)
    __syncthreads();
    if (tid < a.N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      a.rhs_red[(long)subdomain_of(t, blockIdx.x) * a.N + tid] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// F1, rank-2 form (k_f1w; round 4): Q = 2, three row tiles, even N in [34, 40] -- the shape of BASELINE config 3.
// Every element block of df_aa (c_T^{qq'} K_T) and of df_ab (A_ab,T^q) acts through the TWO gradient directions of the P1 shape
// functions: with kappa = L L^T and G_T = [g_0 g_1 g_2],  K_T = Z_T^T Z_T (Z_T = L^T G_T, 2 x 3)  and  A_ab,T = G_T^T W_T, so
//   G_aa^{qq'} = sum_T (Z_T V_T)^T (c_T^{qq'} Z_T V_T),      G_ab^q[:, self] = sum_T (Z_T V_T)^T (W'^q_T R_T),   W' = L^-1 W:
// seven of the eleven column groups (three G_aa pairs, four G_ab) have TWO K rows per element instead of three when the X operand
// is Z V (two rows) instead of V.  A chunk of four elements is then 3 k-steps for the tiles of B_sys / E_red / M_red and 2 k-steps
// for the others: 150 instead of 204 projection MFMAs per chunk.  What changes against k_f1v:
//   * role A's stacked apply carries the two rows of Z_T (template table t.lgz) where it carried the three rows of K_T: Z V leaves
//     the matrix pipe with the system rows, is stored once as the second X operand (Zs) and, times c^{qq'}, as the Y rows of the
//     three G_aa groups;
//   * role B multiplies W'^q_T (2 x 3 per component; k_prep_lds forms it from A_ab with the template table t.hab) with the three
//     flux rows: four A-operand rows, one accumulator register;
//   * the column tiles come in two kinds (3 or 2 K rows per element) and three classes (row tiles needed); which SIMD owns which
//     is a compile-time plan (F1wPlan) that balances the MFMA count: 38 / 38 / 37 / 37 per chunk and SIMD.
struct F1wPlan {
  static constexpr int NL = 7, LA = 2;                 // levels (four column tiles each: tile 4 l + e on SIMD e); role A takes the first LA
  // v = e >> 1 (SIMDs 0, 1 / 2, 3).  level 0: block 0 of the four K3 symmetric groups; 1: their block 1; 2: v0 their packed tails,
  // v1 block 1 of G_aa[0][0] / [1][1]; 3: v0 block 0 of those, v1 K2 tiles of class 3; 4 .. 6: K2 tiles of class 3
  static constexpr int cls(int l, int v) { return l == 0 ? 1 : l == 1 ? 2 : l == 2 ? (v == 0 ? 3 : 2) : l == 3 ? (v == 0 ? 1 : 3) : 3; }
  static constexpr int kr(int l, int v) { return l < 2 ? 3 : l == 2 ? (v == 0 ? 3 : 2) : 2; }      // K rows per element
  static constexpr int first(int role) { return role == 0 ? 0 : LA; }
  static constexpr int count(int role) { return role == 0 ? LA : NL - LA; }
  static constexpr int slots_before(int role, int v, int jt) {
    int n = 0;
    for (int k = 0; k < jt; ++k) n += cls(first(role) + k, v);
    return n;
  }
  static constexpr int slots(int role, int v) { return slots_before(role, v, count(role)); }
  static constexpr int max_slots(int role) { return slots(role, 0) > slots(role, 1) ? slots(role, 0) : slots(role, 1); }
};
template <int V> struct F1wV { static constexpr int value = V; };

template <int ROLE>
__device__ __forceinline__ void f1w_body(const Tmpl& t, const F1Args& a, const double* __restrict__ Wab, double* __restrict__ Xs,
                                         double* __restrict__ Zs, double* __restrict__ Ys, double* __restrict__ red,
                                         int* __restrict__ flag, const int* __restrict__ colmap, const Grp* grp) {
  using LV = F1wPlan;
  constexpr int NTX = 3, QP = 2, NTYS = LV::NL;
  constexpr int LDX = padded_ld(NTX);
  constexpr int LDY = 4 * NTYS * 16 + 16;
  constexpr int NT = LV::count(ROLE), LV0 = LV::first(ROLE);
  constexpr int NS = LV::max_slots(ROLE);
  static_assert(NS <= F1V_SLOTS, "K-split partial layout");
  constexpr int NP = QP * (QP + 1) / 2;
  constexpr int NPAIR = NTX / 2, NSING = NTX % 2;
  const int s = subdomain_of(t, blockIdx.x), tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6), e = wave & 3;
  const int N = a.N, S = a.S, QN = QP * N;
  const int ksplit = gridDim.z;
  const int nchunks = t.nT / EC / ksplit;
  const int T0 = blockIdx.z * nchunks * EC;
  auto pos = [&](int g, int c) { return grp[g].cb[c >> 4] + (c & 15); };
  auto colc = [&](int ct) {
    return ct < 2 * NPAIR ? (32 * (ct >> 1) + 2 * li + 1 < N ? 32 * (ct >> 1) + 2 * li : N - 2) + (ct & 1) : (16 * ct + li < N ? 16 * ct + li : N - 1);
  };
  const double* Vs = (const double*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(a.V + (long)s * t.n * N) >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)(a.V + (long)s * t.n * N)));

  d4 acc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};

  // the projection of one chunk: 3 k-steps over the V rows for the tiles of kind 3, 2 k-steps over the Z rows for the tiles of kind 2
  auto mfma_phase = [&](int c, auto vtag) {
    constexpr int V = decltype(vtag)::value;
    const double* Xb = Xs + (c & 1) * 3 * EC * LDX;
    const double* Zb = Zs + (c & 1) * 2 * EC * LDX;
    const double* Yb = Ys + (c & 1) * 3 * EC * LDY;
    constexpr bool any3 = LV::kr(LV0, V) == 3 || LV::kr(LV0 + (NT > 1 ? 1 : 0), V) == 3;      // (kind-3 tiles are the first of a role's list)
    if constexpr (any3) {
#pragma unroll
      for (int kk = 0; kk < 3 * EC; kk += 4) {
        double av[NTX];
#pragma unroll
        for (int i = 0; i < NTX; ++i) av[i] = Xb[(kk + lk) * LDX + i * 16 + li];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          if (LV::kr(LV0 + jt, V) == 3) {      // (compile time once the loop is unrolled)
            const double bv = Yb[(kk + lk) * LDY + (4 * (LV0 + jt) + e) * 16 + li];
#pragma unroll
            for (int i = 0; i < LV::cls(LV0 + jt, V); ++i)
              acc[LV::slots_before(ROLE, V, jt) + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv, acc[LV::slots_before(ROLE, V, jt) + i], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int kk = 0; kk < 2 * EC; kk += 4) {
        double av[NTX];
#pragma unroll
        for (int i = 0; i < NTX; ++i) av[i] = Zb[(kk + lk) * LDX + i * 16 + li];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          if (LV::kr(LV0 + jt, V) == 2) {
            const double bv = Yb[(kk + lk) * LDY + (4 * (LV0 + jt) + e) * 16 + li];
#pragma unroll
            for (int i = 0; i < LV::cls(LV0 + jt, V); ++i)
              acc[LV::slots_before(ROLE, V, jt) + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv, acc[LV::slots_before(ROLE, V, jt) + i], 0, 0, 0);
          }
        }
      }
    }
  };
  auto tie = [](double& v) { asm volatile("" : "+v"(v)); };
  auto tie2 = [](d2& v) { asm volatile("" : "+v"(v)); };
  double rhs_part[NTX];
#pragma unroll
  for (int ct = 0; ct < NTX; ++ct) rhs_part[ct] = 0.0;
  // rows of role A's stacked apply: 0 .. 8 A_0, A_1, P; 9 .. 11 M; 12, 13 Z; 14 b; 15 zero
  constexpr int R_GRP = 3 * (QP + 2);                   // rows 0 .. R_GRP - 1: A_q V, P V, M V  (column groups 0 .. Q + 1)
  constexpr int R_B = R_GRP + 2, RRZ = R_GRP / 4, KQB = R_B % 4;
  static_assert(R_GRP % 4 == 0 && R_B / 4 == RRZ && R_B < 16, "the Z rows and b . V in one accumulator register");

  if constexpr (ROLE == 0) {
    struct Set {
      double A[3];
      d2 Bp[3][NPAIR];
      double Bs[3];
    };
    const int r16 = li, kq = lk;
    const double* ap[3];
    unsigned ainc = 0;
    int bbk[3];
    unsigned lcp[3][NPAIR], lcs[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int k = 4 * ks + kq, bb = k / 3, cc = k - 3 * bb;
      bbk[ks] = bb;
      const double* src = t.zero64;
      unsigned stride = r16 < R_GRP ? 9 : r16 < R_B ? 6 : r16 == R_B ? 3 : 0;
      if (r16 < 3 * (QP + 1)) {
        const int g = r16 / 3, i = r16 % 3;
        src = (g < QP ? a.A_diag + ((long)g * S + s) * t.nT * 36 : a.P_diag + (long)s * t.nT * 36) + bb * 9 + i * 3 + cc;
        stride = 36;
      } else if (bb == 0 && r16 < R_GRP) {
        src = t.mass9 + (r16 - 3 * (QP + 1)) * 3 + cc;
      } else if (bb == 0 && r16 < R_B) {
        src = t.lgz + (r16 - R_GRP) * 3 + cc;
      } else if (bb == 0 && r16 == R_B) {
        src = a.b + (long)s * t.n + cc;
      }
      ap[ks] = src + (long)(T0 + e) * stride;
      ainc = 8u * EC * stride;
#pragma unroll
      for (int pp = 0; pp < NPAIR; ++pp) {
        const int col = 32 * pp + 2 * li + 1 < N ? 32 * pp + 2 * li : N - 2;
        lcp[ks][pp] = 8u * (unsigned)(cc * N + col);
      }
      const int cs = 16 * (NTX - 1) + li < N ? 16 * (NTX - 1) + li : N - 1;
      lcs[ks] = 8u * (unsigned)(cc * N + cs);
    }
    const unsigned rs3 = 24u * (unsigned)N;
    // LDS offsets (doubles) of the apply's outputs (see k_f1v: lanes beyond N hold duplicates of a column < N and store the same
    // value to the same address).  Rows 0 .. 11 (registers 0 .. 2): every lane has a row; register 3: lanes kq < 2 hold a Z row
    int xoff[NTX], zoff[NTX], yoff[3][NTX], koff[NP][NTX];
    const bool zrow = kq < 2;
#pragma unroll
    for (int ct = 0; ct < NTX; ++ct) {
      const int col = colc(ct);
      xoff[ct] = (3 * e + (kq < 3 ? kq : 0)) * LDX + col;
      zoff[ct] = (2 * e + (zrow ? kq : 0)) * LDX + col;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const int r = kq + 4 * rr;
        yoff[rr][ct] = (3 * e + r % 3) * LDY + pos(r / 3, col);
      }
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) koff[pr][ct] = (2 * e + (zrow ? kq : 0)) * LDY + pos(QP + 2 + pr, col);
    }
    const cint_p nbc = (cint_p)t.nb_elem;
    struct Sc {
      int nb[3];
      double cc[NP];
    };
    auto load_sc = [&](int T, Sc& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) x.nb[f] = nbc[T * 3 + f];
      int pr = 0;
#pragma unroll
      for (int q = 0; q < QP; ++q)
#pragma unroll
        for (int q2 = q; q2 < QP; ++q2) x.cc[pr++] = ((cdbl_p)(a.caa + ((long)(q * QP + q2) * S + s) * t.nT))[T];
    };
    auto load_set = [&](int T, const Sc& sc, Set& x) {
      unsigned eo[4];
      eo[0] = (unsigned)T * rs3;
#pragma unroll
      for (int f = 0; f < 3; ++f) eo[1 + f] = (unsigned)(sc.nb[f] >= 0 ? sc.nb[f] : T) * rs3;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) x.A[ks] = gload_f64(ap[ks]);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const unsigned eoff = bbk[ks] == 0 ? eo[0] : bbk[ks] == 1 ? eo[1] : bbk[ks] == 2 ? eo[2] : eo[3];
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) x.Bp[ks][pp] = gload_s128(Vs, eoff + lcp[ks][pp]);
        if (NSING) x.Bs[ks] = gload_s64(Vs, eoff + lcs[ks]);
      }
    };
    auto tie_set = [&](Set& x) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        tie(x.A[ks]);
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) tie2(x.Bp[ks][pp]);
        if (NSING) tie(x.Bs[ks]);
      }
    };
    auto bop = [&](const Set& x, int ks, int ct) { return ct < 2 * NPAIR ? x.Bp[ks][ct >> 1][ct & 1] : x.Bs[ks]; };
    auto stage = [&](int c, Set& cur, const Sc& sc_cur, Sc& sc_nxt) {
      const int T = T0 + c * EC + e;
      const bool more = c + 1 < nchunks;
      double* Xb = Xs + (c & 1) * 3 * EC * LDX;
      double* Zb = Zs + (c & 1) * 2 * EC * LDX;
      double* Yb = Ys + (c & 1) * 3 * EC * LDY;
      load_sc(more ? T + EC : T, sc_nxt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tie_set(cur);
      d4 D[NTX];
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) D[ct] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A[ks], bop(cur, ks, ct), D[ct], 0, 0, 0);
      if (kq < 3) {
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Xb[xoff[ct]] = bop(cur, 0, ct);
      }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Yb[yoff[rr][ct]] = D[ct][rr];
      if (zrow) {                                  // lane constant: one exec region for the Z rows and the c^{qq'} Z V stores
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Zb[zoff[ct]] = D[ct][RRZ];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr)
#pragma unroll
          for (int ct = 0; ct < NTX; ++ct) Yb[koff[pr][ct]] = sc_cur.cc[pr] * D[ct][RRZ];
      }
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) rhs_part[ct] += D[ct][RRZ];      // meaningful on the lanes kq == KQB
      if (more) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) ap[ks] = (const double*)((const char*)ap[ks] + ainc);
        load_set(T + EC, sc_nxt, cur);
      }
    };
    Set s0;
    Sc c0, c1;
    load_sc(T0 + e, c0);
    load_set(T0 + e, c0, s0);
    auto round = [&](int c, Sc& x0, Sc& x1) {
      stage(c, s0, x0, x1);
      lds_barrier();
      mfma_phase(c, F1wV<0>{});                  // (role A's tiles are the same for every SIMD)
    };
    for (int c = 0; c < nchunks; c += 2) {
      round(c, c0, c1);
      round(c + 1, c1, c0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_set(s0);
    if (a.rhs_red != nullptr && kq == KQB) {
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) {
        const int nat = ct < 2 * NPAIR ? 32 * (ct >> 1) + 2 * li + (ct & 1) : 16 * ct + li;
        if (nat < N) red[e * 64 + nat] = rhs_part[ct];
      }
    }
  } else {
    // ------------------------------------------------------------- role B: Y^{q,q2} = W'^q_T R_T^{q2}, two rows per element
    struct Set {
      double A;
      d2 Bp[QP][NPAIR];
      double Bs[QP];
    };
    const double* Rs = (const double*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(a.Rself + (long)s * t.nrt * QN) >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)(a.Rself + (long)s * t.nrt * QN)));
    const int r16 = li, kq = lk;
    const double* ap = t.zero64;                        // lanes without an entry walk through the zero table with the same stride
    if (r16 < 2 * QP && kq < 3) ap = Wab + ((long)(r16 >> 1) * S + s) * t.nT * 6 + (r16 & 1) * 3 + kq;
    ap += (long)(T0 + e) * 6;
    unsigned lcp[QP][NPAIR], lcs[QP];
    int yo[QP][NTX];                                   // LDS offsets of accumulator register 0: row kq = (q, d) = (kq >> 1, kq & 1)
#pragma unroll
    for (int q2 = 0; q2 < QP; ++q2) {
#pragma unroll
      for (int pp = 0; pp < NPAIR; ++pp) lcp[q2][pp] = 8u * (unsigned)(q2 * N + (32 * pp + 2 * li + 1 < N ? 32 * pp + 2 * li : N - 2));
      lcs[q2] = 8u * (unsigned)(q2 * N + (16 * (NTX - 1) + li < N ? 16 * (NTX - 1) + li : N - 1));
#pragma unroll
      for (int ct = 0; ct < NTX; ++ct) yo[q2][ct] = (2 * e + (kq & 1)) * LDY + pos(QP + 2 + NP + (kq >> 1) * QP + q2, colc(ct));
    }
    const unsigned rsr = 8u * (unsigned)QN;
    const cint_p rtc = (cint_p)t.elem_rt;
    struct Sc {
      int rt[3];
    };
    auto load_sc = [&](int T, Sc& x) {
#pragma unroll
      for (int f = 0; f < 3; ++f) x.rt[f] = rtc[T * 3 + f];
    };
    auto load_set = [&](const Sc& sc, Set& x) {
      const unsigned ro = (unsigned)(kq == 1 ? sc.rt[1] : kq == 2 ? sc.rt[2] : sc.rt[0]) * rsr;
      x.A = gload_f64(ap);
#pragma unroll
      for (int q2 = 0; q2 < QP; ++q2) {
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) x.Bp[q2][pp] = gload_s128(Rs, ro + lcp[q2][pp]);
        if (NSING) x.Bs[q2] = gload_s64(Rs, ro + lcs[q2]);
      }
    };
    auto tie_set = [&](Set& x) {
      tie(x.A);
#pragma unroll
      for (int q2 = 0; q2 < QP; ++q2) {
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) tie2(x.Bp[q2][pp]);
        if (NSING) tie(x.Bs[q2]);
      }
    };
    auto bop = [&](const Set& x, int q2, int ct) { return ct < 2 * NPAIR ? x.Bp[q2][ct >> 1][ct & 1] : x.Bs[q2]; };
    auto stage = [&](int c, Set& cur, Sc& sc_nxt) {
      const int T = T0 + c * EC + e;
      const bool more = c + 1 < nchunks;
      double* Yb = Ys + (c & 1) * 3 * EC * LDY;
      load_sc(more ? T + EC : T, sc_nxt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tie_set(cur);
#pragma unroll
      for (int q2 = 0; q2 < QP; ++q2) {
        d4 D[NTX];
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct)
          D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A, bop(cur, q2, ct), (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < NTX; ++ct) Yb[yo[q2][ct]] = D[ct][0];
      }
      if (more) {
        ap += 6 * EC;
        load_set(sc_nxt, cur);
      }
    };
    // One staging loop per tile plan (SIMDs 0, 1 / 2, 3), each with its OWN prologue behind the wave-uniform branch: with the
    // prologue's asm-managed loads live across the branch, hipcc moved the set's registers on the out-of-line side before the first
    // vmcnt(0), and chunk 0 of SIMDs 2, 3 multiplied what the registers held before the loads had landed (their elements'
    // contributions to G_ab were missing; the ISA guard walks the loops, not the prologue).  (Both plans in ONE loop: 101 spilled VGPRs.)
    auto run = [&](auto vtag) {
      Set s0;
      Sc c0;
      load_sc(T0 + e, c0);
      load_set(c0, s0);
      for (int c = 0; c < nchunks; ++c) {
        stage(c, s0, c0);
        lds_barrier();
        mfma_phase(c, vtag);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tie_set(s0);
    };
    if (e < 2) run(F1wV<0>{});
    else run(F1wV<1>{});
  }
  // ---- K-split (see f1u_body: partial tiles written through, the last arriver sums them in the fixed order of the parts)
  if (ksplit > 1) {
    constexpr int PW = F1V_SLOTS * 4 * 64;
    const long wg = 8L * PW + 64;
    double* mine = a.part + ((long)s * ksplit + blockIdx.z) * wg;
    double* pw = mine + (long)wave * PW + 2 * lane;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) store_sc1_b128(pw + (i * 2 + h) * 128, acc[i][2 * h], acc[i][2 * h + 1]);
    __syncthreads();
    if (a.rhs_red != nullptr && tid < N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      store_sc1_b64(mine + 8L * PW + tid, sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.ticket + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != ksplit - 1) return;
    if (tid == 0) __hip_atomic_store(a.ticket + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double* all = a.part + (long)s * ksplit * wg;
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int z = 0; z < ksplit; ++z) {
      const double* pz = all + z * wg + (long)wave * PW + 2 * lane;
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] += load_sc1_b64(pz + (i * 2 + r / 2) * 128 + r % 2);
    }
    if (a.rhs_red != nullptr && tid < N) {
      double sum = 0.0;
      for (int z = 0; z < ksplit; ++z) sum += load_sc1_b64(all + z * wg + 8L * PW + tid);
      a.rhs_red[(long)s * N + tid] = sum;
    }
  }
  // ---- epilogue (as k_f1v): scatter the tiles through the column map; symmetric groups deliver row <= column and the mirror image
  auto epilogue = [&](auto vtag) {
    constexpr int V = decltype(vtag)::value;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int m = colmap[(4 * (LV0 + jt) + e) * 16 + li];
      const bool live = m >= 0;
      const int g = live ? m >> 8 : 0, jj = m & 255;
      const int ld = grp[g].ld;
      const bool sym = grp[g].sym != 0;
      double* base = grp[g].dst + (long)s * grp[g].sstride;
      double* base_t = grp[g].dst_t ? grp[g].dst_t + (long)s * grp[g].sstride : nullptr;
#pragma unroll
      for (int i = 0; i < LV::cls(LV0 + jt, V); ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i * 16 + lk + 4 * r;
          const double val = acc[LV::slots_before(ROLE, V, jt) + i][r];
          if (live && row < N && (!sym || row <= jj)) {
            base[(long)row * ld + jj] = val;
            if (sym && row < jj) base[(long)jj * ld + row] = val;
            if (base_t) base_t[(long)jj * ld + row] = val;
          }
        }
      }
    }
  };
  if (ROLE == 0 || e < 2) epilogue(F1wV<0>{});
  else epilogue(F1wV<1>{});
}

__global__ __launch_bounds__(512, 2) void k_f1w(Tmpl t, F1Args a, GrpTable gt, const double* __restrict__ Wab) {
  constexpr int NTX = 3, NTYS = F1wPlan::NL;
  constexpr int LDX = padded_ld(NTX);
  constexpr int LDY = 4 * NTYS * 16 + 16;
  __shared__ double Xs[2 * 3 * EC * LDX];
  __shared__ double Zs[2 * 2 * EC * LDX];
  __shared__ double Ys[2 * 3 * EC * LDY];
  __shared__ double red[EC * 64];
  __shared__ int colmap[4 * NTYS * 16];
  __shared__ Grp grp[F1_MAXG];
  __shared__ int flag;
  const int tid = threadIdx.x;
#pragma unroll
  for (int g = 0; g < F1_MAXG; ++g)
    if (tid == g) grp[g] = gt.g[g];
  for (int i = tid; i < 2 * 3 * EC * LDX; i += 512) Xs[i] = 0.0;
  for (int i = tid; i < 2 * 2 * EC * LDX; i += 512) Zs[i] = 0.0;
  for (int i = tid; i < 2 * 3 * EC * LDY; i += 512) Ys[i] = 0.0;
  for (int i = tid; i < 4 * NTYS * 16; i += 512) colmap[i] = -1;
  if (tid < EC * 64) red[tid] = 0.0;
  __syncthreads();
  for (int i = tid; i < gt.n * a.N; i += 512) {
    const int g = i / a.N, c = i - g * a.N;
    colmap[grp[g].cb[c >> 4] + (c & 15)] = (g << 8) | c;
  }
  __syncthreads();
  if (uniform(tid >> 6) < EC)
    f1w_body<0>(t, a, Wab, Xs, Zs, Ys, red, &flag, colmap, grp);
  else
    f1w_body<1>(t, a, Wab, Xs, Zs, Ys, red, &flag, colmap, grp);
  if (gridDim.z == 1 && a.rhs_red != nullptr) {
    __syncthreads();
    if (tid < a.N) {
      double sum = 0.0;
      for (int w = 0; w < EC; ++w) sum += red[w * 64 + tid];
      a.rhs_red[(long)subdomain_of(t, blockIdx.x) * a.N + tid] = sum;
    }
  }
}

// Zeroes every output of k_f1 (the destination blocks of its column groups and rhs_red) before a K-split launch.
__global__ __launch_bounds__(256) void k_f1_zero(GrpTable gt, int S, int N, double* __restrict__ rhs_red, const int* __restrict__ sub_list) {
  const int s = sub_list ? sub_list[blockIdx.x] : blockIdx.x;
  for (int g = 0; g < gt.n; ++g) {
    double* dst = gt.g[g].dst + (long)s * gt.g[g].sstride;
    double* dst_t = gt.g[g].dst_t ? gt.g[g].dst_t + (long)s * gt.g[g].sstride : nullptr;
    const int ld = gt.g[g].ld;
    for (int i = threadIdx.x; i < N * N; i += 256) {
      dst[(long)(i / N) * ld + i % N] = 0.0;
      if (dst_t) dst_t[(long)(i / N) * ld + i % N] = 0.0;
    }
  }
  if (rhs_red)
    for (int i = threadIdx.x; i < N; i += 256) rhs_red[(long)s * N + i] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// F2: X = element-gathered flux image R~ (3 face rows per element) and its divergence d (1 row per element).
// One wave per column tile of the (Q N)-wide self block; NR row tiles.
struct F2Args {
  const double *Rself, *Bbb, *b;
  const int* nbr;
  double *G_bb, *G_rdd, *r_fd;
  int Q, N, S;
  long gstride;   // doubles between the self blocks of consecutive subdomains: 9 QN^2 (block-compact) or QN^2 (factored)
};

// Same producer / consumer structure as k_f1: waves 0-3 stage the face rows R~_T, B_T R~_T, d_T, |T| d_T of element
// 4 c + w into LDS buffer c & 1 (global loads of chunk c + 1 in flight), waves 4 .. 7 share the
// upper-triangular tiles of the (Q N)-wide self blocks of G_bb and G_rdd.  One barrier per chunk.
constexpr int F2_NCW = 4;   // consumer waves of k_f2

template <int NR>
__global__ __launch_bounds__(64 * (F2_NCW + EC)) void k_f2(Tmpl t, F2Args a) {
  constexpr int LD = padded_ld(NR);
  extern __shared__ double dyn[];             // per-element scalars cached once: coef [nT][3], bsum [nT], rt [nT][3] (int)
  __shared__ double Xb[2][3 * EC * LD], Yb[2][3 * EC * LD], Xd[2][EC * LD], Yd[2][EC * LD];
  __shared__ double red[EC * 128];
  const int s = subdomain_of(t, blockIdx.x), tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6);
  const int nthreads = 64 * (F2_NCW + EC);
  const int QN = a.Q * a.N, C = 5 * QN;
  double* coefs = dyn;
  double* bsums = dyn + 3 * t.nT;
  int* rts = reinterpret_cast<int*>(dyn + 4 * t.nT);
  const int* nbr_s = a.nbr + s * 5;
  for (int i = tid; i < 3 * t.nT; i += nthreads) {
    const int T = i / 3, f = i - 3 * T;
    coefs[i] = face_sign_at(t, nbr_s, T, f) * t.face_len[i] / t.area[T];
    rts[i] = t.elem_rt[i];
  }
  for (int T = tid; T < t.nT; T += nthreads) {
    const double* be = a.b + (long)s * t.n + 3 * T;
    bsums[T] = be[0] + be[1] + be[2];
  }
  for (int i = tid; i < 2 * 3 * EC * LD; i += nthreads) (&Xb[0][0])[i] = (&Yb[0][0])[i] = 0.0;
  for (int i = tid; i < 2 * EC * LD; i += nthreads) (&Xd[0][0])[i] = (&Yd[0][0])[i] = 0.0;
  for (int i = tid; i < EC * 128; i += nthreads) red[i] = 0.0;
  __syncthreads();
  const int nchunks = t.nT / EC;
  const double* Rs = a.Rself + (long)s * t.nrt * QN;

  if (wave < EC) {
    // ================================================= producers
    // Prefetch set of one element: its 3 face rows of R~ in two column halves (6 loads; idle lanes re-read the last
    // column) and its B_T block, item i in lane i (1 load).  All seven are asm loads completed by an explicit
    // s_waitcnt vmcnt(7) (see gload_f64): everything else the staging needs comes from LDS or the scalar unit, because
    // ANY compiler-visible vector load in this loop ends in a vmcnt(0) that also waits for the prefetch just issued.
    double rv[2][3], nrv[2][3], Bl, nBl;
    const int cc0 = lane < QN ? lane : QN - 1, cc1 = lane + 64 < QN ? lane + 64 : QN - 1;
    const int bi = lane < 9 ? lane : 8;
    auto load = [&](int T, double (&r)[2][3], double& B) {
      const int r0 = rts[3 * T], r1 = rts[3 * T + 1], r2 = rts[3 * T + 2];
      r[0][0] = gload_f64(Rs + (long)r0 * QN + cc0);
      r[0][1] = gload_f64(Rs + (long)r1 * QN + cc0);
      r[0][2] = gload_f64(Rs + (long)r2 * QN + cc0);
      r[1][0] = gload_f64(Rs + (long)r0 * QN + cc1);
      r[1][1] = gload_f64(Rs + (long)r1 * QN + cc1);
      r[1][2] = gload_f64(Rs + (long)r2 * QN + cc1);
      B = gload_f64(a.Bbb + ((long)s * t.nT + T) * 9 + bi);
    };
    auto wait_set = [&](double (&r)[2][3], double& B) {
      asm volatile("s_waitcnt vmcnt(7)"
                   : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]), "+v"(B));
    };
    auto wait_set_tie = [&](double (&r)[2][3], double& B) {
      asm volatile("" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]), "+v"(B));
    };
    load(wave, rv, Bl);
    double rfd_part[2] = {0.0, 0.0};   // columns lane and lane + 64 (QN <= 128)
    // One pipeline step: issue the loads of chunk c + 1 into the OTHER register set, stage chunk c from this one.
    // The two sets alternate (loop unrolled by two, no register copies).
    auto step = [&](int c, double (&cr)[2][3], double& cBl, double (&nr)[2][3], double& nB) {
      const int T = c * EC + wave, el = wave;
      load(c + 1 < nchunks ? T + EC : T, nr, nB);   // unconditional: the wait below counts on exactly 7 younger loads
      wait_set(cr, cBl);
      double cB[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) cB[i] = bcast_d(cBl, i);
      double* xb = &Xb[c & 1][0];
      double* yb = &Yb[c & 1][0];
      double* xd = &Xd[c & 1][0];
      double* yd = &Yd[c & 1][0];
      const double c0f = coefs[3 * T], c1f = coefs[3 * T + 1], c2f = coefs[3 * T + 2];
      const double bsum = bsums[T], area = ((cdbl_p)t.area)[T];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int cc = lane + 64 * k;
#ifdef F2_NO_STAGE
        if (cc < QN && cr[k][0] == 1.2345e300) {
#else
        if (cc < QN) {
#endif
          const double rv0 = cr[k][0], rv1 = cr[k][1], rv2 = cr[k][2];
          xb[(3 * el) * LD + cc] = rv0;
          xb[(3 * el + 1) * LD + cc] = rv1;
          xb[(3 * el + 2) * LD + cc] = rv2;
          yb[(3 * el) * LD + cc] = cB[0] * rv0 + cB[1] * rv1 + cB[2] * rv2;
          yb[(3 * el + 1) * LD + cc] = cB[3] * rv0 + cB[4] * rv1 + cB[5] * rv2;
          yb[(3 * el + 2) * LD + cc] = cB[6] * rv0 + cB[7] * rv1 + cB[8] * rv2;
          const double d = c0f * rv0 + c1f * rv1 + c2f * rv2;
          xd[el * LD + cc] = d;
          yd[el * LD + cc] = area * d;
          rfd_part[k] += bsum * d;
        }
      }
      lds_barrier();                               // barrier c: buffer c & 1 complete (loads stay in flight)
    };
    for (int c = 0; c < nchunks; c += 2) {         // nT is a multiple of 8, so nchunks is even
      step(c, rv, Bl, nrv, nBl);
      step(c + 1, nrv, nBl, rv, Bl);
    }
    lds_barrier();                                 // final barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last prefetch set is never consumed, but it has to stay
    wait_set_tie(rv, Bl);                              // live until its loads have landed (see k_f1)
    wait_set_tie(nrv, nBl);
    red[wave * 128 + lane] = rfd_part[0];
    red[wave * 128 + 64 + lane] = rfd_part[1];
  } else {
    // ================================================= consumers
    // G_bb[self,self] = R~^T B R~ and G_rdd[self,self] = d^T |T| d are symmetric: only the NR (NR + 1) / 2 tiles on and
    // above the diagonal are computed, dealt round-robin over the consumer waves (static accumulator index k, the
    // tile coordinates of (wave, k) are wave-uniform scalars); the epilogue mirrors the off-diagonal tiles.
    constexpr int NTRI = NR * (NR + 1) / 2;
    constexpr int NCW = F2_NCW;                                // consumer waves (one per SIMD)
    constexpr int TPW = (NTRI + NCW - 1) / NCW;                // tiles per wave
    const int cw = wave - EC;
    int ti[TPW], tj[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      int idx = ((cw + s) & (NCW - 1)) + k * NCW, i = 0;       // idx-th upper-triangular tile, row-major; rotated by the subdomain so
                                                               // that the waves with one tile more are not always on the same SIMD
                                                               // (a SIMD's time is the sum of its waves' MFMA streams: 125.7 -> 120.8 us)
      if (idx >= NTRI) idx = -1;
      int rem = idx;
      while (rem >= NR - i && idx >= 0) {
        rem -= NR - i;
        ++i;
      }
      ti[k] = idx >= 0 ? i : -1;
      tj[k] = idx >= 0 ? i + rem : 0;
    }
    d4 accb[TPW], accd[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) accb[k] = accd[k] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nchunks; ++c) {
      lds_barrier();                               // barrier c
      const double* xb = &Xb[c & 1][0];
      const double* yb = &Yb[c & 1][0];
      const double* xd = &Xd[c & 1][0];
      const double* yd = &Yd[c & 1][0];
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        if (ti[k] < 0) continue;                   // wave-uniform
        const int xo = ti[k] * 16 + li, yo = tj[k] * 16 + li;
#ifndef F2_NO_MFMA
#pragma unroll
        for (int kk = 0; kk < 3 * EC; kk += 4)
          accb[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[(kk + lk) * LD + xo], yb[(kk + lk) * LD + yo], accb[k], 0, 0, 0);
        accd[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[lk * LD + xo], yd[lk * LD + yo], accd[k], 0, 0, 0);
#else
        (void)xb; (void)yb; (void)xd; (void)yd; (void)xo; (void)yo;
#endif
      }
    }
    lds_barrier();                                 // final barrier
    double* gb = a.G_bb + (long)s * a.gstride;              // [self, self]: block 0 of the block-compact layout
    double* gd = a.G_rdd + (long)s * a.gstride;
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      if (ti[k] < 0) continue;
      const int col = tj[k] * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti[k] * 16 + lk + 4 * r;
        const double vb = accb[k][r], vd = accd[k][r];
        if (col < QN && row < QN) {
          gb[(long)row * QN + col] = vb;
          gd[(long)row * QN + col] = vd;
          if (ti[k] != tj[k]) {                    // mirror: exactly symmetric across the off-diagonal tiles (inside a diagonal tile both halves are sums of their own: symmetric to rounding)
            gb[(long)col * QN + row] = vb;
            gd[(long)col * QN + row] = vd;
          }
        }
      }
    }
  }
  // r_fd self block: fixed-order sum over the EC producer waves
  __syncthreads();
  for (int c = tid; c < QN; c += nthreads) {
    double sum = 0.0;
    for (int w = 0; w < EC; ++w) sum += red[w * 128 + c];
    a.r_fd[(long)s * C + 2 * QN + c] = sum;
  }
}

// ---------------------------------------------------------------------------------------------------------
// F3: X = W_self (Oswald interpolation error of the own basis), Y = E W_self.  4 waves; tiles dealt round-robin.
struct F3Args {
  const double *V, *ebar, *AvgSelf, *AvgSide;
  double* G_nc;
  int N, S;
  long gsub;      // doubles between the [self, self] blocks of consecutive subdomains: 25 N^2 (dense) or N^2 (factored)
  int gld, goff;  // their row length and first entry: 5 N, 10 N^2 + 2 N (dense) or N, 0 (factored)
};

#ifndef F3_EW_X
#define F3_EW_X 4
#endif
constexpr int F3_EW = F3_EW_X;   // elements staged per wave and chunk: 16 elements (48 K-rows) per barrier pair

template <int NTX>
__global__ __launch_bounds__(256) void k_f3(Tmpl t, F3Args a) {
  constexpr int LD = padded_ld(NTX);
  constexpr int ECH = 4 * F3_EW;                 // elements per chunk
  constexpr int NTRI = NTX * (NTX + 1) / 2;      // G_nc[self,self] = W^T E W is symmetric: tiles on / above the diagonal
  constexpr int NT = (NTRI + 3) / 4;
  __shared__ double Xs[3 * ECH * LD], Ys[3 * ECH * LD];
  const int s = subdomain_of(t, blockIdx.x), tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6);
  const int N = a.N;
  for (int i = tid; i < 3 * ECH * LD; i += 256) Xs[i] = Ys[i] = 0.0;
  int ti[NT], tj[NT];
  // (Measured and dropped, round 3: rotating the tile dealing by the subdomain index so that the waves with the extra tile of the
  // workgroups sharing a CU sit on different SIMDs -- no change, 58 us: the kernel waits on its loads, not on the matrix pipe; a
  // register double buffer for the next chunk's rows behind LDS-only barriers -- hipcc drains it with vmcnt(0) at the dependent
  // index loads and the 144 VGPRs cost occupancy: 89 us.)
#pragma unroll
  for (int k = 0; k < NT; ++k) {                 // (wave + 4 k)-th upper-triangular tile, row-major
    int idx = wave + 4 * k, i = 0;
    if (idx >= NTRI) idx = -1;
    int rem = idx;
    while (idx >= 0 && rem >= NTX - i) {
      rem -= NTX - i;
      ++i;
    }
    ti[k] = idx >= 0 ? i : -1;
    tj[k] = idx >= 0 ? i + rem : 0;
  }
  __syncthreads();
  d4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* Vs = a.V + (long)s * t.n * N;
  const double* As = a.AvgSelf + (long)s * t.nv * N;
  const double* ebs = a.ebar + (long)s * t.nT;
  const int jc = lane < N ? lane : N - 1;
  for (int c0 = 0; c0 < t.nT; c0 += ECH) {
    // each wave stages F3_EW elements: all their loads are issued before the first use (one exposed latency per chunk)
    double v[F3_EW][3], av[F3_EW][3];
#pragma unroll
    for (int e = 0; e < F3_EW; ++e) {
      const int T = c0 + wave * F3_EW + e;
      const bool ok = T < t.nT;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int r = ok ? 3 * T + i : 0;
        v[e][i] = Vs[(long)r * N + jc];
        av[e][i] = As[(long)t.dof_vertex[r] * N + jc];
      }
    }
#pragma unroll
    for (int e = 0; e < F3_EW; ++e) {
      const int T = c0 + wave * F3_EW + e;
      const bool ok = T < t.nT;
      const double* K = t.stiff + 9 * (ok ? T : 0);
      const double eb = ok ? ebs[T] : 0.0;
      double w[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) w[i] = ok ? v[e][i] - av[e][i] : 0.0;
      if (lane < N) {
        const int row = 3 * (wave * F3_EW + e);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          Xs[(row + i) * LD + lane] = w[i];
          Ys[(row + i) * LD + lane] = eb * (K[i * 3] * w[0] + K[i * 3 + 1] * w[1] + K[i * 3 + 2] * w[2]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      if (ti[k] < 0) continue;                   // wave-uniform
      const int xo = ti[k] * 16 + li, yo = tj[k] * 16 + li;
#pragma unroll
      for (int kk = 0; kk < 3 * ECH; kk += 4)
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(kk + lk) * LD + xo], Ys[(kk + lk) * LD + yo], acc[k], 0, 0, 0);
    }
    __syncthreads();
  }
  double* g = a.G_nc + (long)s * a.gsub + a.goff;
  const int gld = a.gld;
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (ti[k] < 0) continue;
    const int col = tj[k] * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ti[k] * 16 + lk + 4 * r;
      if (col < N && row < N) {
        g[(long)row * gld + col] = acc[k][r];
        if (ti[k] != tj[k]) g[(long)col * gld + row] = acc[k][r];   // mirror: exactly symmetric across the off-diagonal tiles (inside a diagonal tile both halves are sums of their own: symmetric to rounding)
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Thin part of G_nc: workgroup (side a, subdomain s) writes block-row a (all 5 column blocks) and block [self, a].
// K runs over the 3 * ntouch rows of the elements with a vertex on side a.  The two dense blocks [a, self] and
// [a, a] are one small MFMA product  Wa^T [E Ws | E Wa]  (K = 3 ntouch, padded to a multiple of 4); the blocks
// towards the other sides only see the corner elements and are done on the VALU.
template <int NTX>
__device__ __forceinline__ void thin_nc_body(const Tmpl& t, int S, const int* __restrict__ nbr, int N,
                                             const double* __restrict__ V, const double* __restrict__ ebar,
                                             const double* __restrict__ AvgSelf, const double* __restrict__ AvgSide,
                                             double* __restrict__ G_nc, int side, int s) {
  constexpr int NMAX = 16 * NTX;
  constexpr int LDA = padded_ld(NTX);
  constexpr int LDB = LDA;                      // one Y buffer, used twice: E W_self (block [a, self]), then E W_a ([a, a])
  constexpr int NW = 8;                         // waves per workgroup
  constexpr int NACC = (NTX * NTX + NW - 1) / NW;
  constexpr int ITM = 3;                        // staging items per thread (ne * N <= 3 * 512, checked by the launcher)
  extern __shared__ double lds[];               // Wa [KP][LDA], Yc [KP][LDB]: 40 KB at config 3 -> 4 workgroups per CU
  const int slot = side_to_slot(side), tid = threadIdx.x;
  const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6);
  const int W = 5 * N;
  double* G = G_nc + (long)s * W * W;
  const int s2 = nbr[s * 5 + slot];
  if (s2 < 0) {   // no neighbour: the whole block-row and the [self, a] block are zero
    for (int i = tid; i < N * W; i += 512) G[(long)(slot * N + i / W) * W + i % W] = 0.0;
    for (int i = tid; i < N * N; i += 512) G[(long)(2 * N + i / N) * W + slot * N + i % N] = 0.0;
    return;
  }
  const int ne = t.touch_count[side];
  const int KP = (3 * t.ntouch + 3) & ~3;
  double* Wa = lds;
  double* Yc = lds + KP * LDA;
  // Per touching element everything but ebar is template data built once at mesh upload (t.stiff, t.touch_vtx,
  // t.touch_pos, t.touch_mask): no dependent index chains or integer divisions in the staging.
  // They are copied to LDS by all threads (coalesced), with ebar folded into the stiffness blocks.
  double* Ksc = Yc + KP * LDB;                                    // [ne][9]  ebar_T K_T
  int* ttab = reinterpret_cast<int*>(Ksc + 9 * t.ntouch);        // [ne]
  int* vtab = ttab + t.ntouch;                                    // [ne][3]
  int* ptab = vtab + 3 * t.ntouch;                                // [ne][3][4]
  int* side_mask = ptab + 12 * t.ntouch;                          // [ne]
  const double* ebs = ebar + (long)s * t.nT;
  for (int i = tid; i < KP * LDA; i += 512) Wa[i] = 0.0;
  for (int i = tid; i < KP * LDB; i += 512) Yc[i] = 0.0;
  for (int i = tid; i < ne; i += 512) {
    ttab[i] = t.touch_elem[side * t.ntouch + i];
    side_mask[i] = t.touch_mask[side * t.ntouch + i];
  }
  for (int i = tid; i < 3 * ne; i += 512) vtab[i] = t.touch_vtx[side * t.ntouch * 3 + i];
  for (int i = tid; i < 12 * ne; i += 512) ptab[i] = t.touch_pos[side * t.ntouch * 12 + i];
  for (int i = tid; i < 9 * ne; i += 512) {
    const int T = t.touch_elem[side * t.ntouch + i / 9];
    Ksc[i] = ebs[T] * t.stiff[9 * T + i % 9];
  }
  __syncthreads();
  const int nvs = nvs_of(t);
  const double* Vs = V + (long)s * t.n * N;
  const double* As = AvgSelf + (long)s * t.nv * N;
  const double* Aa = AvgSide + ((long)s * 4 + side) * nvs * N;
  // phase 1: rows of the touching elements: Wa (neighbour image) and Y = E W_self; E W_a stays in registers for pass 2
  double ya[ITM][3];
#pragma unroll
  for (int u = 0; u < ITM; ++u) {
    const int it = tid + 512 * u;
    ya[u][0] = ya[u][1] = ya[u][2] = 0.0;
    if (it < ne * N) {
      const int p = it / N, j = it - p * N;
      const int T = ttab[p];
      double wa[3], ws[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int pos = ptab[(3 * p + i) * 4 + side];
        ws[i] = Vs[(long)(3 * T + i) * N + j] - As[(long)vtab[3 * p + i] * N + j];
        wa[i] = pos >= 0 ? -Aa[(long)pos * N + j] : 0.0;
      }
      const double* K = Ksc + 9 * p;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Wa[(3 * p + k) * LDA + j] = wa[k];
        Yc[(3 * p + k) * LDB + j] = K[k * 3] * ws[0] + K[k * 3 + 1] * ws[1] + K[k * 3 + 2] * ws[2];
        ya[u][k] = K[k * 3] * wa[0] + K[k * 3 + 1] * wa[1] + K[k * 3 + 2] * wa[2];
      }
    }
  }
  __syncthreads();
  // phase 2: two MFMA passes over the same Wa, tiles dealt round-robin to the 8 waves:
  //   pass 0: Wa^T (E W_self) = block [a, self] (and its transpose [self, a]);  pass 1: Wa^T (E W_a) = block [a, a]
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      __syncthreads();                          // every wave is done reading Y = E W_self
#pragma unroll
      for (int u = 0; u < ITM; ++u) {
        const int it = tid + 512 * u;
        if (it < ne * N) {
          const int p = it / N, j = it - p * N;
#pragma unroll
          for (int k = 0; k < 3; ++k) Yc[(3 * p + k) * LDB + j] = ya[u][k];
        }
      }
      __syncthreads();
    }
    d4 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < KP; kk += 4) {
#pragma unroll
      for (int k = 0; k < NACC; ++k) {
        const int tile = wave + NW * k;
        if (tile < NTX * NTX) {
          const int ti = tile / NTX, tj = tile - ti * NTX;
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wa[(kk + lk) * LDA + ti * 16 + li], Yc[(kk + lk) * LDB + tj * 16 + li], acc[k], 0, 0, 0);
        }
      }
    }
    const int cslot = pass == 0 ? 2 : slot;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      const int tile = wave + NW * k;
      const int ti = tile / NTX, tj = tile - ti * NTX;
      const int jj = tj * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + lk + 4 * r;
        const double val = acc[k][r];
        if (tile < NTX * NTX && jj < N && row < N) {
          G[(long)(slot * N + row) * W + cslot * N + jj] = val;
          if (pass == 0) G[(long)(2 * N + jj) * W + slot * N + row] = val;   // [self, a] = [a, self]^T
        }
      }
    }
  }
  // phase 3: blocks [a, b] towards the other sides b: only elements touching both sides contribute (the corner ones;
  // none for the opposite side).  Work item = (other slot, column j, chunk of NMAX / 4 rows), all 8 waves busy.
  constexpr int RPC = NMAX / 4;
  for (int it = tid; it < 3 * N * 4; it += 512) {
    const int which = it / (4 * N), rem = it - which * 4 * N, rc = rem / N, j = rem - rc * N;
    int slot2 = -1, seen = 0;                     // which-th slot of {0, 1, 3, 4} without `slot`
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      if (sl == 2 || sl == slot) continue;
      if (seen == which) slot2 = sl;
      ++seen;
    }
    double accv[RPC];
#pragma unroll
    for (int i = 0; i < RPC; ++i) accv[i] = 0.0;
    const int sx = nbr[s * 5 + slot2];
    if (sx >= 0) {
      const int sd2 = slot_to_side(slot2), bit2 = 1 << sd2;
      const double* Ab = AvgSide + ((long)s * 4 + sd2) * nvs * N;
      for (int p = 0; p < ne; ++p) {
        if (!(side_mask[p] & bit2)) continue;
        double w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int pos = ptab[(3 * p + i) * 4 + sd2];
          w[i] = pos >= 0 ? -Ab[(long)pos * N + j] : 0.0;
        }
        const double* K = Ksc + 9 * p;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double y = K[k * 3] * w[0] + K[k * 3 + 1] * w[1] + K[k * 3 + 2] * w[2];
          const double* wa = Wa + (3 * p + k) * LDA + rc * RPC;
#pragma unroll
          for (int i = 0; i < RPC; ++i) accv[i] += wa[i] * y;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RPC; ++i)
      if (rc * RPC + i < N) G[(long)(slot * N + rc * RPC + i) * W + slot2 * N + j] = accv[i];
  }
}

template <int NTX>
// 8 waves per SIMD (<= 64 VGPRs, no spills for NTX <= 3): four 512-thread workgroups per CU instead of three (the
// 40 KB of LDS allow four) -- 140 -> 132 us at config 3
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NTX <= 3 ? 8 : 6, 8))) void k_thin_nc(Tmpl t, int S, const int* __restrict__ nbr, int N,
                                                 const double* __restrict__ V, const double* __restrict__ ebar,
                                                 const double* __restrict__ AvgSelf, const double* __restrict__ AvgSide,
                                                 double* __restrict__ G_nc) {
  thin_nc_body<NTX>(t, S, nbr, N, V, ebar, AvgSelf, AvgSide, G_nc, blockIdx.x, subdomain_of(t, blockIdx.y));
}

// ---------------------------------------------------------------------------------------------------------
// Thin part of G_nc, factored.  The image of neighbour a's basis under the Oswald interpolation error lives on the
// vertices of side a only: W_a = -P_a A_a with A_a [nvs][N] the vertex averages k_vertex_avg computes (Avg_side) and P_a
// the 0/1 matrix that copies side vertex `pos` to every local DoF of a touching element sitting on it.  Every block of
// G_nc that involves slot a is therefore a product of rank <= nvs:
//   G_nc[a, self] = A_a^T C_a,   C_a = -P_a^T E W_self  [nvs][N];      G_nc[a, b] = A_a^T M_ab A_b,   M_ab = P_a^T E P_b [nvs][nvs]
// and the factored layout stores one row per (side, side vertex):  F_nc [S][4][nvs][2 N + 4 nvs] = A_a | C_a | M_a0 .. M_a3
// (26 KB per subdomain at config 3 instead of the 115 KB of the nine N x N blocks; the estimate kernels contract the rows
// with the coefficient vectors directly).  Rows of vertices a side does not have, and of sides without neighbour, are zero.
__host__ __device__ inline int fnc_ld(int nvs, int N, int vertex_patch = 0) { return 2 * N + 4 * nvs + (vertex_patch ? N : 0); }
constexpr int NCF_ROWS = 8;   // rows listed per side vertex in k_thin_ncf (4 in the 8-triangle pattern)

struct ThinNcfArgs {
  const double *V, *ebar, *AvgSelf, *AvgSide;
  const int* nbr;
  double* Fnc;
  int N, S;
};

__device__ __forceinline__ void thin_ncf_body(const Tmpl& t, const ThinNcfArgs& a, int side, int s) {
  extern __shared__ double lds[];
  const int slot = side_to_slot(side), tid = threadIdx.x, N = a.N;
  const int nvs = nvs_of(t), LD = fnc_ld(nvs, N, t.opt_oswald_vertex);
  double* Fs = a.Fnc + ((long)s * 4 + side) * nvs * LD;
  const int s2 = a.nbr[s * 5 + slot];
  if (s2 < 0) {
    for (int i = tid; i < nvs * LD; i += 256) Fs[i] = 0.0;
    return;
  }
  THIN_STAMP(2, 0);
  const int ne = t.touch_count[side];
  double* Y = lds;                                                // [3 ne][N]  rows of E W_self of the touching elements
  double* Ksc = Y + 3 * t.ntouch * N;                            // [ne][9]    ebar_T K_T
  int* ttab = reinterpret_cast<int*>(Ksc + 9 * t.ntouch);        // [ne]
  int* vtab = ttab + t.ntouch;                                    // [ne][3]
  int* ptab = vtab + 3 * t.ntouch;                                // [ne][3][4]
  int* side_mask = ptab + 12 * t.ntouch;                          // [ne]
  int* rlist = side_mask + t.ntouch;                              // [nvs][NCF_ROWS + 1]: rows 3 p + k sitting on side vertex pos (count first)
  const double* ebs = a.ebar + (long)s * t.nT;
  for (int i = tid; i < ne; i += 256) {
    ttab[i] = t.touch_elem[side * t.ntouch + i];
    side_mask[i] = t.touch_mask[side * t.ntouch + i];
  }
  // the rows that meet in a side vertex, in table order: t.touch_rlist (built at mesh upload; a vertex with more than NCF_ROWS rows has
  // count -1 and is scanned where it is used)
  static_assert(NCF_ROWS == 8, "layout of t.touch_rlist");
  for (int i = tid; i < nvs * (NCF_ROWS + 1); i += 256) rlist[i] = t.touch_rlist[side * nvs * (NCF_ROWS + 1) + i];
  for (int i = tid; i < 3 * ne; i += 256) vtab[i] = t.touch_vtx[side * t.ntouch * 3 + i];
  for (int i = tid; i < 12 * ne; i += 256) ptab[i] = t.touch_pos[side * t.ntouch * 12 + i];
  for (int i = tid; i < 9 * ne; i += 256) {
    const int T = t.touch_elem[side * t.ntouch + i / 9];
    Ksc[i] = ebs[T] * t.stiff[9 * T + i % 9];
  }
  THIN_STAMP(2, 1);
  __syncthreads();
  THIN_STAMP(2, 2);
  const double* Vs = a.V + (long)s * t.n * N;
  const double* As = a.AvgSelf + (long)s * t.nv * N;
  // (round 3) every thread requests the rows of a batch of items before it uses the first one: one memory round trip per batch
  // instead of one per item; the neighbour's vertex averages of the output sweep are requested here as well
  const int nside = (side == 0 || side == 3) ? t.nvx : t.nvy;
  const double* Aa = a.AvgSide + ((long)s * 4 + side) * nvs * N;
  constexpr int OT = 2;                             // output items per thread held across the barrier (nvs N <= 512: checked below)
  double aav[OT];
#pragma unroll
  for (int u = 0; u < OT; ++u) {
    const int it = u * 256 + tid, pos = it / N, j = it - pos * N;
    aav[u] = it < nvs * N && pos < nside ? Aa[(long)pos * N + j] : 0.0;
  }
  auto stage_y = [&](auto wtag) {
    constexpr int W = decltype(wtag)::value;
    using VT = typename VecT<W>::T;
    constexpr int IT = W == 2 ? 3 : 5;              // (config 3: all items of a thread in one batch)
    const int NW = N / W;
    for (int base = 0; base < ne * NW; base += IT * 256) {
      VT vv[IT][3], av[IT][3];
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int it = base + u * 256 + tid, itc = it < ne * NW ? it : ne * NW - 1;
        const int p = itc / NW, j = (itc - p * NW) * W, T = ttab[p];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          vv[u][i] = ldv<W>(Vs + (long)(3 * T + i) * N + j);
          av[u][i] = ldv<W>(As + (long)vtab[3 * p + i] * N + j);
        }
      }
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int it = base + u * 256 + tid;
        if (it >= ne * NW) continue;
        const int p = it / NW, j = (it - p * NW) * W;
        VT ws[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) ws[i] = vv[u][i] - av[u][i];
        const double* K = Ksc + 9 * p;
#pragma unroll
        for (int k = 0; k < 3; ++k) stv<W>(Y + (3 * p + k) * N + j, K[k * 3] * ws[0] + K[k * 3 + 1] * ws[1] + K[k * 3 + 2] * ws[2]);
      }
    }
  };
  if (N & 1) stage_y(W1{});
  else stage_y(W2{});
  THIN_STAMP(2, 3);
  __syncthreads();
  THIN_STAMP(2, 4);
  // A_a | C_a: one item per (side vertex, column); the rows that meet in a vertex are summed in table order
  for (int it = tid, u = 0; it < nvs * N; it += 256, ++u) {
    const int pos = it / N, j = it - pos * N;
    double c = 0.0;
    if (pos < nside) {
      const int* rl = rlist + pos * (NCF_ROWS + 1);
      if (rl[0] >= 0) {
        for (int k = 0; k < rl[0]; ++k) c -= Y[rl[1 + k] * N + j];
      } else {
        for (int r = 0; r < 3 * ne; ++r)
          if (ptab[r * 4 + side] == pos) c -= Y[r * N + j];
      }
    }
    Fs[(long)pos * LD + j] = u < OT ? (u == 0 ? aav[0] : aav[1]) : (pos < nside ? Aa[(long)pos * N + j] : 0.0);
    Fs[(long)pos * LD + N + j] = c;
    if (t.opt_oswald_vertex) {
      // A_diag: the diagonal subdomain's share of the vertex average at a cross point, carried by the sides 0 (S) and 3 (N) at
      // their end vertices (corner 0 SW / 1 SE on side 0, 2 NW / 3 NE on side 3); it multiplies the DIAGONAL subdomain's
      // coefficients in the estimate (z_a[pos] += A_diag . u_diag)
      const bool carries = (side == 0 || side == 3) && (pos == 0 || pos == t.nvx - 1);
      const int corner = (side == 0 ? 0 : 2) + (pos == 0 ? 0 : 1);
      const double* Ac = a.AvgSide + (long)a.S * 4 * nvs * N;
      Fs[(long)pos * LD + 2 * N + 4 * nvs + j] = carries ? Ac[((long)s * 4 + corner) * N + j] : 0.0;
    }
  }
  // M_ab: one item per (side vertex, other side b, vertex of b)
  for (int it = tid; it < nvs * 4 * nvs; it += 256) {
    const int pos = it / (4 * nvs), rem = it - pos * 4 * nvs, sb = rem / nvs, pos2 = rem - sb * nvs;
    double m = 0.0;
    // two lattice vertices share an element only if they are lattice neighbours (every triangle of the template spans one lattice
    // step): everything else is zero without a look at the tables
    const int ax = side == 0 || side == 3 ? pos : (side == 1 ? 0 : t.nvx - 1), ay = side == 1 || side == 2 ? pos : (side == 0 ? 0 : t.nvy - 1);
    const int bx = sb == 0 || sb == 3 ? pos2 : (sb == 1 ? 0 : t.nvx - 1), by = sb == 1 || sb == 2 ? pos2 : (sb == 0 ? 0 : t.nvy - 1);
    const bool near = ax - bx <= 1 && bx - ax <= 1 && ay - by <= 1 && by - ay <= 1;
    if (pos < nside && near) {
      const int bit = 1 << sb;
      const int* rl = rlist + pos * (NCF_ROWS + 1);
      const int nr = rl[0] >= 0 ? rl[0] : 3 * ne;
      for (int i = 0; i < nr; ++i) {
        const int r = rl[0] >= 0 ? rl[1 + i] : i, p = r / 3, k = r - 3 * p;
        if (!(side_mask[p] & bit) || ptab[r * 4 + side] != pos) continue;
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2)
          if (ptab[(3 * p + k2) * 4 + sb] == pos2) m += Ksc[9 * p + 3 * k + k2];
      }
    }
    Fs[(long)pos * LD + 2 * N + rem] = m;
  }
  THIN_STAMP(2, 5);
}

__global__ __launch_bounds__(256) void k_thin_ncf(Tmpl t, ThinNcfArgs a) { thin_ncf_body(t, a, blockIdx.x, subdomain_of(t, blockIdx.y)); }

static size_t thin_ncf_lds_bytes(const Tmpl& t, int N) {
  const size_t nvs = (size_t)(t.nvx > t.nvy ? t.nvx : t.nvy);
  return sizeof(double) * (3 * (size_t)t.ntouch * N + 9 * (size_t)t.ntouch) +
         sizeof(int) * (17 * (size_t)t.ntouch + nvs * (NCF_ROWS + 1));
}

// ---------------------------------------------------------------------------------------------------------
// Thin part of G_bb / G_rdd / G_ab / r_fd for side a.  The image of neighbour a's basis on the target subdomain lives
// on the np <= ncf side faces only, so every block of these operators that involves slot a is a rank-<=np product of the
// FACTORS built here, one row per side face p (T = the element of face p, f_p its local face):
//   Ra [p][QN]    flux image of the neighbour on face p                    (= R_side)
//   Yb [p][QN]    (B_T R~_self)[f_p]            G_bb[a, self] = Ra^T Yb,   G_bb[a, a]  = Ra^T diag(B_T[f_p][f_p]) Ra
//   Dp [p][QN]    |T| c_p d_T(R~_self)          G_rdd[a, self] = Ra^T Dp,  G_rdd[a, a] = Ra^T diag(|T| c_p^2) Ra
//   Xab [p][Q][N] ((A_ab^q)^T[:, f_p] V_T)      G_ab^q[:, a] = Xab_q^T Ra
//   sc [p][3]     B_T[f_p][f_p], |T| c_p^2, (b_T . 1) c_p                  r_fd[a] = sc2^T Ra
// The factors ARE the output in the factored layout (lrbms_project_estimate_fused_factored: ~2 KB instead of 205 KB per
// side at config 3, and the reduced estimate consumes them directly); the dense block-compact layout is produced from
// them by k_thin_expand.  Row layout of F_side [S][4][ncf][LD], LD = 4 QN + 4:
//   [0, QN) Ra | [QN, 2QN) Yb | [2QN, 3QN) Dp | [3QN, 4QN) Xab (q, i) | sc0 sc1 sc2 0
struct ThinRtArgs {
  const double *V, *Rself, *Rside, *Bbb, *Aab, *b;
  const int* nbr;
  double *Fside, *r_fd;
  int Q, N, S;
};

__host__ __device__ inline int fside_ld(int Q, int N) { return 4 * Q * N + 4; }

static size_t thin_rt_lds_bytes(const Tmpl& t, int Q, int N) {      // fco [ncf][4], sc2 [ncf], Bl [ncf][3], Aq [ncf][Q][3], pad, Rl [ncf][QN], fidx [ncf][5] ints
  return sizeof(double) * ((size_t)(8 + 3 * Q) * t.ncf + 1 + (size_t)t.ncf * Q * N + 3 * (size_t)t.ncf);
}

__device__ __forceinline__ void thin_rt_body(const Tmpl& t, const ThinRtArgs& a, int side, int s) {
  extern __shared__ double lds[];
  const int slot = side_to_slot(side), tid = threadIdx.x;
  const int Q = a.Q, N = a.N, QN = Q * N, C = 5 * QN, S = a.S, LD = fside_ld(Q, N);
  double* Fs = a.Fside + ((long)s * 4 + side) * t.ncf * LD;
  const int s2 = a.nbr[s * 5 + slot];
  const int np = s2 < 0 ? 0 : t.side_count[side];
  // rows of side faces that do not exist (no neighbour, or fewer faces than ncf on this side): zero factors
  for (int i = tid; i < (t.ncf - np) * LD; i += 256) Fs[(long)np * LD + i] = 0.0;
  if (np == 0) {
    for (int i = tid; i < QN; i += 256) a.r_fd[(long)s * C + slot * QN + i] = 0.0;
    return;
  }
  // per side face p: element, its face on the side, its three RT0 rows and divergence coefficients -- resolved ONCE by
  // np threads into LDS (the index chains side_elem -> nb_elem -> elem_rt / face_len / area are three dependent round trips)
  THIN_STAMP(1, 0);
  double* fco = lds;                                  // [np][4]: sign |e_g| / |T| (g = 0..2), |T|
  double* sc2 = fco + 4 * np;                         // [np]    (b_T . 1) c_p
  double* Bl = sc2 + np;                              // [np][3]    row f_p of B_T
  double* Aq = Bl + 3 * np;                           // [np][Q][3] column f_p of (A_ab^q)_T
  double* Rl = Aq + 3 * Q * np + ((np * (8 + 3 * Q)) & 1);   // [np][QN]   the neighbour's flux image on the side faces (Ra), 16-byte aligned
  int* fidx = reinterpret_cast<int*>(Rl + np * QN);   // [np][5]: T, fp, rt row of face 0..2
  const int* nbr_s = a.nbr + s * 5;
  const double* Rs = a.Rself + (long)s * t.nrt * QN;
  const double* Rsd = a.Rside + ((long)s * 4 + side) * t.ncf * QN;
  // Ra (np QN contiguous doubles) is requested first and parked in LDS: the factor rows and r_fd both read it there
  constexpr int RT_IT = 3;
  double rr[RT_IT];
#pragma unroll
  for (int u = 0; u < RT_IT; ++u) rr[u] = Rsd[u * 256 + tid < np * QN ? u * 256 + tid : 0];
  if (tid < np) {
    // template data of the side face from ONE table row (t.sface_i / t.sface_d, built at mesh upload), then the subdomain's data
    // of that element: two round trips (was a chain of five: side_elem -> nb_elem -> elem_rt / face_len / area -> b, B)
    const int p = tid;
    const int4* ti4 = reinterpret_cast<const int4*>(t.sface_i + (side * t.ncf + p) * 12);
    const int4 i0 = ti4[0], i1 = ti4[1], i2 = ti4[2];      // T fp rt0 rt1 | rt2 sg0 sg1 sg2 | bs0 bs1 bs2 -
    const double* td = t.sface_d + (side * t.ncf + p) * 4;
    const double len[3] = {td[0], td[1], td[2]}, area = td[3];
    const int T = i0.x, fp = i0.y;
    const int rt[3] = {i0.z, i0.w, i1.x}, sg[3] = {i1.y, i1.z, i1.w}, bs[3] = {i2.x, i2.y, i2.z};
    const double* be = a.b + (long)s * t.n + 3 * T;
    const double* B = a.Bbb + ((long)s * t.nT + T) * 9 + fp * 3;
    const double b0 = be[0], b1 = be[1], b2 = be[2], B0 = B[0], B1 = B[1], B2 = B[2];
    double Aqv[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < Q) {
        const double* A = a.Aab + (((long)q * S + s) * t.nT + T) * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) Aqv[q][k] = A[k * 3 + fp];
      }
    fidx[p * 5] = T;
    fidx[p * 5 + 1] = fp;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      fidx[p * 5 + 2 + g] = rt[g];
      const int sign = (bs[g] >= 0 && nbr_s[side_to_slot(bs[g] >= 0 ? bs[g] : 0)] < 0) ? 1 : sg[g];      // face_sign_at
      fco[p * 4 + g] = sign * len[g] / area;
    }
    fco[p * 4 + 3] = area;
    const double cp = fco[p * 4 + fp];
    sc2[p] = (b0 + b1 + b2) * cp;
    double* row = Fs + (long)p * LD + 4 * QN;
    row[0] = fp == 0 ? B0 : fp == 1 ? B1 : B2;
    row[1] = area * cp * cp;
    row[2] = sc2[p];
    row[3] = 0.0;
    Bl[p * 3] = B0;
    Bl[p * 3 + 1] = B1;
    Bl[p * 3 + 2] = B2;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < Q)
#pragma unroll
        for (int k = 0; k < 3; ++k) Aq[(p * Q + q) * 3 + k] = Aqv[q][k];
  }
#pragma unroll
  for (int u = 0; u < RT_IT; ++u)
    if (u * 256 + tid < np * QN) Rl[u * 256 + tid] = rr[u];
  for (int i = RT_IT * 256 + tid; i < np * QN; i += 256) Rl[i] = Rsd[i];      // (larger templates)
  THIN_STAMP(1, 1);
  __syncthreads();
  THIN_STAMP(1, 2);
  // The three sweeps below are latency-bound (a workgroup has a few items per thread, every item a handful of global loads), so each
  // thread requests the loads of ALL its items of a sweep before it uses the first one: one memory round trip per sweep instead
  // of one per item (round 3; the expressions, and with them the results, are unchanged).
  auto factor_rows = [&](auto wtag) {
    constexpr int W = decltype(wtag)::value;
    using VT = typename VecT<W>::T;
    constexpr int IT = W == 2 ? 2 : 3;                // items per thread and batch (7 loads each: the kernel stays below 96 VGPRs)
    const int QW = QN / W;
    for (int base = 0; base < np * QW; base += IT * 256) {
      VT ra[IT], rv[IT][3], xv[IT][3];
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int it = base + u * 256 + tid, itc = it < np * QW ? it : np * QW - 1;
        const int p = itc / QW, c = (itc - p * QW) * W, q = c / N, i = c - q * N;
        const int T = fidx[p * 5];
        ra[u] = ldv<W>(Rl + p * QN + c);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          rv[u][g] = ldv<W>(Rs + (long)fidx[p * 5 + 2 + g] * QN + c);
          xv[u][g] = ldv<W>(a.V + ((long)s * t.n + 3 * T + g) * N + i);
        }
      }
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int it = base + u * 256 + tid;
        if (it >= np * QW) continue;
        const int p = it / QW, c = (it - p * QW) * W, q = c / N;
        const int fp = fidx[p * 5 + 1];
        const double area = fco[p * 4 + 3];
        VT yb = zerov<W>(), d = zerov<W>();
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          yb += Bl[p * 3 + g] * rv[u][g];
          d += fco[p * 4 + g] * rv[u][g];
        }
        VT x = zerov<W>();
#pragma unroll
        for (int k = 0; k < 3; ++k) x += xv[u][k] * Aq[(p * Q + q) * 3 + k];
        double* row = Fs + (long)p * LD;
        stv<W>(row + c, ra[u]);
        stv<W>(row + QN + c, yb);
        stv<W>(row + 2 * QN + c, area * fco[p * 4 + fp] * d);
        stv<W>(row + 3 * QN + c, x);
      }
    }
  };
  if (N & 1) factor_rows(W1{});
  else factor_rows(W2{});
  THIN_STAMP(1, 3);
  for (int c = tid; c < QN; c += 256) {
    double v = 0.0;
    for (int p = 0; p < np; ++p) v += sc2[p] * Rl[p * QN + c];
    a.r_fd[(long)s * C + slot * QN + c] = v;
  }
  THIN_STAMP(1, 4);
}

__global__ __launch_bounds__(256) void k_thin_rt(Tmpl t, ThinRtArgs a) { thin_rt_body(t, a, blockIdx.x, subdomain_of(t, blockIdx.y)); }

// Dense block-compact form of the side blocks from the factors (callers that want the blocks themselves: the reference
// keeps such operators as BlockOperators, block_swipdg.py:336-338): G_bb / G_rdd [S][9][QN][QN] blocks 1 + side = [a, self],
// 5 + side = [a, a]; G_ab [Q][S][N][5QN] columns of slot a.  Bound by its writes (1 GB at config 3).
struct ThinExpandArgs {
  const double* Fside;
  const int* nbr;
  double *G_bb, *G_rdd, *G_ab;
  int Q, N, S;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_thin_expand(Tmpl t, ThinExpandArgs a) {
  extern __shared__ double lds[];
  const int side = blockIdx.x, s = subdomain_of(t, blockIdx.y), slot = side_to_slot(side), tid = threadIdx.x;
  const int Q = a.Q, N = a.N, QN = Q * N, C = 5 * QN, S = a.S, LD = fside_ld(Q, N);
  double* Gb_as = a.G_bb + ((long)s * 9 + 1 + side) * QN * QN;
  double* Gb_aa = a.G_bb + ((long)s * 9 + 5 + side) * QN * QN;
  double* Gd_as = a.G_rdd + ((long)s * 9 + 1 + side) * QN * QN;
  double* Gd_aa = a.G_rdd + ((long)s * 9 + 5 + side) * QN * QN;
  const int s2 = a.nbr[s * 5 + slot];
  const int np = t.side_count[side];
  if (s2 < 0 || np == 0) {
    for (int i = tid; i < QN * QN; i += 256) Gb_as[i] = Gb_aa[i] = Gd_as[i] = Gd_aa[i] = 0.0;
    for (int q = 0; q < Q; ++q)
      for (int i = tid; i < N * QN; i += 256) a.G_ab[(((long)q * S + s) * N + i / QN) * C + slot * QN + i % QN] = 0.0;
    return;
  }
  double* Fl = lds;                 // [np][LD] the factor rows of this side
  const double* Fs = a.Fside + ((long)s * 4 + side) * t.ncf * LD;
  for (int i = tid; i < np * LD; i += 256) Fl[i] = Fs[i];
  __syncthreads();
  const double* Ra = Fl;
  const double* Yb = Fl + QN;
  const double* Dp = Fl + 2 * QN;
  const double* Xab = Fl + 3 * QN;
  const double* sc = Fl + 4 * QN;
  if ((QN & 1) == 0) {
    // Register tiles of 2 rows x 2 columns, lanes over consecutive column PAIRS: every store is 16 bytes per lane and a
    // wave's store instruction covers whole 128-byte lines.
    const int hp = QN / 2, trows = (QN + 1) / 2;
    for (int it = tid; it < trows * hp; it += 256) {
      const int tr = it / hp, cc = 2 * (it - tr * hp), r0 = 2 * tr, r1 = r0 + 1 < QN ? r0 + 1 : QN - 1;
      double2 vb[2], vd[2], wb[2], wd[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) vb[i] = vd[i] = wb[i] = wd[i] = make_double2(0.0, 0.0);
      for (int p = 0; p < np; ++p) {
        const double2 rc = *reinterpret_cast<const double2*>(Ra + p * LD + cc);
        const double2 yc = *reinterpret_cast<const double2*>(Yb + p * LD + cc);
        const double2 dc = *reinterpret_cast<const double2*>(Dp + p * LD + cc);
        const double k0 = sc[p * LD], k1 = sc[p * LD + 1];
        const double ra[2] = {Ra[p * LD + r0], Ra[p * LD + r1]};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          vb[i].x += ra[i] * (k0 * rc.x); vb[i].y += ra[i] * (k0 * rc.y);
          vd[i].x += ra[i] * (k1 * rc.x); vd[i].y += ra[i] * (k1 * rc.y);
          wb[i].x += ra[i] * yc.x;        wb[i].y += ra[i] * yc.y;
          wd[i].x += ra[i] * dc.x;        wd[i].y += ra[i] * dc.y;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (r0 + i < QN) {
          const long o = (long)(r0 + i) * QN + cc;
          *reinterpret_cast<double2*>(Gb_aa + o) = vb[i];
          *reinterpret_cast<double2*>(Gd_aa + o) = vd[i];
          *reinterpret_cast<double2*>(Gb_as + o) = wb[i];
          *reinterpret_cast<double2*>(Gd_as + o) = wd[i];
        }
      }
    }
    constexpr int RT = 4;
    const int irows = (N + RT - 1) / RT;
    for (int it = tid; it < Q * irows * hp; it += 256) {
      const int q = it / (irows * hp), rem = it - q * irows * hp, ir = rem / hp, cc = 2 * (rem - ir * hp), i0 = RT * ir;
      double2 v[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) v[i] = make_double2(0.0, 0.0);
      for (int p = 0; p < np; ++p) {
        const double2 rc = *reinterpret_cast<const double2*>(Ra + p * LD + cc);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const double x = Xab[p * LD + q * N + (i0 + i < N ? i0 + i : N - 1)];
          v[i].x += x * rc.x;
          v[i].y += x * rc.y;
        }
      }
#pragma unroll
      for (int i = 0; i < RT; ++i)
        if (i0 + i < N)
          *reinterpret_cast<double2*>(a.G_ab + (((long)q * S + s) * N + i0 + i) * C + slot * QN + cc) = v[i];
    }
  } else {
    // odd QN: register tiles of 4 rows x 1 column, lanes over consecutive columns (8-byte stores)
    constexpr int RT = 4;
    const int trows = (QN + RT - 1) / RT;
    for (int it = tid; it < trows * QN; it += 256) {
      const int tr = it / QN, cc = it - tr * QN, r0 = RT * tr;
      double vb[RT], vd[RT], wb[RT], wd[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) vb[i] = vd[i] = wb[i] = wd[i] = 0.0;
      for (int p = 0; p < np; ++p) {
        const double rc = Ra[p * LD + cc], yc = Yb[p * LD + cc], dc = Dp[p * LD + cc];
        const double s0 = sc[p * LD] * rc, s1 = sc[p * LD + 1] * rc;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const double ra = Ra[p * LD + (r0 + i < QN ? r0 + i : QN - 1)];
          vb[i] += ra * s0;
          vd[i] += ra * s1;
          wb[i] += ra * yc;
          wd[i] += ra * dc;
        }
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        if (r0 + i < QN) {
          const long o = (long)(r0 + i) * QN + cc;
          Gb_aa[o] = vb[i];
          Gd_aa[o] = vd[i];
          Gb_as[o] = wb[i];
          Gd_as[o] = wd[i];
        }
      }
    }
    const int irows = (N + RT - 1) / RT;
    for (int it = tid; it < Q * irows * QN; it += 256) {
      const int q = it / (irows * QN), rem = it - q * irows * QN, ir = rem / QN, cc = rem - ir * QN, i0 = RT * ir;
      double v[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) v[i] = 0.0;
      for (int p = 0; p < np; ++p) {
        const double rc = Ra[p * LD + cc];
#pragma unroll
        for (int i = 0; i < RT; ++i) v[i] += Xab[p * LD + q * N + (i0 + i < N ? i0 + i : N - 1)] * rc;
      }
#pragma unroll
      for (int i = 0; i < RT; ++i)
        if (i0 + i < N) a.G_ab[(((long)q * S + s) * N + i0 + i) * C + slot * QN + cc] = v[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Off-diagonal blocks of the projected system: B_q[s][slot a] = V_s|side^T (C_q V_nbr|side), K = 3 * (faces on the
// side).  One workgroup per (side, subdomain) handles every q: the rows of V_s at the side are staged once, C_q V_nbr per
// q, and the N x N product is a handful of MFMAs (the VALU version re-read two LDS operands per multiply-add and was
// bound by the LDS pipe: 76 us at config 3 for 105 MB of output).
static size_t coupling_lds_bytes(const Tmpl& t, int ntx, int Q) {      // Xin [KP][LD], Tm [Q][KP][LD]; Cl [Q][ncf][9]
  return sizeof(double) * ((1 + (size_t)Q) * (size_t)((3 * t.ncf + 3) & ~3) * padded_ld(ntx) + (size_t)Q * t.ncf * 9);
}

template <int NTX>
__device__ __forceinline__ void coupling_body(const Tmpl& t, int S, const int* __restrict__ nbr, int Q, int N,
                                              const double* __restrict__ V, const double* __restrict__ A_cpl,
                                              double* __restrict__ B_sys, int side, int s) {
  constexpr int LD = padded_ld(NTX);
  constexpr int NT = (NTX * NTX + 3) / 4;
  extern __shared__ double lds[];
  const int slot = side_to_slot(side), tid = threadIdx.x;
  const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = uniform(tid >> 6);
  const int s2 = nbr[s * 5 + slot];
  const int cnt = t.side_count[side];
  if (s2 < 0 || cnt == 0) {
    for (int q = 0; q < Q; ++q) {
      double* out = B_sys + ((((long)q * S + s) * 5 + slot) * N) * N;
      for (int i = tid; i < N * N; i += 256) out[i] = 0.0;
    }
    return;
  }
  THIN_STAMP(0, 0);
  const int K = 3 * cnt, KP = (K + 3) & ~3;
  double* Xin = lds;             // [KP][LD]  rows of V_s at the side
  double* Tm = lds + KP * LD;    // [Q][KP][LD]  rows of V_nbr at the side (raw, in Tm[0]), then C_q * (those rows)
  double* Cl = Tm + Q * KP * LD; // [Q][cnt][9]
  // One memory round trip (behind the element tables) per workgroup (round 3): every thread requests all its rows before it stores
  // the first one, and both components share the neighbour's rows: they are parked raw in Tm[0], then one LDS pass per side face
  // replaces them by C_q V_nbr for every q -- the expression of the former per-component staging, so the results are unchanged.
  // (Before: rows of V_s, then per component the coupling blocks and the neighbour's rows: six round trips, five barriers.)
  // Columns >= N of the LDS rows are never written: they only reach output entries that are not stored.
  double clv[2];                                    // the coupling blocks: the first 512 entries requested first, stored behind the rows' requests
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = u * 256 + tid, ic = i < Q * cnt * 9 ? i : 0, q = ic / (cnt * 9), r = ic - q * cnt * 9;
    clv[u] = A_cpl[((((long)q * S + s) * 4 + side) * t.ncf) * 9 + r];
  }
  auto stage = [&](auto wtag) {
    constexpr int W = decltype(wtag)::value;
    using VT = typename VecT<W>::T;
    constexpr int IT = W == 2 ? 2 : 4;
    const int NW = N / W;
    for (int base = 0; base < KP * NW; base += IT * 256) {
      VT xv[IT], vv[IT];
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int i = base + u * 256 + tid, ic = i < K * NW ? i : 0;
        const int row = ic / NW, col = (ic - row * NW) * W, pos = row / 3, ii = row - 3 * pos;
        xv[u] = ldv<W>(V + ((long)s * t.n + 3 * t.side_elem[side * t.ncf + pos] + ii) * N + col);
        vv[u] = ldv<W>(V + ((long)s2 * t.n + 3 * t.side_elem_out[side * t.ncf + pos] + ii) * N + col);
      }
      if (base == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (u * 256 + tid < Q * cnt * 9) Cl[u * 256 + tid] = clv[u];
        for (int i = 512 + tid; i < Q * cnt * 9; i += 256) {      // (large templates: the rest of the blocks)
          const int q = i / (cnt * 9), r = i - q * cnt * 9;
          Cl[i] = A_cpl[((((long)q * S + s) * 4 + side) * t.ncf) * 9 + r];
        }
      }
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int i = base + u * 256 + tid;
        if (i >= KP * NW) continue;
        const int row = i / NW, col = (i - row * NW) * W;
        stv<W>(Xin + row * LD + col, i < K * NW ? xv[u] : zerov<W>());      // the padding rows K .. KP - 1 multiply every entry: zeros
        stv<W>(Tm + row * LD + col, i < K * NW ? vv[u] : zerov<W>());
        for (int q = 1; q < Q; ++q)
          if (i >= K * NW) stv<W>(Tm + (q * KP + row) * LD + col, zerov<W>());
      }
    }
    THIN_STAMP(0, 1);
    __syncthreads();
    for (int i = tid; i < cnt * NW; i += 256) {          // (side face, columns): the three rows of the neighbour's element -> C_q * rows
      const int pos = i / NW, col = (i - pos * NW) * W;
      VT v2[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) v2[k] = ldv<W>(Tm + (3 * pos + k) * LD + col);
      for (int q = 0; q < Q; ++q) {
        const double* C = Cl + (q * cnt + pos) * 9;
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
          stv<W>(Tm + (q * KP + 3 * pos + ii) * LD + col, C[ii * 3] * v2[0] + C[ii * 3 + 1] * v2[1] + C[ii * 3 + 2] * v2[2]);
      }
    }
  };
  if (N & 1) stage(W1{});
  else stage(W2{});
  __syncthreads();
  THIN_STAMP(0, 2);
  for (int q = 0; q < Q; ++q) {
    double* out = B_sys + ((((long)q * S + s) * 5 + slot) * N) * N;
    const double* Tq = Tm + q * KP * LD;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int tile = wave + 4 * k;
      if (tile >= NTX * NTX) continue;           // wave-uniform
      const int ti = tile / NTX, tj = tile - ti * NTX;
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      // (unrolled by 8 k-steps with the operands of all of them requested first: a rolled loop pays one LDS round trip per MFMA)
      for (int k0 = 0; k0 < KP; k0 += 32) {
        double xa[8], yb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int kr = k0 + 4 * u + lk < KP ? k0 + 4 * u + lk : KP - 1;
          xa[u] = Xin[kr * LD + ti * 16 + li];
          yb[u] = Tq[kr * LD + tj * 16 + li];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 4 * u < KP) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], yb[u], acc, 0, 0, 0);      // wave-uniform
      }
      const int col = tj * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + lk + 4 * r;
        if (row < N && col < N) out[(long)row * N + col] = acc[r];
      }
    }
  }
  THIN_STAMP(0, 3);
}

template <int NTX>
__global__ __launch_bounds__(256) void k_coupling(Tmpl t, int S, const int* __restrict__ nbr, int Q, int N,
                                                  const double* __restrict__ V, const double* __restrict__ A_cpl,
                                                  double* __restrict__ B_sys) {
  coupling_body<NTX>(t, S, nbr, Q, N, V, A_cpl, B_sys, blockIdx.x, subdomain_of(t, blockIdx.y));
}

// Factored layout: all three thin kernels are 256-thread workgroups -- one launch, grid (4, S, 3).
template <int NTX>
__global__ __launch_bounds__(256) void k_thin3(Tmpl t, ThinRtArgs a, ThinNcfArgs f, const double* __restrict__ A_cpl,
                                               double* __restrict__ B_sys) {
  // (an XCD-aware 1-D grid -- the twelve workgroups of a subdomain on one XCD, kinds interleaved -- was measured: no change)
  const int side = blockIdx.x, s = subdomain_of(t, blockIdx.y), z = blockIdx.z;
  if (z == 0)
    coupling_body<NTX>(t, a.S, a.nbr, a.Q, a.N, a.V, A_cpl, B_sys, side, s);
  else if (z == 1)
    thin_rt_body(t, a, side, s);
  else
    thin_ncf_body(t, f, side, s);
}

// The three thin kernels of a pass in one launch, for small subdomain counts: grid (4, S, 3), z = 0 the nonconformity
// side blocks (512 threads), z = 1 the coupling projection, z = 2 the flux side factors (256 threads each: the upper four
// waves leave at once -- a finished wave is not waited for by s_barrier).  Long workgroups first.
struct ThinNcArgs {
  const double *ebar, *AvgSelf, *AvgSide, *A_cpl;
  double *G_nc, *B_sys;
  double* Fnc;   // factored layout: the side factors of G_nc instead of its side blocks
};
template <int NTX>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NTX <= 3 ? 5 : 4, 8))) void k_thin(Tmpl t, ThinRtArgs a, ThinNcArgs c) {
  const int side = blockIdx.x, s = subdomain_of(t, blockIdx.y);
  if (blockIdx.z == 0 && c.Fnc == nullptr) {
    thin_nc_body<NTX>(t, a.S, a.nbr, a.N, a.V, c.ebar, c.AvgSelf, c.AvgSide, c.G_nc, side, s);
    return;
  }
  if (threadIdx.x >= 256) return;
  if (blockIdx.z == 0) {
    const ThinNcfArgs f{a.V, c.ebar, c.AvgSelf, c.AvgSide, a.nbr, c.Fnc, a.N, a.S};
    thin_ncf_body(t, f, side, s);
    return;
  }
  if (blockIdx.z == 1)
    coupling_body<NTX>(t, a.S, a.nbr, a.Q, a.N, a.V, c.A_cpl, c.B_sys, side, s);
  else
    thin_rt_body(t, a, side, s);
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace


int64_t fused_work_size(lrbms_ctx* ctx, int Q, int N) {
  const Tmpl& t = ctx->t;
  const long nvs = t.nvx > t.nvy ? t.nvx : t.nvy;
  return (long)ctx->S * t.nrt * Q * N + (long)ctx->S * 4 * t.ncf * Q * N + (long)ctx->S * t.nv * N + (long)ctx->S * 4 * nvs * N +
         (long)ctx->S * 4 * t.ncf * fside_ld(Q, N) +   // side factors of the dense layout (the factored layout returns them)
         (long)Q * ctx->S * t.nT * 6;                  // W'^q_T, the rank-2 factors of the df_ab element blocks (k_prep_lds -> k_f1w)
}

int64_t fused_fside_size(lrbms_ctx* ctx, int Q, int N) { return (long)ctx->S * 4 * ctx->t.ncf * fside_ld(Q, N); }

int64_t fused_fnc_size(lrbms_ctx* ctx, int N) {
  const int nvs = ctx->t.nvx > ctx->t.nvy ? ctx->t.nvx : ctx->t.nvy;
  return (long)ctx->S * 4 * nvs * fnc_ld(nvs, N, ctx->t.opt_oswald_vertex);
}

namespace {
__global__ __launch_bounds__(256) void k_build_tables(Tmpl t, double* __restrict__ stiff, int* __restrict__ tvtx,
                                                      int* __restrict__ tpos, int* __restrict__ tmask, int* __restrict__ sfi,
                                                      double* __restrict__ sfd, int* __restrict__ trl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nvs_b = t.nvx > t.nvy ? t.nvx : t.nvy;
  if (i < 4 * nvs_b) {                            // the rows that meet in a side vertex, in table order (k_thin_ncf scanned them per workgroup)
    const int sd = i / nvs_b, pos = i - sd * nvs_b;
    int cnt = 0;
    for (int r = 0; r < 3 * t.touch_count[sd]; ++r) {
      const int T = t.touch_elem[sd * t.ntouch + r / 3];
      const int v = t.dof_vertex[3 * T + r % 3], lx = v % t.nvx, ly = v / t.nvx;
      const int ps = sd == 0 ? (ly == 0 ? lx : -1) : sd == 1 ? (lx == 0 ? ly : -1) : sd == 2 ? (lx == t.nvx - 1 ? ly : -1) : (ly == t.nvy - 1 ? lx : -1);
      if (ps == pos) {
        if (cnt < 8) trl[i * 9 + 1 + cnt] = r;
        ++cnt;
      }
    }
    for (int k = cnt; k < 8; ++k) trl[i * 9 + 1 + k] = 0;
    trl[i * 9] = cnt <= 8 ? cnt : -1;
  }
  if (i < 4 * t.ncf) {                            // side faces: what k_thin_rt resolved by three dependent index chains per workgroup
    const int sd = i / t.ncf, p = i - sd * t.ncf;
    int* o = sfi + i * 12;
    double* d = sfd + i * 4;
    for (int k = 0; k < 12; ++k) o[k] = 0;
    for (int k = 0; k < 4; ++k) d[k] = 0.0;
    if (p < t.side_count[sd]) {
      const int T = t.side_elem[i];
      int fp = 0;
      for (int f = 0; f < 3; ++f)
        if (t.nb_elem[T * 3 + f] == -(1 + sd)) fp = f;
      o[0] = T;
      o[1] = fp;
      for (int g = 0; g < 3; ++g) {
        const int nb = t.nb_elem[T * 3 + g];
        o[2 + g] = t.elem_rt[T * 3 + g];
        o[5 + g] = t.face_sign[T * 3 + g];
        o[8 + g] = nb < 0 ? -1 - nb : -1;
        d[g] = t.face_len[T * 3 + g];
      }
      d[3] = t.area[T];
    }
  }
  if (i < t.nT) {
    double K[9];
    stiffness3(t, i, K);
    for (int k = 0; k < 9; ++k) stiff[i * 9 + k] = K[k];
    double* mass9 = stiff + 9 * t.nT;
    const double m = t.area[i] * (1.0 / 12.0);
    for (int k = 0; k < 9; ++k) mass9[i * 9 + k] = (k % 4 == 0) ? m + m : m;
    // rank-2 factors of the element blocks that act through the gradients (k_f1w): kappa = L L^T (lower Cholesky factor of the
    // symmetrised tensor), Z = L^T G and H = L^-1 (G G^T)^-1 G with G = [g_0 g_1 g_2] (2 x 3)
    double* lgz = stiff + 27 * t.nT + 64;
    double* hab = lgz + 6 * t.nT;
    const double ksym = 0.5 * (t.kappa[1] + t.kappa[2]);
    const double l00 = sqrt(t.kappa[0]), l10 = ksym / l00, l11 = sqrt(t.kappa[3] - l10 * l10);
    double gx[3], gy[3], m00 = 0.0, m01 = 0.0, m11 = 0.0;
    for (int k = 0; k < 3; ++k) {
      gx[k] = t.grad[(i * 3 + k) * 2];
      gy[k] = t.grad[(i * 3 + k) * 2 + 1];
      m00 += gx[k] * gx[k];
      m01 += gx[k] * gy[k];
      m11 += gy[k] * gy[k];
    }
    const double det = m00 * m11 - m01 * m01;
    for (int k = 0; k < 3; ++k) {
      lgz[i * 6 + k] = l00 * gx[k] + l10 * gy[k];
      lgz[i * 6 + 3 + k] = l11 * gy[k];
      const double ux = (m11 * gx[k] - m01 * gy[k]) / det, uy = (m00 * gy[k] - m01 * gx[k]) / det;      // (G G^T)^-1 g_k
      hab[i * 6 + k] = ux / l00;                                                                            // L^-1 u: forward substitution
      hab[i * 6 + 3 + k] = (uy - l10 * (ux / l00)) / l11;
    }
  }
  for (int k = i; k < 9 * t.nT + 64; k += gridDim.x * blockDim.x) stiff[18 * t.nT + k] = 0.0;
  if (i < 4 * t.ntouch) {
    const int sd = i / t.ntouch, p = i - sd * t.ntouch;
    int m = 0;
    if (p < t.touch_count[sd]) {
      const int T = t.touch_elem[i];
      for (int d = 0; d < 3; ++d) {
        const int v = t.dof_vertex[3 * T + d], lx = v % t.nvx, ly = v / t.nvx;
        m |= (ly == 0 ? 1 : 0) | (lx == 0 ? 2 : 0) | (lx == t.nvx - 1 ? 4 : 0) | (ly == t.nvy - 1 ? 8 : 0);
        tvtx[i * 3 + d] = v;
        tpos[(i * 3 + d) * 4 + 0] = ly == 0 ? lx : -1;
        tpos[(i * 3 + d) * 4 + 1] = lx == 0 ? ly : -1;
        tpos[(i * 3 + d) * 4 + 2] = lx == t.nvx - 1 ? ly : -1;
        tpos[(i * 3 + d) * 4 + 3] = ly == t.nvy - 1 ? lx : -1;
      }
    } else {
      for (int d = 0; d < 3; ++d) {
        tvtx[i * 3 + d] = 0;
        for (int q = 0; q < 4; ++q) tpos[(i * 3 + d) * 4 + q] = -1;
      }
    }
    tmask[i] = m;
  }
}
}  // namespace

int build_template_tables(lrbms_ctx* ctx) {
  Tmpl& t = ctx->t;
  double* stiff = nullptr;
  int *tvtx = nullptr, *tpos = nullptr, *tmask = nullptr;
  LRBMS_HIP_CHECK(ctx, hipMalloc(&stiff, sizeof(double) * (39 * (size_t)t.nT + 64)));   // stiff | mass9 | zeros [9 nT + 64] | lgz [6 nT] | hab [6 nT]
  ctx->owned.push_back(stiff);
  LRBMS_HIP_CHECK(ctx, hipMalloc(&tvtx, sizeof(int) * 12 * (size_t)t.ntouch));
  ctx->owned.push_back(tvtx);
  LRBMS_HIP_CHECK(ctx, hipMalloc(&tpos, sizeof(int) * 48 * (size_t)t.ntouch));
  ctx->owned.push_back(tpos);
  LRBMS_HIP_CHECK(ctx, hipMalloc(&tmask, sizeof(int) * 4 * (size_t)t.ntouch));
  ctx->owned.push_back(tmask);
  int* sfi = nullptr;
  double* sfd = nullptr;
  LRBMS_HIP_CHECK(ctx, hipMalloc(&sfi, sizeof(int) * 48 * (size_t)t.ncf));
  ctx->owned.push_back(sfi);
  LRBMS_HIP_CHECK(ctx, hipMalloc(&sfd, sizeof(double) * 16 * (size_t)t.ncf));
  ctx->owned.push_back(sfd);
  const int nvs_b = t.nvx > t.nvy ? t.nvx : t.nvy;
  int* trl = nullptr;
  LRBMS_HIP_CHECK(ctx, hipMalloc(&trl, sizeof(int) * 36 * (size_t)nvs_b));
  ctx->owned.push_back(trl);
  const int total = std::max(std::max(64, std::max(4 * t.ncf, 4 * nvs_b)), t.nT > 4 * t.ntouch ? t.nT : 4 * t.ntouch);
  hipLaunchKernelGGL(k_build_tables, dim3((total + 255) / 256), dim3(256), 0, nullptr, t, stiff, tvtx, tpos, tmask, sfi, sfd, trl);
  LRBMS_LAUNCH_CHECK(ctx);
  LRBMS_HIP_CHECK(ctx, hipDeviceSynchronize());
  t.stiff = stiff;
  t.mass9 = stiff + 9 * (size_t)t.nT;
  t.zero64 = stiff + 18 * (size_t)t.nT;
  t.lgz = stiff + 27 * (size_t)t.nT + 64;
  t.hab = t.lgz + 6 * (size_t)t.nT;
  t.touch_vtx = tvtx;
  t.touch_pos = tpos;
  t.touch_mask = tmask;
  t.sface_i = sfi;
  t.sface_d = sfd;
  t.touch_rlist = trl;
  return LRBMS_OK;
}

// (row tiles, Q, levels with 1 / 2 / 3 live row tiles) combinations of the lean projection kernel k_f1v that are compiled in;
// everything else runs k_f1u.  Q = 2, N = 20 (config 2).  (Round 3 also compiled <3,2,1,2,4> and <3,2,1,2,3> for N = 34 .. 40: that
// shape is k_f1w's now -- the rank-2 form, same structure, 26 % fewer projection MFMAs -- and the two instantiations are retired.)
#define LRBMS_F1V_LIST(X) X(2, 2, 1, 3, 0)
static bool f1v_instantiated(int ntx, int Q, const int lv[3]) {
#define LRBMS_F1V_HAS(A, B, C, D, E) if (ntx == A && Q == B && lv[0] == C && lv[1] == D && lv[2] == E) return true;
  LRBMS_F1V_LIST(LRBMS_F1V_HAS)
#undef LRBMS_F1V_HAS
  return false;
}

// Column layout of Y for k_f1v.  A symmetric group (B_sys diagonal, E_red, M_red, G_aa[q][q]) is cut into blocks of 16 columns;
// block j needs the row tiles 0 .. j only (entries below the diagonal are mirror images), so its tile has "class" j + 1; the
// short last blocks (N - 16 (ntx - 1) columns) of all symmetric groups are packed back to back into shared tiles of class ntx,
// the unsymmetric groups follow contiguously (class ntx).  Tiles in ascending class, four per level (one per SIMD); a level's
// class is that of its last tile.  Fills Grp::cb / sym and lv = levels of class 1 / 2 / 3; false if the columns do not fit.
static bool f1v_layout(std::vector<Grp>& groups, int N, int ntx, int lv[3]) {
  int ns = 0, nu = 0;
  for (Grp& g : groups) {
    g.sym = (g.kind == G_SYS || g.kind == G_ENERGY || g.kind == G_MASS || (g.kind == G_AA && g.q == g.q2)) ? 1 : 0;
    (g.sym ? ns : nu) += 1;
  }
  const int tail = N - 16 * (ntx - 1);
  const int ntail = (ns * tail + 15) / 16, nuns = (nu * N + 15) / 16;
  const int nblock = ns * (ntx - 1), T = nblock + ntail + nuns;
  if (T > 4 * F1_NTY || ntx > 3) return false;
  int si = 0, ui = 0;
  for (Grp& g : groups) {
    if (g.sym) {
      for (int j = 0; j < 4; ++j) g.cb[j] = j < ntx - 1 ? 16 * (j * ns + si) : 16 * nblock + si * tail;
      ++si;
    } else {
      for (int j = 0; j < 4; ++j) g.cb[j] = 16 * (nblock + ntail) + ui * N + 16 * j;
      ++ui;
    }
  }
  lv[0] = lv[1] = lv[2] = 0;
  for (int l = 0; 4 * l < T; ++l) {
    const int k = std::min(4 * l + 3, T - 1);
    lv[(k < nblock ? k / ns + 1 : ntx) - 1] += 1;
  }
  return true;
}

// W'^q_T = hab_T A_ab,T^q: one thread per (q, s, T, d, f).  Launched by lrbms_assemble_products beside Aab, and by a fused pass that
// is handed an Aab the context holds no factors for.
namespace {
__global__ __launch_bounds__(256) void k_wab(Tmpl t, long total, const double* __restrict__ Aab, double* __restrict__ Wab) {
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const long blk = it / 6;                           // (q, s, T)
    const int k = (int)(it - blk * 6), d = k / 3, f = k - d * 3, T = (int)(blk % t.nT);
    const double* ab = Aab + blk * 9;
    const double* h = t.hab + T * 6 + d * 3;
    Wab[it] = (h[0] * ab[f] + h[1] * ab[3 + f]) + h[2] * ab[6 + f];
  }
}
}  // namespace
int launch_wab(lrbms_ctx* ctx, int Q, const double* Aab, double* Wab, hipStream_t st) {
  const long total = (long)Q * ctx->S * ctx->t.nT * 6;
  KScope ks(ctx, "k_wab", st);
  hipLaunchKernelGGL(k_wab, dim3(grid_for(total)), dim3(256), 0, st, ctx->t, total, Aab, Wab);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

static bool f1w_usable(lrbms_ctx* ctx, int Q, int N) { return Q == 2 && N % 2 == 0; }

// Column layout of Y for k_f1w (Q = 2, three row tiles, even N in [34, 40]): the tiles of F1wPlan.  Symmetric groups of kind 3
// (B_sys diagonal q = 0, 1, E_red, M_red; index si): block 0 -> tile si (level 0), block 1 -> tile 4 + si (level 1), the tails packed
// into tiles 8, 9 (level 2, SIMDs 0, 1).  Symmetric groups of kind 2 (G_aa[0][0], G_aa[1][1]; index sj): block 0 -> tile 12 + sj
// (level 3, SIMDs 0, 1), block 1 -> tile 10 + sj (level 2, SIMDs 2, 3), the tails packed into tile 14.  The unsymmetric groups
// (G_aa[0][1], the four G_ab) contiguously from tile 15 on.  false if the shape is not the plan's.
static bool f1w_layout(std::vector<Grp>& groups, int N, int Q) {
  const int ntx = (N + 15) / 16, tail = N - 32;
  if (Q != 2 || ntx != 3 || N % 2 != 0 || tail < 2 || tail > 8 || groups.size() != 11) return false;
  int s3 = 0, s2 = 0, ui = 0;
  for (Grp& g : groups) {
    g.sym = (g.kind == G_SYS || g.kind == G_ENERGY || g.kind == G_MASS || (g.kind == G_AA && g.q == g.q2)) ? 1 : 0;
    for (int j = 0; j < 4; ++j) g.cb[j] = 0;
    if (g.sym && g.kind != G_AA) {
      g.cb[0] = 16 * s3;
      g.cb[1] = 16 * (4 + s3);
      g.cb[2] = 16 * 8 + s3 * tail;
      ++s3;
    } else if (g.sym) {
      g.cb[0] = 16 * (12 + s2);
      g.cb[1] = 16 * (10 + s2);
      g.cb[2] = 16 * 14 + s2 * tail;
      ++s2;
    } else {
      for (int j = 0; j < 4; ++j) g.cb[j] = 16 * 15 + ui * N + 16 * j;
      ++ui;
    }
  }
  return s3 == 4 && s2 == 2 && ui == 5 && 16 * 15 + 5 * N <= 16 * 28;
}

// v_mfma_f64_16x16x4_f64 instructions the dense projection kernel (k_f1v / k_f1u / k_f1, whichever the launcher takes for this
// shape and these options) executes per subdomain; 0 if the fused pass does not support the shape.  For the roofline of bench.py.
long f1_mfma_per_subdomain(lrbms_ctx* ctx, int Q, int N) {
  const Tmpl& t = ctx->t;
  if (N < 1 || N > 64 || Q < 1 || Q > 4) return 0;
  const int ntx = (N + 15) / 16, ng = Q + 2 + Q * (Q + 1) / 2 + Q * Q, nch = t.nT / EC;
  std::vector<Grp> groups;
  for (int q = 0; q < Q; ++q) groups.push_back({G_SYS, q, 0});
  groups.push_back({G_ENERGY, 0, 0});
  groups.push_back({G_MASS, 0, 0});
  for (int q = 0; q < Q; ++q)
    for (int q2 = q; q2 < Q; ++q2) groups.push_back({G_AA, q, q2});
  for (int q = 0; q < Q; ++q)
    for (int q2 = 0; q2 < Q; ++q2) groups.push_back({G_AB, q, q2});
  const bool one_slice = ng <= std::min(F1_MAXG, (4 * F1_NTY * 16) / N);
  const bool unified = (Q == 1 || Q == 2) && one_slice && ntx <= 3 && ctx->opt_f1_legacy != 1;
  int lv[3] = {0, 0, 0};
  if (unified && ctx->opt_f1_legacy == 0 && f1w_usable(ctx, Q, N) && f1w_layout(groups, N, Q))
    return (long)nch * 150 + (long)t.nT * (3 * ntx + Q * ntx);      // k_f1w: 18 tile rows x 3 + 48 x 2 k-steps per chunk; the two applies
  if (unified && ctx->opt_f1_legacy != 2 && N % 2 == 0 && N >= 2 && f1v_layout(groups, N, ntx, lv) && f1v_instantiated(ntx, Q, lv))
    return (long)nch * 3 * 4 * (lv[0] + 2 * lv[1] + 3 * lv[2]) + (long)t.nT * (3 * ntx + Q * ntx);      // projection + the two applies
  const int tiles = (ng * N + 15) / 16;                         // k_f1u skips the column tiles beyond the last column
  return (long)nch * 3 * ntx * tiles + (long)t.nT * 3 * ntx;
}

static size_t thin_nc_lds_bytes(const Tmpl& t, int ntx) {   // Wa, Yc, Ksc, ttab / vtab / ptab / side_mask of k_thin_nc
  const size_t kp = (size_t)((3 * t.ntouch + 3) & ~3);
  return sizeof(double) * (kp * 2 * padded_ld(ntx) + 9 * (size_t)t.ntouch) + sizeof(int) * 17 * (size_t)t.ntouch;
}

// Which (template, Q, N) the fused pass can run, per output layout.  Everything is a question of LDS (160 KB per
// workgroup on gfx950): the factored layout needs less of it (no N x N side blocks are formed), so large templates
// (k_c = 16: 2 048 elements per subdomain) run fused in that layout only.
bool fused_supported(lrbms_ctx* ctx, int Q, int N, bool factored) {
  const Tmpl& t = ctx->t;
  constexpr size_t LDS_MAX = 160 * 1024;
  if (N > 64 || Q > 4 || Q * N > 128 || t.nT % 8 != 0 || t.ntouch > 256) return false;
  const int ntx = (N + 15) / 16, nr = (Q * N + 15) / 16, nch = t.nT / EC;
  // k_f1: the unified kernel splits the element range until its stiffness table fits beside the staging buffers; the
  // producer / consumer kernel (N > 48 or Q > 2) keeps the whole adjacency + stiffness tables (96 bytes per element)
  const int ngroups = Q + 2 + Q * (Q + 1) / 2 + Q * Q;
  const bool unified = (Q == 1 || Q == 2) && ntx <= 3 && ngroups <= std::min(F1_MAXG, (4 * F1_NTY * 16) / N);
  if (unified) {
    int ks = 1;
    while (72 * (size_t)(t.nT / ks) > 56 * 1024 && nch % (4 * ks) == 0) ks *= 2;
    if (72 * (size_t)(t.nT / ks) > 56 * 1024) return false;
  } else if (96 * (size_t)t.nT > 48 * 1024) {
    return false;
  }
  static const size_t f2_static[8] = {12288, 28672, 28672, 45056, 45056, 61440, 61440, 77824};   // k_f2<NR> (see the .s)
  if (44 * (size_t)t.nT + f2_static[nr - 1] > LDS_MAX) return false;
  if (coupling_lds_bytes(t, ntx, Q) > LDS_MAX) return false;
  if (factored) return thin_ncf_lds_bytes(t, N) <= LDS_MAX;
  if (thin_nc_lds_bytes(t, ntx) > LDS_MAX || t.ntouch * N > 3 * 512) return false;   // k_thin_nc: LDS, items per thread
  if ((size_t)t.ncf * fside_ld(Q, N) * sizeof(double) > LDS_MAX) return false;       // k_thin_expand
  return true;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per (device, kernel): raised once per process, not set on every launch of the
// pass (a runtime call of ~2 us on the chain preparation -> projection kernel of every step)
static hipError_t raise_max_lds(int device, const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> raised;
  std::lock_guard<std::mutex> lock(mu);
  int& cur = raised[{device, fn}];
  if (cur >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) cur = bytes;
  return e;
}

int launch_project_estimate_fused(lrbms_ctx* ctx, int Q, int N, const double* V, const double* F, const double* A_diag,
                                  const double* A_cpl, const double* P_diag, const double* b, const double* ebar,
                                  const double* caa, const double* Aab, const double* Bbb, double* work, double* B_sys,
                                  double* rhs_red, double* E_red, double* M_red, double* G_nc, double* r_fd, double* G_rdd,
                                  double* G_bb, double* G_ab, double* G_aa, double* Fside, double* Fnc, int phase,
                                  hipStream_t st) {
  // Fside == nullptr: dense layout -- G_rdd / G_bb block-compact [S][9][QN][QN], G_ab [Q][S][N][5QN]; the side factors
  // stay in `work` and k_thin_expand writes the side blocks from them.
  // Fside != nullptr: factored layout -- G_rdd / G_bb [S][QN][QN] and G_ab [Q][S][N][QN] hold the self parts only, every
  // block that involves a neighbour slot is represented by F_side [S][4][ncf][4 QN + 4] (see k_thin_rt); G_nc [S][N][N] is
  // its [self, self] block and F_nc [S][4][nvs][2 N + 4 nvs] the factors of the others (see k_thin_ncf).
  // phase 0: the whole pass.  phase 1 / 2: its halo-independent / halo-dependent halves, for a sharded run that overlaps
  // the halo exchange with phase 1 (everything that reads only the rank's own basis slabs: R_self, Avg_self, k_f1,
  // k_f2, k_f3 -- more than half of the pass); phase 2 then needs the halo slabs of V (R_side, Avg_side, the thin
  // kernels, the coupling blocks).  1 followed by 2 gives bit-identical results to 0.
  if (!fused_supported(ctx, Q, N, Fside != nullptr))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "fused pass: unsupported N / Q / template size for this output layout");
  // phase 3 / 4: phase 1 split once more into its preparation (R_self, Avg_self) and its dense kernels (k_f1, k_f2,
  // k_f3), so that a host can record an event between them and start phase 2 on another stream as soon as the halo has
  // arrived, while the dense kernels are still running (Engine.project_and_estimate with `halo=`).
  // phase 5: 1 and 2 in ONE call for the sharded step -- phase 1 on the caller's stream, phase 2 on library stream 0 behind whatever
  // the caller has queued there (the wait for the halo exchange and its unpack), both joined into the caller's stream at the end:
  // one fork and one join for the step instead of a fork / join per call and an event pair of the host's (the host side of a step is
  // what bounds a rank with few subdomains, tools/host_step_time.py).  The same kernels with the same arguments as 1 then 2.
  if (phase < 0 || phase > 5) return lrbms_fail(ctx, LRBMS_E_INVALID, "fused pass: phase must be 0 .. 5");
  if (phase == 5 && st == ctx->aux[0])
    return lrbms_fail(ctx, LRBMS_E_INVALID, "fused pass: phase 5 runs its halo-dependent half on library stream 0; the caller's stream must be another one");
  ctx->pass_ran = true;
  const bool both = phase == 5;
  const bool do_prep = phase == 0 || phase == 1 || phase == 3 || both;
  const bool do_a = phase == 0 || phase == 1 || phase == 4 || both;      // the dense, halo-independent kernels
  const bool do_b = phase == 0 || phase == 2 || both;
  // incremental re-projection (lrbms_fused_set_subset): Sg workgroup rows for the listed subdomains; S stays the stride of every
  // array.  Launch POLICY (forked launches, two preparation workgroups per subdomain) follows Sg -- every policy gives the same
  // bits --, the K-split of the projection kernel follows S: a split changes the summation order, and a subset pass must
  // reproduce the bits of the whole pass.
  Tmpl t = ctx->t;
  t.sub_list = ctx->subset_n > 0 ? ctx->subset : nullptr;
  t.sub_count = ctx->subset_n;
  const int S = ctx->S, QN = Q * N, C = 5 * QN;
  const int Sg = t.sub_list ? t.sub_count : S;
  const long nvs = t.nvx > t.nvy ? t.nvx : t.nvy;
  double* Rself = work;
  double* Rside = Rself + (long)S * t.nrt * QN;
  double* AvgSelf = Rside + (long)S * 4 * t.ncf * QN;
  double* AvgSide = AvgSelf + (long)S * t.nv * N;
  const bool factored = Fside != nullptr;
  if (factored != (Fnc != nullptr)) return lrbms_fail(ctx, LRBMS_E_INVALID, "fused pass: F_side and F_nc go together");
  if (t.opt_oswald_vertex && !factored)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "LRBMS_OPT_OSWALD_VERTEX_PATCH: the diagonal subdomains enter through the factored layout "
                                            "only (F_nc)");
  if (t.opt_oswald_vertex && ctx->S_ext != ctx->S && !ctx->diag_explicit)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "LRBMS_OPT_OSWALD_VERTEX_PATCH on a sharded grid: the diagonal subdomains must be halo "
                                            "slabs named by lrbms_set_diagonal_neighbours");
  if (!factored) Fside = AvgSide + (long)S * 4 * nvs * N;
  const long gstride = factored ? (long)QN * QN : (long)9 * QN * QN;      // self blocks of G_bb / G_rdd
  const int abld = factored ? QN : C;                                     // row length of G_ab
  const int aboff = factored ? 0 : 2 * QN;                                // column offset of its self part
  // Small per-rank counts ("forked" mode, S < 192): no kernel fills the chip alone and the pass is bound by launches and
  // by the longest dependency chain, not by work.  The two preparation sweeps then go out as ONE launch (k_prep /
  // k_prep_side), the three thin kernels as ONE launch (k_thin) on a library stream, k_f2 and k_f3 on two more, k_f1 on
  // the caller's stream: five launches and one fork / join instead of twelve launches and two forks (config 2: 86 us of
  // host enqueue per pass for 104 us of device time before).
  const bool forked = ctx->opt_streams >= 0 ? ctx->opt_streams != 0 : Sg < 192;      // LRBMS_OPT_STREAMS
  const int gy_flux = (t.nrt * N + 255) / 256, gy_vtx = (t.nv * N + 255) / 256;   // (2 - 8 items per thread instead: no faster)
  // (k_prep only there: at 1 024 subdomains it takes 182 us against 113 + 59 us for the two sweeps on their own)
  // factored layout: the three thin kernels are 256-thread workgroups and always share one launch (k_thin3: 141 us at
  // 1 024 subdomains against 65 + 49 + 41 us one after the other -- they are latency-bound and fill each other's gaps)
  const bool merge_thin = forked || factored;
  // LRBMS_OPT_PREP_LDS: the preparation sweeps from one copy of the basis slab in LDS; with it G_nc[self, self] is folded into the same
  // kernel (k_f3 is not launched) whenever N <= 48 and the call covers both the preparation and the dense kernels
  const int ntx_p = (N + 15) / 16;
  const bool prep_ok = ctx->opt_prep_lds != 0 && N % 2 == 0;
  const bool gnc_fold = prep_ok && ctx->opt_prep_lds != 2 && ntx_p <= 3 && prep_lds_bytes(t, Q, N, true) <= 160 * 1024;      // (3: as 1)
  const size_t prep_lds = prep_lds_bytes(t, Q, N, gnc_fold);
  const bool prep_from_lds = prep_ok && prep_lds <= 160 * 1024;
  bool side_from_lds = false;      // phase 2: k_prep_lds<side> wrote R_side AND Avg_side
  // the rank-2 form of the projection kernel (k_f1w; LRBMS_OPT_F1_FORM 0 only) and the factors W' it multiplies the flux rows with:
  // those lrbms_assemble_products left with the context for this Aab, or formed here (work buffer) for any other
  const double* Wab = nullptr;
  bool f1w_ok = false;
  if (ctx->opt_f1_legacy == 0 && do_a) {
    std::vector<Grp> probe;
    for (int i = 0; i < 11; ++i) probe.push_back({i < 2 ? G_SYS : i == 2 ? G_ENERGY : i == 3 ? G_MASS : i < 7 ? G_AA : G_AB, i == 6 ? 1 : 0, i == 5 || i == 6 ? 1 : 0});
    f1w_ok = Q == 2 && f1w_layout(probe, N, Q);
    if (f1w_ok) {
      if (ctx->wab != nullptr && ctx->wab_src == Aab && ctx->wab_Q == Q) {
        Wab = ctx->wab;
      } else {
        double* wloc = AvgSide + (long)S * 4 * nvs * N + (long)S * 4 * t.ncf * fside_ld(Q, N);
        if (int rc = launch_wab(ctx, Q, Aab, wloc, st)) return rc;
        Wab = wloc;
      }
    }
  }
  if (do_prep) {
    if (prep_from_lds) {
      KScope ks(ctx, "k_prep_lds", st);
      if (ctx->num_cus == 0) {
        int ncu = 0;
        LRBMS_HIP_CHECK(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
        ctx->num_cus = ncu > 0 ? ncu : 256;
      }
      const int prep_parts = 2 * Sg <= ctx->num_cus ? 2 : 1;      // one workgroup per CU: two per subdomain while that leaves none waiting
      const GncArgs ga{ebar, gnc_fold ? G_nc : nullptr, factored ? (long)N * N : (long)25 * N * N, factored ? N : 5 * N,
                       factored ? 0 : 10 * N * N + 2 * N};
      // More subdomains than CUs: one workgroup per CU that takes its subdomains one after the other and requests the next slab
      // while it works on the current one (see the kernel); LRBMS_OPT_PREP_LDS 3 keeps one workgroup per subdomain (the same bits).
      const size_t prep_lds_p = prep_lds_bytes_persistent(t, Q, N, gnc_fold);
      // (with the G_nc fold only, and every wave with a K part of its own in it: the waves of the neighbours' shares request from
      // inside its MFMA loop)
      const bool persist = prep_parts == 1 && Sg > ctx->num_cus && ctx->opt_prep_lds != 3 && prep_lds_p <= 160 * 1024 && gnc_fold &&
                           7 * ((t.nT / 2 + 7) / 8) < t.nT / 2 &&
                           (size_t)t.n * (N / 2) <= 8 * (size_t)PREP_LDS_THREADS && (size_t)Q * t.nrt * 3 <= 2 * (size_t)PREP_LDS_THREADS &&
                           (size_t)Q * S * t.nrt * 48 < ((size_t)1 << 31);
      const size_t lds_used = persist ? prep_lds_p : prep_lds;
      const int gx = persist ? ctx->num_cus : Sg;
#define LRBMS_PREP(NTXV)                                                                                                              \
  do {                                                                                                                                \
    LRBMS_HIP_CHECK(ctx, raise_max_lds(ctx->device, (const void*)k_prep_lds<NTXV>, (int)lds_used));                                   \
    hipLaunchKernelGGL(k_prep_lds<NTXV>, dim3(gx, prep_parts), dim3(PREP_LDS_THREADS), lds_used, st, t, S, ctx->nbr, Q, N, F, V, Rself, Rside, AvgSelf, \
                       AvgSide, phase == 0 ? 1 : 0, ga, persist ? 1 : 0);                                                             \
  } while (0)
      if (ntx_p == 1) LRBMS_PREP(1);
      else if (ntx_p == 2) LRBMS_PREP(2);
      else LRBMS_PREP(3);
#undef LRBMS_PREP
    } else {      // (odd N, or a slab beyond the LDS: the two streaming sweeps, at every subdomain count -- their merged form k_prep is retired)
      {
        KScope ks(ctx, "k_flux_compact", st);
        hipLaunchKernelGGL(k_flux_compact, dim3(Sg, gy_flux), dim3(256), 0, st, t, S, ctx->nbr, Q, N, F, V, Rself, Rside,
                           phase == 0 ? 1 : 0);
      }
      KScope ks(ctx, "k_vertex_avg", st);
      hipLaunchKernelGGL(k_vertex_avg, dim3(Sg, gy_vtx), dim3(256), 0, st, t, S, ctx->nbr, N, V, AvgSelf, AvgSide,
                         phase == 0 ? 1 : 0);
    }
    // sharded pass: phase 2 (on another stream, once the halo is there) reads what the preparation wrote; the library orders
    // the two itself, so that the host needs neither a call boundary nor an event of its own between preparation and dense kernels
    // (recorded below, where the fork event of a forked call -- the same point of the stream -- serves as well)
  }
  // the neighbours' shares of the flux image and of the vertex averages (phase 2; phase 5: on library stream 0, behind the fork)
  auto side_prep = [&](hipStream_t ss) -> int {
    // (the same predicate as the whole pass and phase 1: where THEY prepare by the streaming sweeps -- a slab beyond the LDS -- the
    // neighbours' shares come from the streaming code too, so "1 then 2 == 0" compares one code with itself)
    if (prep_from_lds && prep_lds_bytes(t, Q, N, false) - prep_lds_slab_bytes(t, N, false) <= 64 * 1024) {
      side_from_lds = true;
      // the neighbours' shares by the side threads' code of k_prep_lds in its slab-less form (one 256-thread workgroup per subdomain,
      // per-vertex data and row tables resolved once per workgroup; several workgroups per CU, so it also fits beside the dense
      // kernels) -- bit-identical to the whole pass by construction
      KScope ks(ctx, "k_prep_lds<side>", ss);
      const GncArgs ga{ebar, nullptr, 0, 0, 0};
      const size_t side_lds = prep_lds_bytes(t, Q, N, false) - prep_lds_slab_bytes(t, N, false);      // tables + coefficients: no slab
      hipLaunchKernelGGL((k_prep_lds<1, 256>), dim3(Sg, 1), dim3(256), side_lds, ss, t, S, ctx->nbr, Q, N, F, V, Rself, Rside, AvgSelf,
                         AvgSide, 2, ga, 0);
    } else {
      {
        KScope ks(ctx, "k_flux_side", ss);
        hipLaunchKernelGGL(k_flux_side, dim3(grid_for((long)Sg * 4 * t.ncf * N)), dim3(256), 0, ss, t, S, ctx->nbr, Q, N, F, V, Rside);
      }
      // Avg_side: the dense layout launches k_vertex_side right in front of its only reader, k_thin_nc (below); in the factored
      // layout the reader is k_thin3, which is launched from the merged branch -- so it goes out here.  (Round 3: it went out
      // NOWHERE in that combination -- factored layout, >= 192 subdomains per rank, phase 2 -- and the phased-vs-whole test did not
      // see it because it reused a work buffer that still held the averages of the whole pass.)
      if (merge_thin) {      // (forked or factored: the reader is k_thin / k_thin3)
        KScope ks(ctx, "k_vertex_side", ss);
        hipLaunchKernelGGL(k_vertex_side, dim3(grid_for((long)Sg * 4 * nvs * N)), dim3(256), 0, ss, t, S, ctx->nbr, N, V, AvgSide);
      }
    }
    return LRBMS_OK;
  };
  if (phase == 2) {
    LRBMS_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->prep_done ? ctx->prep_done : ctx->ev_prep, 0));      // (a no-op if never recorded / same stream)
    if (int rc = side_prep(st)) return rc;
  }
  LRBMS_LAUNCH_CHECK(ctx);
  // fork: F2 / F3, the thin kernels and the coupling projection are independent of each other and of F1 (they all read
  // only V and the two prepare kernels' outputs); on separate streams their workgroups interleave on the CUs, which
  // hides the latencies each of them exposes when it runs alone (they are latency-, not throughput-bound)
  // Measured on MI355X / ROCm 7.2 (config 3 tiles): S = 128: 0.32 ms forked vs 0.41 ms serial (no kernel fills 256 CUs alone);
  // S = 256: 0.53 vs 0.51; S = 512: 0.97 vs 0.93; S = 1024: 1.92 vs 1.81 -> fork only below 192 subdomains per rank.
  // LRBMS_OPT_STREAMS 0 / 1 overrides.
  const bool multi = (do_a || do_b) && forked;      // (phase 5: its halo-dependent half goes to library stream 0 under either policy)
  // forked: caller's stream k_f1 | aux0 k_thin (nonconformity side blocks, coupling projection, flux side factors), k_f3 |
  // aux1 k_f2.  k_f3 goes behind k_thin on library stream 0, not on a stream of its own: one join less, and k_f2 (one long
  // workgroup per subdomain) is not crowded out of the CUs k_f1 leaves free (128 subdomains: 171 vs 178 / 185 us per pass
  // with k_f3 behind k_f2 / on a third stream)
  hipStream_t s_rt = multi || both ? ctx->aux[0] : st, s_f23 = multi ? ctx->aux[1] : st, s_nc = multi || both ? ctx->aux[0] : st;
  hipStream_t s_f3 = multi ? ctx->aux[0] : st;
  // only the library streams that get work in this call are forked and joined (an event operation costs host time, and
  // a sharded step makes three calls on ~0.2 ms of device work); a library stream that IS the caller's stream (the
  // sharded choreography runs phase 2 on stream 0) needs neither.
  // (Replaying the phases as captured hipGraphs instead was measured too: one graph launch costs ~35 us of host time and
  // the replay loses the overlap between the branches, 272 us per step.)
  const bool f3_runs = do_a && !(prep_from_lds && gnc_fold);      // library stream 0 carries the thin kernels and k_f3
  const bool use_aux[3] = {(both || (multi && (do_b || f3_runs))) && ctx->aux[0] != st, multi && do_a && ctx->aux[1] != st, false};
  const bool forks = use_aux[0] || use_aux[1] || use_aux[2];
  if (forks) {
    LRBMS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, st));
    for (int i = 0; i < 3; ++i)
      if (use_aux[i]) LRBMS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->aux[i], ctx->ev_fork, 0));
  }
  if (phase == 1 || phase == 3) {      // what phase 2 waits for: the point behind the preparation (one event packet, not two)
    if (!forks) LRBMS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_prep, st));
    ctx->prep_done = forks ? ctx->ev_fork : ctx->ev_prep;
  }

  // ---- F1: build the column-group list and launch in slices of at most 4 * F1_NTY * 16 / N groups
  std::vector<Grp> groups;
  if (do_a) {
  for (int q = 0; q < Q; ++q) groups.push_back({G_SYS, q, 0, N, B_sys + ((long)q * S * 5 + 2) * N * N, nullptr, (long)5 * N * N});
  groups.push_back({G_ENERGY, 0, 0, N, E_red, nullptr, (long)N * N});
  groups.push_back({G_MASS, 0, 0, N, M_red, nullptr, (long)N * N});
  for (int q = 0; q < Q; ++q)
    for (int q2 = q; q2 < Q; ++q2)   // c^{q q'} is symmetric in (q, q'): G_aa[q'][q] = G_aa[q][q']^T
      groups.push_back({G_AA, q, q2, N, G_aa + ((long)q * Q + q2) * S * N * N,
                        q2 != q ? G_aa + ((long)q2 * Q + q) * S * N * N : nullptr, (long)N * N});
  for (int q = 0; q < Q; ++q)
    for (int q2 = 0; q2 < Q; ++q2)
      groups.push_back({G_AB, q, q2, abld, G_ab + (long)q * S * N * abld + aboff + q2 * N, nullptr, (long)N * abld});
  {
    constexpr int NTY = 7;                                   // 4 consumer waves x 7 column tiles = 448 columns per slice
    const int per = std::min(F1_MAXG, (4 * NTY * 16) / N);
    const int ntx = (N + 15) / 16;
    const bool one_slice = groups.size() <= (size_t)per;   // the straight-line (compile-time Q) producer needs every group
    const size_t ldsf1 = sizeof(int) * 6 * t.nT + sizeof(double) * 9 * t.nT;   // 6 nT ints: 8-byte aligned (nT % 8 == 0)
    for (size_t g0 = 0; g0 < groups.size(); g0 += 3 * (size_t)per) {   // up to three slices per launch (grid.y)
      GrpTable gt[3];
      int nsl = 0;
      for (int sl = 0; sl < 3; ++sl) {
        gt[sl].n = 0;
        const size_t b0 = g0 + (size_t)sl * per;
        if (b0 >= groups.size()) continue;
        gt[sl].n = (int)std::min<size_t>(per, groups.size() - b0);
        for (int i = 0; i < gt[sl].n; ++i) gt[sl].g[i] = groups[b0 + i];
        nsl = sl + 1;
      }
      F1Args a{V, A_diag, P_diag, caa, Aab, Rself, b, g0 == 0 ? rhs_red : nullptr, Q, N, S, nullptr, nullptr};
      const bool legacy = ctx->opt_f1_legacy == 1;   // LRBMS_OPT_F1_FORM 1: the producer / consumer form of the same kernel
      const bool unified = (Q == 1 || Q == 2) && one_slice && ntx <= 3 && !legacy;
      // the lean form (k_f1v): even N, and one of the instantiated (row tiles, Q, column tiles per SIMD) combinations
      int lv[3] = {0, 0, 0};
      const bool lean2 = unified && f1w_ok && f1w_layout(groups, N, Q);      // the rank-2 form (config 3's shape)
      const bool lean = lean2 || (unified && ctx->opt_f1_legacy != 2 && N % 2 == 0 && N >= 2 && f1v_layout(groups, N, ntx, lv) &&
                                  f1v_instantiated(ntx, Q, lv));
      if (lean)
        for (int i = 0; i < gt[0].n; ++i) gt[0].g[i] = groups[g0 + i];      // with the column map filled in
      // K-split: a rank with few subdomains spreads the element range of a subdomain over up to four workgroups (k_f1u:
      // partial tiles + "last one sums in fixed order"; the producer / consumer kernel: two halves that meet by atomic
      // add on zeroed outputs).  Every part keeps an even number of chunks (the stage loops are unrolled by two).  A
      // template whose stiffness table does not fit beside the staging buffers in LDS is split for that reason alone.
      const int nch = t.nT / EC;
      int ksplit = 1;
      if (lean) {
        const int want = ctx->opt_f1_ksplit > 0 ? ctx->opt_f1_ksplit : (S <= 64 ? 2 : 1);      // LRBMS_OPT_F1_KSPLIT
        while (ksplit < want && nch % (4 * ksplit) == 0) ksplit *= 2;
      } else if (unified) {
        // (measured, one MI355X: 64 subdomains 98 / 102 us per pass split in 2 / 4, 110 unsplit; 128 subdomains 171 unsplit, 185 / 199 split)
        // (an UNEVEN two-way split at 128 subdomains -- long parts first, short parts beside the other kernels -- was measured
        // too: 160 - 190 us per pass for 28 .. 16 of the 32 chunks in the long part, against 150 unsplit)
        const int want = ctx->opt_f1_ksplit > 0 ? ctx->opt_f1_ksplit : (S <= 64 ? 2 : 1);      // LRBMS_OPT_F1_KSPLIT
        while (ksplit < want && nch % (4 * ksplit) == 0) ksplit *= 2;
        while (72 * (size_t)(t.nT / ksplit) > 56 * 1024 && nch % (4 * ksplit) == 0) ksplit *= 2;
        if (72 * (size_t)(t.nT / ksplit) > 56 * 1024)
          return lrbms_fail(ctx, LRBMS_E_INVALID, "fused pass: template too large for k_f1u");
      } else {
        const bool split_ok = nch % 4 == 0;
        ksplit = split_ok && (ctx->opt_f1_ksplit > 0 ? ctx->opt_f1_ksplit == 2 : 4 * S * nsl <= 256) ? 2 : 1;
      }
      if (ksplit > 1 && unified) {
        const long need = (long)S * ksplit * (lean ? f1v_part_size(ntx) : f1u_part_size(ntx));
        if (ctx->ksp_part_cap < need) {
          if (ctx->ksp_part) LRBMS_HIP_CHECK(ctx, hipFree(ctx->ksp_part));
          ctx->ksp_part = nullptr;
          ctx->ksp_part_cap = 0;
          LRBMS_HIP_CHECK(ctx, hipMalloc(&ctx->ksp_part, sizeof(double) * (size_t)need));
          ctx->ksp_part_cap = need;
        }
        if (ctx->ksp_ticket_cap < S) {
          if (ctx->ksp_ticket) LRBMS_HIP_CHECK(ctx, hipFree(ctx->ksp_ticket));
          ctx->ksp_ticket = nullptr;
          ctx->ksp_ticket_cap = 0;
          LRBMS_HIP_CHECK(ctx, hipMalloc(&ctx->ksp_ticket, sizeof(int) * (size_t)S));
          LRBMS_HIP_CHECK(ctx, hipMemset(ctx->ksp_ticket, 0, sizeof(int) * (size_t)S));
          ctx->ksp_ticket_cap = S;
        }
        a.part = ctx->ksp_part;
        a.ticket = ctx->ksp_ticket;
      }
      if (ksplit > 1 && !unified) {
        for (int sl = 0; sl < nsl; ++sl)
          hipLaunchKernelGGL(k_f1_zero, dim3(Sg), dim3(256), 0, st, gt[sl], S, N, sl == 0 ? a.rhs_red : nullptr, t.sub_list);
      }
      const dim3 grid(Sg, nsl, ksplit);
      // the timing name tells the form that ran (tests assert it; bench.py files all three under k_f1)
      KScope ks(ctx, lean2 ? "k_f1w" : lean ? "k_f1v" : unified ? "k_f1u" : "k_f1", st);
      if (lean2) {
        hipLaunchKernelGGL(k_f1w, grid, dim3(512), 0, st, t, a, gt[0], Wab);
        LRBMS_LAUNCH_CHECK(ctx);
        continue;
      }
      if (lean) {
#define LRBMS_F1V(A, B, C, D, E)                                               \
  if (ntx == A && Q == B && lv[0] == C && lv[1] == D && lv[2] == E)            \
    hipLaunchKernelGGL((k_f1v<A, B, C, D, E>), grid, dim3(512), 0, st, t, a, gt[0]);
        LRBMS_F1V_LIST(LRBMS_F1V)
#undef LRBMS_F1V
        LRBMS_LAUNCH_CHECK(ctx);
        continue;
      }
      const size_t ldsf1u = sizeof(double) * 9 * (t.nT / ksplit);
#define LRBMS_F1(NTXV)                                                                                              \
  do {                                                                                                              \
    if (Q == 1 && unified) hipLaunchKernelGGL((k_f1u<NTXV, 1>), grid, dim3(512), ldsf1u, st, t, a, gt[0]);            \
    else if (Q == 2 && unified) hipLaunchKernelGGL((k_f1u<NTXV, 2>), grid, dim3(512), ldsf1u, st, t, a, gt[0]);       \
    else if (Q == 1 && one_slice) hipLaunchKernelGGL((k_f1<NTXV, NTY, 1>), grid, dim3(512), ldsf1, st, t, a, gt[0], gt[1], gt[2]); \
    else if (Q == 2 && one_slice) hipLaunchKernelGGL((k_f1<NTXV, NTY, 2>), grid, dim3(512), ldsf1, st, t, a, gt[0], gt[1], gt[2]); \
    else hipLaunchKernelGGL((k_f1<NTXV, NTY, 0>), grid, dim3(512), ldsf1, st, t, a, gt[0], gt[1], gt[2]);             \
  } while (0)
      switch (ntx) {
        case 1: LRBMS_F1(1); break;
        case 2: LRBMS_F1(2); break;
        case 3: LRBMS_F1(3); break;
        default:   // N = 49 .. 64: 16 accumulator tiles + two prefetch sets per wave would spill in the unified form
          if (Q == 1 && one_slice) hipLaunchKernelGGL((k_f1<4, NTY, 1>), grid, dim3(512), ldsf1, st, t, a, gt[0], gt[1], gt[2]);
          else if (Q == 2 && one_slice) hipLaunchKernelGGL((k_f1<4, NTY, 2>), grid, dim3(512), ldsf1, st, t, a, gt[0], gt[1], gt[2]);
          else hipLaunchKernelGGL((k_f1<4, NTY, 0>), grid, dim3(512), ldsf1, st, t, a, gt[0], gt[1], gt[2]);
          break;
      }
#undef LRBMS_F1
      LRBMS_LAUNCH_CHECK(ctx);
    }
  }
  }
  // ---- thin parts
  if (both)      // library stream 0 is behind the preparation (the fork) and behind the caller's halo; the projection kernel is out already
    if (int rc = side_prep(ctx->aux[0])) return rc;
  if (do_b && merge_thin) {
    const int ntx = (N + 15) / 16;
    ThinRtArgs a{V, Rself, Rside, Bbb, Aab, b, ctx->nbr, Fside, r_fd, Q, N, S};
    ThinNcArgs c{ebar, AvgSelf, AvgSide, A_cpl, G_nc, B_sys, Fnc};
    size_t lds = factored ? thin_ncf_lds_bytes(t, N) : thin_nc_lds_bytes(t, ntx);
    lds = std::max(lds, thin_rt_lds_bytes(t, Q, N));
    lds = std::max(lds, coupling_lds_bytes(t, ntx, Q));
#define LRBMS_THIN(NTX)                                                                                                      \
  do {                                                                                                                       \
    if (lds > 64 * 1024)                                                                                                     \
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_thin<NTX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k_thin<NTX>, dim3(4, Sg, 3), dim3(512), lds, s_rt, t, a, c);                                           \
  } while (0)
#define LRBMS_THIN3(NTX)                                                                                                     \
  do {                                                                                                                       \
    if (lds > 64 * 1024)                                                                                                     \
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_thin3<NTX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k_thin3<NTX>, dim3(4, Sg, 3), dim3(256), lds, s_rt, t, a, f, A_cpl, B_sys);                            \
  } while (0)
    if (factored) {
      const ThinNcfArgs f{V, ebar, AvgSelf, AvgSide, ctx->nbr, Fnc, N, S};
      KScope ks(ctx, "k_thin3", s_rt);
      switch (ntx) {
        case 1: LRBMS_THIN3(1); break;
        case 2: LRBMS_THIN3(2); break;
        case 3: LRBMS_THIN3(3); break;
        default: LRBMS_THIN3(4); break;
      }
    } else {
      KScope ks(ctx, "k_thin", s_rt);
      switch (ntx) {
        case 1: LRBMS_THIN(1); break;
        case 2: LRBMS_THIN(2); break;
        case 3: LRBMS_THIN(3); break;
        default: LRBMS_THIN(4); break;
      }
    }
#undef LRBMS_THIN3
#undef LRBMS_THIN
    LRBMS_LAUNCH_CHECK(ctx);
    if (!factored) {
      ThinExpandArgs e{Fside, ctx->nbr, G_bb, G_rdd, G_ab, Q, N, S};
      const size_t lds3 = sizeof(double) * (size_t)t.ncf * fside_ld(Q, N);
      KScope ks(ctx, "k_thin_expand", s_rt);
      hipLaunchKernelGGL(k_thin_expand, dim3(4, Sg), dim3(256), lds3, s_rt, t, e);
      LRBMS_LAUNCH_CHECK(ctx);
    }
  }
  if (do_b && !merge_thin) {
    hipStream_t side = s_nc;
    if (phase != 0 && !side_from_lds) {      // (phase 0: the preparation wrote the neighbours' shares as well)
      KScope ks(ctx, "k_vertex_side", side);
      hipLaunchKernelGGL(k_vertex_side, dim3(grid_for((long)Sg * 4 * nvs * N)), dim3(256), 0, side, t, S, ctx->nbr, N, V, AvgSide);
    }
    const int ntx = (N + 15) / 16;
    if (factored) {
      const ThinNcfArgs f{V, ebar, AvgSelf, AvgSide, ctx->nbr, Fnc, N, S};
      const size_t ldsf = thin_ncf_lds_bytes(t, N);
      if (ldsf > 64 * 1024)
        LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_thin_ncf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
      KScope ks(ctx, "k_thin_ncf", side);
      hipLaunchKernelGGL(k_thin_ncf, dim3(4, Sg), dim3(256), ldsf, side, t, f);
    } else {
    const size_t lds = thin_nc_lds_bytes(t, ntx);
    // templates with more than ~24 touching elements per side (k_c = 8: 78 KB at N = 40) need the opt-in for > 64 KB of LDS
#define LRBMS_THIN_NC(NTX)                                                                                                       \
  do {                                                                                                                           \
    if (lds > 64 * 1024)                                                                                                         \
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_thin_nc<NTX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k_thin_nc<NTX>, dim3(4, Sg), dim3(512), lds, side, t, S, ctx->nbr, N, V, ebar, AvgSelf, AvgSide, G_nc);    \
  } while (0)
    {
    KScope ks(ctx, "k_thin_nc", side);
    switch (ntx) {
      case 1: LRBMS_THIN_NC(1); break;
      case 2: LRBMS_THIN_NC(2); break;
      case 3: LRBMS_THIN_NC(3); break;
      default: LRBMS_THIN_NC(4); break;
    }
    }
#undef LRBMS_THIN_NC
    }
    LRBMS_LAUNCH_CHECK(ctx);
    ThinRtArgs a{V, Rself, Rside, Bbb, Aab, b, ctx->nbr, Fside, r_fd, Q, N, S};
    const size_t lds2 = thin_rt_lds_bytes(t, Q, N);
    {
      KScope ks(ctx, "k_thin_rt", s_rt);
      hipLaunchKernelGGL(k_thin_rt, dim3(4, Sg), dim3(256), lds2, s_rt, t, a);
    }
    LRBMS_LAUNCH_CHECK(ctx);
    if (!factored) {
      ThinExpandArgs e{Fside, ctx->nbr, G_bb, G_rdd, G_ab, Q, N, S};
      const size_t lds3 = sizeof(double) * (size_t)t.ncf * fside_ld(Q, N);
      KScope ks(ctx, "k_thin_expand", s_rt);
      hipLaunchKernelGGL(k_thin_expand, dim3(4, Sg), dim3(256), lds3, s_rt, t, e);
      LRBMS_LAUNCH_CHECK(ctx);
    }
  }
  // ---- F2
  if (do_a) {
    F2Args a{Rself, Bbb, b, ctx->nbr, G_bb, G_rdd, r_fd, Q, N, S, gstride};
    const int nr = (QN + 15) / 16;
    const size_t ldsf2 = sizeof(double) * 4 * t.nT + sizeof(int) * 3 * t.nT;
#define LRBMS_F2(NRV)                                                                                                       \
  do {                                                                                                                       \
    if (ldsf2 > 64 * 1024)                                                                                                   \
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_f2<NRV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf2)); \
    hipLaunchKernelGGL(k_f2<NRV>, dim3(Sg), dim3(64 * (F2_NCW + EC)), ldsf2, s_f23, t, a);                                    \
  } while (0)
    // (A unified-role form of this kernel as for k_f1 -- all eight waves stage, chunks of eight elements -- was measured
    // and dropped: its 98 KB of LDS leave one workgroup per CU, 136 us against 126 us for three co-resident workgroups of
    // this producer / consumer form; here the MFMA share is small, so other workgroups do fill the producers' latencies.)
    KScope ks(ctx, "k_f2", s_f23);
    switch (nr) {
      case 1: LRBMS_F2(1); break;
      case 2: LRBMS_F2(2); break;
      case 3: LRBMS_F2(3); break;
      case 4: LRBMS_F2(4); break;
      case 5: LRBMS_F2(5); break;
      case 6: LRBMS_F2(6); break;
      case 7: LRBMS_F2(7); break;
      default: LRBMS_F2(8); break;
    }
#undef LRBMS_F2
    LRBMS_LAUNCH_CHECK(ctx);
  }
  // ---- F3 (unless k_prep_lds has produced G_nc[self, self] from its copy of the slab)
  if (do_a && !(prep_from_lds && gnc_fold)) {
    F3Args a{V, ebar, AvgSelf, AvgSide, G_nc, N, S, factored ? (long)N * N : 25L * N * N, factored ? N : 5 * N,
             factored ? 0 : 10 * N * N + 2 * N};
    const int ntx = (N + 15) / 16;
    KScope ks(ctx, "k_f3", s_f3);
    switch (ntx) {
      case 1: hipLaunchKernelGGL(k_f3<1>, dim3(Sg), dim3(256), 0, s_f3, t, a); break;
      case 2: hipLaunchKernelGGL(k_f3<2>, dim3(Sg), dim3(256), 0, s_f3, t, a); break;
      case 3: hipLaunchKernelGGL(k_f3<3>, dim3(Sg), dim3(256), 0, s_f3, t, a); break;
      default: hipLaunchKernelGGL(k_f3<4>, dim3(Sg), dim3(256), 0, s_f3, t, a); break;
    }
    LRBMS_LAUNCH_CHECK(ctx);
  }
  if (do_b && !merge_thin) {   // off-diagonal blocks of B_sys
    const int ntx = (N + 15) / 16;
    const size_t ldsc = coupling_lds_bytes(t, ntx, Q);
    KScope ks(ctx, "k_coupling", s_rt);
#define LRBMS_CPL(NTXV)                                                                                                      \
  do {                                                                                                                       \
    if (ldsc > 64 * 1024)                                                                                                    \
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_coupling<NTXV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc)); \
    hipLaunchKernelGGL(k_coupling<NTXV>, dim3(4, Sg), dim3(256), ldsc, s_rt, t, S, ctx->nbr, Q, N, V, A_cpl, B_sys);          \
  } while (0)
    switch (ntx) {
      case 1: LRBMS_CPL(1); break;
      case 2: LRBMS_CPL(2); break;
      case 3: LRBMS_CPL(3); break;
      default: LRBMS_CPL(4); break;
    }
#undef LRBMS_CPL
    LRBMS_LAUNCH_CHECK(ctx);
  }
  for (int i = 0; i < 3; ++i)
    if (use_aux[i]) {
      LRBMS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[i], ctx->aux[i]));
      LRBMS_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_join[i], 0));
    }
  return LRBMS_OK;
}
