// Full-order solve (SURVEY.md section 8f "next" #2: snapshot generation).
//
// Reference: DuneDiscretization._solve (discretize_elliptic_block_swipdg.py:219-225) hands the assembled global matrix
// to ISTL (bicgstab.ilut, python/scripts/online_adaptive_lrbms.py:71).  Here the block operator is never assembled:
// the matvec works on the block-ELL data of the discretization (A_diag: diagonal blocks of every subdomain, A_cpl:
// coupling blocks on the side faces) and the solver is a preconditioned CG -- the SWIPDG operator is symmetric positive
// definite -- with the 3x3 element blocks as block-Jacobi preconditioner.  Per iteration: two launches
//   k_fom_cg_matvec   beta = rz' / rz, p = z + beta p (own and neighbouring elements, on the fly), y = A(mu) p, partial p.y
//   k_fom_cg_update   alpha = rz / pAp, x += alpha p, r -= alpha y, z = M^-1 r, partial r.z and r.r
// A dependent kernel costs ~6 us on this part whatever it does, so the scalar reductions are no kernels of their own:
// every workgroup sums the per-workgroup partials of the previous kernel itself (a few KB from L2, the same fixed
// order everywhere, so every workgroup gets the same bits; 4 launches per iteration: 31.8 us, 2 launches: see DESIGN.md).
// The host looks at |r| where the observed convergence rate predicts the tolerance, at most 100 iterations apart.
// The theta-weighted blocks are combined once per solve.
#include <type_traits>
#include "lrbms_dev.h"

namespace {

struct QVecF { double v[8]; };

// sum over the 64 lanes of a wave, returned to every lane: DPP moves inside the rows of 16 lanes, then the four row sums
// through scalar registers (fixed order; no LDS crossbar traffic as with __shfl_xor)
__device__ inline double wave_sum_dpp(double v) {
  auto dpp = [](double x, auto ctrl_c) {
    constexpr int CTRL = decltype(ctrl_c)::value;
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  double rows[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    rows[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
  return (rows[0] + rows[1]) + (rows[2] + rows[3]);
}

__device__ inline double block_sum_256(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

// sums of a[0..na) and b[0..nb) by a 256-thread workgroup: thread-strided partial sums (all loads of a 2048-entry chunk
// issued before the first add: a plain `acc += a[i]` loop waits one L2 round trip per entry), then one fixed-order tree
// for both (same bits in every workgroup).  red: 512 doubles.
__device__ inline void block_sum_arrays_256(const double* __restrict__ a, int na, const double* __restrict__ b, int nb, double* red,
                                            double& sa, double& sb) {
  const int tid = threadIdx.x;
  double acc_a = 0.0, acc_b = 0.0;
  const int n = na > nb ? na : nb;
  for (int base = 0; base < n; base += 2048) {
    double va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + tid + 256 * k;
      va[k] = i < na ? a[i] : 0.0;
      vb[k] = i < nb ? b[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc_a += va[k];
      acc_b += vb[k];
    }
  }
  red[tid] = acc_a;
  red[256 + tid] = acc_b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      red[tid] += red[tid + off];
      red[256 + tid] += red[256 + tid + off];
    }
    __syncthreads();
  }
  sa = red[0];
  sb = red[256];
  __syncthreads();
}

// Amu_d [S][nT][4][9] = sum_q theta_q A_diag_q (+ mass * |T|/12 (1 + delta_ij) on the diagonal block),
// Amu_c [S][4][ncf][9] = sum_q theta_q A_cpl_q, Minv [S][nT][9] = inverse of the diagonal 3x3 block
__global__ __launch_bounds__(256) void k_fom_combine(Tmpl t, int S, int Q, QVecF th, double mass, const double* __restrict__ A_diag,
                                                     const double* __restrict__ A_cpl, double* __restrict__ Amu_d,
                                                     double* __restrict__ Amu_c, double* __restrict__ Minv) {
  const long nd = (long)S * t.nT * 36, nc = (long)S * 4 * t.ncf * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nd + nc; i += (long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    if (i < nd) {
      for (int q = 0; q < Q; ++q) acc += th.v[q] * A_diag[(long)q * nd + i];
      const int k = (int)(i % 36);
      if (mass != 0.0 && k < 9) acc += mass * t.area[(i / 36) % t.nT] / 12.0 * (k % 4 == 0 ? 2.0 : 1.0);
      Amu_d[i] = acc;
    } else {
      for (int q = 0; q < Q; ++q) acc += th.v[q] * A_cpl[(long)q * nc + (i - nd)];
      Amu_c[i - nd] = acc;
    }
  }
  const long ne = (long)S * t.nT;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (long)gridDim.x * blockDim.x) {
    double a[9];
    for (int k = 0; k < 9; ++k) {
      a[k] = 0.0;
      for (int q = 0; q < Q; ++q) a[k] += th.v[q] * A_diag[(long)q * nd + e * 36 + k];
      if (mass != 0.0) a[k] += mass * t.area[e % t.nT] / 12.0 * (k % 4 == 0 ? 2.0 : 1.0);
    }
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double id = 1.0 / (a[0] * c0 + a[1] * c1 + a[2] * c2);
    double* o = Minv + e * 9;
    o[0] = c0 * id;
    o[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id;
    o[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id;
    o[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  }
}

// Four lanes per (subdomain, element): lane b applies block b of the element's block row (its own 3x3 block and the
// three face neighbours'), so a wave reads 16 x 288 contiguous bytes of the block-ELL data; the three row sums are
// combined by two butterflies.  beta = rz_new / rz_old from the partials prz_new / prz_old [npart] (first: 0).
// 64 elements per workgroup; partial[blockIdx.x] = sum over its elements of p . y.
__global__ __launch_bounds__(256) void k_fom_cg_matvec(Tmpl t, int S, const int* __restrict__ nbr, const double* __restrict__ Amu_d,
                                                       const double* __restrict__ Amu_c, const double* __restrict__ z,
                                                       const double* __restrict__ p_old, const double* __restrict__ prz_new,
                                                       const double* __restrict__ prz_old, int npart, int first,
                                                       const double* __restrict__ c0, double* __restrict__ p_new,
                                                       double* __restrict__ y, double* __restrict__ partial) {
  __shared__ double red[512];
  const long ne = (long)S * t.nT;
  const long idx = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int b = threadIdx.x & 3;
  // all loads first (block, neighbour values): none of them depends on beta, and the reduction below synchronises
  double blk[9], zv[3] = {0.0, 0.0, 0.0}, po[3] = {0.0, 0.0, 0.0};
  bool live = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) blk[k] = 0.0;
  if (idx < ne) {
    const int e = (int)(idx % t.nT), s = (int)(idx / t.nT);
    int e2 = e, s2 = s;
    const double* bp = Amu_d + idx * 36 + b * 9;
    if (b > 0) {
      const int nb = t.nb_elem[e * 3 + b - 1];
      if (nb >= 0) {
        e2 = nb;
      } else {
        const int side = -1 - nb;
        s2 = nbr[s * 5 + side_to_slot(side)];
        e2 = t.nb_elem_out[e * 3 + b - 1];
        bp = Amu_c + (((long)s * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + b - 1]) * 9;
      }
    }
    if (s2 >= 0) {
      live = true;
      const long g = ((long)s2 * t.nT + e2) * 3;
#pragma unroll
      for (int k = 0; k < 9; ++k) blk[k] = bp[k];
      const double cz = c0 ? c0[s2] : 0.0;                // coarse part of the preconditioned residual (constant per subdomain)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        zv[i] = z[g + i] + cz;
        if (!first) po[i] = p_old[g + i];
      }
    }
  }
  double beta = 0.0;
  if (!first) {
    double rz_new, rz_old;
    block_sum_arrays_256(prz_new, npart, prz_old, npart, red, rz_new, rz_old);
    beta = rz_old != 0.0 ? rz_new / rz_old : 0.0;
  }
  double dot = 0.0;
  if (idx < ne) {
    double acc[3] = {0.0, 0.0, 0.0}, pv[3] = {0.0, 0.0, 0.0};
    if (live) {
      for (int i = 0; i < 3; ++i) pv[i] = zv[i] + beta * po[i];
      for (int i = 0; i < 3; ++i) acc[i] = blk[i * 3] * pv[0] + blk[i * 3 + 1] * pv[1] + blk[i * 3 + 2] * pv[2];
    }
    for (int i = 0; i < 3; ++i) {
      acc[i] += __shfl_xor(acc[i], 1, 64);
      acc[i] += __shfl_xor(acc[i], 2, 64);
    }
    if (b == 0)
      for (int i = 0; i < 3; ++i) {
        p_new[idx * 3 + i] = pv[i];
        y[idx * 3 + i] = acc[i];
        dot += pv[i] * acc[i];
      }
  }
  const double sum = block_sum_256(dot, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = sum;
}

// alpha = rz / pAp from the partials prz_in [npart] / ppap [npap] (first: 0 -- only z and the partials of the start residual)
__global__ __launch_bounds__(256) void k_fom_cg_update(long ne, const double* __restrict__ Minv, const double* __restrict__ prz_in,
                                                       const double* __restrict__ ppap, int npart, int npap, int first,
                                                       double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                       const double* __restrict__ y, double* __restrict__ z,
                                                       double* __restrict__ prz_out, double* __restrict__ prr,
                                                       double* __restrict__ r0w) {
  __shared__ double red[512];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // all loads first: none of them depends on alpha, and the reduction below synchronises
  double xv[3] = {0.0, 0.0, 0.0}, rv[3] = {0.0, 0.0, 0.0}, pv[3] = {0.0, 0.0, 0.0}, yv[3] = {0.0, 0.0, 0.0}, M[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = 0.0;
  if (idx < ne) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const long g = idx * 3 + i;
      rv[i] = r[g];
      if (!first) {
        xv[i] = x[g];
        pv[i] = p[g];
        yv[i] = y[g];
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] = Minv[idx * 9 + k];
  }
  double alpha = 0.0;
  if (!first) {
    double rz, pap;
    block_sum_arrays_256(prz_in, npart, ppap, npap, red, rz, pap);
    alpha = pap != 0.0 ? rz / pap : 0.0;
  }
  double rz = 0.0, rr = 0.0, rsum = 0.0;
  if (idx < ne) {
    for (int i = 0; i < 3; ++i) {
      const long g = idx * 3 + i;
      if (!first) {
        x[g] = xv[i] + alpha * pv[i];
        rv[i] -= alpha * yv[i];
        r[g] = rv[i];
      }
    }
    rsum = rv[0] + rv[1] + rv[2];
    for (int i = 0; i < 3; ++i) {
      const double zi = M[i * 3] * rv[0] + M[i * 3 + 1] * rv[1] + M[i * 3 + 2] * rv[2];
      z[idx * 3 + i] = zi;
      rz += rv[i] * zi;
      rr += rv[i] * rv[i];
    }
  }
  if (r0w) {
    // restriction to the coarse space on the way: the sum of the new residual over this wave's 64 elements (waves do not
    // straddle subdomains when 64 divides the elements per subdomain); k_fom_coarse1 adds the waves of a subdomain
    const double ws = wave_sum_dpp(rsum);
    if ((threadIdx.x & 63) == 0) r0w[idx >> 6] = ws;
  }
  const double s1 = block_sum_256(rz, red);
  const double s2 = block_sum_256(rr, red);
  if (threadIdx.x == 0) {
    prz_out[blockIdx.x] = s1;
    prr[blockIdx.x] = s2;
  }
}

// implicit Euler step, warm start x = u_k:  r = M u_k + dt b - y  with y = (M + dt A) u_k;  partial = |M u_k + dt b|^2
__global__ __launch_bounds__(256) void k_fom_step_residual(Tmpl t, long ne, double dt, const double* __restrict__ uk,
                                                           const double* __restrict__ b, const double* __restrict__ y,
                                                           double* __restrict__ r, double* __restrict__ partial) {
  __shared__ double red[256];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (idx < ne) {
    const double m = t.area[idx % t.nT] / 12.0;
    const double u0 = uk[idx * 3], u1 = uk[idx * 3 + 1], u2 = uk[idx * 3 + 2];
    const double sum = u0 + u1 + u2;
    const double uv[3] = {u0, u1, u2};
    for (int i = 0; i < 3; ++i) {
      const double rhs = m * (sum + uv[i]) + dt * b[idx * 3 + i];
      r[idx * 3 + i] = rhs - y[idx * 3 + i];
      acc += rhs * rhs;
    }
  }
  const double s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Coarse level (see online.hip): the span of the subdomain indicator functions.  A0[s][t] = 1^T A_st 1: the sum of all
// entries of the combined blocks between subdomains s and t.  One workgroup per subdomain.
__global__ __launch_bounds__(256) void k_fom_coarse_entries(Tmpl t, int S, const int* __restrict__ nbr, const double* __restrict__ Amu_d,
                                                            const double* __restrict__ Amu_c, double* __restrict__ A0) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < t.nT * 4; i += 256) {
    const int e = i >> 2, b = i & 3;
    if (b > 0 && t.nb_elem[e * 3 + b - 1] < 0) continue;             // that neighbour lives in another subdomain (Amu_c)
    const double* blk = Amu_d + ((long)s * t.nT * 4 + i) * 9;
    for (int k = 0; k < 9; ++k) acc += blk[k];
  }
  const double diag = block_sum_256(acc, red);
  if (threadIdx.x == 0) A0[(long)s * S + s] = diag;
  for (int side = 0; side < 4; ++side) {
    const int s2 = nbr[s * 5 + side_to_slot(side)];
    if (s2 < 0) continue;
    double a = 0.0;
    for (int i = threadIdx.x; i < t.side_count[side] * 9; i += 256) a += Amu_c[((long)s * 4 + side) * t.ncf * 9 + i];
    const double off = block_sum_256(a, red);
    if (threadIdx.x == 0) A0[(long)s * S + s2] = off;
  }
}

// Coarse part of the preconditioned residual with the restriction folded in: r0[t] = sum of the wps wave sums of
// subdomain t (written by k_fom_cg_update), c0[s] = (A0^-1 r0)_s, prz_c[s] = r0[s] c0[s].  One wave per coarse row.
__global__ __launch_bounds__(256) void k_fom_coarse1(int S, int wps, const double* __restrict__ A0inv, const double* __restrict__ r0w,
                                                     double* __restrict__ c0, double* __restrict__ prz_c) {
  const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  const double* row = A0inv + (long)s * S;
  double acc = 0.0;
  for (int base = 0; base < S; base += 512) {
    double a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + lane + 64 * k;
      a[k] = i < S ? row[i] : 0.0;
      b[k] = 0.0;
    }
    for (int w = 0; w < wps; ++w) {
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = base + lane + 64 * k;
        t[k] = i < S ? r0w[(long)i * wps + w] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) b[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += a[k] * b[k];
  }
  const double sum = wave_sum_dpp(acc);
  if (lane == 0) {
    double own = 0.0;
    for (int w = 0; w < wps; ++w) own += r0w[(long)s * wps + w];
    c0[s] = sum;
    prz_c[s] = own * sum;
  }
}

// r0[s] = sum of the residual over subdomain s (restriction to the coarse space); zeroes the coarse solution and the
// coarse part of the r.z partials, which k_coarse_apply accumulates into.  One workgroup per subdomain.
__global__ __launch_bounds__(256) void k_fom_restrict(int n, const double* __restrict__ r, double* __restrict__ r0,
                                                      double* __restrict__ c0, double* __restrict__ prz_c) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  double v[8];
  double acc = 0.0;
  for (int base = 0; base < n; base += 2048) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + threadIdx.x + 256 * k;
      v[k] = i < n ? r[(long)s * n + i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  const double sum = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    r0[s] = sum;
    c0[s] = 0.0;
    prz_c[s] = 0.0;
  }
}

}  // namespace

int64_t fom_solve_work_size(lrbms_ctx* ctx) {
  const Tmpl& t = ctx->t;
  const int64_t S = ctx->S, ne = S * t.nT;
  const int64_t nblk = (ne + 255) / 256, nmv = (ne + 63) / 64;
  return ne * 36 + S * 4 * t.ncf * 9 + ne * 9 + 5 * ne * 3 + 3 * nblk + 2 * nmv + 4 * S + 16;
}

namespace {

struct FomCg {
  double *Amu_d, *Amu_c, *Minv, *r, *z, *p[2], *y, *prz[2], *ppap, *prr, *r0, *c0, *r0w;
  int wps = 0;        // waves per subdomain of the update kernel if 64 divides the elements per subdomain (fused restriction), else 0
  const double* A0inv = nullptr;   // coarse level, or nullptr
  long ne, nv;
  int nblk, nmv;      // workgroups (= partial sums) of the update kernel (256 elements each) / the matvec kernel (64 each)
  void carve(lrbms_ctx* ctx, double* work) {
    const Tmpl& t = ctx->t;
    const long S = ctx->S;
    ne = S * t.nT;
    nv = ne * 3;
    nblk = (int)((ne + 255) / 256);
    nmv = (int)((ne + 63) / 64);
    Amu_d = work;
    Amu_c = Amu_d + ne * 36;
    Minv = Amu_c + S * 4 * t.ncf * 9;
    r = Minv + ne * 9;
    z = r + nv;
    p[0] = z + nv;
    p[1] = p[0] + nv;
    y = p[1] + nv;
    prz[0] = y + nv;              // [nblk + S]: r.z partials of the update kernel, then the coarse parts per subdomain
    prz[1] = prz[0] + nblk + S;
    prr = prz[1] + nblk + S;
    ppap = prr + nblk;   // [nmv]
    r0 = ppap + nmv;
    c0 = r0 + S;
    r0w = c0 + 3 * S;             // [nmv] residual sums per wave of the update kernel
    wps = t.nT % 64 == 0 ? t.nT / 64 : 0;
  }
};

int fom_host_sum(lrbms_ctx* ctx, const double* dev, std::vector<double>& host, double* out, hipStream_t st) {
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host.data(), dev, sizeof(double) * host.size(), hipMemcpyDeviceToHost, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  double acc = 0.0;
  for (double v : host) acc += v;
  *out = acc;
  return LRBMS_OK;
}

// PCG on the combined operator: on entry x holds the start value and b.r the start residual; iterates until
// |r| <= rtol * sqrt(ref2) (ref2 < 0: relative to the start residual).
// the coarse part of z = M^-1 r for the residual just written by k_fom_cg_update: restriction, dense coarse solve
// (k_coarse_apply of online.hip with one column and one unknown per subdomain); c0 is added to z by the next matvec
int fom_coarse_step(lrbms_ctx* ctx, FomCg& b, double* prz, hipStream_t st) {
  if (!b.A0inv) return LRBMS_OK;
  if (b.wps) {                                           // the update kernel left the residual sums per wave
    hipLaunchKernelGGL(k_fom_coarse1, dim3((ctx->S + 3) / 4), dim3(256), 0, st, ctx->S, b.wps, b.A0inv, b.r0w, b.c0, prz + b.nblk);
    LRBMS_LAUNCH_CHECK(ctx);
    return LRBMS_OK;
  }
  hipLaunchKernelGGL(k_fom_restrict, dim3(ctx->S), dim3(256), 0, st, ctx->t.n, b.r, b.r0, b.c0, prz + b.nblk);
  LRBMS_LAUNCH_CHECK(ctx);
  return launch_coarse_apply(ctx, 1, 1, b.A0inv, b.r0, b.c0, prz + b.nblk, st);
}

// Builds the coarse level for the combined blocks of `b` (b.A0inv stays nullptr if it is not available)
int fom_coarse_setup(lrbms_ctx* ctx, FomCg& b, hipStream_t st) {
  b.A0inv = nullptr;
  LRBMS_HIP_CHECK(ctx, hipMemsetAsync(b.prz[0], 0, sizeof(double) * 2 * (b.nblk + ctx->S), st));
  double* A0 = nullptr;
  if (int rc = coarse_begin(ctx, &A0, st)) return rc;
  if (!A0) return LRBMS_OK;
  hipLaunchKernelGGL(k_fom_coarse_entries, dim3(ctx->S), dim3(256), 0, st, ctx->t, ctx->S, ctx->nbr, b.Amu_d, b.Amu_c, A0);
  LRBMS_LAUNCH_CHECK(ctx);
  return coarse_finish(ctx, &b.A0inv, st);
}

// PCG on the combined operator: on entry x holds the start value and b.r the start residual; iterates until
// |r| <= rtol * sqrt(ref2) (ref2 < 0: relative to the start residual).  Preconditioner: inverse 3x3 element blocks plus
// the coarse level on the subdomain indicator functions (element-block Jacobi alone needs 3 800 iterations at config 3,
// doubling with the number of subdomains per direction; with the coarse level ~450).
int fom_cg_run(lrbms_ctx* ctx, FomCg& b, double* x, double ref2, double rtol, int max_iter, int* its, double* rel_out, hipStream_t st) {
  const Tmpl& t = ctx->t;
  const int S = ctx->S, nblk = b.nblk;
  const int npart = b.A0inv ? nblk + S : nblk;          // r.z partials: per update workgroup (+ the coarse parts per subdomain)
  const double* c0 = b.A0inv ? b.c0 : nullptr;
  std::vector<double> host(nblk);
  hipLaunchKernelGGL(k_fom_cg_update, dim3(nblk), dim3(256), 0, st, b.ne, b.Minv, b.prz[0], b.ppap, npart, b.nmv, 1, x, b.r, b.p[0], b.y, b.z,
                     b.prz[0], b.prr, b.A0inv && b.wps ? b.r0w : nullptr);
  LRBMS_LAUNCH_CHECK(ctx);
  if (int rc = fom_coarse_step(ctx, b, b.prz[0], st)) return rc;
  double rr = 0.0;
  if (int rc = fom_host_sum(ctx, b.prr, host, &rr, st)) return rc;
  if (ref2 < 0.0) ref2 = rr;
  *its = 0;
  *rel_out = 0.0;
  if (rr == 0.0 || ref2 == 0.0) return LRBMS_OK;
  double rel = sqrt(rr / ref2);
  int it = 0, block = 10;
  while (rel > rtol && it < max_iter) {
    if (block > max_iter - it) block = max_iter - it;
    for (int k = 0; k < block; ++k, ++it) {
      const int c = it & 1, o = c ^ 1;
      hipLaunchKernelGGL(k_fom_cg_matvec, dim3(b.nmv), dim3(256), 0, st, t, S, ctx->nbr, b.Amu_d, b.Amu_c, b.z, b.p[o], b.prz[c], b.prz[o],
                         npart, it == 0 ? 1 : 0, c0, b.p[c], b.y, b.ppap);
      hipLaunchKernelGGL(k_fom_cg_update, dim3(nblk), dim3(256), 0, st, b.ne, b.Minv, b.prz[c], b.ppap, npart, b.nmv, 0, x, b.r, b.p[c], b.y, b.z,
                         b.prz[o], b.prr, b.A0inv && b.wps ? b.r0w : nullptr);
      if (int rc = fom_coarse_step(ctx, b, b.prz[o], st)) return rc;
    }
    LRBMS_LAUNCH_CHECK(ctx);
    if (int rc = fom_host_sum(ctx, b.prr, host, &rr, st)) return rc;
    rel = sqrt(rr / ref2);
    if (!(rel == rel)) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "fom CG: NaN residual (operator not SPD?)");
    const double rate = log(rel) / it;
    block = 25;
    if (rel > rtol && rate < 0.0) {                     // aim a little short: CG converges superlinearly
      const double need = 0.8 * (log(rtol) - log(rel)) / rate;
      block = need < 4.0 ? 4 : need > 100.0 ? 100 : (int)need;
    }
  }
  *its = it;
  *rel_out = rel;
  return LRBMS_OK;
}

}  // namespace

int launch_fom_solve(lrbms_ctx* ctx, int Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                     double* work, double* x, double rtol, int max_iter, double* info, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "fom_solve needs all subdomains on one rank");
  if (Q < 1 || Q > 8 || max_iter < 1 || !(rtol > 0.0)) return lrbms_fail(ctx, LRBMS_E_INVALID, "fom_solve: bad Q / max_iter / rtol");
  QVecF th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  FomCg c;
  c.carve(ctx, work);
  hipLaunchKernelGGL(k_fom_combine, dim3(4096), dim3(256), 0, st, ctx->t, ctx->S, Q, th, 0.0, A_diag, A_cpl, c.Amu_d, c.Amu_c, c.Minv);
  LRBMS_LAUNCH_CHECK(ctx);
  if (int rc = fom_coarse_setup(ctx, c, st)) return rc;
  LRBMS_HIP_CHECK(ctx, hipMemsetAsync(x, 0, sizeof(double) * c.nv, st));
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(c.r, b, sizeof(double) * c.nv, hipMemcpyDeviceToDevice, st));
  int it = 0;
  double rel = 0.0;
  if (int rc = fom_cg_run(ctx, c, x, -1.0, rtol, max_iter, &it, &rel, st)) return rc;
  if (info) { info[0] = it; info[1] = rel; }
  if (rel > rtol) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "fom_solve: CG did not reach rtol");
  return LRBMS_OK;
}

// Implicit Euler for M u' + A(mu) u = b (pyMOR ImplicitEulerTimeStepper as used at discretize_parabolic_block_swipdg.py:87;
// InstationaryDuneDiscretization._solve :28-40):  (M + dt A(mu)) u_{k+1} = M u_k + dt b,  nt steps in ONE call: the
// theta-weighted blocks (+ the mass on the diagonal 3x3 blocks) are combined once, every step is a warm-started CG with
// the kernels of lrbms_fom_solve.  U [nt+1][S][n]: U[0] is the initial value (input), U[1..nt] are written.
// info[0] = CG iterations over all steps, info[1] = worst final residual relative to |M u_k + dt b|.
int launch_fom_implicit_euler(lrbms_ctx* ctx, int Q, const double* theta, double dt, int nt, const double* A_diag, const double* A_cpl,
                              const double* b, double* work, double* U, double rtol, int max_iter, double* info, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "fom_implicit_euler needs all subdomains on one rank");
  if (Q < 1 || Q > 8 || max_iter < 1 || !(rtol > 0.0) || nt < 1 || !(dt > 0.0))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "fom_implicit_euler: bad Q / max_iter / rtol / nt / dt");
  QVecF th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? dt * theta[q] : 0.0;
  FomCg c;
  c.carve(ctx, work);
  const int nblk = c.nblk;
  hipLaunchKernelGGL(k_fom_combine, dim3(4096), dim3(256), 0, st, ctx->t, ctx->S, Q, th, 1.0, A_diag, A_cpl, c.Amu_d, c.Amu_c, c.Minv);
  LRBMS_LAUNCH_CHECK(ctx);
  if (int rc = fom_coarse_setup(ctx, c, st)) return rc;
  std::vector<double> host(nblk);
  long total_it = 0;
  double worst = 0.0;
  for (int step = 0; step < nt; ++step) {
    const double* uk = U + (long)step * c.nv;
    double* x = U + (long)(step + 1) * c.nv;
    LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(x, uk, sizeof(double) * c.nv, hipMemcpyDeviceToDevice, st));
    // y = (M + dt A) u_k  (the matvec kernel with first = 1 takes its direction from `z`)
    hipLaunchKernelGGL(k_fom_cg_matvec, dim3(c.nmv), dim3(256), 0, st, ctx->t, ctx->S, ctx->nbr, c.Amu_d, c.Amu_c, uk, c.p[1], c.prz[0],
                       c.prz[1], nblk, 1, (const double*)nullptr, c.p[0], c.y, c.ppap);
    hipLaunchKernelGGL(k_fom_step_residual, dim3(nblk), dim3(256), 0, st, ctx->t, c.ne, dt, uk, b, c.y, c.r, c.ppap);
    LRBMS_LAUNCH_CHECK(ctx);
    double ref2 = 0.0;
    if (int rc = fom_host_sum(ctx, c.ppap, host, &ref2, st)) return rc;
    if (ref2 == 0.0) continue;      // zero right-hand side: u_{k+1} = u_k = 0
    int it = 0;
    double rel = 0.0;
    if (int rc = fom_cg_run(ctx, c, x, ref2, rtol, max_iter, &it, &rel, st)) return rc;
    total_it += it;
    if (rel > worst) worst = rel;
    if (rel > rtol) {
      if (info) { info[0] = (double)total_it; info[1] = worst; }
      return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "fom_implicit_euler: CG did not reach rtol");
    }
  }
  if (info) { info[0] = (double)total_it; info[1] = worst; }
  return LRBMS_OK;
}
