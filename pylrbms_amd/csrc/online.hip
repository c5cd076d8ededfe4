// Online kernels.
//
// E1 streams every projected estimator operator of a subdomain once (HBM-bound: ~3.3 MB per subdomain at N = 40)
// and forms the quadratic forms with fixed-order reductions.  O1 solves the block-sparse reduced system by
// block-Jacobi preconditioned CG; the matrix (S x 5 blocks of N x N) is assembled once per mu and then lives in
// the Infinity Cache for the iteration.  All reductions are fixed-order trees (no fp64 atomics) so that results are
// bitwise reproducible run to run.
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#include <rocsolver/rocsolver.h>

#include "lrbms_dev.h"

struct QVec { double v[8]; };

namespace {

__device__ inline double block_reduce_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  double out = red[0];
  __syncthreads();
  return out;
}

// sum_r x[r] sum_c G[r][c] y[c]: waves over rows, lanes over columns (coalesced), per-thread partial
__device__ inline double quad_partial(const double* __restrict__ G, int ld, int rows, int cols, const double* x,
                                      const double* y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  double acc = 0.0;
  for (int r = wave; r < rows; r += nw) {
    const double xr = x[r];
    if (xr == 0.0) continue;
    const double* row = G + (long)r * ld;
    double s = 0.0;
    for (int c = lane; c < cols; c += 64) s += row[c] * y[c];
    acc += xr * s;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------------
// E1: one workgroup per subdomain.
__global__ __launch_bounds__(256) void k_reduced_estimate(int S, const int* __restrict__ nbr, int Q, int N, QVec theta,
                                                          const double* __restrict__ u, const double* __restrict__ G_nc,
                                                          const double* __restrict__ r_fd, const double* __restrict__ G_rdd,
                                                          const double* __restrict__ G_bb, const double* __restrict__ G_ab,
                                                          const double* __restrict__ G_aa, const double* __restrict__ Fside,
                                                          const double* __restrict__ Fnc, int ncf, int nvs,
                                                          const double* __restrict__ f2, const double* __restrict__ ceps,
                                                          double hdiam, double* __restrict__ eta_loc, int nvx_patch,
                                                         const int* __restrict__ nbr_diag) {
  // nvx_patch > 0: LRBMS_OPT_OSWALD_VERTEX_PATCH (value: vertices per x-side) -- rows of F_nc carry A_diag, see k_thin_ncf
  extern __shared__ double lds[];
  const int s = blockIdx.x;
  const int W = 5 * N, C = 5 * Q * N;
  double* uo = lds;            // [W]
  double* ur = lds + W;        // [C]
  double* red = ur + C;        // [256]
  double* fac = red + 256;     // [4][ncf][3 + Q]  (factored layout only)
  double* facn = fac + 4 * ncf * (3 + Q);   // [4][nvs][2]  (factored layout only): z_a = A_a u_a, C_a u_s per side vertex
  for (int i = threadIdx.x; i < W; i += blockDim.x) {
    const int slot = i / N, j = i % N;
    const int s2 = nbr[s * 5 + slot];
    const double val = s2 >= 0 ? u[(long)s2 * N + j] : 0.0;
    uo[i] = val;
    for (int q = 0; q < Q; ++q) ur[(slot * Q + q) * N + j] = theta.v[q] * val;
  }
  __syncthreads();
  const double* ui = uo + 2 * N;
  // nonconformity: u^T G_nc u over the five slots, either from the dense [S][5N][5N] operator or (factored layout) from
  // its [self, self] block [S][N][N] and the side factors F_nc (k_thin_ncf in fused.hip):
  //   u_s^T G_ss u_s + sum_a z_a^T (2 C_a u_s + sum_b M_ab z_b),   z_a = A_a u_a
  double p_nc = Fnc == nullptr ? quad_partial(G_nc + (long)s * W * W, W, W, W, uo, uo)
                               : quad_partial(G_nc + (long)s * N * N, N, N, N, ui, ui);
  // z^T G z = z_s^T G_ss z_s + sum_a (2 z_a^T G_as z_s + z_a^T G_aa z_a) for G_rdd / G_bb, either from the block-compact
  // layout [S][9][QN][QN] or (Fside != nullptr) from the self blocks [S][QN][QN] plus the side factors (see k_thin_rt)
  const int QN = Q * N;
  const bool factored = Fside != nullptr;
  const long gstride = factored ? (long)QN * QN : (long)9 * QN * QN;
  const int abld = factored ? QN : C;
  const double* zs = ur + 2 * QN;
  const double* Gd = G_rdd + (long)s * gstride;
  const double* Gb = G_bb + (long)s * gstride;
  double p_rdd = quad_partial(Gd, QN, QN, QN, zs, zs);
  double p_bb = quad_partial(Gb, QN, QN, QN, zs, zs);
  double p_ab = 0.0, p_aa = 0.0;
  if (!factored) {
    for (int side = 0; side < 4; ++side) {
      const double* za = ur + (side < 2 ? side : side + 1) * QN;
      p_rdd += 2.0 * quad_partial(Gd + (long)(1 + side) * QN * QN, QN, QN, QN, za, zs) +
               quad_partial(Gd + (long)(5 + side) * QN * QN, QN, QN, QN, za, za);
      p_bb += 2.0 * quad_partial(Gb + (long)(1 + side) * QN * QN, QN, QN, QN, za, zs) +
              quad_partial(Gb + (long)(5 + side) * QN * QN, QN, QN, QN, za, za);
    }
  } else {
    // one dot product per (side, side face p, factor k): k = 0: Ra . z_a, 1: Yb . z_s, 2: Dp . z_s, 3 + q: Xab_q . u_i
    const int LD = 4 * QN + 4, nk = 3 + Q, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int it = wave; it < 4 * ncf * nk; it += nw) {
      const int side = it / (ncf * nk), rem = it - side * ncf * nk, p = rem / nk, k = rem - p * nk;
      const double* row = Fside + (((long)s * 4 + side) * ncf + p) * LD;
      const double* x;
      const double* y;
      int len;
      if (k == 0) { x = row; y = ur + (side < 2 ? side : side + 1) * QN; len = QN; }
      else if (k < 3) { x = row + k * QN; y = zs; len = QN; }
      else { x = row + 3 * QN + (k - 3) * N; y = ui; len = N; }
      double d = 0.0;
      for (int c = lane; c < len; c += 64) d += x[c] * y[c];
      for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
      if (lane == 0) fac[it] = d;
    }
    const int LDn = 2 * N + 4 * nvs + (nvx_patch > 0 ? N : 0);
    for (int it = wave; it < 4 * nvs * 2; it += nw) {
      const int row = it >> 1, k = it & 1, side = row / nvs;
      const double* x = Fnc + ((long)s * 4 * nvs + row) * LDn + k * N;
      const double* y = k == 0 ? uo + (side < 2 ? side : side + 1) * N : ui;
      double d = 0.0;
      for (int c = lane; c < N; c += 64) d += x[c] * y[c];
      for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
      if (lane == 0) facn[it] = d;
    }
    __syncthreads();
    if (nvx_patch > 0) {
      // cross points: the diagonal subdomain's share of the vertex average joins z of the side that carries the corner
      // (z_a[pos] += A_diag . u_diag); corners 0 SW / 1 SE on side 0, 2 NW / 3 NE on side 3
      if (wave < 4) {
        const int corner = wave, side = corner < 2 ? 0 : 3, pos = (corner & 1) ? nvx_patch - 1 : 0;
        const int sd = nbr[s * 5 + (corner < 2 ? 0 : 4)] >= 0 ? nbr_diag[s * 4 + corner] : -1;   // the diagonal subdomain (Tmpl::nbr_diag)
        const int row = side * nvs + pos;
        double d = 0.0;
        if (sd >= 0) {
          const double* x = Fnc + ((long)s * 4 * nvs + row) * LDn + 2 * N + 4 * nvs;
          const double* y = u + (long)sd * N;
          for (int c = lane; c < N; c += 64) d += x[c] * y[c];
        }
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (lane == 0) facn[2 * row] += d;
      }
      __syncthreads();
    }
    for (int row = threadIdx.x; row < 4 * nvs; row += blockDim.x) {
      const double* m = Fnc + ((long)s * 4 * nvs + row) * LDn + 2 * N;
      double mz = 0.0;
      for (int c = 0; c < 4 * nvs; ++c) mz += m[c] * facn[2 * c];
      p_nc += facn[2 * row] * (2.0 * facn[2 * row + 1] + mz);
    }
    for (int it = threadIdx.x; it < 4 * ncf; it += blockDim.x) {
      const double* f = fac + it * nk;
      const double* sc = Fside + ((long)s * 4 * ncf + it) * LD + 4 * QN;
      const double ra = f[0];
      p_bb += sc[0] * ra * ra + 2.0 * ra * f[1];
      p_rdd += sc[1] * ra * ra + 2.0 * ra * f[2];
      for (int q = 0; q < Q; ++q) p_ab += theta.v[q] * f[3 + q] * ra;
    }
  }
  double p_rfd = 0.0;
  for (int c = threadIdx.x; c < C; c += blockDim.x) p_rfd += r_fd[(long)s * C + c] * ur[c];
  for (int q = 0; q < Q; ++q) {
    p_ab += theta.v[q] * quad_partial(G_ab + ((long)q * S + s) * N * abld, abld, N, abld, ui, factored ? zs : ur);
    for (int q2 = 0; q2 < Q; ++q2)
      p_aa += theta.v[q] * theta.v[q2] * quad_partial(G_aa + (((long)q * Q + q2) * S + s) * N * N, N, N, N, ui, ui);
  }
  const double nc = block_reduce_sum(p_nc, red);
  const double rdd = block_reduce_sum(p_rdd, red);
  const double bb = block_reduce_sum(p_bb, red);
  const double rfd = block_reduce_sum(p_rfd, red);
  const double ab = block_reduce_sum(p_ab, red);
  const double aa = block_reduce_sum(p_aa, red);
  if (threadIdx.x == 0) {
    const double pi = 3.14159265358979323846;
    eta_loc[s] = nc;
    eta_loc[S + s] = (f2[s] - 2.0 * rfd + rdd) * ((1.0 / (pi * pi)) / ceps[s]) * hdiam * hdiam;   // estimators.py:88-91
    eta_loc[2 * S + s] = aa + bb + 2.0 * ab;
  }
}

// ---------------------------------------------------------------------------------------------------------
// O1 pieces
// Amu[s][slot] = sum_q theta_q B_sys[q][s][slot]
__global__ __launch_bounds__(256) void k_assemble_mu(long per_q, int Q, QVec theta, const double* __restrict__ B_sys,
                                                     double* __restrict__ Amu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_q; i += (long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) acc += theta.v[q] * B_sys[(long)q * per_q + i];
    Amu[i] = acc;
  }
}

// Dinv[s] = inverse of the SPD diagonal block Amu[s][2] by Gauss-Jordan without pivoting in LDS (N <= 64).
// Four lanes share a row (columns c = part, part + 4, ...), rows are ld = N | 1 doubles apart (with the even stride N every
// lane of a wave hit the same two banks), and the pivot row is NOT normalised: step k subtracts (A[r][k] / A[k][k]) x row k
// from every other row, which leaves row k alone, so one barrier per step is enough; A ends as a diagonal matrix and the
// inverse is I[r][:] / A[r][r].  All LDS reads of a step are issued before the first write (a read-modify-write loop
// would wait one LDS round trip per entry).  198 -> 108 us for 1024 blocks of 40 x 40; what remains is LDS bandwidth
// (every step streams both matrices through the LDS pipe: 77 KB per block and step).
__global__ __launch_bounds__(256) void k_block_inverse(int N, const double* __restrict__ Amu, double* __restrict__ Dinv,
                                                       int blocks_per_s, int block_off) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, ld = N | 1;
  double* A = lds;           // [N][ld]
  double* I = lds + N * ld;  // [N][ld]
  const double* src = Amu + ((long)s * blocks_per_s + block_off) * N * N;
  const int r = tid >> 2, part = tid & 3;
  constexpr int CT = 16;     // columns per lane, N <= 64
  if (r < N) {
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = part + 4 * t;
      if (c < N) {
        const double a = src[r * N + c];
        // a zero-padded basis column (ragged local basis sizes after online enrichment) has an exactly zero row and
        // column: put 1 on its diagonal so that the padded unknown decouples and stays 0
        A[r * ld + c] = (r == c && a == 0.0) ? 1.0 : a;
        I[r * ld + c] = r == c ? 1.0 : 0.0;
      }
    }
  }
  __syncthreads();
  const int rc = r < N ? r : 0;
  for (int k = 0; k < N; ++k) {
    const double f = A[rc * ld + k] / A[k * ld + k];
    double ar[CT], ak[CT], ir[CT], ik[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = part + 4 * t, cc = c < N ? c : 0;
      ar[t] = A[rc * ld + cc];
      ak[t] = A[k * ld + cc];
      ir[t] = I[rc * ld + cc];
      ik[t] = I[k * ld + cc];
    }
    if (r < N && r != k) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int c = part + 4 * t;
        if (c < N) {
          A[r * ld + c] = ar[t] - f * ak[t];
          I[r * ld + c] = ir[t] - f * ik[t];
        }
      }
    }
    __syncthreads();
  }
  if (r < N) {
    const double dinv = 1.0 / A[r * ld + r];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = part + 4 * t;
      if (c < N) Dinv[(long)s * N * N + r * N + c] = I[r * ld + c] * dinv;
    }
  }
}

// launch of k_block_inverse: src block (s, off) of bps blocks per subdomain -> dst[s]
static int launch_block_inverse(lrbms_ctx* ctx, int S, int N, const double* src, double* dst, int bps, int off, hipStream_t st) {
  const size_t lds = sizeof(double) * 2 * N * (N | 1);
  if (lds > 64 * 1024)
    LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_block_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_block_inverse, dim3((unsigned)S), dim3(256), lds, st, N, src, dst, bps, off);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// y_s = sum_slot Amu[s][slot] p_{nbr(s,slot)};  partial[s] = p_s . y_s.  One wave per subdomain.
__global__ __launch_bounds__(64) void k_cg_matvec(const int* __restrict__ nbr, int N, const double* __restrict__ Amu,
                                                  const double* __restrict__ p, double* __restrict__ y,
                                                  double* __restrict__ partial) {
  extern __shared__ double lds[];
  const int s = blockIdx.x;
  double* ps = lds;  // [5][N]
  for (int i = threadIdx.x; i < 5 * N; i += 64) {
    const int s2 = nbr[s * 5 + i / N];
    ps[i] = s2 >= 0 ? p[(long)s2 * N + i % N] : 0.0;
  }
  __syncthreads();
  double dot = 0.0;
  for (int r = threadIdx.x; r < N; r += 64) {
    double acc = 0.0;
    for (int slot = 0; slot < 5; ++slot) {
      if (nbr[s * 5 + slot] < 0) continue;
      const double* row = Amu + (((long)s * 5 + slot) * N + r) * N;
      for (int c = 0; c < N; ++c) acc += row[c] * ps[slot * N + c];
    }
    y[(long)s * N + r] = acc;
    dot += acc * ps[2 * N + r];
  }
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
  if (threadIdx.x == 0) partial[s] = dot;
}

// ---- two-kernel CG iteration -------------------------------------------------------------------------------
// A dependent kernel costs ~6 us on this part whatever it does (measured: 6 kernels per iteration = 35 us, graph replay
// or not), so the scalar reductions are not kernels of their own: every workgroup sums the per-subdomain partials of the
// previous kernel itself (S doubles from L2, the same fixed order in every workgroup, so all of them get the same bits).
// Sums of a[0..n) and b[0..n) by one wave: lane-strided partial sums, then a butterfly.  All loads of a 1024-entry
// chunk are issued before the first add (a plain `acc += a[i]` loop waits one L2 round trip per entry: 16 x ~0.7 us).
__device__ inline void wave_sum_arrays(const double* __restrict__ a, const double* __restrict__ b, int n, double& sa, double& sb) {
  const int lane = threadIdx.x & 63;
  double acc_a = 0.0, acc_b = 0.0;
  for (int base = 0; base < n; base += 1024) {
    double va[16], vb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = base + lane + 64 * k;
      va[k] = i < n ? a[i] : 0.0;
      vb[k] = i < n ? b[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      acc_a += va[k];
      acc_b += vb[k];
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc_a += __shfl_xor(acc_a, off, 64);
    acc_b += __shfl_xor(acc_b, off, 64);
  }
  sa = acc_a;
  sb = acc_b;
}

// p_new = z + beta p_old (beta = rz_new / rz_old from the partials; own + neighbour rows on the fly, own rows stored),
// y_s = sum_slot Amu[s][slot] p_new[nbr(s, slot)],  ppap[s] = p_s . y_s.   One 256-thread workgroup per subdomain: the
// 5 N rows of the subdomain's blocks are dealt to groups of 4 lanes, each lane takes every 4th pair of doubles of its
// rows (16-byte loads, the 4 lanes cover 64 contiguous bytes per step).  The matrix (65 MB at config 3) does not fit
// the L2s and streams from the Infinity Cache: 17 us = 3.8 TB/s.  (Measured alternatives: one wave per subdomain with a
// row per lane: the same 17 us; all 40 row loads of a lane hoisted in front of the reduction: 41 us per iteration
// instead of 28 -- the few loads everything waits for then queue behind the bulk of every workgroup on the CU.)
template <bool EVEN>
__global__ __launch_bounds__(256) void k_cg2_matvec(const int* __restrict__ nbr, int S, int N, const double* __restrict__ Amu,
                                                   const double* __restrict__ z, const double* __restrict__ p_old,
                                                   const double* __restrict__ prz_new, const double* __restrict__ prz_old,
                                                   int first, double* __restrict__ p_new, double* __restrict__ y,
                                                   double* __restrict__ ppap) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x;
  double beta = 0.0;
  if (!first) {
    double rz_new, rz_old;
    wave_sum_arrays(prz_new, prz_old, S, rz_new, rz_old);
    beta = rz_old != 0.0 ? rz_new / rz_old : 0.0;
  }
  double* ps = lds;            // [5][N]   direction on the neighbourhood
  double* ya = lds + 5 * N;    // [5][N]   per-slot row sums
  int slot_mask = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) slot_mask |= (nbr[s * 5 + k] >= 0) << k;
  for (int i = tid; i < 5 * N; i += 256) {
    const int s2 = nbr[s * 5 + i / N];
    double v = 0.0;
    if (s2 >= 0) {
      const long g = (long)s2 * N + i % N;
      v = first ? z[g] : z[g] + beta * p_old[g];
      if (i / N == 2) p_new[g] = v;
    }
    ps[i] = v;
  }
  __syncthreads();
  const int q4 = tid & 3;
  const double* base = Amu + (long)s * 5 * N * N;
  for (int pair = tid >> 2; pair < 5 * N; pair += 64) {      // pair = slot * N + row
    const int slot = pair / N;
    const double* row = base + (long)pair * N;
    const double* pv = ps + slot * N;
    double acc = 0.0;
    if ((slot_mask >> slot) & 1) {
      if (EVEN) {
        for (int c = 2 * q4; c < N; c += 8) {
          const double2 a = *reinterpret_cast<const double2*>(row + c);
          acc += a.x * pv[c] + a.y * pv[c + 1];
        }
      } else {
        for (int c = q4; c < N; c += 4) acc += row[c] * pv[c];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (q4 == 0) ya[pair] = acc;
  }
  __syncthreads();
  double dot = 0.0;
  if (tid < N) {
    const double v = ya[tid] + ya[N + tid] + ya[2 * N + tid] + ya[3 * N + tid] + ya[4 * N + tid];
    y[(long)s * N + tid] = v;
    dot = v * ps[2 * N + tid];
  }
  if (tid < 64) {                                            // N <= 64: the first wave holds every row
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (tid == 0) ppap[s] = dot;
  }
}

// alpha = rz / pAp from the partials;  x += alpha p;  r -= alpha y;  z = Dinv r;  prz_out[s] = r_s . z_s;  prr[s] = r_s . r_s.
// first: alpha = 0 (only z and the partials of the start residual are formed).  256 threads per subdomain, 4 lanes per row.
template <bool EVEN>
__global__ __launch_bounds__(256) void k_cg2_update(int S, int N, const double* __restrict__ Dinv, const double* __restrict__ prz_in,
                                                   const double* __restrict__ ppap, int first, double* __restrict__ x,
                                                   double* __restrict__ r, const double* __restrict__ p,
                                                   const double* __restrict__ y, double* __restrict__ z,
                                                   double* __restrict__ prz_out, double* __restrict__ prr) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x;
  double alpha = 0.0;
  if (!first) {
    double rz, pap;
    wave_sum_arrays(prz_in, ppap, S, rz, pap);
    alpha = pap != 0.0 ? rz / pap : 0.0;
  }
  double* rs = lds;        // [N]
  double* zs = lds + N;    // [N]
  if (tid < N) {
    const long g = (long)s * N + tid;
    double rv = r[g];
    if (!first) {
      x[g] += alpha * p[g];
      rv -= alpha * y[g];
      r[g] = rv;
    }
    rs[tid] = rv;
  }
  __syncthreads();
  const int q4 = tid & 3, row_i = tid >> 2;
  if (row_i < N) {
    const double* row = Dinv + ((long)s * N + row_i) * N;
    double acc = 0.0;
    if (EVEN) {
      for (int c = 2 * q4; c < N; c += 8) {
        const double2 a = *reinterpret_cast<const double2*>(row + c);
        acc += a.x * rs[c] + a.y * rs[c + 1];
      }
    } else {
      for (int c = q4; c < N; c += 4) acc += row[c] * rs[c];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (q4 == 0) {
      zs[row_i] = acc;
      z[(long)s * N + row_i] = acc;
    }
  }
  __syncthreads();
  if (tid < 64) {
    double dot = 0.0, rr = 0.0;
    if (tid < N) {
      dot = zs[tid] * rs[tid];
      rr = rs[tid] * rs[tid];
    }
    for (int off = 32; off > 0; off >>= 1) {
      dot += __shfl_down(dot, off, 64);
      rr += __shfl_down(rr, off, 64);
    }
    if (tid == 0) {
      prz_out[s] = dot;
      prr[s] = rr;
    }
  }
}

// Amu[s][slot] = sum_q theta_q B_sys[q][s][slot] (+ M_red[s] on the self slot): the operator of a reduced implicit Euler step
__global__ __launch_bounds__(256) void k_assemble_mu_mass(long per_q, int Q, int N, QVec theta, const double* __restrict__ B_sys,
                                                          const double* __restrict__ M_red, double* __restrict__ Amu) {
  const long nn = (long)N * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_q; i += (long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) acc += theta.v[q] * B_sys[(long)q * per_q + i];
    const long blk = i / nn;
    if (blk % 5 == 2) acc += M_red[(blk / 5) * nn + i % nn];
    Amu[i] = acc;
  }
}

// warm-started step: r_s = M_red[s] u_s + dt b_s - y_s  (y = (M + dt A) u_k);  partial[s] = |M_red[s] u_s + dt b_s|^2
__global__ __launch_bounds__(64) void k_red_step_residual(int N, double dt, const double* __restrict__ M_red,
                                                          const double* __restrict__ uk, const double* __restrict__ b,
                                                          const double* __restrict__ y, double* __restrict__ r,
                                                          double* __restrict__ partial) {
  extern __shared__ double lds[];
  const int s = blockIdx.x;
  for (int i = threadIdx.x; i < N; i += 64) lds[i] = uk[(long)s * N + i];
  __syncthreads();
  double acc = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) {
    const double* row = M_red + ((long)s * N + i) * N;
    double rhs = dt * b[(long)s * N + i];
    for (int c = 0; c < N; ++c) rhs += row[c] * lds[c];
    r[(long)s * N + i] = rhs - y[(long)s * N + i];
    acc += rhs * rhs;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) partial[s] = acc;
}

// out[s] = y_s^T Dinv[s] y_s  (one wave per subdomain)
__global__ __launch_bounds__(64) void k_red_inv_norm2(int N, const double* __restrict__ Dinv, const double* __restrict__ y,
                                                      double* __restrict__ out) {
  extern __shared__ double lds[];
  const int s = blockIdx.x;
  for (int i = threadIdx.x; i < N; i += 64) lds[i] = y[(long)s * N + i];
  __syncthreads();
  double acc = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) {
    const double* row = Dinv + ((long)s * N + i) * N;
    double z = 0.0;
    for (int c = 0; c < N; ++c) z += row[c] * lds[c];
    acc += z * lds[i];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) out[s] = acc;
}

}  // namespace

int launch_reduced_estimate(lrbms_ctx* ctx, int Q, int N, const double* theta, const double* u, const double* G_nc,
                            const double* r_fd, const double* G_rdd, const double* G_bb, const double* G_ab,
                            const double* G_aa, const double* Fside, const double* Fnc, const double* f2, const double* ceps,
                            double hdiam, double* eta_loc, hipStream_t st) {
  QVec th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  if ((Fside != nullptr) != (Fnc != nullptr)) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate: F_side and F_nc go together");
  if (ctx->t.opt_oswald_vertex && (Fnc == nullptr || (ctx->S_ext != ctx->S && !ctx->diag_explicit)))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "LRBMS_OPT_OSWALD_VERTEX_PATCH: factored layout; sharded grids need lrbms_set_diagonal_neighbours");
  const int nvs = ctx->t.nvx > ctx->t.nvy ? ctx->t.nvx : ctx->t.nvy;
  const size_t lds = sizeof(double) * (5 * N + 5 * Q * N + 256 + 4 * ctx->t.ncf * (3 + Q) + 8 * nvs);
  hipLaunchKernelGGL(k_reduced_estimate, dim3(ctx->S), dim3(256), lds, st, ctx->S, ctx->nbr, Q, N, th, u, G_nc, r_fd,
                     G_rdd, G_bb, G_ab, G_aa, Fside, Fnc, ctx->t.ncf, nvs, f2, ceps, hdiam, eta_loc,
                     ctx->t.opt_oswald_vertex ? ctx->t.nvx : 0, ctx->t.nbr_diag);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// =========================================================================================================
// Coarse level of the reduced solvers' preconditioner.
// Block-Jacobi alone has no global coupling: its iteration count doubles with the number of subdomains per direction
// (36 / 75 / 170 iterations at 8x8 / 16x16 / 32x32 subdomains).  Adding the Galerkin coarse problem on the span of the
// FIRST local basis vector of every subdomain (the constant shape function the reference starts every basis with,
// reductor.py:29-31) -- M^-1 = blockdiag(A_ss)^-1 + R0^T (R0 A R0^T)^-1 R0, additive, symmetric positive definite --
// makes it independent of the subdomain count (~27 iterations).  The coarse matrix is the 5-point S x S matrix of the
// (0, 0) entries of the combined blocks; it is factorised and inverted once per solve by rocSOLVER (dpotrf + dpotrs on
// the identity: 2.9 + 0.6 ms at S = 1024 -- kept for subdomain numberings with a band wider than 64; the usual case is
// the hand-written block-tridiagonal factorisation k_bt_factor / k_bt_inverse below) and applied as a dense
// product per iteration (k_coarse_apply).  If the factorisation fails (a zero first basis vector) the solvers run with
// block-Jacobi alone.  LRBMS_NO_COARSE=1 switches the coarse level off.  Because the factorisation costs about as much
// as a whole batched solve, a preconditioner can be built once for a reduced model at a reference parameter
// (lrbms_reduced_precond_build) and handed to every solve (lrbms_reduced_precond_use): any symmetric positive definite
// preconditioner is admissible, and one built at the middle of the parameter range serves the whole range.
namespace {

__global__ __launch_bounds__(256) void k_coarse_init(long S, double* __restrict__ A0, double* __restrict__ A0inv) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < S * S; i += (long)gridDim.x * blockDim.x) {
    A0[i] = 0.0;
    A0inv[i] = (i / S == i % S) ? 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(256) void k_coarse_entries(int S, int N, const int* __restrict__ nbr, const double* __restrict__ Amu,
                                                        double* __restrict__ A0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * 5) return;
  const int s = i / 5, slot = i - s * 5, t = nbr[i];
  if (t >= 0) A0[(long)s * S + t] = Amu[((long)s * 5 + slot) * N * N];      // entry (0, 0) of block [s][slot]
}

// c = A0inv R0 r for nmu <= 16 columns on the fp64 matrix cores:  z[s][0][m] += c[s][m],
// prz[m][s] += r[s][0][m] c[s][m]  (the coarse part of r . z, added to the partial the update kernel has just written).
// One workgroup per 16 subdomains (rows of A0inv), 16 waves splitting K = S: every MFMA step takes its A operand from
// A0inv[k][row] (= A0inv[row][k], the matrix is symmetric: 16 contiguous doubles per k) and its B operand from
// r[k][0][m]; a wave issues the loads of 8 steps (16 per lane) before its first MFMA; the 16 partial tiles meet in LDS.
// (A VALU form with a thread per (m, row, k-part) was bound by its load instructions -- 4 useful addresses each: 17.6 us.)
typedef double d4c __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_coarse_apply(int S, int N, int nmu, const double* __restrict__ A0inv,
                                                       const double* __restrict__ r, double* __restrict__ z,
                                                       double* __restrict__ prz) {
  __shared__ double red[16 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4, wave = tid >> 6;
  const int row0 = blockIdx.x * 16;
  const long NM = (long)N * nmu;
  const int nsteps = (S + 3) / 4;
  const int row = row0 + li;
  const int col = blockIdx.y * 16 + li;                  // panels of 32 parameters: grid.y = 2, one column half each
  d4c acc = (d4c){0.0, 0.0, 0.0, 0.0};
  for (int i0 = 0; wave + 16 * i0 < nsteps; i0 += 8) {   // 8 steps = 16 loads per lane in flight (16 steps spill at 1024 threads)
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = 4 * (wave + 16 * (i0 + u)) + lk;
      const bool in = k < S;
      a[u] = (in && row < S) ? A0inv[(long)k * S + row] : 0.0;
      b[u] = (in && col < nmu) ? r[(long)k * NM + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[(wave * 4 + q) * 64 + lane] = acc[q];
  __syncthreads();
  if (tid < 256) {
    const int q = tid >> 6;                              // D layout: lane holds rows lk + 4 q of the tile, column li
    double c = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) c += red[(w * 4 + q) * 64 + lane];
    const int s = row0 + lk + 4 * q;
    if (s < S && col < nmu) {
      const long g = (long)s * NM + col;
      z[g] += c;
      prz[(long)col * S + s] += r[g] * c;
    }
  }
}

// The same for ONE column (single-parameter reduced solve, full-order solve): one wave per coarse row, every lane holds
// 16 entries of the row and of the coarse residual (32 loads in flight, one round trip), then a butterfly.
__global__ __launch_bounds__(256) void k_coarse_apply1(int S, long NM, const double* __restrict__ A0inv, const double* __restrict__ r,
                                                       double* __restrict__ z, double* __restrict__ prz) {
  const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  const double* row = A0inv + (long)s * S;
  double acc = 0.0;
  for (int base = 0; base < S; base += 1024) {
    double a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = base + lane + 64 * k;
      a[k] = i < S ? row[i] : 0.0;
      b[k] = i < S ? r[(long)i * NM] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a[k] * b[k];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    const long g = (long)s * NM;
    z[g] += acc;
    prz[s] += r[g] * acc;
  }
}

}  // namespace

void coarse_release(lrbms_ctx* ctx) {
  if (ctx->blas) (void)rocblas_destroy_handle((rocblas_handle)ctx->blas);
  if (ctx->coarse) (void)hipFree(ctx->coarse);
  ctx->blas = nullptr;
  ctx->coarse = nullptr;
  ctx->coarse_cap = 0;
}

namespace {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {    // cross-lane move inside a row of 16 lanes without the LDS crossbar
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// ---- hand-written factorisation of the coarse matrix --------------------------------------------------------
// With the subdomains numbered row by row the coarse matrix is block tridiagonal with blocks of b = (band half-width)
// rows: D_i on the diagonal, E_i = A0[block i + 1, block i] below it.  Block Cholesky  L_ii L_ii^T = D_i - F_{i-1} F_{i-1}^T,
// F_i = E_i L_ii^-T  in ONE workgroup with the three b x b blocks of a step in LDS; the inverses Li_i = L_ii^-1 come out of
// the same sweep as the Cholesky factor (chol_n below), so that the solves in k_bt_inverse are small dense products without
// sequential substitution.  Lout [nb][2][b][b | 1]: Li_i, F_i (row major, padded rows).  flag[0] = 1 if a pivot is not
// positive.  Rows >= S (last block) act as identity.
// rocSOLVER needs 3.5 ms for dpotrf + dpotrs at S = 1024 (latency-bound panel factorisations of a DENSE matrix); this
// kernel needs 0.62 ms (per 32 x 32 block: 13.6 us factor + inverse, 2.9 us F, 2.8 us Schur update, measured with
// wall_clock64 laps) and k_bt_inverse 0.35 ms, because they never touch the zero blocks.
__global__ __launch_bounds__(256) void k_bt_factor(int S, int b, int nb, const double* __restrict__ A0, double* __restrict__ Lout,
                                                   int* __restrict__ flag) {
  extern __shared__ double lds[];
  // rows of the LDS blocks are ld = b | 1 doubles apart: with an even stride the column accesses below (one row per lane)
  // all fall into the same banks, which made every phase of this kernel 3-5x slower
  const int ld = b | 1;
  double* Dm = lds;              // [b][ld] current diagonal block -> L (lower)
  double* Li = Dm + b * ld;      // [b][ld] L^-1 (lower)
  double* Fm = Li + b * ld;      // [b][ld] E_i, then F_i
  const int tid = threadIdx.x;
  if (tid == 0) flag[0] = 0;
  auto a0 = [&](long r, long c) -> double {            // symmetric full matrix, column major; identity beyond S
    if (r >= S || c >= S) return r == c ? 1.0 : 0.0;
    return A0[c * (long)S + r];
  };
  constexpr int PT = 16;                               // b * b <= 4096 entries, 256 threads
  // (row, column) of this thread's u-th entry of a b x b block, once: an integer division by the runtime b costs ~60
  // instructions, and index arithmetic of that kind inside the loops below made this kernel five times slower
  short ri[PT], ci[PT], li[PT];                        // li: position in an LDS block
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const int i = tid + 256 * u;
    ri[u] = (short)(i / b);
    ci[u] = (short)(i - (i / b) * b);
    li[u] = (short)(ri[u] * ld + ci[u]);
  }
  double en[PT], dn[PT];                               // E_blk and D_{blk+1}: loaded while block blk is factorised
  auto fetch = [&](double (&dst)[PT], long ro, long co) {
#pragma unroll
    for (int u = 0; u < PT; ++u) dst[u] = tid + 256 * u < b * b ? a0(ro + ri[u], co + ci[u]) : 0.0;
  };
  // pacc[u] = sum_k A[ri[u]][k] B[ci[u]][k] for this thread's entries: all of them advance together over k (independent
  // accumulators, the LDS reads of a step in flight at once); the number of live entries is uniform, so it selects an
  // instantiation instead of predicating sixteen slots (predicated slots wait for their loads one by one)
  short rl[PT], cl[PT];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const bool live = tid + 256 * u < b * b;
    rl[u] = live ? (short)(ri[u] * ld) : (short)0;
    cl[u] = live ? (short)(ci[u] * ld) : (short)0;
  }
  double pacc[PT];
  auto prod_n = [&](auto nu_c, const double* A, const double* Bm) {
    constexpr int NU = decltype(nu_c)::value;
#pragma unroll
    for (int u = 0; u < PT; ++u) pacc[u] = 0.0;
#pragma unroll 2
    for (int k = 0; k < b; ++k) {
#pragma unroll
      for (int u = 0; u < NU; ++u) pacc[u] += A[rl[u] + k] * Bm[cl[u] + k];
    }
  };
  const int nlive = (b * b + 255) / 256;
  auto prod = [&](const double* A, const double* Bm) {
    if (nlive <= 2) prod_n(std::integral_constant<int, 2>{}, A, Bm);
    else if (nlive <= 4) prod_n(std::integral_constant<int, 4>{}, A, Bm);
    else if (nlive <= 8) prod_n(std::integral_constant<int, 8>{}, A, Bm);
    else prod_n(std::integral_constant<int, 16>{}, A, Bm);
  };
  // Cholesky of the block in Dm together with Li = L^-1.  The block A and W (= identity at the start) live in REGISTERS, a
  // 16 x 16 thread tile owning entries (tr + 16 m, tc + 16 n).  Step k: the owners publish column k of A and row k of W
  // through a double-buffered LDS line, one barrier, then every thread applies the two rank-1 updates
  //   A[r][c] -= A[r][k] A[c][k] / d   (r, c > k),     W[r][c] -= A[r][k] / d * W[k][c]   (r > k),
  // and row k of Li is W[k][:] / sqrt(d).  (An LDS-resident right-looking Cholesky followed by a substitution for Li
  // needed 3 barriers and chains of dependent LDS round trips per column: 37 us per block against ~6 us.)
  double* cbuf = Fm + b * ld;    // [2][2][64]
  auto chol_n = [&](auto nt_c) {
    constexpr int NT = decltype(nt_c)::value;
    const int tr = tid >> 4, tc = tid & 15;
    double A[NT][NT], W[NT][NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int r = tr + 16 * m, c = tc + 16 * n;
        A[m][n] = (r < b && c <= r) ? Dm[r * ld + c] : 0.0;
        W[m][n] = r == c ? 1.0 : 0.0;
      }
    for (int k = 0; k < b; ++k) {
      double* col = cbuf + (k & 1) * 128;
      double* roww = col + 64;
      const int kt = k >> 4, kl = k & 15;
      if (tc == kl) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (n == kt) {
#pragma unroll
            for (int m = 0; m < NT; ++m) col[tr + 16 * m] = A[m][n];
          }
      }
      if (tr == kl) {
#pragma unroll
        for (int m = 0; m < NT; ++m)
          if (m == kt) {
#pragma unroll
            for (int n = 0; n < NT; ++n) roww[tc + 16 * n] = W[m][n];
          }
      }
      __syncthreads();
      // every LDS read of the step is issued before the first use (all indices stay inside the two 64-entry lines; what
      // a read must not contribute is masked afterwards): one LDS round trip per step instead of one per operand
      const double d = col[k];
      const double rowme = roww[tid & 63];
      double cr[NT], cc[NT], rw[NT];
#pragma unroll
      for (int m = 0; m < NT; ++m) cr[m] = col[tr + 16 * m];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        cc[n] = col[tc + 16 * n];
        rw[n] = roww[tc + 16 * n];
      }
      const double rinv = rsqrt(d > 0.0 ? d : 1.0), rinv2 = rinv * rinv;
      if (tid == 0 && !(d > 0.0)) flag[0] = 1;
      if (tid < b) Li[k * ld + tid] = rowme * rinv;
#pragma unroll
      for (int m = 0; m < NT; ++m) cr[m] = tr + 16 * m > k ? cr[m] * rinv2 : 0.0;
#pragma unroll
      for (int n = 0; n < NT; ++n) cc[n] = tc + 16 * n > k ? cc[n] : 0.0;
#pragma unroll
      for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          A[m][n] -= cr[m] * cc[n];
          W[m][n] -= cr[m] * rw[n];
        }
    }
    __syncthreads();
  };
  fetch(dn, 0, 0);
#pragma unroll
  for (int u = 0; u < PT; ++u)
    if (tid + 256 * u < b * b) Dm[li[u]] = dn[u];
  __syncthreads();
  for (int blk = 0; blk < nb; ++blk) {
    const long o = (long)blk * b;
    if (blk + 1 < nb) {                                // the next blocks do not depend on this step: loads in flight below
      fetch(en, o + b, o);
      fetch(dn, o + b, o + b);
    }
    // ---- Cholesky of Dm and L^-1 in one sweep, one barrier per column (see chol_n above)
    if (b <= 16) chol_n(std::integral_constant<int, 1>{});
    else if (b <= 32) chol_n(std::integral_constant<int, 2>{});
    else if (b <= 48) chol_n(std::integral_constant<int, 3>{});
    else chol_n(std::integral_constant<int, 4>{});
    double* out = Lout + (long)blk * 2 * b * ld;       // blocks keep the padded row stride: the consumer stages them flat
    for (int i = tid; i < b * ld; i += 256) out[i] = Li[i];
    if (blk + 1 < nb) {
      // ---- F = E Li^T  (E = A0[block blk + 1, block blk]), then the next diagonal block D - F F^T
#pragma unroll
      for (int u = 0; u < PT; ++u)
        if (tid + 256 * u < b * b) Dm[li[u]] = en[u];
      __syncthreads();
      prod(Dm, Li);                                    // Li is zero above its diagonal: the full k range is correct
#pragma unroll
      for (int u = 0; u < PT; ++u)
        if (tid + 256 * u < b * b) Fm[li[u]] = pacc[u];
      __syncthreads();
      prod(Fm, Fm);
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        if (tid + 256 * u < b * b) {
          out[b * ld + li[u]] = Fm[li[u]];
          Dm[li[u]] = dn[u] - pacc[u];
        }
      }
      __syncthreads();
    }
  }
}

// A0^-1 from the block factor: every workgroup owns 16 columns of the identity.  Forward  y_i = Li_i (e_i - F_{i-1} y_{i-1}),
// backward  x_i = Li_i^T (y_i - F_i^T x_{i+1}); the two b x b blocks of a step are staged in LDS with all loads in flight,
// y / x travel through the output columns.  X column major [S][S] (symmetric result).
__global__ __launch_bounds__(256) void k_bt_inverse(int S, int b, int nb, const double* __restrict__ Lf, double* __restrict__ X) {
  extern __shared__ double lds[];
  const int ld = b | 1;          // padded row stride of the factor blocks (as written by k_bt_factor)
  double* Lb = lds;              // [b][ld] Li_i
  double* Fb = Lb + b * ld;      // [b][ld] F block of the step
  double* v = Fb + b * ld;        // [b][16] previous block of the solution
  double* w = v + b * 16;        // [b][16] work
  const int tid = threadIdx.x, c0 = blockIdx.x * 16;
  auto stage = [&](double* dst, const double* src) {
    for (int base = 0; base < b * ld; base += 2048) {
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = base + tid + 256 * k;
        t[k] = i < b * ld ? src[i] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = base + tid + 256 * k;
        if (i < b * ld) dst[i] = t[k];
      }
    }
  };
  // ---- forward
  for (int blk = 0; blk < nb; ++blk) {
    const double* f = Lf + (long)blk * 2 * b * ld;
    __syncthreads();
    stage(Lb, f);
    if (blk > 0) stage(Fb, f - b * ld);                 // F_{blk-1}
    __syncthreads();
    for (int i = tid; i < b * 16; i += 256) {          // w = e_blk - F_{blk-1} y_{blk-1}
      const int r = i / 16, c = i % 16;
      double acc = ((long)blk * b + r == c0 + c) ? 1.0 : 0.0;
      if (blk > 0)
#pragma unroll 8
        for (int k = 0; k < b; ++k) acc -= Fb[r * ld + k] * v[k * 16 + c];
      w[i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < b * 16; i += 256) {          // y = Li w (Li lower triangular)
      const int r = i / 16, c = i % 16;
      double acc = 0.0;
#pragma unroll 8
      for (int k = 0; k <= r; ++k) acc += Lb[r * ld + k] * w[k * 16 + c];
      v[i] = acc;
      const long row = (long)blk * b + r;
      if (row < S && c0 + c < S) X[(long)(c0 + c) * S + row] = acc;
    }
  }
  // ---- backward
  for (int blk = nb - 1; blk >= 0; --blk) {
    const double* f = Lf + (long)blk * 2 * b * ld;
    __syncthreads();
    stage(Lb, f);
    if (blk + 1 < nb) stage(Fb, f + b * ld);            // F_blk
    __syncthreads();
    for (int i = tid; i < b * 16; i += 256) {          // w = y_blk - F_blk^T x_{blk+1}
      const int r = i / 16, c = i % 16;
      const long row = (long)blk * b + r;
      double acc = (row < S && c0 + c < S) ? X[(long)(c0 + c) * S + row] : 0.0;
      if (blk + 1 < nb)
#pragma unroll 8
        for (int k = 0; k < b; ++k) acc -= Fb[k * ld + r] * v[k * 16 + c];
      w[i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < b * 16; i += 256) {          // x = Li^T w
      const int r = i / 16, c = i % 16;
      double acc = 0.0;
#pragma unroll 8
      for (int k = r; k < b; ++k) acc += Lb[k * ld + r] * w[k * 16 + c];
      v[i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < b * 16; i += 256) {
      const int r = i / 16, c = i % 16;
      const long row = (long)blk * b + r;
      if (row < S && c0 + c < S) X[(long)(c0 + c) * S + row] = v[i];
    }
  }
}

}  // namespace

// The dense coarse machinery, shared with the full-order solver (fom.hip):
//   coarse_begin   scratch A0 (zeroed) for the caller to fill with the S x S coarse matrix (either triangle, column major);
//                  *A0 = nullptr if the coarse level is not available (switched off, S too small or too large)
//   coarse_finish  factorises A0 and forms its inverse; *A0inv_out = nullptr if A0 is not positive definite.  Synchronises.
int coarse_begin(lrbms_ctx* ctx, double** A0_out, hipStream_t st) {
  *A0_out = nullptr;
  const long S = ctx->S;
  if (ctx->opt_coarse == 0 || S < 4 || S > 4096) return LRBMS_OK;      // LRBMS_OPT_COARSE
  const long need = 2 * S * S + 16 + 2 * (S + 64) * 65;     // A0, A0inv, info / flag, block factor
  if (ctx->coarse_cap < need) {
    if (ctx->coarse) (void)hipFree(ctx->coarse);
    ctx->coarse = nullptr;
    ctx->coarse_cap = 0;
    LRBMS_HIP_CHECK(ctx, hipMalloc((void**)&ctx->coarse, sizeof(double) * need));
    ctx->coarse_cap = need;
  }
  hipLaunchKernelGGL(k_coarse_init, dim3(2048), dim3(256), 0, st, S, ctx->coarse, ctx->coarse + S * S);
  LRBMS_LAUNCH_CHECK(ctx);
  *A0_out = ctx->coarse;
  return LRBMS_OK;
}

int coarse_finish(lrbms_ctx* ctx, const double** A0inv_out, hipStream_t st) {
  *A0inv_out = nullptr;
  const long S = ctx->S;
  {
    // band half-width of the coarse matrix = largest distance to a neighbour in the subdomain numbering (the row length
    // of a Cartesian subdomain grid): up to 64 the block-tridiagonal kernels do the job, beyond that rocSOLVER
    int bw = 1;
    for (long i = 0; i < S * 5; ++i) {
      const int t = ctx->nbr_host[i];
      if (t >= 0 && t < S) {
        const int dist = t > (int)(i / 5) ? t - (int)(i / 5) : (int)(i / 5) - t;
        bw = dist > bw ? dist : bw;
      }
    }
    if (bw <= 64 && ctx->opt_coarse != 2) {
      const int b = bw, nb = (int)((S + b - 1) / b);
      double* A0 = ctx->coarse;
      double* A0inv = A0 + S * S;
      int* flag = reinterpret_cast<int*>(A0inv + S * S);
      double* Lf = A0inv + S * S + 16;
      const size_t lds_f = sizeof(double) * (3 * b * (b | 1) + 256), lds_i = sizeof(double) * (2 * b * (b | 1) + 32 * b);
      if (lds_f > 64 * 1024)
        LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bt_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      if (lds_i > 64 * 1024)
        LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bt_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_i));
      hipLaunchKernelGGL(k_bt_factor, dim3(1), dim3(256), lds_f, st, (int)S, b, nb, A0, Lf, flag);
      hipLaunchKernelGGL(k_bt_inverse, dim3((unsigned)((S + 15) / 16)), dim3(256), lds_i, st, (int)S, b, nb, Lf, A0inv);
      LRBMS_LAUNCH_CHECK(ctx);
      int hflag = 0;
      LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st));
      LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
      if (hflag == 0) *A0inv_out = A0inv;              // otherwise not positive definite: no coarse level
      return LRBMS_OK;
    }
  }
  if (!ctx->blas) {
    rocblas_handle h = nullptr;
    if (rocblas_create_handle(&h) != rocblas_status_success) return lrbms_fail(ctx, LRBMS_E_HIP, "rocblas_create_handle failed");
    ctx->blas = h;
  }
  rocblas_handle h = (rocblas_handle)ctx->blas;
  if (rocblas_set_stream(h, st) != rocblas_status_success) return lrbms_fail(ctx, LRBMS_E_HIP, "rocblas_set_stream failed");
  double* A0 = ctx->coarse;
  double* A0inv = A0 + S * S;
  rocblas_int* info = (rocblas_int*)(A0inv + S * S);
  if (rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)S, A0, (rocblas_int)S, info) != rocblas_status_success)
    return lrbms_fail(ctx, LRBMS_E_HIP, "rocsolver_dpotrf failed");
  rocblas_int hinfo = 0;
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(&hinfo, info, sizeof(rocblas_int), hipMemcpyDeviceToHost, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (hinfo != 0) return LRBMS_OK;                       // not positive definite (zero first basis vector): no coarse level
  if (rocsolver_dpotrs(h, rocblas_fill_lower, (rocblas_int)S, (rocblas_int)S, A0, (rocblas_int)S, A0inv, (rocblas_int)S) !=
      rocblas_status_success)
    return lrbms_fail(ctx, LRBMS_E_HIP, "rocsolver_dpotrs failed");
  *A0inv_out = A0inv;
  return LRBMS_OK;
}

int launch_coarse_apply(lrbms_ctx* ctx, int N, int nmu, const double* A0inv, const double* r, double* z, double* prz, hipStream_t st) {
  if (nmu == 1)
    hipLaunchKernelGGL(k_coarse_apply1, dim3((ctx->S + 3) / 4), dim3(256), 0, st, ctx->S, (long)N, A0inv, r, z, prz);
  else
    hipLaunchKernelGGL(k_coarse_apply, dim3((ctx->S + 15) / 16, (nmu + 15) / 16), dim3(1024), 0, st, ctx->S, N, nmu, A0inv, r, z, prz);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// Coarse inverse for the combined reduced blocks Amu (nullptr if not available).  Synchronises `st`.
static int coarse_setup(lrbms_ctx* ctx, int N, const double* Amu, const double** A0inv_out, hipStream_t st) {
  *A0inv_out = nullptr;
  double* A0 = nullptr;
  if (int rc = coarse_begin(ctx, &A0, st)) return rc;
  if (!A0) return LRBMS_OK;
  const long S = ctx->S;
  hipLaunchKernelGGL(k_coarse_entries, dim3((unsigned)((S * 5 + 255) / 256)), dim3(256), 0, st, (int)S, N, ctx->nbr, Amu, A0);
  LRBMS_LAUNCH_CHECK(ctx);
  return coarse_finish(ctx, A0inv_out, st);
}

// Prebuilt preconditioner, caller-owned: pc[0] = 1 if the coarse inverse is present, pc[1] = N, then Dinv [S][N][N], A0inv [S][S]
int64_t reduced_precond_size(lrbms_ctx* ctx, int N) { return 2 + (int64_t)ctx->S * N * N + (int64_t)ctx->S * ctx->S; }

int launch_reduced_precond_build(lrbms_ctx* ctx, int Q, int N, const double* theta, const double* B_sys, double* work, double* pc,
                                 hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_precond_build needs all subdomains on one rank");
  if (N > 64 || N < 1 || Q < 1 || Q > 8) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_precond_build: bad N / Q");
  const long S = ctx->S;
  QVec th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  double* Amu = work;                                   // lrbms_reduced_solve_work_size doubles are enough
  double* Dinv = pc + 2;
  double* A0inv = Dinv + S * N * N;
  const long per_q = S * 5 * N * N;
  hipLaunchKernelGGL(k_assemble_mu, dim3((unsigned)((per_q + 255) / 256 > 8192 ? 8192 : (per_q + 255) / 256)), dim3(256), 0, st,
                     per_q, Q, th, B_sys, Amu);
  if (int rc = launch_block_inverse(ctx, (int)S, N, Amu, Dinv, 5, 2, st)) return rc;
  LRBMS_LAUNCH_CHECK(ctx);
  const double* built = nullptr;
  if (int rc = coarse_setup(ctx, N, Amu, &built, st)) return rc;
  const double head[2] = {built ? 1.0 : 0.0, (double)N};
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(pc, head, sizeof(head), hipMemcpyHostToDevice, st));
  if (built) LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(A0inv, built, sizeof(double) * S * S, hipMemcpyDeviceToDevice, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return LRBMS_OK;
}

// Dinv / A0inv of the preconditioner in use (ctx->user_pc) for basis size N, or nullptrs
static int user_precond(lrbms_ctx* ctx, int N, const double** Dinv, const double** A0inv, hipStream_t st) {
  *Dinv = nullptr;
  *A0inv = nullptr;
  if (!ctx->user_pc || ctx->user_pc_N != N) return LRBMS_OK;
  double head[2];
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(head, ctx->user_pc, sizeof(head), hipMemcpyDeviceToHost, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if ((int)head[1] != N) return lrbms_fail(ctx, LRBMS_E_INVALID, "the preconditioner in use was built for another basis size");
  *Dinv = ctx->user_pc + 2;
  if (head[0] != 0.0 && ctx->opt_coarse != 0) *A0inv = *Dinv + (long)ctx->S * N * N;
  return LRBMS_OK;
}

int64_t reduced_solve_work_size(lrbms_ctx* ctx, int N) {
  const long S = ctx->S;
  return S * 5 * N * N + S * N * N + 5 * S * N + 4 * S + 16;
}

namespace {

struct RedCg {
  double *Amu, *Dinv, *r, *z, *p[2], *y, *prz[2], *ppap, *prr;
  const double* A0inv = nullptr;      // coarse level (coarse_setup), or nullptr
  void carve(double* work, long S, int N) {
    Amu = work;
    Dinv = Amu + S * 5 * N * N;
    r = Dinv + S * N * N;
    z = r + S * N;
    p[0] = z + S * N;
    p[1] = p[0] + S * N;
    y = p[1] + S * N;
    prz[0] = y + S * N;
    prz[1] = prz[0] + S;
    ppap = prz[1] + S;
    prr = ppap + S;
  }
};

// sum of a device array of S doubles on the host (fixed order); synchronises `st`
int host_sum(lrbms_ctx* ctx, const double* dev, std::vector<double>& host, double* out, hipStream_t st) {
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host.data(), dev, sizeof(double) * host.size(), hipMemcpyDeviceToHost, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  double acc = 0.0;
  for (double v : host) acc += v;
  *out = acc;
  return LRBMS_OK;
}

// Block-Jacobi PCG on Amu x = rhs: on entry x holds the start value and b.r the start residual; iterates until
// |r| <= rtol * sqrt(ref2) (ref2 < 0: relative to the start residual).  Two kernels per iteration (see k_cg2_*); the
// convergence check (8 KB to the host) is placed where the observed rate predicts convergence, at most 40 iterations apart.
int red_cg_run(lrbms_ctx* ctx, int N, RedCg& b, double* x, double ref2, double rtol, int max_iter, int* its, double* rel_out,
               hipStream_t st) {
  const int S = ctx->S;
  std::vector<double> host(S);
  const size_t lds_mv = sizeof(double) * 10 * N, lds_up = sizeof(double) * 2 * N;
  if (N % 2 == 0)
    hipLaunchKernelGGL(k_cg2_update<true>, dim3(S), dim3(256), lds_up, st, S, N, b.Dinv, b.prz[0], b.ppap, 1, x, b.r, b.p[0], b.y, b.z,
                       b.prz[0], b.prr);
  else
    hipLaunchKernelGGL(k_cg2_update<false>, dim3(S), dim3(256), lds_up, st, S, N, b.Dinv, b.prz[0], b.ppap, 1, x, b.r, b.p[0], b.y, b.z,
                       b.prz[0], b.prr);
  if (b.A0inv)
    if (int rc = launch_coarse_apply(ctx, N, 1, b.A0inv, b.r, b.z, b.prz[0], st)) return rc;
  LRBMS_LAUNCH_CHECK(ctx);
  double rr = 0.0;
  if (int rc = host_sum(ctx, b.prr, host, &rr, st)) return rc;
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
      if (N % 2 == 0) {
        hipLaunchKernelGGL(k_cg2_matvec<true>, dim3(S), dim3(256), lds_mv, st, ctx->nbr, S, N, b.Amu, b.z, b.p[o], b.prz[c], b.prz[o],
                           it == 0 ? 1 : 0, b.p[c], b.y, b.ppap);
        hipLaunchKernelGGL(k_cg2_update<true>, dim3(S), dim3(256), lds_up, st, S, N, b.Dinv, b.prz[c], b.ppap, 0, x, b.r, b.p[c], b.y,
                           b.z, b.prz[o], b.prr);
      } else {
        hipLaunchKernelGGL(k_cg2_matvec<false>, dim3(S), dim3(256), lds_mv, st, ctx->nbr, S, N, b.Amu, b.z, b.p[o], b.prz[c], b.prz[o],
                           it == 0 ? 1 : 0, b.p[c], b.y, b.ppap);
        hipLaunchKernelGGL(k_cg2_update<false>, dim3(S), dim3(256), lds_up, st, S, N, b.Dinv, b.prz[c], b.ppap, 0, x, b.r, b.p[c], b.y,
                           b.z, b.prz[o], b.prr);
      }
      if (b.A0inv)
        if (int rc = launch_coarse_apply(ctx, N, 1, b.A0inv, b.r, b.z, b.prz[o], st)) return rc;
    }
    LRBMS_LAUNCH_CHECK(ctx);
    if (int rc = host_sum(ctx, b.prr, host, &rr, st)) return rc;
    rel = sqrt(rr / ref2);
    if (!(rel == rel)) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "reduced CG: NaN residual (system not SPD?)");
    const double rate = log(rel) / it;                 // log-residual per iteration so far (relative to ref2)
    block = 10;
    if (rel > rtol && rate < 0.0) {                    // aim a little short: CG converges superlinearly
      const double need = 0.8 * (log(rtol) - log(rel)) / rate;
      block = need < 2.0 ? 2 : need > 40.0 ? 40 : (int)need;
    }
  }
  *its = it;
  *rel_out = rel;
  return LRBMS_OK;
}

}  // namespace

int launch_reduced_solve(lrbms_ctx* ctx, int Q, int N, const double* theta, const double* B_sys, const double* rhs_red,
                         double* work, double* u, double rtol, int max_iter, double* info, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_solve needs all subdomains on one rank");
  if (N > 64) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_solve: N > 64 not supported by the block inverse");
  const int S = ctx->S;
  QVec th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  RedCg b;
  b.carve(work, S, N);
  const long per_q = (long)S * 5 * N * N;
  const long vec = (long)S * N;
  hipLaunchKernelGGL(k_assemble_mu, dim3((unsigned)((per_q + 255) / 256 > 8192 ? 8192 : (per_q + 255) / 256)), dim3(256),
                     0, st, per_q, Q, th, B_sys, b.Amu);
  LRBMS_LAUNCH_CHECK(ctx);
  LRBMS_LAUNCH_CHECK(ctx);
  const double* pcD = nullptr;
  if (int rc = user_precond(ctx, N, &pcD, &b.A0inv, st)) return rc;
  if (pcD) {
    b.Dinv = const_cast<double*>(pcD);                   // read only below
  } else {
    if (int rc = launch_block_inverse(ctx, (int)S, N, b.Amu, b.Dinv, 5, 2, st)) return rc;
    LRBMS_LAUNCH_CHECK(ctx);
    if (int rc = coarse_setup(ctx, N, b.Amu, &b.A0inv, st)) return rc;
  }
  // x0 = 0, r0 = b
  LRBMS_HIP_CHECK(ctx, hipMemsetAsync(u, 0, sizeof(double) * vec, st));
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(b.r, rhs_red, sizeof(double) * vec, hipMemcpyDeviceToDevice, st));
  int it = 0;
  double rel = 0.0;
  if (int rc = red_cg_run(ctx, N, b, u, -1.0, rtol, max_iter, &it, &rel, st)) return rc;
  if (info) { info[0] = it; info[1] = rel; }
  if (rel > rtol) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve: CG did not reach rtol");
  return LRBMS_OK;
}

// Reduced implicit Euler (the reduced counterpart of InstationaryDuneDiscretization._solve,
// discretize_parabolic_block_swipdg.py:28-40):  (M_red + dt sum_q theta_q B_q) u_{k+1} = M_red u_k + dt rhs_red.
// The step operator and its block-Jacobi preconditioner are built once, every step is a warm-started PCG with the
// kernels of lrbms_reduced_solve.  U [nt+1][S][N]: U[0] initial value (input), U[1..nt] written.
int launch_reduced_implicit_euler(lrbms_ctx* ctx, int Q, int N, const double* theta, double dt, int nt, const double* B_sys,
                                  const double* M_red, const double* rhs_red, double* work, double* U, double rtol,
                                  int max_iter, double* info, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_implicit_euler needs all subdomains on one rank");
  if (N > 64 || N < 1) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_implicit_euler: N > 64 not supported by the block inverse");
  if (Q < 1 || Q > 8 || nt < 1 || !(dt > 0.0) || !(rtol > 0.0) || max_iter < 1)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_implicit_euler: bad Q / nt / dt / rtol / max_iter");
  const int S = ctx->S;
  QVec th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? dt * theta[q] : 0.0;
  RedCg b;
  b.carve(work, S, N);
  const long per_q = (long)S * 5 * N * N;
  const long vec = (long)S * N;
  hipLaunchKernelGGL(k_assemble_mu_mass, dim3((unsigned)((per_q + 255) / 256 > 8192 ? 8192 : (per_q + 255) / 256)), dim3(256),
                     0, st, per_q, Q, N, th, B_sys, M_red, b.Amu);
  LRBMS_LAUNCH_CHECK(ctx);
  if (int rc = launch_block_inverse(ctx, (int)S, N, b.Amu, b.Dinv, 5, 2, st)) return rc;
  LRBMS_LAUNCH_CHECK(ctx);
  if (int rc = coarse_setup(ctx, N, b.Amu, &b.A0inv, st)) return rc;
  std::vector<double> host(S);
  long total_it = 0;
  double worst = 0.0;
  for (int step = 0; step < nt; ++step) {
    const double* uk = U + (long)step * vec;
    double* u = U + (long)(step + 1) * vec;
    LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(u, uk, sizeof(double) * vec, hipMemcpyDeviceToDevice, st));
    // warm start: r = M_red u_k + dt b - (M_red + dt A) u_k; the matvec kernel with first = 1 takes its direction from `z`
    if (N % 2 == 0)
      hipLaunchKernelGGL(k_cg2_matvec<true>, dim3(S), dim3(256), sizeof(double) * 10 * N, st, ctx->nbr, S, N, b.Amu, uk, b.p[1], b.prz[0],
                         b.prz[1], 1, b.p[0], b.y, b.ppap);
    else
      hipLaunchKernelGGL(k_cg2_matvec<false>, dim3(S), dim3(256), sizeof(double) * 10 * N, st, ctx->nbr, S, N, b.Amu, uk, b.p[1], b.prz[0],
                         b.prz[1], 1, b.p[0], b.y, b.ppap);
    hipLaunchKernelGGL(k_red_step_residual, dim3(S), dim3(64), sizeof(double) * N, st, N, dt, M_red, uk, rhs_red, b.y, b.r, b.ppap);
    LRBMS_LAUNCH_CHECK(ctx);
    double ref2 = 0.0;
    if (int rc = host_sum(ctx, b.ppap, host, &ref2, st)) return rc;
    if (ref2 == 0.0) continue;                           // zero right-hand side: u_{k+1} = u_k = 0
    int it = 0;
    double rel = 0.0;
    if (int rc = red_cg_run(ctx, N, b, u, ref2, rtol, max_iter, &it, &rel, st)) return rc;
    total_it += it;
    if (rel > worst) worst = rel;
    if (rel > rtol) {
      if (info) { info[0] = (double)total_it; info[1] = worst; }
      return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "reduced_implicit_euler: CG did not reach rtol");
    }
  }
  if (info) { info[0] = (double)total_it; info[1] = worst; }
  return LRBMS_OK;
}

// Time-stepping residual of the reduced model (ParabolicEstimator.estimate, estimators.py:146-148, with d = rd):
// out[l][s] = y^T M_red[s]^{-1} y,  y = (sum_q theta_q B_q dU_l)_s, for L difference vectors dU [L][S][N].
// work: S*5*N*N + S*N*N + S*N + S doubles.
int launch_reduced_time_residual(lrbms_ctx* ctx, int Q, int N, int L, const double* theta, const double* B_sys, const double* M_red,
                                 const double* dU, double* work, double* out, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_time_residual needs all subdomains on one rank");
  if (N > 64 || N < 1 || L < 1 || Q < 1 || Q > 8) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_time_residual: bad N / L / Q");
  const int S = ctx->S;
  QVec th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  double* Amu = work;
  double* Minv = Amu + (long)S * 5 * N * N;
  double* y = Minv + (long)S * N * N;
  double* partial = y + (long)S * N;
  const long per_q = (long)S * 5 * N * N;
  hipLaunchKernelGGL(k_assemble_mu, dim3((unsigned)((per_q + 255) / 256 > 8192 ? 8192 : (per_q + 255) / 256)), dim3(256),
                     0, st, per_q, Q, th, B_sys, Amu);
  if (int rc = launch_block_inverse(ctx, (int)S, N, M_red, Minv, 1, 0, st)) return rc;
  LRBMS_LAUNCH_CHECK(ctx);
  for (int l = 0; l < L; ++l) {
    hipLaunchKernelGGL(k_cg_matvec, dim3(S), dim3(64), sizeof(double) * 5 * N, st, ctx->nbr, N, Amu, dU + (long)l * S * N, y, partial);
    hipLaunchKernelGGL(k_red_inv_norm2, dim3(S), dim3(64), sizeof(double) * N, st, N, Minv, y, out + (long)l * S);
  }
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// Elliptic-reconstruction terms of _estimate_elliptic with d = rd (estimators.py:65-68, :80-83), per subdomain:
//   out = y^T Minv y - b^T Minv b - 2 (Minv (y - b))^T G_ud ur,   y = (A_red(mu) u)_s, b = rhs_red[s], Minv = M_red[s]^-1,
//   ur = theta_q u_nbr(slot)[j] in column order (slot, q, j), G_ud [S][N][5 Q N] = V^T M Div Rt (projected r_ud_s).
// One wave per subdomain.
__global__ __launch_bounds__(64) void k_red_recon_terms(const int* __restrict__ nbr, int Q, int N, QVec theta,
                                                        const double* __restrict__ Minv, const double* __restrict__ y,
                                                        const double* __restrict__ b, const double* __restrict__ G_ud,
                                                        const double* __restrict__ u, double* __restrict__ out) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, lane = threadIdx.x, C = 5 * Q * N;
  double* ys = lds;            // [N]  y
  double* ds = ys + N;         // [N]  y - b
  double* bs = ds + N;         // [N]  b
  double* ur = bs + N;         // [C]
  for (int i = lane; i < N; i += 64) {
    const double yv = y[(long)s * N + i], bv = b[(long)s * N + i];
    ys[i] = yv;
    bs[i] = bv;
    ds[i] = yv - bv;
  }
  for (int c = lane; c < C; c += 64) {
    const int slot = c / (Q * N), rem = c - slot * Q * N, q = rem / N, j = rem - q * N;
    const int s2 = nbr[s * 5 + slot];
    ur[c] = s2 >= 0 ? theta.v[q] * u[(long)s2 * N + j] : 0.0;
  }
  __syncthreads();
  double acc = 0.0;
  for (int r = lane; r < N; r += 64) {
    const double* row = Minv + ((long)s * N + r) * N;
    double zy = 0.0, zb = 0.0, w = 0.0;
    for (int c = 0; c < N; ++c) {
      zy += row[c] * ys[c];
      zb += row[c] * bs[c];
      w += row[c] * ds[c];
    }
    const double* g = G_ud + ((long)s * N + r) * C;
    double gu = 0.0;
    for (int c = 0; c < C; ++c) gu += g[c] * ur[c];
    acc += zy * ys[r] - zb * bs[r] - 2.0 * w * gu;      // Minv is symmetric: (Minv d)_r (G_ud ur)_r summed over r
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[s] = acc;
}

int launch_reduced_reconstruction_terms(lrbms_ctx* ctx, int Q, int N, int L, const double* theta, const double* B_sys,
                                        const double* M_red, const double* rhs_red, const double* G_ud, const double* U,
                                        double* work, double* out, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_reconstruction_terms needs all subdomains on one rank");
  if (N > 64 || N < 1 || L < 1 || Q < 1 || Q > 8) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_reconstruction_terms: bad N / L / Q");
  const int S = ctx->S;
  QVec th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  double* Amu = work;                                  // lrbms_reduced_time_residual_work_size doubles
  double* Minv = Amu + (long)S * 5 * N * N;
  double* y = Minv + (long)S * N * N;
  double* partial = y + (long)S * N;
  const long per_q = (long)S * 5 * N * N;
  hipLaunchKernelGGL(k_assemble_mu, dim3((unsigned)((per_q + 255) / 256 > 8192 ? 8192 : (per_q + 255) / 256)), dim3(256),
                     0, st, per_q, Q, th, B_sys, Amu);
  if (int rc = launch_block_inverse(ctx, (int)S, N, M_red, Minv, 1, 0, st)) return rc;
  LRBMS_LAUNCH_CHECK(ctx);
  for (int l = 0; l < L; ++l) {
    const double* ul = U + (long)l * S * N;
    hipLaunchKernelGGL(k_cg_matvec, dim3(S), dim3(64), sizeof(double) * 5 * N, st, ctx->nbr, N, Amu, ul, y, partial);
    hipLaunchKernelGGL(k_red_recon_terms, dim3(S), dim3(64), sizeof(double) * (3 * N + 5 * Q * N), st, ctx->nbr, Q, N, th, Minv, y,
                       rhs_red, G_ud, ul, out + (long)l * S);
  }
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// =========================================================================================================
// Batched reduced solve: nmu parameters at once.  The affine structure A(mu) = sum_q theta_q(mu) B_q means the
// block-sparse matvec of ALL parameters reads each projected block B_q[s][slot] exactly once per iteration
// (131 MB at config 3, Infinity-Cache / HBM bound) and applies it to an [N x nmu] panel, so the per-iteration cost
// and the launch overhead are amortised over the batch: this is what turns "one mu-solve" into mu-solves / s.
// One block-Jacobi preconditioner (inverse diagonal blocks at the batch-mean theta) serves every mu: it is SPD, so
// CG converges for each column; columns that converged early keep iterating harmlessly (alpha = 0 when rz = 0).
// Vectors are [S][N][nmu] (mu fastest); all reductions are fixed-order (per-subdomain partials + one-workgroup sum).
namespace {

constexpr int BMAX = 64;   // max parameters per group of the batched solve (stride of its scalar arrays)
constexpr int TBMAX = 32;  // parameters a by-value ThetaBatch holds (the VALU cross-check kernels and the estimate passes use <= 16)
constexpr int BCG_KMAX = 5;   // outputs per thread of the batched matvec: N * nmu <= 256 * BCG_KMAX = 1280

struct ThetaBatch { double v[TBMAX * 8]; };   // theta[m][q], q < 8

// direction + matvec:  p_new = z + beta p_old (own + neighbour rows, into LDS; own rows written to p_out),
// y_s = sum_slot sum_q theta_q B_q[s][slot] p_new[nbr(s, slot)],  partial[s][m] = p_new_s . y_s
template <int BCG_K>
__global__ __launch_bounds__(256) void k_bcg_matvec(int S, const int* __restrict__ nbr, int Q, int N, int nmu, ThetaBatch th,
                                                    const double* __restrict__ B_sys, const double* __restrict__ z,
                                                    const double* __restrict__ p_old, const double* __restrict__ beta, int first,
                                                    double* __restrict__ p_out, double* __restrict__ y,
                                                    double* __restrict__ partial) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int NM = N * nmu;
  double* Pt = lds;              // [5][N][nmu]
  double* Bs = Pt + 5 * NM;      // [N][N]
  double* red = Bs + N * N;      // [256]
  for (int i = tid; i < 5 * NM; i += 256) {
    const int slot = i / NM, rem = i - slot * NM, m = rem % nmu;
    const int s2 = nbr[s * 5 + slot];
    double v = 0.0;
    if (s2 >= 0) {
      const long g = (long)s2 * NM + rem;
      v = first ? z[g] : z[g] + beta[m] * p_old[g];
      if (slot == 2) p_out[g] = v;
    }
    Pt[i] = v;
  }
  double acc[BCG_K];                               // outputs it = tid + 256 k < N nmu  (N nmu <= 256 BCG_K)
#pragma unroll
  for (int k = 0; k < BCG_K; ++k) acc[k] = 0.0;
  for (int slot = 0; slot < 5; ++slot) {
    if (nbr[s * 5 + slot] < 0) continue;
    for (int q = 0; q < Q; ++q) {
      __syncthreads();
      const double* B = B_sys + ((((long)q * S + s) * 5 + slot) * N) * N;
      for (int i = tid; i < N * N; i += 256) Bs[i] = B[i];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < BCG_K; ++k) {
        const int it = tid + 256 * k;
        if (it < NM) {
          const int r = it / nmu, m = it - r * nmu;
          const double* pc = Pt + slot * NM + m;
          double sum = 0.0;
          for (int c = 0; c < N; ++c) sum += Bs[r * N + c] * pc[c * nmu];
          acc[k] += th.v[m * 8 + q] * sum;
        }
      }
    }
  }
  __syncthreads();
  double* Ys = Bs;   // reuse (N * nmu <= N * N is not guaranteed): use red-free region of Pt slot 0 instead
  (void)Ys;
#pragma unroll
  for (int k = 0; k < BCG_K; ++k) {
    const int it = tid + 256 * k;
    if (it < NM) y[(long)s * NM + it] = acc[k];
  }
  // partial[s][m] = sum_r p_new[s][r][m] * y[s][r][m]: stage products in LDS slot 0 of Pt, then sum over r per m
  __syncthreads();
#pragma unroll
  for (int k = 0; k < BCG_K; ++k) {
    const int it = tid + 256 * k;
    if (it < NM) Pt[it] = acc[k] * Pt[2 * NM + it];
  }
  __syncthreads();
  if (tid < nmu) {
    double sum = 0.0;
    for (int r = 0; r < N; ++r) sum += Pt[r * nmu + tid];
    partial[(long)tid * gridDim.x + s] = sum;          // [m][S]: the reduce reads contiguous runs
  }
  (void)red;
}

// The same step on the fp64 matrix cores, for batches of at most 16 parameters (the panel is padded to 16 columns):
// y_s (N x 16) = sum_{slot, q} B_q[s][slot] (N x N) * (P_slot diag(theta_q)) (N x 16).  Wave w owns row tile w.  Each
// block is staged once into LDS with coalesced loads (the next block's loads are issued before the MFMAs of the current
// one), and read from there as MFMA A operand: 1 LDS read per MFMA, where the VALU form above reads two LDS operands
// per multiply-add and is bound by the LDS pipe (78 us per iteration at config 3 for 131 MB of blocks; this form:
// 766 -> 968 mu-solves/s).  (Folding the two single-workgroup reductions of an iteration into the producing kernels
// with a "last workgroup reduces" ticket was measured too: the device-scope fences it needs cost far more on this
// multi-XCD part than the two launches they save: 348 mu-solves/s.  All 10 block loads of a subdomain in flight at once
// (inline-asm loads, 70 per lane, consumed block by block behind exact s_waitcnt vmcnt(n)): 1 030 vs 1 190 mu-solves/s --
// 176 VGPRs leave two workgroups per CU instead of three, and the kernel is not short of loads in flight but of
// bandwidth: 131 MB per launch stream at 2.5 TB/s here, 3.8 TB/s is the most any kernel of this library reaches.)
// NC = 32 (round 4): panels of 32 parameters -- every block read serves twice as many solves; eight waves: wave & 3 = row tile,
// wave >> 2 = column half (the same per-wave work as the 16-column form, twice the waves to hide the block loads behind).
typedef double d4m __attribute__((ext_vector_type(4)));
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt(0), i.e. it waits for the block loads just
// requested for the NEXT step -- the prefetch would overlap with nothing (the same idiom as lds_barrier in fused.hip).
__device__ inline void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NC>
__global__ __launch_bounds__(16 * NC) void k_bcg_matvec_mfma(int S, const int* __restrict__ nbr, int Q, int N, int nmu, ThetaBatch th,
                                                         const double* __restrict__ B_sys, const double* __restrict__ z,
                                                         const double* __restrict__ p_old, const double* __restrict__ beta,
                                                         int first, double* __restrict__ p_out, double* __restrict__ y,
                                                         double* __restrict__ partial) {
  extern __shared__ double lds[];
  constexpr int NTH = 16 * NC;                         // threads: 256 / 512
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = tid >> 6, rt = wave & 3, ch = wave >> 2;      // row tile, column half
  const int NM = N * nmu;
  const int KP = (N + 3) & ~3;                         // K padded to a multiple of 4 (zero rows / columns)
  const int LDB = N + ((4 - N % 8) + 8) % 8;           // row stride == 4 (mod 8) doubles: conflict-free A-operand reads
  // LDS (NC = 16): 40 032 B at N = 40, so that FOUR workgroups fit a CU and all 1 024 of config 3 are resident at once (with the
  // 47.6 KB of a tile-padded block and a separate product buffer only three fit: a second, third-full round of workgroups)
  double* Pt = lds;                                    // [5][KP][NC]
  double* Bs = Pt + 5 * KP * NC;                       // [N + 1][LDB]: the block, row N stays zero (rows of a partial tile)
  double* prod = Bs;                                   // [N][NC], after the last block has been multiplied
  for (int i = tid; i < 5 * KP * NC; i += NTH) {
    const int slot = i / (KP * NC), rem = i - slot * KP * NC, c = rem / NC, m = rem % NC;
    const int s2 = nbr[s * 5 + slot];
    double v = 0.0;
    if (s2 >= 0 && c < N && m < nmu) {
      const long g = (long)s2 * NM + c * nmu + m;
      v = first ? z[g] : z[g] + beta[m] * p_old[g];
      if (slot == 2) p_out[g] = v;
    }
    Pt[i] = v;
  }
  for (int i = tid; i < (N + 1) * LDB; i += NTH) Bs[i] = 0.0;
  const int col = ch * 16 + li;                        // this lane's parameter
  double thq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) thq[q] = col < nmu ? th.v[col * 8 + q] : 0.0;
  // block list of this subdomain: (slot, q) for every existing neighbour slot; register prefetch of the next block
  constexpr int PF = 4096 / NTH;                       // N * N <= 4096 = NTH threads x PF
  double pf[PF];
  auto load_block = [&](int slot, int q) {
    const double* B = B_sys + ((((long)q * S + s) * 5 + slot) * N) * N;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = tid + NTH * k;
      pf[k] = i < N * N ? B[i] : 0.0;
    }
  };
  int slots[5], ns = 0;
  for (int slot = 0; slot < 5; ++slot)
    if (nbr[s * 5 + slot] >= 0) slots[ns++] = slot;
  const int nblk = ns * Q;
  d4m acc = (d4m){0.0, 0.0, 0.0, 0.0};
  const bool active = rt * 16 < N;                     // this wave's row tile exists
  if (nblk > 0) load_block(slots[0], 0);
  for (int b = 0; b < nblk; ++b) {
    const int slot = slots[b / Q], q = b - (b / Q) * Q;
    __syncthreads();                                   // previous block's MFMAs are done reading Bs (and Pt is complete)
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = tid + NTH * k;
      if (i < N * N) Bs[(i / N) * LDB + i % N] = pf[k];
    }
    if (b + 1 < nblk) load_block(slots[(b + 1) / Q], (b + 1) - ((b + 1) / Q) * Q);
    __syncthreads();
    if (active) {
      const double* pslot = Pt + slot * KP * NC;
      double thv = thq[0];
#pragma unroll
      for (int qq = 1; qq < 8; ++qq)
        if (q == qq) thv = thq[qq];
      const int ra = rt * 16 + li < N ? rt * 16 + li : N;
      for (int kk = 0; kk < KP; kk += 4) {
        const double a = Bs[ra * LDB + kk + lk];
        const double bv = thv * pslot[(kk + lk) * NC + col];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
      }
    }
  }
  __syncthreads();                                     // all reads of Bs are done: it becomes the product buffer
  // D layout: lane holds rows lk + 4 r of its tile, column li
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rt * 16 + lk + 4 * r;
    if (active && row < N && col < nmu) {
      y[(long)s * NM + row * nmu + col] = acc[r];
      prod[row * NC + col] = acc[r] * Pt[2 * KP * NC + row * NC + col];
    }
  }
  __syncthreads();
  if (tid < nmu) {
    double sum = 0.0;
    for (int r = 0; r < N; ++r) sum += prod[r * NC + tid];
    partial[(long)tid * gridDim.x + s] = sum;          // [m][S]: the reduce reads contiguous runs
  }
}

// Round 4: the panel matvec for panels of 32 and 64 parameters.
// The 16-column kernel above keeps the direction panels of all five slots in LDS (40 KB at N = 40); at 32 columns that is 65 KB
// per workgroup, two workgroups per CU with ONE 12.8 KB block in flight each, and the kernel got slower per block than it gained
// per parameter (72 us for 32 columns against 40 us for 16: 1.8 TB/s).  Here a wave owns (row tile rt, column tile ch); the
// direction panel of ONE slot at a time is staged in LDS ([N][NC], two copies: the next slot's rows are requested with contiguous
// loads -- z + beta p_old, N nmu consecutive doubles -- at the first block of the current slot and parked after its MFMAs), every
// wave takes its B operand P_slot[k][its 16 columns] into KP / 4 registers once per slot, and two copies of the current block
// give ONE barrier per block.  69 KB of LDS at N = 40 and 64 columns: two workgroups of 16 waves per CU.  Every projected block
// then serves 64 solves per read instead of 16.  theta comes from device memory (a by-value table of 64 x 8 doubles would fill
// the kernel arguments).  (First attempt, dropped: the B operands straight from global memory into registers, one slot ahead --
// 2 x 3 x KP / 4 registers per lane; 217 VGPRs, or spills under the 128 of a 1 024-thread workgroup: 382 us per launch.)
// CT column tiles per wave.  Measured at 64 columns: CT = 2 (512 threads, the A operand of a k-step read from LDS once for two
// MFMAs, two workgroups of 8 waves per CU instead of one of 16) needs 150 VGPRs, spills under the 128 that two workgroups allow,
// and takes 131 us against 102 us for CT = 1: every instantiation the launcher takes has CT = 1.
template <int NC, int KSC, int CT>      // KSC: k-steps compiled in (4, 8, 10, 12, 16 for N <= 16, 32, 40, 48, 64)
// (Forcing two workgroups per CU -- 80 VGPRs at 12 waves per workgroup, 52 bytes of scratch -- was measured: 115 us against 102 us.
// The launch moves ~380 MB -- 131 MB of blocks, 210 MB of direction rows (z and p_old of five slots), 42 MB of results -- in 102 us,
// 3.7 TB/s: as fast as any streaming kernel of this library gets from the Infinity Cache / HBM.)
__global__ __launch_bounds__(64 * (NC / 16 / CT) * ((KSC + 3) / 4)) void k_bcg_matvec_panel(int S, const int* __restrict__ nbr, int Q, int N, int nmu,
                                                              const double* __restrict__ theta,      // [nmu][8] device
                                                              const double* __restrict__ B_sys, const double* __restrict__ z,
                                                              const double* __restrict__ p_old, const double* __restrict__ beta,
                                                              int first, double* __restrict__ p_out, double* __restrict__ y,
                                                              double* __restrict__ partial) {
  extern __shared__ double lds[];
  // one wave per (row tile that exists for this KSC, group of CT column tiles); the column group is the FAST index, so that the
  // waves of a row tile -- and with them the idle lanes of a partial last tile -- spread over the four SIMDs (wave w runs on SIMD
  // w % 4: with the row tile fast and N = 40 every wave of the empty fourth tile sat on SIMD 3 and the MFMAs on the other three)
  constexpr int NCG = NC / 16 / CT, RTN = (KSC + 3) / 4, NTH = 64 * NCG * RTN;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), rt = wave / NCG, col = (wave % NCG) * 16 * CT + li;      // first of CT columns, 16 apart
  const int NM = N * nmu;
  const int LDB = N + ((4 - N % 8) + 8) % 8;           // row stride == 4 (mod 8) doubles: conflict-free A-operand reads
  const int BSZ = (N + 1) * LDB + 16;                  // (+ 16: the k-steps beyond N of the last row read zeros, not the other copy)
  const int PSZ = 4 * KSC * NC;                        // one direction panel [4 KSC][NC]: rows >= N and columns >= nmu stay zero
  double* Bs = lds;                                    // [2][N + 1][LDB]: row N and the columns >= N stay zero
  double* Ps = Bs + 2 * BSZ;                           // [2][4 KSC][NC]
  double* bl = Ps + 2 * PSZ;                           // [NC] beta | [4][NC] theta_q (Q <= 4 with the wide panels: checked by the launcher)
  double* tl = bl + NC;                                // | one dump slot (entries beyond the block / the panel)
  const int DUMP = 2 * BSZ + 2 * PSZ + 5 * NC;
  for (int i = tid; i < 2 * BSZ + 2 * PSZ; i += NTH) lds[i] = 0.0;
  if (tid < NC) bl[tid] = (tid < nmu && !first) ? beta[tid] : 0.0;
  // (theta through LDS, not registers loaded in front of the loop: hipcc's wait-count state at the loop header keeps a load that was
  // never waited for on the entry path "pending" in every iteration, i.e. a vmcnt(0) in front of its first use -- behind the prefetch)
  for (int i = tid; i < 4 * NC; i += NTH) tl[i] = (i % NC < nmu) ? theta[(i % NC) * 8 + i / NC] : 0.0;
  constexpr int PF = (16 * KSC * KSC + NTH - 1) / NTH; // N * N <= (4 KSC)^2 <= NTH threads x PF
  constexpr int PP = (4 * KSC * NC + NTH - 1) / NTH;   // N nmu <= 4 KSC NC <= NTH threads x PP
  double pf[PF];                                       // (two sets, blocks b + 1 and b + 2 in flight, were measured: no faster -- the loads are not what a step waits for)
  // LDS offset of this thread's k-th block entry (divisions once); entries beyond the block go to a dump slot UNCONDITIONALLY: a
  // store under a condition leaves a path without the wait for its load, and hipcc then guards the registers' reuse in the next
  // step with waits that also cover the loads just issued (seen in the ISA: vmcnt(0) behind every prefetch)
  int boff[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int i = tid + NTH * k;
    boff[k] = i < N * N ? (i / N) * LDB + i % N : -1;
  }
  auto load_block = [&](int slot, int q) {             // unconditional loads (clamped index): a conditional one compiles to a branch with a wait
    const double* B = B_sys + ((((long)q * S + s) * 5 + slot) * N) * N;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = tid + NTH * k;
      pf[k] = B[i < N * N ? i : N * N - 1];
    }
  };
  // rows of the direction panel of a slot: element e = c nmu + m of the neighbour's [N][nmu] array, NM consecutive doubles
  double zz[PP], pp[PP];
  int poff[PP], pm[PP];                                // LDS offset c NC + m and column m of this thread's k-th panel entry (-1 beyond)
#pragma unroll
  for (int k = 0; k < PP; ++k) {
    const int e = tid + NTH * k, c = e / nmu;
    pm[k] = e < NM ? e - c * nmu : 0;
    poff[k] = e < NM ? c * NC + pm[k] : -1;
  }
  // neighbour table row and the list of existing slots: wave-uniform scalars read ONCE (a vector load of nbr inside the block loop
  // would wait, in order, for every block load in flight), the list packed four bits per entry (no dynamically indexed array)
  const int nb0 = __builtin_amdgcn_readfirstlane(nbr[s * 5]), nb1 = __builtin_amdgcn_readfirstlane(nbr[s * 5 + 1]),
            nb3 = __builtin_amdgcn_readfirstlane(nbr[s * 5 + 3]), nb4 = __builtin_amdgcn_readfirstlane(nbr[s * 5 + 4]);
  auto nb_of = [&](int slot) { return slot == 0 ? nb0 : slot == 1 ? nb1 : slot == 2 ? s : slot == 3 ? nb3 : nb4; };
  int ns = 0, packed = 0;
#pragma unroll
  for (int slot = 0; slot < 5; ++slot)
    if (nb_of(slot) >= 0) packed |= slot << (4 * ns), ++ns;
  auto slot_at = [&](int i) { return (packed >> (4 * i)) & 7; };
  auto request_panel = [&](int slot) {
    const long base = (long)nb_of(slot) * NM;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      const int e = tid + NTH * k;
      zz[k] = (z + base)[e < NM ? e : NM - 1];
      pp[k] = (p_old + base)[e < NM ? e : NM - 1];     // (first: an allocated but unwritten array; the value is dropped)
    }
  };
  auto park_panel = [&](double* dst) {
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      const double v = first ? zz[k] : zz[k] + bl[pm[k]] * pp[k];
      (poff[k] >= 0 ? dst + poff[k] : lds + DUMP)[0] = v;
    }
  };
  const int nblk = ns * Q;
  d4m acc[CT];
  double pv[CT][KSC];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = (d4m){0.0, 0.0, 0.0, 0.0};
  const bool active = rt * 16 < N;                     // this wave's row tile exists
  __syncthreads();                                     // the zeros and beta are in place
  if (nblk > 0) {
    load_block(slot_at(0), 0);
    request_panel(slot_at(0));
    park_panel(Ps);
  }
  int si = 0, q = 0, slot = nblk > 0 ? slot_at(0) : 2;
  for (int b = 0; b < nblk; ++b) {
    // (the previous reader of this copy was block b - 2: every wave finished it before it arrived at the barrier of block b - 1)
#pragma unroll
    for (int k = 0; k < PF; ++k) lds[boff[k] >= 0 ? (b & 1) * BSZ + boff[k] : DUMP] = pf[k];
    const double* Bc = Bs + (b & 1) * BSZ;
    const int qn = q + 1 < Q ? q + 1 : 0, sin = q + 1 < Q ? si : si + 1;
    const int slotn = sin < ns ? slot_at(sin) : slot;
    const bool newslot = q == 0, stage = newslot && si + 1 < ns;
    const int slot_next = stage ? slot_at(si + 1) : slot;
    // the next slot's panel rows are requested IN FRONT of the block prefetch: hipcc guards the reuse of the zz / pp registers with
    // waits that cover every load issued before them (seen in the ISA: vmcnt(0) right behind the block prefetch, which then
    // overlapped with nothing); in this order those waits see no block load of this step
    if (stage) request_panel(slot_next);               // lands during the MFMAs below
    // UNCONDITIONAL (the last step fetches the last block again): with the prefetch under a condition hipcc counts the waits of
    // the panel rows below for the path without it, i.e. they also wait for the block loads just issued
    load_block(slotn, sin < ns ? qn : q);
    lds_only_barrier();                                // this block and (first block of a slot) its direction panel are complete
    if (newslot) {
      const double* Pc = Ps + (si & 1) * PSZ + lk * NC + col;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) pv[t][ks] = Pc[4 * ks * NC + 16 * t];
    }
    if (active) {
      double thv[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) thv[t] = tl[q * NC + col + 16 * t];
      const double* arow = Bc + (rt * 16 + li < N ? rt * 16 + li : N) * LDB + lk;
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        const double a = arow[4 * ks];
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, thv[t] * pv[t][ks], acc[t], 0, 0, 0);
      }
    }
    // (the other panel copy was last read at the first block of slot si - 1, in front of the barrier every wave has passed since)
    if (stage) park_panel(Ps + ((si + 1) & 1) * PSZ);
    q = qn, si = sin, slot = slotn;
  }
  __syncthreads();                                     // all reads of Bs are done: it becomes the reduction buffer [4][NC]
  // D layout: lane holds rows rt 16 + lk + 4 r, column col.  p . A p per column: rows of this lane, the four lk groups of the wave
  // (shuffles), the four row-tile waves (LDS) -- a fixed order
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int ct = col + 16 * t;
    const long base = (long)s * NM;
    double dot = 0.0, zo[4], po[4];
    const int cc = ct < nmu ? ct : nmu - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                      // the own direction at this lane's D elements, as the panel had it: the same expression, the same bits
      const int row = rt * 16 + lk + 4 * r, rc = row < N ? row : N - 1;
      zo[r] = (z + base)[rc * nmu + cc];
      po[r] = (p_old + base)[rc * nmu + cc];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + lk + 4 * r;
      if (active && ct < nmu && row < N) {
        const double pw = first ? zo[r] : zo[r] + bl[ct] * po[r];
        (y + base)[row * nmu + ct] = acc[t][r];
        (p_out + base)[row * nmu + ct] = pw;             // the new direction of the own slot
        dot += acc[t][r] * pw;
      }
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (lk == 0) lds[rt * NC + ct] = dot;
  }
  __syncthreads();
  if (tid < nmu) {
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < RTN; ++r) sum += lds[r * NC + tid];
    partial[(long)tid * gridDim.x + s] = sum;          // [m][S]
  }
}

// out[m] = sum_s partial[s][m]; mode 1: pAp -> alpha = rz / pAp; mode 2: rz_new -> beta = rz_new / rz, rz = rz_new;
// mode 0: rz (initial).  scal layout: rz [BMAX], alpha [BMAX], beta [BMAX], rr [BMAX]
__global__ __launch_bounds__(1024) void k_bcg_reduce(int S, int nmu, const double* __restrict__ partial,
                                                     const double* __restrict__ partial2, double* __restrict__ scal, int mode) {
  const int lane = threadIdx.x & 63, nw = (blockDim.x >> 6) * gridDim.x;      // (grid: one workgroup of 16 waves per 16 parameters)
  for (int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); m < nmu; m += nw) {       // wave m sums parameter m: lane-strided, then a fixed shuffle tree
    double a = 0.0, b = 0.0;
    // all loads of a 1024-entry chunk are issued before the first add (an `a += partial[i]` loop waits one L2 round
    // trip per entry: this single-workgroup kernel took 7.6 us, 19 % of an iteration of the batched solve)
    for (int base = 0; base < S; base += 1024) {
      double va[16], vb[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = base + lane + 64 * k;
        va[k] = i < S ? partial[(long)m * S + i] : 0.0;
        vb[k] = (partial2 && i < S) ? partial2[(long)m * S + i] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        a += va[k];
        b += vb[k];
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off, 64);
      b += __shfl_down(b, off, 64);
    }
    if (lane == 0) {
      if (mode == 0) {
        scal[m] = a;
      } else if (mode == 1) {
        scal[BMAX + m] = a != 0.0 ? scal[m] / a : 0.0;
      } else {
        scal[2 * BMAX + m] = scal[m] != 0.0 ? a / scal[m] : 0.0;
        scal[m] = a;
      }
      if (partial2) scal[3 * BMAX + m] = b;
    }
  }
}

// x += alpha p; r -= alpha y; z = Dinv r; partial[s][m] = r.z, partial2[s][m] = r.r
__global__ __launch_bounds__(256) void k_bcg_update(int N, int nmu, const double* __restrict__ Dinv, const double* __restrict__ scal,
                                                    int first, double* __restrict__ x, double* __restrict__ r,
                                                    const double* __restrict__ p, const double* __restrict__ y,
                                                    double* __restrict__ z, double* __restrict__ partial,
                                                    double* __restrict__ partial2) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, NM = N * nmu;
  double* rs = lds;          // [N][nmu]
  double* prod = rs + NM;    // [2][N][nmu]
  for (int i = tid; i < NM; i += 256) {
    const long g = (long)s * NM + i;
    const double alpha = first ? 0.0 : scal[BMAX + i % nmu];
    if (!first) x[g] += alpha * p[g];
    const double rv = first ? r[g] : r[g] - alpha * y[g];
    r[g] = rv;
    rs[i] = rv;
  }
  __syncthreads();
  for (int i = tid; i < NM; i += 256) {
    const int row = i / nmu, m = i - row * nmu;
    const double* D = Dinv + ((long)s * N + row) * N;
    double acc = 0.0;
    for (int c = 0; c < N; ++c) acc += D[c] * rs[c * nmu + m];
    z[(long)s * NM + i] = acc;
    prod[i] = acc * rs[i];
    prod[NM + i] = rs[i] * rs[i];
  }
  __syncthreads();
  if (tid < nmu) {
    double a = 0.0, b = 0.0;
    for (int row = 0; row < N; ++row) {
      a += prod[row * nmu + tid];
      b += prod[NM + row * nmu + tid];
    }
    partial[(long)tid * gridDim.x + s] = a;            // [m][S]
    partial2[(long)tid * gridDim.x + s] = b;
  }
}

// The same update with z = Dinv r as a 16-column MFMA product (batches of at most 16 parameters): wave w owns row tile w,
// the inverse diagonal block is the A operand straight from global memory, r (padded to 16 columns) the B operand in LDS.
template <int NC>
__global__ __launch_bounds__(16 * NC) void k_bcg_update_mfma(int N, int nmu, const double* __restrict__ Dinv, const double* __restrict__ scal,
                                                         int first, double* __restrict__ x, double* __restrict__ r,
                                                         const double* __restrict__ p, const double* __restrict__ y,
                                                         double* __restrict__ z, double* __restrict__ partial,
                                                         double* __restrict__ partial2) {
  extern __shared__ double lds[];
  constexpr int NTH = 16 * NC;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4, wave = tid >> 6;
  const int rt = wave & 3, col = (wave >> 2) * 16 + li;      // row tile, this lane's parameter
  const int NM = N * nmu, KP = (N + 3) & ~3;
  double* rs = lds;              // [KP][NC]  (columns >= nmu and rows >= N zero)
  double* prod = rs + KP * NC;   // [2][N][NC]
  for (int i = tid; i < KP * NC; i += NTH) rs[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < NM; i += NTH) {
    const long g = (long)s * NM + i;
    const int row = i / nmu, m = i - row * nmu;
    const double alpha = first ? 0.0 : scal[BMAX + m];
    if (!first) x[g] += alpha * p[g];
    const double rv = first ? r[g] : r[g] - alpha * y[g];
    r[g] = rv;
    rs[row * NC + m] = rv;
  }
  __syncthreads();
  if (rt * 16 < N) {
    const int ra = rt * 16 + li < N ? rt * 16 + li : N - 1;
    const double* D = Dinv + ((long)s * N + ra) * N;
    d4m T = (d4m){0.0, 0.0, 0.0, 0.0};
    // (requesting all k-steps of the row strip in front of the first MFMA was measured: 41 vs 35 us at 64 columns -- the kernel moves
    // 160 MB of vectors per launch and is bound by that, not by this chain)
    for (int kk = 0; kk < KP; kk += 4) {
      const bool in = kk + lk < N;
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? D[kk + lk] : 0.0, rs[(kk + lk) * NC + col], T, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = rt * 16 + lk + 4 * rr;
      if (row < N) {
        const double rv = rs[row * NC + col];
        if (col < nmu) z[(long)s * NM + row * nmu + col] = T[rr];
        prod[row * NC + col] = T[rr] * rv;
        prod[(N + row) * NC + col] = rv * rv;
      }
    }
  }
  __syncthreads();
  if (tid < nmu) {
    double a = 0.0, b = 0.0;
    for (int row = 0; row < N; ++row) {
      a += prod[row * NC + tid];
      b += prod[(N + row) * NC + tid];
    }
    partial[(long)tid * gridDim.x + s] = a;
    partial2[(long)tid * gridDim.x + s] = b;
  }
}

__global__ __launch_bounds__(256) void k_bcg_init(long total, int nmu, const double* __restrict__ rhs, double* __restrict__ x,
                                                  double* __restrict__ r) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    x[i] = 0.0;
    r[i] = rhs[i / nmu];
  }
}

}  // namespace

// doubles of work per group of <= 16 parameters of the batched reduced solve: u (groups of a multi-group call solve into their
// own [S][N][nm] array), r, z, p0, p1, y, two partial arrays, scalars
// (sized for the widest group, 64 parameters)
static long reduced_batch_group_size(long S, int N, int W) { return 6 * S * N * W + 2 * S * W + 4 * BMAX + 8 * BMAX; }
// parameters per group (= columns of the MFMA panel): calls of more than 16 parameters run panels of 32 -- every projected block
// streamed by the panel matvec then serves 32 solves -- unless the VALU cross-check form is asked for (its kernels hold <= 16)
static int reduced_batch_group_width(const lrbms_ctx* ctx, int nmu) {
  if (ctx->opt_solve_valu != 0 || nmu <= 16) return 16;
  return nmu <= 32 ? 32 : 64;
}

int64_t reduced_solve_batch_work_size(lrbms_ctx* ctx, int N, int nmu) {
  const long S = ctx->S;
  // room for either grouping of the call: panels of 16 (LRBMS_OPT_SOLVE_VALU, nmu <= 16) or the width reduced_batch_group_width picks
  const int W = nmu <= 16 ? 16 : nmu <= 32 ? 32 : 64;
  const long a = (long)((nmu + 15) / 16) * reduced_batch_group_size(S, N, 16), b = (long)((nmu + W - 1) / W) * reduced_batch_group_size(S, N, W);
  return S * 5 * N * N + S * N * N + (a > b ? a : b) + 16;
}

namespace {
__global__ __launch_bounds__(256) void k_bcg_scatter(long rows, int nm, int nmu, const double* __restrict__ ug, double* __restrict__ u) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * nm; i += (long)gridDim.x * blockDim.x)
    u[(i / nm) * nmu + i % nm] = ug[i];
}
}  // namespace

// nmu <= 64 parameters per call in groups of <= 16, each group an independent CG on its own stream (the caller's and the
// library's three side streams), launches interleaved iteration by iteration, residuals looked at together: a group's kernels
// are S latency-bound workgroups, so the groups share the chip at little cost to each other (config 3: 6 100 mu-solves/s with
// one group at a time, 8 700 with three in flight).  One preconditioner per call, read-only while the groups run: the prebuilt
// one (lrbms_reduced_precond_use) or the inverse diagonal blocks + coarse level at the mean theta of the call.
int launch_reduced_solve_batch(lrbms_ctx* ctx, int Q, int N, int nmu, const double* theta, const double* B_sys,
                               const double* rhs_red, double* work, double* u, double rtol, int max_iter, double* info,
                               hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_solve_batch needs all subdomains on one rank");
  if (N > 64 || nmu < 1 || nmu > 64)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_solve_batch: need N <= 64, nmu <= 64");
  const int S = ctx->S;
  const int GW = reduced_batch_group_width(ctx, nmu);
  const int ng = (nmu + GW - 1) / GW;
  const long per_q = (long)S * 5 * N * N;
  QVec mean;
  for (int q = 0; q < 8; ++q) mean.v[q] = 0.0;
  for (int m = 0; m < nmu; ++m)
    for (int q = 0; q < Q; ++q) mean.v[q] += theta[m * Q + q] / nmu;
  double* Amu = work;                                   // blocks at the mean theta of the call (only the diagonal is inverted)
  double* Dinv = Amu + per_q;
  double* gwork = Dinv + (long)S * N * N;
  const long gsize = reduced_batch_group_size(S, N, GW);
  struct Group {
    int nm, m0, it;
    ThetaBatch th;
    double thd[BMAX * 8];                               // theta [m][8] of the group, copied to the device for the wide panels
    double *ug, *r, *z, *pin, *pout, *y, *partial, *partial2, *scal, *theta_dev;
    hipStream_t st;
    bool done;
    double rel, rr0[BMAX];
  } g[4];
  for (int k = 0; k < ng; ++k) {
    Group& G = g[k];
    G.m0 = GW * k;
    G.nm = nmu - G.m0 < GW ? nmu - G.m0 : GW;
    G.st = k == 0 ? st : ctx->aux[k - 1];
    G.it = 0;
    G.done = false;
    G.rel = 0.0;
    const long vec = (long)S * N * G.nm;
    double* w = gwork + k * gsize;
    G.ug = ng == 1 ? u : w;                             // a single group solves straight into the caller's array
    G.r = w + (long)S * N * GW;
    G.z = G.r + vec;
    G.pin = G.z + vec;
    G.pout = G.pin + vec;
    G.y = G.pout + vec;
    G.partial = w + 6L * S * N * GW;
    G.partial2 = G.partial + (long)S * GW;
    G.scal = G.partial2 + (long)S * GW;                 // rz, alpha, beta, rr (BMAX each)
    G.theta_dev = G.scal + 4 * BMAX;                    // [BMAX][8]
    for (int m = 0; m < TBMAX; ++m)
      for (int q = 0; q < 8; ++q) G.th.v[m * 8 + q] = (m < G.nm && q < Q) ? theta[(G.m0 + m) * Q + q] : 0.0;
    for (int m = 0; m < BMAX; ++m)
      for (int q = 0; q < 8; ++q) G.thd[m * 8 + q] = (m < G.nm && q < Q) ? theta[(G.m0 + m) * Q + q] : 0.0;
  }
  // ---- the preconditioner of the call, on the caller's stream, before the groups fork
  const double* A0inv = nullptr;
  const double* pcD = nullptr;
  if (int rc = user_precond(ctx, N, &pcD, &A0inv, st)) return rc;
  if (pcD) {
    Dinv = const_cast<double*>(pcD);                     // prebuilt (lrbms_reduced_precond_build / _use): read only below
  } else {
    hipLaunchKernelGGL(k_assemble_mu, dim3((unsigned)((per_q + 255) / 256 > 8192 ? 8192 : (per_q + 255) / 256)), dim3(256), 0, st,
                       per_q, Q, mean, B_sys, Amu);
    if (int rc = launch_block_inverse(ctx, (int)S, N, Amu, Dinv, 5, 2, st)) return rc;
    LRBMS_LAUNCH_CHECK(ctx);
    if (int rc = coarse_setup(ctx, N, Amu, &A0inv, st)) return rc;
  }
  // matrix-core form of the panel matvec and of the preconditioner (LRBMS_OPT_SOLVE_VALU forces the VALU form)
  const int kp = (N + 3) & ~3, ldb = N + ((4 - N % 8) + 8) % 8;
  const size_t bs_lds = (size_t)(N + 1) * ldb > (size_t)N * 16 ? (size_t)(N + 1) * ldb : (size_t)N * 16;   // block, later the products
  // panels of 16: the direction panels of the five slots in LDS (k_bcg_matvec_mfma); wider panels: in registers, two copies of
  // the current block in LDS (k_bcg_matvec_panel)
  const int ksc_n = N <= 16 ? 4 : N <= 32 ? 8 : N <= 40 ? 10 : N <= 48 ? 12 : 16;      // k-steps compiled into k_bcg_matvec_panel
  const size_t lds_panel = sizeof(double) * (2 * ((size_t)(N + 1) * ldb + 16) + 2 * (size_t)4 * ksc_n * GW + 5 * GW + 1);
  if (GW > 16 && Q > 4) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_solve_batch: more than 16 parameters per call need Q <= 4");
  const size_t lds_mfma = GW == 16 ? sizeof(double) * ((size_t)5 * kp * 16 + bs_lds) : lds_panel;
  const bool use_mfma = ctx->opt_solve_valu == 0;
  if (use_mfma && lds_mfma > 160 * 1024) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_solve_batch: panels exceed the LDS");
#define LRBMS_PANEL_DISPATCH(X)                                                                                    \
  do {                                                                                                             \
    if (GW == 64) {                                                                                                \
      if (ksc_n == 4) X(64, 4, 1); else if (ksc_n == 8) X(64, 8, 1); else if (ksc_n == 10) X(64, 10, 1); else if (ksc_n == 12) X(64, 12, 1); else X(64, 16, 1); \
    } else {                                                                                                       \
      if (ksc_n == 4) X(32, 4, 1); else if (ksc_n == 8) X(32, 8, 1); else if (ksc_n == 10) X(32, 10, 1); else if (ksc_n == 12) X(32, 12, 1); else X(32, 16, 1); \
    }                                                                                                              \
  } while (0)
  if (use_mfma && lds_mfma > 64 * 1024) {
    if (GW == 16) {
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bcg_matvec_mfma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
    } else {
#define LRBMS_PANEL_ATTR(NCV, KSV, CTV) \
  LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bcg_matvec_panel<NCV, KSV, CTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma))
      LRBMS_PANEL_DISPATCH(LRBMS_PANEL_ATTR);
#undef LRBMS_PANEL_ATTR
    }
  }
  const size_t lds_mv = sizeof(double) * (5 * (size_t)N * 16 + (size_t)N * N + 256);
  if (!use_mfma && lds_mv > 64 * 1024)
    LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bcg_matvec<BCG_KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mv));
  const size_t lds_upd_mfma = sizeof(double) * ((size_t)kp * GW + (size_t)2 * N * GW);
  if (use_mfma && lds_upd_mfma > 64 * 1024) {
    if (GW == 64)
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bcg_update_mfma<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_upd_mfma));
    else
      LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_bcg_update_mfma<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_upd_mfma));
  }
  if (ng > 1) {
    LRBMS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, st));
    for (int k = 1; k < ng; ++k) LRBMS_HIP_CHECK(ctx, hipStreamWaitEvent(g[k].st, ctx->ev_fork, 0));
  }
  auto join = [&]() -> int {                             // the side streams joined into the caller's
    for (int k = 1; k < ng; ++k) {
      LRBMS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[k - 1], g[k].st));
      LRBMS_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_join[k - 1], 0));
    }
    return LRBMS_OK;
  };
  // every EARLY return below (a failed HIP call between the fork and the join) leaves through this guard: the group streams are
  // drained -- their asynchronous copies target `host` in this stack frame -- and joined into the caller's stream, which the
  // fused pass and the next call share with them
  struct ExitGuard {
    lrbms_ctx* ctx;
    hipStream_t st;
    hipStream_t gs[4];
    int ng;
    bool armed;
    ~ExitGuard() {
      if (!armed) return;
      for (int k = 0; k < ng; ++k) (void)hipStreamSynchronize(gs[k]);
      for (int k = 1; k < ng; ++k) {
        (void)hipEventRecord(ctx->ev_join[k - 1], gs[k]);
        (void)hipStreamWaitEvent(st, ctx->ev_join[k - 1], 0);
      }
    }
  } guard{ctx, st, {g[0].st, ng > 1 ? g[1].st : st, ng > 2 ? g[2].st : st, ng > 3 ? g[3].st : st}, ng, true};
  auto update = [&](Group& G, int first) {
    const size_t lds_upd = sizeof(double) * 3 * (size_t)N * G.nm;
    if (use_mfma && GW == 64)
      hipLaunchKernelGGL(k_bcg_update_mfma<64>, dim3(S), dim3(1024), lds_upd_mfma, G.st, N, G.nm, Dinv, G.scal, first, G.ug, G.r,
                         first ? G.pin : G.pout, G.y, G.z, G.partial, G.partial2);
    else if (use_mfma && GW == 32)
      hipLaunchKernelGGL(k_bcg_update_mfma<32>, dim3(S), dim3(512), lds_upd_mfma, G.st, N, G.nm, Dinv, G.scal, first, G.ug, G.r,
                         first ? G.pin : G.pout, G.y, G.z, G.partial, G.partial2);
    else if (use_mfma)
      hipLaunchKernelGGL(k_bcg_update_mfma<16>, dim3(S), dim3(256), lds_upd_mfma, G.st, N, G.nm, Dinv, G.scal, first, G.ug, G.r,
                         first ? G.pin : G.pout, G.y, G.z, G.partial, G.partial2);
    else
      hipLaunchKernelGGL(k_bcg_update, dim3(S), dim3(256), lds_upd, G.st, N, G.nm, Dinv, G.scal, first, G.ug, G.r,
                         first ? G.pin : G.pout, G.y, G.z, G.partial, G.partial2);
    if (A0inv) hipLaunchKernelGGL(k_coarse_apply, dim3((S + 15) / 16, (G.nm + 15) / 16), dim3(1024), 0, G.st, S, N, G.nm, A0inv, G.r, G.z, G.partial);
    hipLaunchKernelGGL(k_bcg_reduce, dim3((G.nm + 15) / 16), dim3(1024), 0, G.st, S, G.nm, G.partial, G.partial2, G.scal, first ? 0 : 2);
  };
  double host[4][4 * BMAX];
  for (int k = 0; k < ng; ++k) {
    Group& G = g[k];
    const long vec = (long)S * N * G.nm;
    if (GW > 16) LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(G.theta_dev, G.thd, sizeof(double) * 8 * BMAX, hipMemcpyHostToDevice, G.st));
    hipLaunchKernelGGL(k_bcg_init, dim3((unsigned)((vec + 255) / 256 > 4096 ? 4096 : (vec + 255) / 256)), dim3(256), 0, G.st, vec, G.nm,
                       rhs_red, G.ug, G.r);
    update(G, 1);
    LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host[k], G.scal, sizeof(double) * 4 * BMAX, hipMemcpyDeviceToHost, G.st));
  }
  LRBMS_LAUNCH_CHECK(ctx);
  bool all_done = true;
  for (int k = 0; k < ng; ++k) {
    Group& G = g[k];
    LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(G.st));
    bool all_zero = true;
    for (int m = 0; m < G.nm; ++m) {
      G.rr0[m] = host[k][3 * BMAX + m];
      all_zero &= G.rr0[m] == 0.0;
    }
    G.done = all_zero;                                   // zero right-hand side: x = 0 is the solution
    all_done &= G.done;
  }
  int block = 10;                                        // iterations until the next look at the residuals (see red_cg_run)
  int rc = LRBMS_OK;
  while (!all_done) {
    for (int c = 0; c < block; ++c)
      for (int k = 0; k < ng; ++k) {
        Group& G = g[k];
        if (G.done || G.it >= max_iter) continue;
        const int first = G.it == 0 ? 1 : 0;
        const long NM = (long)N * G.nm;
        if (use_mfma && GW > 16) {
#define LRBMS_PANEL(NCV, KSV, CTV)                                                                                                  \
  hipLaunchKernelGGL((k_bcg_matvec_panel<NCV, KSV, CTV>), dim3(S), dim3(64 * (NCV / 16 / CTV) * ((KSV + 3) / 4)), lds_mfma, G.st, S, ctx->nbr, Q, N, G.nm,   \
                     G.theta_dev, B_sys, G.z, G.pin, G.scal + 2 * BMAX, first, G.pout, G.y, G.partial)
          LRBMS_PANEL_DISPATCH(LRBMS_PANEL);
#undef LRBMS_PANEL
        }
        else if (use_mfma)
          hipLaunchKernelGGL(k_bcg_matvec_mfma<16>, dim3(S), dim3(256), lds_mfma, G.st, S, ctx->nbr, Q, N, G.nm, G.th, B_sys, G.z, G.pin,
                             G.scal + 2 * BMAX, first, G.pout, G.y, G.partial);
        else if (NM <= 768)   // three outputs per thread: fewer registers, measurably faster for the usual batch of 16
          hipLaunchKernelGGL(k_bcg_matvec<3>, dim3(S), dim3(256), lds_mv, G.st, S, ctx->nbr, Q, N, G.nm, G.th, B_sys, G.z, G.pin,
                             G.scal + 2 * BMAX, first, G.pout, G.y, G.partial);
        else
          hipLaunchKernelGGL(k_bcg_matvec<BCG_KMAX>, dim3(S), dim3(256), lds_mv, G.st, S, ctx->nbr, Q, N, G.nm, G.th, B_sys, G.z, G.pin,
                             G.scal + 2 * BMAX, first, G.pout, G.y, G.partial);
        hipLaunchKernelGGL(k_bcg_reduce, dim3((G.nm + 15) / 16), dim3(1024), 0, G.st, S, G.nm, G.partial, (const double*)nullptr, G.scal, 1);
        update(G, 0);
        double* tmp = G.pin; G.pin = G.pout; G.pout = tmp;
        ++G.it;
      }
    LRBMS_LAUNCH_CHECK(ctx);
    for (int k = 0; k < ng; ++k)
      if (!g[k].done)
        LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host[k], g[k].scal, sizeof(double) * 4 * BMAX, hipMemcpyDeviceToHost, g[k].st));
    all_done = true;
    double need_max = 0.0;
    for (int k = 0; k < ng; ++k) {
      Group& G = g[k];
      if (G.done) continue;
      LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(G.st));
      G.rel = 0.0;
      for (int m = 0; m < G.nm; ++m) {
        const double rm = G.rr0[m] > 0.0 ? sqrt(host[k][3 * BMAX + m] / G.rr0[m]) : 0.0;
        if (!(rm == rm)) rc = LRBMS_E_NOT_CONVERGED;
        G.rel = rm > G.rel ? rm : G.rel;
      }
      if (G.rel <= rtol || G.it >= max_iter || rc != LRBMS_OK) {
        G.done = true;
        continue;
      }
      all_done = false;
      // CG converges superlinearly, so the average rate so far overestimates what is left: aim a little short (a further
      // look costs one host round trip, a wasted iteration three to four kernels)
      const double rate = log(G.rel) / G.it;
      double need = 10.0;
      if (rate < 0.0) need = 0.8 * (log(rtol) - log(G.rel)) / rate;
      need_max = need > need_max ? need : need_max;
    }
    block = need_max < 2.0 ? 2 : need_max > 40.0 ? 40 : (int)need_max;
  }
  if (ng > 1)
    for (int k = 0; k < ng; ++k) {
      const long rows = (long)S * N;
      hipLaunchKernelGGL(k_bcg_scatter, dim3((unsigned)((rows * g[k].nm + 255) / 256 > 4096 ? 4096 : (rows * g[k].nm + 255) / 256)),
                         dim3(256), 0, g[k].st, rows, g[k].nm, nmu, g[k].ug, u + g[k].m0);
    }
  LRBMS_LAUNCH_CHECK(ctx);
  if (int jrc = join()) return jrc;
  guard.armed = false;
  int it = 0;
  double rel = 0.0;
  for (int k = 0; k < ng; ++k) {
    it = g[k].it > it ? g[k].it : it;
    rel = g[k].rel > rel ? g[k].rel : rel;
  }
  if (info) { info[0] = it; info[1] = rel; }
  if (rc != LRBMS_OK) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve_batch: NaN residual (system not SPD?)");
  if (rel > rtol) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "reduced_solve_batch: CG did not reach rtol");
  return LRBMS_OK;
}

// =========================================================================================================
// E1, throughput form: the local estimator terms of nmu <= 16 reduced solutions in one pass over the projected
// operators.  Every operator entry is loaded once and used for all parameters (2 nmu flops per 8 bytes), so the
// batch costs about one single-parameter estimate in HBM traffic.  Lanes own columns (their coefficient panel lives
// in registers), waves stride over rows (row coefficients are LDS broadcasts); fixed-order wave + block reductions.
namespace {

constexpr int EB = 16;   // max parameters per estimate batch

// acc[m] += w[m] * sum_{r, c} x[r][m] G[r][c] y[c][m]   (x, y in LDS as [row][nmu]; w optional per-parameter weight)
__device__ inline void quad_batch(const double* __restrict__ G, int ld, int rows, int cols, const double* x, const double* y,
                                  int nmu, const double* w, double factor, double (&acc)[EB]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int c = lane; c < cols; c += 64) {
    double yc[EB];
#pragma unroll
    for (int m = 0; m < EB; ++m) yc[m] = m < nmu ? factor * (w ? w[m] : 1.0) * y[c * nmu + m] : 0.0;
    for (int r = wave; r < rows; r += nw) {
      const double g = G[(long)r * ld + c];
      const double* xr = x + r * nmu;
#pragma unroll
      for (int m = 0; m < EB; ++m)
        if (m < nmu) acc[m] += (xr[m] * yc[m]) * g;
    }
  }
}

__global__ __launch_bounds__(256) void k_reduced_estimate_batch(int S, const int* __restrict__ nbr, int Q, int N, int nmu,
                                                                ThetaBatch th, const double* __restrict__ u,
                                                                const double* __restrict__ G_nc, const double* __restrict__ r_fd,
                                                                const double* __restrict__ G_rdd, const double* __restrict__ G_bb,
                                                                const double* __restrict__ G_ab, const double* __restrict__ G_aa,
                                                                const double* __restrict__ f2, const double* __restrict__ ceps,
                                                                double hdiam, double* __restrict__ eta_loc, int ldu, int m0) {
  // (u, eta_loc: arrays of ldu columns; this launch handles the nmu <= 16 columns from m0 on)
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = 5 * N, QN = Q * N, C = 5 * QN;
  double* uo = lds;                 // [W][nmu]   coefficients of the own + neighbour bases
  double* ur = uo + W * nmu;        // [C][nmu]   theta_q(mu_m) * coefficient, column order (slot, q, j)
  double* thq = ur + C * nmu;       // [Q][nmu]   theta_q per parameter (weights of the aa / ab terms)
  double* red = thq + 8 * EB;       // [4][3][EB]
  for (int i = tid; i < W * nmu; i += 256) {
    const int row = i / nmu, m = i - row * nmu, slot = row / N, j = row - slot * N;
    const int s2 = nbr[s * 5 + slot];
    const double val = s2 >= 0 ? u[((long)s2 * N + j) * ldu + m0 + m] : 0.0;
    uo[i] = val;
    for (int q = 0; q < Q; ++q) ur[((slot * Q + q) * N + j) * nmu + m] = th.v[m * 8 + q] * val;
  }
  for (int i = tid; i < Q * nmu; i += 256) thq[i] = th.v[(i % nmu) * 8 + i / nmu];
  __syncthreads();
  double a_nc[EB], a_r[EB], a_df[EB];
#pragma unroll
  for (int m = 0; m < EB; ++m) a_nc[m] = a_r[m] = a_df[m] = 0.0;
  const double* ui = uo + 2 * N * nmu;
  const double* zs = ur + 2 * QN * nmu;
  quad_batch(G_nc + (long)s * W * W, W, W, W, uo, uo, nmu, nullptr, 1.0, a_nc);
  const double* Gd = G_rdd + (long)s * 9 * QN * QN;
  const double* Gb = G_bb + (long)s * 9 * QN * QN;
  quad_batch(Gd, QN, QN, QN, zs, zs, nmu, nullptr, 1.0, a_r);
  quad_batch(Gb, QN, QN, QN, zs, zs, nmu, nullptr, 1.0, a_df);
  for (int side = 0; side < 4; ++side) {
    const double* za = ur + (side < 2 ? side : side + 1) * QN * nmu;
    quad_batch(Gd + (long)(1 + side) * QN * QN, QN, QN, QN, za, zs, nmu, nullptr, 2.0, a_r);
    quad_batch(Gd + (long)(5 + side) * QN * QN, QN, QN, QN, za, za, nmu, nullptr, 1.0, a_r);
    quad_batch(Gb + (long)(1 + side) * QN * QN, QN, QN, QN, za, zs, nmu, nullptr, 2.0, a_df);
    quad_batch(Gb + (long)(5 + side) * QN * QN, QN, QN, QN, za, za, nmu, nullptr, 1.0, a_df);
  }
  // - 2 r_fd . ur  (a 1 x C "matrix" against the constant row coefficient 1)
  if (wave == 0)
    for (int c = lane; c < C; c += 64) {
      const double g = r_fd[(long)s * C + c];
#pragma unroll
      for (int m = 0; m < EB; ++m)
        if (m < nmu) a_r[m] -= 2.0 * g * ur[c * nmu + m];
    }
  for (int q = 0; q < Q; ++q) {
    quad_batch(G_ab + ((long)q * S + s) * N * C, C, N, C, ui, ur, nmu, thq + q * nmu, 2.0, a_df);
    for (int q2 = 0; q2 < Q; ++q2) {
      double wq[EB];
#pragma unroll
      for (int m = 0; m < EB; ++m) wq[m] = m < nmu ? thq[q * nmu + m] * thq[q2 * nmu + m] : 0.0;
      quad_batch(G_aa + (((long)q * Q + q2) * S + s) * N * N, N, N, N, ui, ui, nmu, wq, 1.0, a_df);
    }
  }
  // fixed-order reductions: 64 lanes by shuffles, then the 4 waves through LDS
#pragma unroll
  for (int m = 0; m < EB; ++m) {
    double v0 = a_nc[m], v1 = a_r[m], v2 = a_df[m];
    for (int off = 32; off > 0; off >>= 1) {
      v0 += __shfl_down(v0, off, 64);
      v1 += __shfl_down(v1, off, 64);
      v2 += __shfl_down(v2, off, 64);
    }
    if (lane == 0) {
      red[(wave * 3 + 0) * EB + m] = v0;
      red[(wave * 3 + 1) * EB + m] = v1;
      red[(wave * 3 + 2) * EB + m] = v2;
    }
  }
  __syncthreads();
  if (tid < nmu) {
    const double pi = 3.14159265358979323846;
    double nc = 0.0, rr = 0.0, df = 0.0;
    for (int w = 0; w < 4; ++w) {
      nc += red[(w * 3 + 0) * EB + tid];
      rr += red[(w * 3 + 1) * EB + tid];
      df += red[(w * 3 + 2) * EB + tid];
    }
    eta_loc[((long)0 * S + s) * ldu + m0 + tid] = nc;
    eta_loc[((long)1 * S + s) * ldu + m0 + tid] = (f2[s] + rr) * ((1.0 / (pi * pi)) / ceps[s]) * hdiam * hdiam;   // estimators.py:88-91
    eta_loc[((long)2 * S + s) * ldu + m0 + tid] = df;
  }
}

// The same estimates on the fp64 matrix cores.  For every block G (R x C) of a projected operator the batch of quadratic
// forms  x_m^T G y_m  is  sum_r x[r][m] (G Y)[r][m]:  T = G Y is a 16-column MFMA product whose A operand is loaded from
// global memory directly in lane layout (row = lane & 15, k = lane >> 4: every entry of G is read exactly once, 8 bytes
// per lane, four k-steps consume one 128-byte line per row), Y comes from LDS with the per-parameter weights folded in,
// and each lane finishes with four multiply-adds against X.  The VALU form above needs 16 LDS operands per loaded entry
// and runs at 15 % of the HBM rate this kernel is bound by.
// waves per workgroup of the batched estimate: its coefficient panels take 77 KB of LDS at config 3 (two workgroups per
// CU), and every wave works through its tiles in batches of 8 loads, one round trip each -- eight waves instead of four
// double the loads in flight per CU at the same LDS footprint
constexpr int EST_NW = 8;
// Tiles are dealt to the waves through a running unit counter (`unit`, the same sequence in every wave).  Dense layout:
// round-robin.  Factored layout: waves 0-3 each carry the side factors of one side (the longest chain of the kernel),
// so out of every 12 tiles waves 4-7 take two and waves 0-3 one.
__device__ inline int est_owner(int unit, bool factored) {
  if (!factored) return unit % EST_NW;
  const int k = unit % 12;
  return k < 8 ? 4 + (k & 3) : k - 8;
}

__device__ inline void quad_mfma(const double* __restrict__ G, int ld, int R, int C, const double* X, const double* Y, double wl,
                                 double& acc, int& unit, bool factored) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wave = threadIdx.x >> 6;
  const int ntile = (R + 15) >> 4;
  for (int tile = 0; tile < ntile; ++tile) {
    if (est_owner(unit++, factored) != wave) continue;
    const int ra = tile * 16 + li < R ? tile * 16 + li : R - 1;           // rows >= R repeat the last row (their X is 0)
    const double* grow = G + (long)ra * ld;
    d4m T = (d4m){0.0, 0.0, 0.0, 0.0};
    int kk = 0;
    for (; kk + 32 <= C; kk += 32) {                                      // 8 loads in flight per lane
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = grow[kk + 4 * u + lk];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], wl * Y[(kk + 4 * u + lk) * 16 + li], T, 0, 0, 0);
    }
    for (; kk < C; kk += 4) {
      const bool in = kk + lk < C;                                        // columns >= C contribute nothing
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(grow[in ? kk + lk : C - 1], in ? wl * Y[(kk + lk) * 16 + li] : 0.0, T, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = tile * 16 + lk + 4 * rr;
      if (row < R) acc += X[row * 16 + li] * T[rr];
    }
  }
}

// T = G_rows Y for ONE tile of 16 rows: lane (li, lk) streams row pointer `grow` (its row of G, K entries), Y [K][16] from LDS
__device__ inline d4m tile_mfma(const double* __restrict__ grow, int K, const double* Y, double wl) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  d4m T = (d4m){0.0, 0.0, 0.0, 0.0};
  int kk = 0;
  for (; kk + 32 <= K; kk += 32) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = grow[kk + 4 * u + lk];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], wl * Y[(kk + 4 * u + lk) * 16 + li], T, 0, 0, 0);
  }
  for (; kk < K; kk += 4) {
    const bool in = kk + lk < K;
    T = __builtin_amdgcn_mfma_f64_16x16x4f64(grow[in ? kk + lk : K - 1], in ? wl * Y[(kk + lk) * 16 + li] : 0.0, T, 0, 0, 0);
  }
  return T;
}

// Side blocks of G_bb / G_rdd / G_ab from the factors F_side (see k_thin_rt in fused.hip): for side a with coefficient
// panels z_a (neighbour), z_s, u_i (own) the contributions are sum_p [sc0_p ra_p^2 + 2 ra_p (Yb z_s)_p + 2 theta_q (Xab_q u_i)_p ra_p]
// (diffusive flux) and sum_p [sc1_p ra_p^2 + 2 ra_p (Dp z_s)_p] (residual) with ra = Ra z_a -- 3 + Q thin products per side.
// (`ua`: the neighbour's coefficient panel [N][16]; z_a = theta_q u_a is formed in the B operand, lane li = parameter)
__device__ inline void side_factored(const double* __restrict__ Fs, int ncf, int Q, int N, const double* ua, const double* zs,
                                     const double* ui, const double* thl, double& a_r, double& a_df) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const int QN = Q * N, LD = 4 * QN + 4;
  for (int tile = 0; tile * 16 < ncf; ++tile) {
    const int ra_row = tile * 16 + li < ncf ? tile * 16 + li : ncf - 1;     // rows >= ncf repeat the last row (masked below)
    const double* grow = Fs + (long)ra_row * LD;
    d4m Tra = (d4m){0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < Q; ++q) {
      double tq = thl[0];
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (q == k) tq = thl[k];
      const d4m Tq = tile_mfma(grow + q * N, N, ua, tq);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Tra[rr] += Tq[rr];
    }
    const d4m Tyb = tile_mfma(grow + QN, QN, zs, 1.0);
    const d4m Tdd = tile_mfma(grow + 2 * QN, QN, zs, 1.0);
    d4m Tx = (d4m){0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < Q; ++q) {
      double tq = thl[0];
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (q == k) tq = thl[k];
      const d4m Tq = tile_mfma(grow + 3 * QN + q * N, N, ui, tq);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Tx[rr] += Tq[rr];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = tile * 16 + lk + 4 * rr;
      if (row < ncf) {
        const double* sc = Fs + (long)row * LD + 4 * QN;
        const double ra = Tra[rr];
        a_df += sc[0] * ra * ra + 2.0 * ra * (Tyb[rr] + Tx[rr]);
        a_r += sc[1] * ra * ra + 2.0 * ra * Tdd[rr];
      }
    }
  }
}

__global__ __launch_bounds__(64 * EST_NW) void k_reduced_estimate_batch_mfma(int S, const int* __restrict__ nbr, int Q, int N, int nmu,
                                                                     ThetaBatch th, const double* __restrict__ u,
                                                                     const double* __restrict__ G_nc, const double* __restrict__ r_fd,
                                                                     const double* __restrict__ G_rdd, const double* __restrict__ G_bb,
                                                                     const double* __restrict__ G_ab, const double* __restrict__ G_aa,
                                                                     const double* __restrict__ Fside, const double* __restrict__ Fnc,
                                                                     int ncf, int nvs, const double* __restrict__ f2,
                                                                     const double* __restrict__ ceps, double hdiam,
                                                                     double* __restrict__ eta_loc, int ldu, int m0, int nvx_patch,
                                                                     const int* __restrict__ nbr_diag) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4, wave = tid >> 6;
  const int W = 5 * N, QN = Q * N, C = 5 * QN;
  const bool factored = Fside != nullptr;
  // factored layout: only the own rows of the theta-weighted panel are kept (the neighbours' enter through the side
  // factors, with theta folded into the MFMA operand): 40 instead of 80 KB of LDS at config 3, three workgroups per CU
  const int Wp = (W + 3) & ~3, Cp = factored ? (QN + 3) & ~3 : (C + 3) & ~3, Zp = factored ? (4 * nvs + 3) & ~3 : 0;
  double* uo = lds;                 // [Wp][16]  coefficients of the own + neighbour bases (columns >= nmu and pad rows zero)
  double* ur = uo + Wp * 16;        // [Cp][16]  theta_q(mu_m) * coefficient: row order (slot, q, j), factored: (q, j) of the own slot
  double* red = ur + Cp * 16;       // [EST_NW][3][16]
  double* zn = red + EST_NW * 3 * 16;   // [Zp][16]  (factored layout) z_a = A_a u_a per side vertex, rows (side, vertex)
  double* dcn = zn + Zp * 16;           // [4][16]   LRBMS_OPT_OSWALD_VERTEX_PATCH: A_diag . u_diag per corner and parameter
  for (int i = tid; i < (Wp + Cp) * 16; i += 64 * EST_NW) lds[i] = 0.0;
  for (int i = tid; i < Zp * 16; i += 64 * EST_NW) zn[i] = 0.0;
  if (nvx_patch > 0 && tid < 64) {
    // cross points (see k_reduced_estimate): thread = (corner, parameter)
    const int corner = tid >> 4, m = tid & 15, side = corner < 2 ? 0 : 3, pos = (corner & 1) ? nvx_patch - 1 : 0;
    const int sd = nbr[s * 5 + (corner < 2 ? 0 : 4)] >= 0 ? nbr_diag[s * 4 + corner] : -1;
    double d = 0.0;
    if (sd >= 0 && m < nmu) {
      const double* x = Fnc + ((long)s * 4 * nvs + side * nvs + pos) * (3 * N + 4 * nvs) + 2 * N + 4 * nvs;
      for (int c = 0; c < N; ++c) d += x[c] * u[((long)sd * N + c) * ldu + m0 + m];
    }
    dcn[tid] = d;
  }
  __syncthreads();
  for (int i = tid; i < W * nmu; i += 64 * EST_NW) {
    const int row = i / nmu, m = i - row * nmu, slot = row / N, j = row - slot * N;
    const int s2 = nbr[s * 5 + slot];
    const double val = s2 >= 0 ? u[((long)s2 * N + j) * ldu + m0 + m] : 0.0;
    uo[row * 16 + m] = val;
    if (!factored) {
      for (int q = 0; q < Q; ++q) ur[((slot * Q + q) * N + j) * 16 + m] = th.v[m * 8 + q] * val;
    } else if (slot == 2) {
      for (int q = 0; q < Q; ++q) ur[(q * N + j) * 16 + m] = th.v[m * 8 + q] * val;
    }
  }
  __syncthreads();
  double thl[8];                    // theta_q of this lane's parameter
#pragma unroll
  for (int q = 0; q < 8; ++q) thl[q] = li < nmu ? th.v[li * 8 + q] : 0.0;
  double a_nc = 0.0, a_r = 0.0, a_df = 0.0;
  const double* ui = uo + 2 * N * 16;
  const double* zs = factored ? ur : ur + 2 * QN * 16;
  const long gstride = factored ? (long)QN * QN : (long)9 * QN * QN;
  const int abld = factored ? QN : C;
  // factored layout: the four sides go to waves 0-3, which get fewer tiles of the self operators below (est_owner)
  int unit = 0;
  if (factored) {
    // nonconformity side terms from F_nc (k_thin_ncf): z_a = A_a u_a, then sum_a z_a^T (2 C_a u_s + sum_b M_ab z_b); the
    // z panel of all four sides goes through LDS (every side needs every other side's)
    const int LDn = 2 * N + 4 * nvs + (nvx_patch > 0 ? N : 0);
    for (int side = wave; side < 4; side += EST_NW) {
      const double* ua = uo + (side < 2 ? side : side + 1) * N * 16;
      side_factored(Fside + ((long)s * 4 + side) * ncf * (4 * QN + 4), ncf, Q, N, ua, zs, ui, thl, a_r, a_df);
      const double* Fn = Fnc + ((long)s * 4 + side) * nvs * LDn;
      for (int tile = 0; tile * 16 < nvs; ++tile) {
        const int rr0 = tile * 16 + li < nvs ? tile * 16 + li : nvs - 1;
        const d4m Tz = tile_mfma(Fn + (long)rr0 * LDn, N, ua, 1.0);
        const d4m Tc = tile_mfma(Fn + (long)rr0 * LDn + N, N, ui, 1.0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = tile * 16 + lk + 4 * rr;
          if (row < nvs) {
            double zz = Tz[rr];
            if (nvx_patch > 0 && (side == 0 || side == 3) && (row == 0 || row == nvx_patch - 1))
              zz += dcn[((side == 0 ? 0 : 2) + (row == 0 ? 0 : 1)) * 16 + li];      // the diagonal subdomain's share at a cross point
            zn[(side * nvs + row) * 16 + li] = zz;
            a_nc += 2.0 * zz * Tc[rr];
          }
        }
      }
    }
    quad_mfma(G_nc + (long)s * N * N, N, N, N, ui, ui, 1.0, a_nc, unit, factored);
  } else {
    quad_mfma(G_nc + (long)s * W * W, W, W, W, uo, uo, 1.0, a_nc, unit, factored);
  }
  const double* Gd = G_rdd + (long)s * gstride;
  const double* Gb = G_bb + (long)s * gstride;
  quad_mfma(Gd, QN, QN, QN, zs, zs, 1.0, a_r, unit, factored);
  quad_mfma(Gb, QN, QN, QN, zs, zs, 1.0, a_df, unit, factored);
  if (!factored) {
    for (int side = 0; side < 4; ++side) {
      const double* za = ur + (side < 2 ? side : side + 1) * QN * 16;
      quad_mfma(Gd + (long)(1 + side) * QN * QN, QN, QN, QN, za, zs, 2.0, a_r, unit, factored);
      quad_mfma(Gd + (long)(5 + side) * QN * QN, QN, QN, QN, za, za, 1.0, a_r, unit, factored);
      quad_mfma(Gb + (long)(1 + side) * QN * QN, QN, QN, QN, za, zs, 2.0, a_df, unit, factored);
      quad_mfma(Gb + (long)(5 + side) * QN * QN, QN, QN, QN, za, za, 1.0, a_df, unit, factored);
    }
  }
  // - 2 r_fd . ur: lanes over (16 rows of c) x parameter
  if (!factored) {
    for (int c = wave * 4 + (lane >> 4); c < C; c += 4 * EST_NW) a_r -= 2.0 * r_fd[(long)s * C + c] * ur[c * 16 + li];
  } else {
    for (int c = wave * 4 + (lane >> 4); c < C; c += 4 * EST_NW) {
      const int slot = c / QN, rem = c - slot * QN, q = rem / N, j = rem - q * N;
      double tq = thl[0];
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (q == k) tq = thl[k];
      a_r -= 2.0 * r_fd[(long)s * C + c] * tq * uo[(slot * N + j) * 16 + li];
    }
  }
  for (int q = 0; q < Q; ++q) {
    double tq = thl[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (q == k) tq = thl[k];
    quad_mfma(G_ab + ((long)q * S + s) * N * abld, abld, N, abld, ui, factored ? zs : ur, 2.0 * tq, a_df, unit, factored);
    for (int q2 = 0; q2 < Q; ++q2) {
      double tq2 = thl[0];
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (q2 == k) tq2 = thl[k];
      quad_mfma(G_aa + (((long)q * Q + q2) * S + s) * N * N, N, N, N, ui, ui, tq * tq2, a_df, unit, factored);
    }
  }
  if (factored) {
    // second half of the nonconformity side terms, z_a^T M_ab z_b: needs the z panel of every side, so it comes behind a
    // barrier -- placed here, after the tiles of the self operators, so that no wave waits for the side waves' first half
    __syncthreads();
    const int LDn = 2 * N + 4 * nvs + (nvx_patch > 0 ? N : 0);
    for (int side = wave - 4; side >= 0 && side < 4; side += EST_NW) {
      const double* Fn = Fnc + ((long)s * 4 + side) * nvs * LDn + 2 * N;
      for (int tile = 0; tile * 16 < nvs; ++tile) {
        const int rr0 = tile * 16 + li < nvs ? tile * 16 + li : nvs - 1;
        const d4m Tm = tile_mfma(Fn + (long)rr0 * LDn, 4 * nvs, zn, 1.0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = tile * 16 + lk + 4 * rr;
          if (row < nvs) a_nc += zn[(side * nvs + row) * 16 + li] * Tm[rr];
        }
      }
    }
  }
  // fixed-order reductions: the four k-groups of a wave by shuffles (lanes li, li + 16, li + 32, li + 48 share a
  // parameter), then the waves through LDS
  for (int off = 32; off >= 16; off >>= 1) {
    a_nc += __shfl_down(a_nc, off, 64);
    a_r += __shfl_down(a_r, off, 64);
    a_df += __shfl_down(a_df, off, 64);
  }
  if (lane < 16) {
    red[(wave * 3 + 0) * 16 + lane] = a_nc;
    red[(wave * 3 + 1) * 16 + lane] = a_r;
    red[(wave * 3 + 2) * 16 + lane] = a_df;
  }
  __syncthreads();
  if (tid < nmu) {
    const double pi = 3.14159265358979323846;
    double nc = 0.0, rr = 0.0, df = 0.0;
    for (int w = 0; w < EST_NW; ++w) {
      nc += red[(w * 3 + 0) * 16 + tid];
      rr += red[(w * 3 + 1) * 16 + tid];
      df += red[(w * 3 + 2) * 16 + tid];
    }
    eta_loc[((long)0 * S + s) * ldu + m0 + tid] = nc;
    eta_loc[((long)1 * S + s) * ldu + m0 + tid] = (f2[s] + rr) * ((1.0 / (pi * pi)) / ceps[s]) * hdiam * hdiam;   // estimators.py:88-91
    eta_loc[((long)2 * S + s) * ldu + m0 + tid] = df;
  }
}

}  // namespace

int launch_reduced_estimate_batch(lrbms_ctx* ctx, int Q, int N, int nmu, const double* theta, const double* u, const double* G_nc,
                                  const double* r_fd, const double* G_rdd, const double* G_bb, const double* G_ab,
                                  const double* G_aa, const double* Fside, const double* Fnc, const double* f2, const double* ceps,
                                  double hdiam, double* eta_loc, hipStream_t st) {
  if ((Fside != nullptr) != (Fnc != nullptr)) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate_batch: F_side and F_nc go together");
  if (ctx->t.opt_oswald_vertex && (Fnc == nullptr || (ctx->S_ext != ctx->S && !ctx->diag_explicit)))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "LRBMS_OPT_OSWALD_VERTEX_PATCH: factored layout; sharded grids need lrbms_set_diagonal_neighbours");
  const int nvs = ctx->t.nvx > ctx->t.nvy ? ctx->t.nvx : ctx->t.nvy;
  if (nmu < 1 || nmu > 64) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate_batch: need 1 <= nmu <= 64");
  // passes of <= 16 parameters over the same u / eta_loc arrays (column offset m0): the layout lrbms_reduced_solve_batch returns
  // is consumed as it is, no repacking
  for (int m0 = 0; m0 < nmu; m0 += EB) {
    const int nm = nmu - m0 < EB ? nmu - m0 : EB;
    const size_t lds = sizeof(double) * ((size_t)(5 * N + 5 * Q * N) * nm + 8 * EB + 4 * 3 * EB);
    if (lds > 160 * 1024) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate_batch: coefficient panels exceed the LDS");
    ThetaBatch th;
    for (int m = 0; m < TBMAX; ++m)
      for (int q = 0; q < 8; ++q) th.v[m * 8 + q] = (m < nm && q < Q) ? theta[(m0 + m) * Q + q] : 0.0;
    if (Fside != nullptr || ctx->opt_estimate_valu == 0) {   // matrix-core form (default; the only one for the factored layout)
      const size_t ldm = sizeof(double) * ((size_t)(((5 * N + 3) & ~3) + (Fside ? ((Q * N + 3) & ~3) + ((4 * nvs + 3) & ~3) : (5 * Q * N + 3) & ~3)) * 16 +
                                           EST_NW * 3 * 16 + 64);
      if (ldm > 160 * 1024) return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate_batch: coefficient panels exceed the LDS");
      if (ldm > 64 * 1024)
        LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_reduced_estimate_batch_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)ldm));
      hipLaunchKernelGGL(k_reduced_estimate_batch_mfma, dim3(ctx->S), dim3(64 * EST_NW), ldm, st, ctx->S, ctx->nbr, Q, N, nm, th, u, G_nc, r_fd,
                         G_rdd, G_bb, G_ab, G_aa, Fside, Fnc, ctx->t.ncf, nvs, f2, ceps, hdiam, eta_loc, nmu, m0,
                         ctx->t.opt_oswald_vertex ? ctx->t.nvx : 0, ctx->t.nbr_diag);
    } else {
      if (lds > 64 * 1024)
        LRBMS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_reduced_estimate_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_reduced_estimate_batch, dim3(ctx->S), dim3(256), lds, st, ctx->S, ctx->nbr, Q, N, nm, th, u, G_nc, r_fd, G_rdd,
                         G_bb, G_ab, G_aa, f2, ceps, hdiam, eta_loc, nmu, m0);
    }
  }
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}
