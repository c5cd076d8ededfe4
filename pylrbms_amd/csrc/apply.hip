// Template-sparse x dense "apply" kernels of the timed project+estimate region
// (SURVEY.md section 8a rows K7, K8 and the SpMM halves of P1/P2).
//
// Every operator here is a few non-zeros per row whose positions come from the shared subdomain template, so the
// kernels are pure streaming: the flattened (subdomain, row, column) index runs with the basis column fastest, which
// makes every global load/store of V / Wt / Rt a contiguous run of N (or 5N, 5QN) doubles.  HBM-bound.
#include "lrbms_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// K7: Wt[s][r][slot*N + j] = delta_{slot,self} V[s][r][j] - I_os^{s}[V_kk extended by zero](r)[j]
// I_os averages over all elements of neighborhood_of(s) sharing the vertex of DoF r; vertices on the physical
// boundary interpolate to 0.
__global__ __launch_bounds__(256) void k_oswald(Tmpl t, int S, const int* __restrict__ nbr, int N,
                                                const double* __restrict__ V, double* __restrict__ Wt) {
  const long total = (long)S * t.n * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % N);
    const long sr = idx / N;
    const int r = (int)(sr % t.n), s = (int)(sr / t.n);
    const int v = t.dof_vertex[r];
    const int lx = v % t.nvx, ly = v / t.nvx;
    // which sides does the vertex lie on, and the matching vertex in that neighbour
    int vside[4];
    vside[0] = (ly == 0) ? lx + t.nvx * (t.nvy - 1) : -1;
    vside[1] = (lx == 0) ? (t.nvx - 1) + t.nvx * ly : -1;
    vside[2] = (lx == t.nvx - 1) ? 0 + t.nvx * ly : -1;
    vside[3] = (ly == t.nvy - 1) ? lx : -1;
    int cnt = t.vdof_ptr[v + 1] - t.vdof_ptr[v];
    bool dirichlet = false;
    for (int sd = 0; sd < 4; ++sd) {
      if (vside[sd] < 0) continue;
      if (nbr[s * 5 + side_to_slot(sd)] < 0 || t.opt_oswald_subdomain)
        dirichlet = true;
      else
        cnt += t.vdof_ptr[vside[sd] + 1] - t.vdof_ptr[vside[sd]];
    }
    const double inv = dirichlet ? 0.0 : 1.0 / (double)cnt;
    double* out = Wt + ((long)s * t.n + r) * (5 * N) + j;
    // self slot
    {
      double acc = 0.0;
      for (int p = t.vdof_ptr[v]; p < t.vdof_ptr[v + 1]; ++p) acc += V[((long)s * t.n + t.vdof_idx[p]) * N + j];
      out[2 * N] = V[((long)s * t.n + r) * N + j] - inv * acc;
    }
    for (int sd = 0; sd < 4; ++sd) {
      const int slot = side_to_slot(sd);
      double val = 0.0;
      const int s2 = nbr[s * 5 + slot];
      if (vside[sd] >= 0 && s2 >= 0 && !dirichlet) {
        double acc = 0.0;
        const int v2 = vside[sd];
        for (int p = t.vdof_ptr[v2]; p < t.vdof_ptr[v2 + 1]; ++p) acc += V[((long)s2 * t.n + t.vdof_idx[p]) * N + j];
        val = -inv * acc;
      }
      out[slot * N] = val;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// K8: Rt[s][r][(slot*Q + q)*N + j]: RT0 DoF r of the flux reconstruction of V_kk (kk = neighbour in `slot`),
// extended by zero, restricted to subdomain s.
__global__ __launch_bounds__(256) void k_flux(Tmpl t, int S, const int* __restrict__ nbr, int Q, int N,
                                              const double* __restrict__ F, const double* __restrict__ V,
                                              double* __restrict__ Rt) {
  const long total = (long)S * t.nrt * N;
  const int C = 5 * Q * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % N);
    const long sr = idx / N;
    const int r = (int)(sr % t.nrt), s = (int)(sr / t.nrt);
    const int e0 = t.rt_e0[r], e1 = t.rt_e1[r], side = t.rt_side[r];
    double* out = Rt + ((long)s * t.nrt + r) * C + j;
    const int side_slot = side >= 0 ? side_to_slot(side) : -1;
    const int s2 = side >= 0 ? nbr[s * 5 + side_slot] : -1;
    double v0[3], v1[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) v0[i] = V[((long)s * t.n + 3 * e0 + i) * N + j];
    if (side < 0)
      for (int i = 0; i < 3; ++i) v1[i] = V[((long)s * t.n + 3 * e1 + i) * N + j];
    else if (s2 >= 0)
      for (int i = 0; i < 3; ++i) v1[i] = V[((long)s2 * t.n + 3 * e1 + i) * N + j];
    for (int q = 0; q < Q; ++q) {
      const double* f = F + (((long)q * S + s) * t.nrt + r) * 6;
      double self = f[0] * v0[0] + f[1] * v0[1] + f[2] * v0[2];
      double other = f[3] * v1[0] + f[4] * v1[1] + f[5] * v1[2];
      for (int slot = 0; slot < 5; ++slot) {
        double val = 0.0;
        if (slot == 2)
          val = side < 0 ? self + other : self;
        else if (slot == side_slot && s2 >= 0)
          val = other;
        out[(slot * Q + q) * N] = val;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// y[s][3e+i][c] = sum_b sum_j A[s][e][b][i][j] x[s][3 nb(e,b) + j][c]   (block-ELL, diagonal blocks of a subdomain)
__global__ __launch_bounds__(256) void k_blockell_apply(Tmpl t, int S, int M, const double* __restrict__ A, long sA,
                                                        const double* __restrict__ x, double* __restrict__ y) {
  const long total = (long)S * t.nT * M;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % M);
    const long se = idx / M;
    const int e = (int)(se % t.nT), s = (int)(se / t.nT);
    const double* blk = A + (long)s * sA + (long)e * 36;
    double acc[3] = {0, 0, 0};
    for (int b = 0; b < 4; ++b) {
      const int e2 = b == 0 ? e : t.nb_elem[e * 3 + b - 1];
      if (e2 < 0) continue;
      const double* xr = x + ((long)s * t.n + 3 * e2) * M + c;
      const double x0 = xr[0], x1 = xr[M], x2 = xr[2 * M];
      for (int i = 0; i < 3; ++i) acc[i] += blk[b * 9 + i * 3] * x0 + blk[b * 9 + i * 3 + 1] * x1 + blk[b * 9 + i * 3 + 2] * x2;
    }
    double* yr = y + ((long)s * t.n + 3 * e) * M + c;
    yr[0] = acc[0];
    yr[M] = acc[1];
    yr[2 * M] = acc[2];
  }
}

// Full-order block operator: y_s = sum_q theta_q (A_diag_q x_s + coupling blocks x_neighbour)
struct QVecA { double v[8]; };
__global__ __launch_bounds__(256) void k_fom_apply(Tmpl t, int S, const int* __restrict__ nbr, int Q, QVecA theta, int M,
                                                   const double* __restrict__ A_diag, const double* __restrict__ A_cpl,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  const long total = (long)S * t.nT * M;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % M);
    const long se = idx / M;
    const int e = (int)(se % t.nT), s = (int)(se / t.nT);
    double acc[3] = {0, 0, 0};
    for (int b = 0; b < 4; ++b) {
      int e2 = e, s2 = s, side = -1;
      if (b > 0) {
        const int nb = t.nb_elem[e * 3 + b - 1];
        if (nb >= 0) {
          e2 = nb;
        } else {
          side = -1 - nb;
          s2 = nbr[s * 5 + side_to_slot(side)];
          if (s2 < 0) continue;
          e2 = t.nb_elem_out[e * 3 + b - 1];
        }
      }
      const double* xr = x + ((long)s2 * t.n + 3 * e2) * M + c;
      const double x0 = xr[0], x1 = xr[M], x2 = xr[2 * M];
      for (int q = 0; q < Q; ++q) {
        const double* blk = side < 0 ? A_diag + (((long)q * S + s) * t.nT + e) * 36 + b * 9
                                     : A_cpl + ((((long)q * S + s) * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + b - 1]) * 9;
        const double th = theta.v[q];
        for (int i = 0; i < 3; ++i) acc[i] += th * (blk[i * 3] * x0 + blk[i * 3 + 1] * x1 + blk[i * 3 + 2] * x2);
      }
    }
    double* yr = y + ((long)s * t.n + 3 * e) * M + c;
    yr[0] = acc[0];
    yr[M] = acc[1];
    yr[2 * M] = acc[2];
  }
}

// y[s][3e+i][c] = scal[s][e] * sum_j K_e[i][j] x[s][3e+j][c]  (mode 0: K_e = stiffness template;
// mode 1: K_e = mass template |T|/12 (1 + delta_ij), scal ignored)
__global__ __launch_bounds__(256) void k_elemdiag_apply(Tmpl t, int S, int M, int mode, const double* __restrict__ scal,
                                                        const double* __restrict__ x, double* __restrict__ y) {
  const long total = (long)S * t.nT * M;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % M);
    const long se = idx / M;
    const int e = (int)(se % t.nT), s = (int)(se / t.nT);
    const double* xr = x + ((long)s * t.n + 3 * e) * M + c;
    const double xv[3] = {xr[0], xr[M], xr[2 * M]};
    double* yr = y + ((long)s * t.n + 3 * e) * M + c;
    if (mode == 0) {
      const double sc = scal[(long)s * t.nT + e];
      for (int i = 0; i < 3; ++i) {
        const double gx = t.grad[(e * 3 + i) * 2], gy = t.grad[(e * 3 + i) * 2 + 1];
        double acc = 0.0;
        for (int j = 0; j < 3; ++j) {
          const double hx = t.grad[(e * 3 + j) * 2], hy = t.grad[(e * 3 + j) * 2 + 1];
          acc += (gx * (t.kappa[0] * hx + t.kappa[1] * hy) + gy * (t.kappa[2] * hx + t.kappa[3] * hy)) * xv[j];
        }
        yr[i * M] = sc * acc;
      }
    } else {
      const double m = t.area[e] / 12.0;
      const double sum = xv[0] + xv[1] + xv[2];
      for (int i = 0; i < 3; ++i) yr[i * M] = m * (sum + xv[i]);
    }
  }
}

// out[s][c] = y_s^T M_s^{-1} y_s for column c of Y [S][n][C]: the P1 mass matrix is block diagonal with the element
// blocks |T|/12 (I + J), whose inverse is 12/|T| (I - J/4)   (l2_product.apply_inverse(...).pairwise_dot(...),
// estimators.py:148)
__global__ __launch_bounds__(256) void k_mass_inv_norm2(Tmpl t, int S, int C, const double* __restrict__ Y, double* __restrict__ out) {
  const long total = (long)S * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C), s = (int)(idx / C);
    double acc = 0.0;
    for (int e = 0; e < t.nT; ++e) {
      const double* yr = Y + ((long)s * t.n + 3 * e) * C + c;
      const double y0 = yr[0], y1 = yr[C], y2 = yr[2 * (long)C];
      const double sum = y0 + y1 + y2;
      acc += 12.0 / t.area[e] * (y0 * y0 + y1 * y1 + y2 * y2 - 0.25 * sum * sum);
    }
    out[idx] = acc;
  }
}

// D[s][e][c] = sum_f sign_f |e_f| / |T| Rt[s][rt(e,f)][c]   (div of the RT0 function, one value per element)
__global__ __launch_bounds__(256) void k_div_apply(Tmpl t, int S, const int* __restrict__ nbr, int C,
                                                   const double* __restrict__ Rt, double* __restrict__ D) {
  const long total = (long)S * t.nT * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long se = idx / C;
    const int e = (int)(se % t.nT), s = (int)(se / t.nT);
    double acc = 0.0;
    for (int f = 0; f < 3; ++f) {
      int sign = t.face_sign[e * 3 + f];
      const int nb = t.nb_elem[e * 3 + f];
      if (nb < 0 && nbr[s * 5 + side_to_slot(-1 - nb)] < 0) sign = 1;
      acc += sign * t.face_len[e * 3 + f] * Rt[((long)s * t.nrt + t.elem_rt[e * 3 + f]) * C + c];
    }
    D[idx] = acc / t.area[e];
  }
}

// BR[s][r][c] = sum_{T contains r} sum_g Bbb[s][T][f_r][g] Rt[s][rt(T,g)][c]
__global__ __launch_bounds__(256) void k_bb_apply(Tmpl t, int S, int C, const double* __restrict__ Bbb,
                                                  const double* __restrict__ Rt, double* __restrict__ BR) {
  const long total = (long)S * t.nrt * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long sr = idx / C;
    const int r = (int)(sr % t.nrt), s = (int)(sr / t.nrt);
    double acc = 0.0;
    for (int w = 0; w < 2; ++w) {
      int e, f;
      if (w == 0) {
        e = t.rt_e0[r];
        f = t.rt_f0[r];
      } else {
        if (t.rt_side[r] >= 0) break;  // the second element lives in another subdomain
        e = t.rt_e1[r];
        f = t.rt_f1[r];
      }
      const double* B = Bbb + ((long)s * t.nT + e) * 9 + f * 3;
      for (int g = 0; g < 3; ++g) acc += B[g] * Rt[((long)s * t.nrt + t.elem_rt[e * 3 + g]) * C + c];
    }
    BR[idx] = acc;
  }
}

// AR[s][3e+i][c] = sum_f Aab[s][e][i][f] Rt[s][rt(e,f)][c]
__global__ __launch_bounds__(256) void k_ab_apply(Tmpl t, int S, int C, const double* __restrict__ Aab,
                                                  const double* __restrict__ Rt, double* __restrict__ AR) {
  const long total = (long)S * t.nT * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long se = idx / C;
    const int e = (int)(se % t.nT), s = (int)(se / t.nT);
    const double* A = Aab + ((long)s * t.nT + e) * 9;
    double rv[3];
    for (int f = 0; f < 3; ++f) rv[f] = Rt[((long)s * t.nrt + t.elem_rt[e * 3 + f]) * C + c];
    double* out = AR + ((long)s * t.n + 3 * e) * C + c;
    for (int i = 0; i < 3; ++i) out[(long)i * C] = A[i * 3] * rv[0] + A[i * 3 + 1] * rv[1] + A[i * 3 + 2] * rv[2];
  }
}

// out[s][c] = sum_T bsum[s][T] D[s][T][c]  with bsum = sum_i b[s][3T+i]   (r_fd); one workgroup column-chunk per s
__global__ __launch_bounds__(256) void k_rfd(Tmpl t, int C, const double* __restrict__ b, const double* __restrict__ D,
                                             double* __restrict__ out) {
  const int s = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double acc = 0.0;
  for (int e = 0; e < t.nT; ++e) {
    const double* be = b + (long)s * t.n + 3 * e;
    acc += (be[0] + be[1] + be[2]) * D[((long)s * t.nT + e) * C + c];
  }
  out[(long)s * C + c] = acc;
}

// out[s][j] = sum_r b[s][r] V[s][r][j]
__global__ __launch_bounds__(64) void k_project_rhs(Tmpl t, int N, const double* __restrict__ b,
                                                    const double* __restrict__ V, double* __restrict__ out) {
  const int s = blockIdx.x;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    double acc = 0.0;
    for (int r = 0; r < t.n; ++r) acc += b[(long)s * t.n + r] * V[((long)s * t.n + r) * N + j];
    out[(long)s * N + j] = acc;
  }
}

// Coupling blocks of the projected system: out[q][s][slot][a][b] = sum_pos sum_ij V_s[3 e_in + i][a] C[pos][i][j]
// V_s2[3 e_out + j][b].  One workgroup per (side, s, q); K = 3 ncf rows staged through LDS.
__global__ __launch_bounds__(256) void k_project_coupling(Tmpl t, int S, const int* __restrict__ nbr, int N,
                                                          const double* __restrict__ V, const double* __restrict__ A_cpl,
                                                          double* __restrict__ B_sys) {
  extern __shared__ double lds[];
  const int side = blockIdx.x, s = blockIdx.y, q = blockIdx.z;
  const int slot = side_to_slot(side);
  double* out = B_sys + ((((long)q * S + s) * 5 + slot) * N) * N;
  const int s2 = nbr[s * 5 + slot];
  if (s2 < 0) {
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) out[i] = 0.0;
    return;
  }
  const int cnt = t.side_count[side];
  const int K = 3 * cnt;
  double* Xin = lds;           // [K][N]  rows of V_s at the side
  double* Tm = lds + K * N;    // [K][N]  C * rows of V_s2
  for (int i = threadIdx.x; i < K * N; i += blockDim.x) {
    const int row = i / N, col = i % N;
    const int pos = row / 3, ii = row % 3;
    const int ein = t.side_elem[side * t.ncf + pos], eout = t.side_elem_out[side * t.ncf + pos];
    Xin[i] = V[((long)s * t.n + 3 * ein + ii) * N + col];
    const double* C = A_cpl + ((((long)q * S + s) * 4 + side) * t.ncf + pos) * 9 + ii * 3;
    const double* v2 = V + ((long)s2 * t.n + 3 * eout) * N + col;
    Tm[i] = C[0] * v2[0] + C[1] * v2[N] + C[2] * v2[2 * N];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    const int a = i / N, b = i % N;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += Xin[k * N + a] * Tm[k * N + b];
    out[i] = acc;
  }
}

// dense [S][C][C] (fixed 5-slot layout) -> block-compact [S][9][QN][QN]: block 0 = [self,self], 1 + side = [a,self],
// 5 + side = [a,a]; every other block of the dense matrix is structurally zero (tests/common.py checks that)
__global__ __launch_bounds__(256) void k_extract_blocks(int S, int QN, const double* __restrict__ dense, double* __restrict__ blocks) {
  const int C = 5 * QN;
  const long total = (long)S * 9 * QN * QN;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % QN);
    long rest = idx / QN;
    const int r = (int)(rest % QN);
    rest /= QN;
    const int b = (int)(rest % 9), s = (int)(rest / 9);
    int rslot = 2, cslot = 2;
    if (b >= 1 && b <= 4) rslot = side_to_slot(b - 1);
    if (b >= 5) rslot = cslot = side_to_slot(b - 5);
    blocks[idx] = dense[((long)s * C + rslot * QN + r) * C + cslot * QN + c];
  }
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

int launch_oswald(lrbms_ctx* ctx, int N, const double* V, double* Wt, hipStream_t st) {
  const Tmpl& t = ctx->t;
  if (t.opt_oswald_vertex)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "LRBMS_OPT_OSWALD_VERTEX_PATCH: the image basis Wt has five slots per subdomain, the "
                                            "diagonal subdomains enter through the factored layout of the fused pass only");
  hipLaunchKernelGGL(k_oswald, dim3(grid_for((long)ctx->S * t.n * N)), dim3(256), 0, st, t, ctx->S, ctx->nbr, N, V, Wt);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_flux(lrbms_ctx* ctx, int Q, int N, const double* F, const double* V, double* Rt, hipStream_t st) {
  const Tmpl& t = ctx->t;
  hipLaunchKernelGGL(k_flux, dim3(grid_for((long)ctx->S * t.nrt * N)), dim3(256), 0, st, t, ctx->S, ctx->nbr, Q, N, F, V, Rt);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_mass_inverse_norm2(lrbms_ctx* ctx, int C, const double* Y, double* out, hipStream_t st) {
  if (C < 1) return lrbms_fail(ctx, LRBMS_E_INVALID, "mass_inverse_norm2: C < 1");
  const long total = (long)ctx->S * C;
  hipLaunchKernelGGL(k_mass_inv_norm2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ctx->t, ctx->S, C, Y, out);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// out[s][3e+i][c] = (M_s Div_s Rt_s)[3e+i][c] = |T|/3 * div_T(c)   (the mass matrix applied to the constant (d, d, d))
__global__ __launch_bounds__(256) void k_mass_div_apply(Tmpl t, int S, const int* __restrict__ nbr, int C,
                                                        const double* __restrict__ Rt, double* __restrict__ MD) {
  const long total = (long)S * t.nT * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long se = idx / C;
    const int e = (int)(se % t.nT), s = (int)(se / t.nT);
    double acc = 0.0;
    for (int f = 0; f < 3; ++f) {
      int sign = t.face_sign[e * 3 + f];
      const int nb = t.nb_elem[e * 3 + f];
      if (nb < 0 && nbr[s * 5 + side_to_slot(-1 - nb)] < 0) sign = 1;
      acc += sign * t.face_len[e * 3 + f] * Rt[((long)s * t.nrt + t.elem_rt[e * 3 + f]) * C + c];
    }
    const double v = acc / 3.0;
    double* o = MD + ((long)s * t.n + 3 * e) * C + c;
    o[0] = v;
    o[C] = v;
    o[2 * (long)C] = v;
  }
}

// out[s][l] = sum_T (g_{3T} + g_{3T+1} + g_{3T+2})[l] * sum_{slot, q} theta_q D[s][T][(slot Q + q) L + l]:
// r_ud_s(M^-1 g, U_r) of estimators.py:83 for full-order vectors (the mass matrix and its inverse cancel; D = Div Rt)
__global__ __launch_bounds__(256) void k_div_pairing(Tmpl t, int S, int Q, int L, QVecA theta, const double* __restrict__ D,
                                                     const double* __restrict__ G, double* __restrict__ out) {
  const long total = (long)S * L;
  const int C = 5 * Q * L;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx % L), s = (int)(idx / L);
    double acc = 0.0;
    for (int e = 0; e < t.nT; ++e) {
      const double* g = G + ((long)s * t.n + 3 * e) * L + l;
      const double gs = g[0] + g[L] + g[2 * (long)L];
      const double* d = D + ((long)s * t.nT + e) * C + l;
      double div = 0.0;
      for (int slot = 0; slot < 5; ++slot)
        for (int q = 0; q < Q; ++q) div += theta.v[q] * d[(long)(slot * Q + q) * L];
      acc += gs * div;
    }
    out[idx] = acc;
  }
}

int launch_div_apply(lrbms_ctx* ctx, int C, int mode, const double* Rt, double* out, hipStream_t st) {
  if (C < 1 || mode < 0 || mode > 1) return lrbms_fail(ctx, LRBMS_E_INVALID, "div_apply: bad C / mode");
  const long total = (long)ctx->S * ctx->t.nT * C;
  const unsigned grid = (unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  if (mode == 0)
    hipLaunchKernelGGL(k_div_apply, dim3(grid), dim3(256), 0, st, ctx->t, ctx->S, ctx->nbr, C, Rt, out);
  else
    hipLaunchKernelGGL(k_mass_div_apply, dim3(grid), dim3(256), 0, st, ctx->t, ctx->S, ctx->nbr, C, Rt, out);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_div_pairing(lrbms_ctx* ctx, int Q, int L, const double* theta, const double* D, const double* G, double* out, hipStream_t st) {
  if (Q < 1 || Q > 8 || L < 1) return lrbms_fail(ctx, LRBMS_E_INVALID, "div_pairing: bad Q / L");
  QVecA th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  const long total = (long)ctx->S * L;
  hipLaunchKernelGGL(k_div_pairing, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ctx->t, ctx->S, Q, L, th, D, G, out);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_blockell_apply(lrbms_ctx* ctx, int S, int M, const double* A, long sA, const double* x, double* y,
                          hipStream_t st) {
  const Tmpl& t = ctx->t;
  hipLaunchKernelGGL(k_blockell_apply, dim3(grid_for((long)S * t.nT * M)), dim3(256), 0, st, t, S, M, A, sA, x, y);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_project_coupling(lrbms_ctx* ctx, int Q, int N, const double* V, const double* A_cpl, double* B_sys, hipStream_t st) {
  const Tmpl& t = ctx->t;
  const size_t lds = sizeof(double) * 2 * 3 * t.ncf * N;
  hipLaunchKernelGGL(k_project_coupling, dim3(4, ctx->S, Q), dim3(256), lds, st, t, ctx->S, ctx->nbr, N, V, A_cpl, B_sys);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_fom_apply(lrbms_ctx* ctx, int Q, int M, const double* theta, const double* A_diag, const double* A_cpl,
                     const double* x, double* y, hipStream_t st) {
  const Tmpl& t = ctx->t;
  QVecA th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  hipLaunchKernelGGL(k_fom_apply, dim3(grid_for((long)ctx->S * t.nT * M)), dim3(256), 0, st, t, ctx->S, ctx->nbr, Q, th, M,
                     A_diag, A_cpl, x, y);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// P1 driver: work holds one [S][n][N] slab that is reused for every operator.
int launch_project_system(lrbms_ctx* ctx, int Q, int N, const double* V, const double* A_diag, const double* A_cpl,
                          const double* P_diag, const double* b, double* work, double* B_sys, double* rhs_red,
                          double* E_red, double* M_red, hipStream_t st) {
  const Tmpl& t = ctx->t;
  const int S = ctx->S;
  const long slab = (long)S * t.n * N;
  int rc;
  for (int q = 0; q < Q; ++q) {
    double* AV = work + (long)q * slab;
    rc = launch_blockell_apply(ctx, S, N, A_diag + (long)q * S * t.nT * 36, (long)t.nT * 36, V, AV, st);
    if (rc) return rc;
    // diagonal block -> slot 2
    rc = launch_gemm_tn(ctx, S, t.n, N, N, V, (long)t.n * N, N, AV, (long)t.n * N, N,
                        B_sys + ((long)q * S * 5 + 2) * N * N, (long)5 * N * N, N, nullptr, 1.0, st);
    if (rc) return rc;
  }
  const size_t lds = sizeof(double) * 2 * 3 * t.ncf * N;
  hipLaunchKernelGGL(k_project_coupling, dim3(4, S, Q), dim3(256), lds, st, t, S, ctx->nbr, N, V, A_cpl, B_sys);
  LRBMS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(k_project_rhs, dim3(S), dim3(64), 0, st, t, N, b, V, rhs_red);
  LRBMS_LAUNCH_CHECK(ctx);
  double* tmp = work;
  rc = launch_blockell_apply(ctx, S, N, P_diag, (long)t.nT * 36, V, tmp, st);
  if (rc) return rc;
  rc = launch_gemm_tn(ctx, S, t.n, N, N, V, (long)t.n * N, N, tmp, (long)t.n * N, N, E_red, (long)N * N, N, nullptr, 1.0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_elemdiag_apply, dim3(grid_for((long)S * t.nT * N)), dim3(256), 0, st, t, S, N, 1, nullptr, V, tmp);
  LRBMS_LAUNCH_CHECK(ctx);
  return launch_gemm_tn(ctx, S, t.n, N, N, V, (long)t.n * N, N, tmp, (long)t.n * N, N, M_red, (long)N * N, N, nullptr, 1.0, st);
}

// ---------------------------------------------------------------------------------------------------------
// P2 driver.  Scratch layout (doubles): Y [S][n][C] (largest operand), D [S][n_T][C].
int64_t estimator_work_size(lrbms_ctx* ctx, int Q, int N) {
  const Tmpl& t = ctx->t;
  const long C = 5L * Q * N;
  return (long)ctx->S * t.n * C + (long)ctx->S * t.nT * C + (long)ctx->S * C * C;   // Y, D, one dense Gram
}

int launch_estimator_grams(lrbms_ctx* ctx, int Q, int N, const double* V, const double* Wt, const double* Rt,
                           const double* ebar, const double* caa, const double* Aab, const double* Bbb, const double* b,
                           double* work, double* G_nc, double* r_fd, double* G_rdd, double* G_bb, double* G_ab,
                           double* G_aa, hipStream_t st) {
  const Tmpl& t = ctx->t;
  const int S = ctx->S;
  const int W = 5 * N, C = 5 * Q * N;
  double* Y = work;
  double* D = work + (long)S * t.n * C;
  double* dense = D + (long)S * t.nT * C;      // dense [S][C][C] scratch, extracted into the block-compact outputs
  const int QN = Q * N;
  int rc;
  // nc_ii = Wt^T E_ii Wt  (block_swipdg.py:733)
  hipLaunchKernelGGL(k_elemdiag_apply, dim3(grid_for((long)S * t.nT * W)), dim3(256), 0, st, t, S, W, 0, ebar, Wt, Y);
  LRBMS_LAUNCH_CHECK(ctx);
  rc = launch_gemm_tn(ctx, S, t.n, W, W, Wt, (long)t.n * W, W, Y, (long)t.n * W, W, G_nc, (long)W * W, W, nullptr, 1.0, st);
  if (rc) return rc;
  // residual: D = div Rt; r_fd = b . D (:744); r_dd = D^T M D = sum_T |T| d_T d_T^T (:747)
  hipLaunchKernelGGL(k_div_apply, dim3(grid_for((long)S * t.nT * C)), dim3(256), 0, st, t, S, ctx->nbr, C, Rt, D);
  LRBMS_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(k_rfd, dim3((C + 255) / 256, S), dim3(256), 0, st, t, C, b, D, r_fd);
  LRBMS_LAUNCH_CHECK(ctx);
  rc = launch_gemm_tn(ctx, S, t.nT, C, C, D, (long)t.nT * C, C, D, (long)t.nT * C, C, dense, (long)C * C, C, t.area, 1.0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_extract_blocks, dim3(grid_for((long)S * 9 * QN * QN)), dim3(256), 0, st, S, QN, dense, G_rdd);
  LRBMS_LAUNCH_CHECK(ctx);
  // df_bb = Rt^T B Rt (:762)
  hipLaunchKernelGGL(k_bb_apply, dim3(grid_for((long)S * t.nrt * C)), dim3(256), 0, st, t, S, C, Bbb, Rt, Y);
  LRBMS_LAUNCH_CHECK(ctx);
  rc = launch_gemm_tn(ctx, S, t.nrt, C, C, Rt, (long)t.nrt * C, C, Y, (long)t.nrt * C, C, dense, (long)C * C, C, nullptr, 1.0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_extract_blocks, dim3(grid_for((long)S * 9 * QN * QN)), dim3(256), 0, st, S, QN, dense, G_bb);
  LRBMS_LAUNCH_CHECK(ctx);
  // df_ab^q = V^T A_ab^q Rt (:765-770)
  for (int q = 0; q < Q; ++q) {
    hipLaunchKernelGGL(k_ab_apply, dim3(grid_for((long)S * t.nT * C)), dim3(256), 0, st, t, S, C,
                       Aab + (long)q * S * t.nT * 9, Rt, Y);
    LRBMS_LAUNCH_CHECK(ctx);
    rc = launch_gemm_tn(ctx, S, t.n, N, C, V, (long)t.n * N, N, Y, (long)t.n * C, C, G_ab + (long)q * S * N * C,
                        (long)N * C, C, nullptr, 1.0, st);
    if (rc) return rc;
  }
  // df_aa^{q q'} = V^T (c^{qq'} K) V (:752-760)
  for (int q = 0; q < Q; ++q)
    for (int q2 = 0; q2 < Q; ++q2) {
      hipLaunchKernelGGL(k_elemdiag_apply, dim3(grid_for((long)S * t.nT * N)), dim3(256), 0, st, t, S, N, 0,
                         caa + ((long)q * Q + q2) * S * t.nT, V, Y);
      LRBMS_LAUNCH_CHECK(ctx);
      rc = launch_gemm_tn(ctx, S, t.n, N, N, V, (long)t.n * N, N, Y, (long)t.n * N, N,
                          G_aa + ((long)q * Q + q2) * S * N * N, (long)N * N, N, nullptr, 1.0, st);
      if (rc) return rc;
    }
  return LRBMS_OK;
}
