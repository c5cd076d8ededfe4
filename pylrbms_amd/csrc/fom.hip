// Full-order solve (SURVEY.md section 8f "next" #2: snapshot generation).
//
// Reference: DuneDiscretization._solve (discretize_elliptic_block_swipdg.py:219-225) hands the assembled global matrix
// to ISTL (bicgstab.ilut, python/scripts/online_adaptive_lrbms.py:71).  Here the block operator is never assembled:
// the matvec works on the block-ELL data of the discretization (A_diag: diagonal blocks of every subdomain, A_cpl:
// coupling blocks on the side faces) and the solver is a preconditioned CG -- the SWIPDG operator is symmetric positive
// definite -- with the 3x3 element blocks as block-Jacobi preconditioner.  Per iteration: four launches
//   k_fom_cg_matvec   p = z + beta p (own and neighbouring elements, on the fly), y = A(mu) p, partial p.y per workgroup
//   k_fom_cg_reduce   alpha = rz / pAp
//   k_fom_cg_update   x += alpha p, r -= alpha y, z = M^-1 r, partial r.z and r.r
//   k_fom_cg_reduce   beta = rz' / rz
// all reductions fixed-order (per-workgroup partials + one-workgroup tree), no host synchronisation except the
// convergence check every 25 iterations.  The theta-weighted blocks are combined once per solve.
#include "lrbms_dev.h"

namespace {

struct QVecF { double v[8]; };

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

// Amu_d [S][nT][4][9] = sum_q theta_q A_diag_q, Amu_c [S][4][ncf][9] = sum_q theta_q A_cpl_q,
// Minv [S][nT][9] = inverse of the diagonal 3x3 block
__global__ __launch_bounds__(256) void k_fom_combine(Tmpl t, int S, int Q, QVecF th, const double* __restrict__ A_diag,
                                                     const double* __restrict__ A_cpl, double* __restrict__ Amu_d,
                                                     double* __restrict__ Amu_c, double* __restrict__ Minv) {
  const long nd = (long)S * t.nT * 36, nc = (long)S * 4 * t.ncf * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nd + nc; i += (long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    if (i < nd) {
      for (int q = 0; q < Q; ++q) acc += th.v[q] * A_diag[(long)q * nd + i];
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

// one thread per (subdomain, element); scal: [0] rz, [1] pAp, [2] alpha, [3] beta, [4] rr
__global__ __launch_bounds__(256) void k_fom_cg_matvec(Tmpl t, int S, const int* __restrict__ nbr, const double* __restrict__ Amu_d,
                                                       const double* __restrict__ Amu_c, const double* __restrict__ z,
                                                       const double* __restrict__ p_old, const double* __restrict__ scal, int first,
                                                       double* __restrict__ p_new, double* __restrict__ y, double* __restrict__ partial) {
  __shared__ double red[256];
  const long ne = (long)S * t.nT;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const double beta = first ? 0.0 : scal[3];
  double dot = 0.0;
  if (idx < ne) {
    const int e = (int)(idx % t.nT), s = (int)(idx / t.nT);
    double acc[3] = {0.0, 0.0, 0.0}, pe[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < 4; ++b) {
      int e2 = e, s2 = s;
      const double* blk = Amu_d + idx * 36 + b * 9;
      if (b > 0) {
        const int nb = t.nb_elem[e * 3 + b - 1];
        if (nb >= 0) {
          e2 = nb;
        } else {
          const int side = -1 - nb;
          s2 = nbr[s * 5 + side_to_slot(side)];
          if (s2 < 0) continue;
          e2 = t.nb_elem_out[e * 3 + b - 1];
          blk = Amu_c + (((long)s * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + b - 1]) * 9;
        }
      }
      const long g = ((long)s2 * t.nT + e2) * 3;
      double pv[3];
      for (int i = 0; i < 3; ++i) pv[i] = first ? z[g + i] : z[g + i] + beta * p_old[g + i];
      if (b == 0)
        for (int i = 0; i < 3; ++i) pe[i] = pv[i];
      for (int i = 0; i < 3; ++i) acc[i] += blk[i * 3] * pv[0] + blk[i * 3 + 1] * pv[1] + blk[i * 3 + 2] * pv[2];
    }
    for (int i = 0; i < 3; ++i) {
      p_new[idx * 3 + i] = pe[i];
      y[idx * 3 + i] = acc[i];
      dot += pe[i] * acc[i];
    }
  }
  const double sum = block_sum_256(dot, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = sum;
}

// mode 0: scal[0] = sum(partial), scal[4] = sum(partial2) (start);  mode 1: pAp -> alpha;  mode 2: rz' -> beta, rz; rr
__global__ __launch_bounds__(1024) void k_fom_cg_reduce(int n, const double* __restrict__ partial, const double* __restrict__ partial2,
                                                        double* __restrict__ scal, int mode) {
  __shared__ double red[1024], red2[1024];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    a += partial[i];
    if (partial2) b += partial2[i];
  }
  red[threadIdx.x] = a;
  red2[threadIdx.x] = b;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[threadIdx.x] += red[threadIdx.x + off];
      red2[threadIdx.x] += red2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (mode == 0) {
      scal[0] = red[0];
      scal[4] = red2[0];
    } else if (mode == 1) {
      scal[1] = red[0];
      scal[2] = red[0] != 0.0 ? scal[0] / red[0] : 0.0;
    } else {
      scal[3] = scal[0] != 0.0 ? red[0] / scal[0] : 0.0;
      scal[0] = red[0];
      scal[4] = red2[0];
    }
  }
}

__global__ __launch_bounds__(256) void k_fom_cg_update(long ne, const double* __restrict__ Minv, const double* __restrict__ scal, int first,
                                                       double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                       const double* __restrict__ y, double* __restrict__ z,
                                                       double* __restrict__ partial, double* __restrict__ partial2) {
  __shared__ double red[256];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const double alpha = first ? 0.0 : scal[2];
  double rz = 0.0, rr = 0.0;
  if (idx < ne) {
    double rv[3];
    for (int i = 0; i < 3; ++i) {
      const long g = idx * 3 + i;
      if (!first) x[g] += alpha * p[g];
      rv[i] = first ? r[g] : r[g] - alpha * y[g];
      r[g] = rv[i];
    }
    const double* M = Minv + idx * 9;
    for (int i = 0; i < 3; ++i) {
      const double zi = M[i * 3] * rv[0] + M[i * 3 + 1] * rv[1] + M[i * 3 + 2] * rv[2];
      z[idx * 3 + i] = zi;
      rz += rv[i] * zi;
      rr += rv[i] * rv[i];
    }
  }
  const double s1 = block_sum_256(rz, red);
  const double s2 = block_sum_256(rr, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s1;
    partial2[blockIdx.x] = s2;
  }
}

}  // namespace

int64_t fom_solve_work_size(lrbms_ctx* ctx) {
  const Tmpl& t = ctx->t;
  const int64_t S = ctx->S, ne = S * t.nT;
  const int64_t nblk = (ne + 255) / 256;
  return ne * 36 + S * 4 * t.ncf * 9 + ne * 9 + 5 * ne * 3 + 2 * nblk + 16;
}

int launch_fom_solve(lrbms_ctx* ctx, int Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                     double* work, double* x, double rtol, int max_iter, double* info, hipStream_t st) {
  if (ctx->S_ext != ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "fom_solve needs all subdomains on one rank");
  if (Q < 1 || Q > 8 || max_iter < 1 || !(rtol > 0.0)) return lrbms_fail(ctx, LRBMS_E_INVALID, "fom_solve: bad Q / max_iter / rtol");
  const Tmpl& t = ctx->t;
  const int S = ctx->S;
  const long ne = (long)S * t.nT, nv = ne * 3;
  const int nblk = (int)((ne + 255) / 256);
  QVecF th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  double* Amu_d = work;
  double* Amu_c = Amu_d + ne * 36;
  double* Minv = Amu_c + (long)S * 4 * t.ncf * 9;
  double* r = Minv + ne * 9;
  double* z = r + nv;
  double* p0 = z + nv;
  double* p1 = p0 + nv;
  double* y = p1 + nv;
  double* partial = y + nv;
  double* partial2 = partial + nblk;
  double* scal = partial2 + nblk;
  hipLaunchKernelGGL(k_fom_combine, dim3(4096), dim3(256), 0, st, t, S, Q, th, A_diag, A_cpl, Amu_d, Amu_c, Minv);
  LRBMS_LAUNCH_CHECK(ctx);
  LRBMS_HIP_CHECK(ctx, hipMemsetAsync(x, 0, sizeof(double) * nv, st));
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(r, b, sizeof(double) * nv, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_fom_cg_update, dim3(nblk), dim3(256), 0, st, ne, Minv, scal, 1, x, r, p0, y, z, partial, partial2);
  hipLaunchKernelGGL(k_fom_cg_reduce, dim3(1), dim3(1024), 0, st, nblk, partial, partial2, scal, 0);
  LRBMS_LAUNCH_CHECK(ctx);
  double host[8];
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host, scal, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const double rr0 = host[4];
  if (rr0 == 0.0) {
    if (info) { info[0] = 0; info[1] = 0.0; }
    return LRBMS_OK;
  }
  double rel = 1.0;
  int it = 0;
  double* pin = p0;
  double* pout = p1;
  const int check_every = 25;
  while (it < max_iter) {
    for (int k = 0; k < check_every && it < max_iter; ++k, ++it) {
      hipLaunchKernelGGL(k_fom_cg_matvec, dim3(nblk), dim3(256), 0, st, t, S, ctx->nbr, Amu_d, Amu_c, z, pin, scal, it == 0 ? 1 : 0, pout,
                         y, partial);
      hipLaunchKernelGGL(k_fom_cg_reduce, dim3(1), dim3(1024), 0, st, nblk, partial, (const double*)nullptr, scal, 1);
      hipLaunchKernelGGL(k_fom_cg_update, dim3(nblk), dim3(256), 0, st, ne, Minv, scal, 0, x, r, pout, y, z, partial, partial2);
      hipLaunchKernelGGL(k_fom_cg_reduce, dim3(1), dim3(1024), 0, st, nblk, partial, partial2, scal, 2);
      double* tmp = pin;
      pin = pout;
      pout = tmp;
    }
    LRBMS_LAUNCH_CHECK(ctx);
    LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host, scal, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
    LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
    rel = sqrt(host[4] / rr0);
    if (!(rel == rel)) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "fom_solve: NaN residual (operator not SPD?)");
    if (rel <= rtol) break;
  }
  if (info) { info[0] = it; info[1] = rel; }
  if (rel > rtol) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "fom_solve: CG did not reach rtol");
  return LRBMS_OK;
}
