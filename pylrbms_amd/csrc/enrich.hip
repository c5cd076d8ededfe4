// Online enrichment: batched local corrector solves (SURVEY.md section 8f "next" #1).
//
// Reference: DuneDiscretization.solve_for_local_correction (discretize_elliptic_block_swipdg.py:227-316) assembles
// the SWIPDG operator of the neighbourhood N(ii) = {ii and its face neighbours} with an all-Dirichlet local boundary
// (:240-247, :794-795), the L2 functional of f as right-hand side (:263-268), solves with ISTL and keeps the part of
// the solution that lives on subdomain ii (:303-316).  online_enrichment.py:49-50 does this for one marked subdomain
// after the other.
//
// Here every marked subdomain is one workgroup that runs its whole preconditioned CG without leaving the CU:
//  * the neighbourhood operator is never assembled: it is the block-ELL data the discretization already holds
//    (A_diag, A_cpl) plus the Dirichlet correction blocks D_corr on the outer coupling faces;
//  * each thread owns one element (EPT == 1, the 384-DoF subdomains of the benchmark configs): its four theta-weighted
//    3x3 blocks, its inverse diagonal block, x and r stay in registers for all iterations;
//  * the search direction p (5 n doubles) lives in LDS, where the neighbouring elements read it;
//  * the three dot products of an iteration are fixed-order wave + workgroup reductions (deterministic), three
//    barriers per iteration, no global synchronisation and no HBM traffic inside the loop.
// Larger templates fall back to EPT > 1 elements per thread with the blocks re-read (L2) every iteration.
#include "lrbms_dev.h"

namespace {

struct QVecE { double v[8]; };

// theta-weighted block `bidx` (0 = diagonal, 1 + f = face f) of element e of neighbourhood member `slot` (subdomain kk)
// of marked subdomain ii.  Returns the LDS offset of the source element's first DoF, or -1 if the block is absent.
__device__ inline int hood_block(const Tmpl& t, int S, const int* __restrict__ nbr, int Q, const QVecE& th,
                                 const double* __restrict__ A_diag, const double* __restrict__ A_cpl,
                                 const double* __restrict__ D_corr, int ii, int slot, int kk, int e, int bidx, double Hb[9]) {
  const int nT = t.nT, n = t.n;
  const double* base;
  long qs;
  int src;
  if (bidx == 0) {
    base = A_diag + ((long)kk * nT + e) * 36;
    qs = (long)S * nT * 36;
    src = slot * n + 3 * e;
  } else {
    const int f = bidx - 1;
    const int nb = t.nb_elem[e * 3 + f];
    if (nb >= 0) {
      base = A_diag + ((long)kk * nT + e) * 36 + bidx * 9;
      qs = (long)S * nT * 36;
      src = slot * n + 3 * nb;
    } else {
      const int side = -1 - nb, sl2 = side_to_slot(side);
      int srcslot;
      if (slot == 2) {
        if (nbr[ii * 5 + sl2] < 0) return -1;   // physical boundary: already in A_diag
        srcslot = sl2;
      } else {
        if (sl2 != 4 - slot) return -1;         // outer boundary of the neighbourhood: Dirichlet, no coupling
        srcslot = 2;
      }
      base = A_cpl + (((long)kk * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + f]) * 9;
      qs = (long)S * 4 * t.ncf * 9;
      src = srcslot * n + 3 * t.nb_elem_out[e * 3 + f];
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) Hb[i] = 0.0;
  for (int q = 0; q < Q; ++q) {
    const double w = th.v[q];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hb[i] += w * base[q * qs + i];
  }
  if (bidx == 0 && slot != 2) {
    const long dqs = (long)S * 4 * t.ncf * 9;
    for (int f = 0; f < 3; ++f) {
      const int nb = t.nb_elem[e * 3 + f];
      if (nb >= 0) continue;
      const int side = -1 - nb, sl2 = side_to_slot(side);
      if (sl2 == 4 - slot || nbr[kk * 5 + sl2] < 0) continue;
      const double* dc = D_corr + (((long)kk * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + f]) * 9;
      for (int q = 0; q < Q; ++q) {
        const double w = th.v[q];
#pragma unroll
        for (int i = 0; i < 9; ++i) Hb[i] += w * dc[q * dqs + i];
      }
    }
  }
  return src;
}

__device__ inline void inv3(const double a[9], double o[9]) {
  const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
  const double id = 1.0 / (a[0] * c0 + a[1] * c1 + a[2] * c2);
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

template <int NW>
__device__ inline double block_sum(double v, double* red, int lane, int wave) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += red[w];
  return s;
}

template <int NW>
__device__ inline void block_sum2(double& a, double& b, double* red, int lane, int wave) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  if (lane == 0) {
    red[wave] = a;
    red[NW + wave] = b;
  }
  __syncthreads();
  double sa = 0.0, sb = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    sa += red[w];
    sb += red[NW + w];
  }
  a = sa;
  b = sb;
}

template <int EPT, int BT>
__global__ __launch_bounds__(BT) void k_hood_pcg(Tmpl t, int S, const int* __restrict__ nbr, int Q, QVecE th,
                                                 const int* __restrict__ marked, const double* __restrict__ A_diag,
                                                 const double* __restrict__ A_cpl, const double* __restrict__ D_corr,
                                                 const double* __restrict__ bvec, double* __restrict__ corr,
                                                 double rtol2, int max_iter, double* __restrict__ info) {
  extern __shared__ double lds[];
  constexpr int NW = BT / 64;
  constexpr bool REG = EPT == 1;
  const int n = t.n, nT = t.nT, nel = 5 * nT;
  double* P = lds;               // [5][n] search direction
  double* redA = lds + 5 * n;    // [NW]
  double* redB = redA + NW;      // [2][NW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ii = marked[blockIdx.x];

  int slot[EPT], kk[EPT], el[EPT];
  bool act[EPT];
  double x[EPT][3], r[EPT][3], p[EPT][3], Mi[EPT][9];
  double H[REG ? 4 : 1][9];
  int src[REG ? 4 : 1];

  for (int i = tid; i < 5 * n; i += BT) P[i] = 0.0;
  __syncthreads();

  double rz = 0.0, rr = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int j = tid + k * BT;
    slot[k] = j < nel ? j / nT : 0;
    el[k] = j < nel ? j - slot[k] * nT : 0;
    kk[k] = j < nel ? nbr[ii * 5 + slot[k]] : -1;
    act[k] = kk[k] >= 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) x[k][i] = r[k][i] = p[k][i] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) Mi[k][i] = 0.0;
    if constexpr (REG) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        src[b] = -1;
#pragma unroll
        for (int i = 0; i < 9; ++i) H[b][i] = 0.0;
      }
    }
    if (act[k]) {
      double H0[9];
      if constexpr (REG) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          double Hb[9];
          const int sb = hood_block(t, S, nbr, Q, th, A_diag, A_cpl, D_corr, ii, slot[k], kk[k], el[k], b, Hb);
          src[b] = sb;
          if (sb >= 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) H[b][i] = Hb[i];
          }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) H0[i] = H[0][i];
      } else {
        hood_block(t, S, nbr, Q, th, A_diag, A_cpl, D_corr, ii, slot[k], kk[k], el[k], 0, H0);
      }
      inv3(H0, Mi[k]);
      const double* bs = bvec + (long)kk[k] * n + 3 * el[k];
#pragma unroll
      for (int i = 0; i < 3; ++i) r[k][i] = bs[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        p[k][i] = Mi[k][i * 3] * r[k][0] + Mi[k][i * 3 + 1] * r[k][1] + Mi[k][i * 3 + 2] * r[k][2];   // z = M^-1 r
        rz += r[k][i] * p[k][i];
        rr += r[k][i] * r[k][i];
        P[slot[k] * n + 3 * el[k] + i] = p[k][i];
      }
    }
  }
  block_sum2<NW>(rz, rr, redB, lane, wave);
  const double rr0 = rr;
  int iters = 0;
  bool bad = false;
  if (rr0 > 0.0) {
    for (int it = 0; it < max_iter; ++it) {
      __syncthreads();                                   // P complete
      double Ap[EPT][3];
      double pAp = 0.0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
#pragma unroll
        for (int i = 0; i < 3; ++i) Ap[k][i] = 0.0;
        if constexpr (REG) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int sb = src[b] >= 0 ? src[b] : 0;     // absent blocks are zero: read any valid address
            const double p0 = P[sb], p1 = P[sb + 1], p2 = P[sb + 2];
#pragma unroll
            for (int i = 0; i < 3; ++i) Ap[k][i] += H[b][i * 3] * p0 + H[b][i * 3 + 1] * p1 + H[b][i * 3 + 2] * p2;
          }
        } else if (act[k]) {
          for (int b = 0; b < 4; ++b) {
            double Hb[9];
            const int sb = hood_block(t, S, nbr, Q, th, A_diag, A_cpl, D_corr, ii, slot[k], kk[k], el[k], b, Hb);
            if (sb < 0) continue;
            const double p0 = P[sb], p1 = P[sb + 1], p2 = P[sb + 2];
#pragma unroll
            for (int i = 0; i < 3; ++i) Ap[k][i] += Hb[i * 3] * p0 + Hb[i * 3 + 1] * p1 + Hb[i * 3 + 2] * p2;
          }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) pAp += p[k][i] * Ap[k][i];
      }
      pAp = block_sum<NW>(pAp, redA, lane, wave);
      iters = it + 1;
      if (!(pAp > 0.0)) {
        bad = true;
        break;
      }
      const double alpha = rz / pAp;
      double rz_new = 0.0;
      rr = 0.0;
      double z[EPT][3];
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          x[k][i] += alpha * p[k][i];
          r[k][i] -= alpha * Ap[k][i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          z[k][i] = Mi[k][i * 3] * r[k][0] + Mi[k][i * 3 + 1] * r[k][1] + Mi[k][i * 3 + 2] * r[k][2];
          rz_new += r[k][i] * z[k][i];
          rr += r[k][i] * r[k][i];
        }
      }
      block_sum2<NW>(rz_new, rr, redB, lane, wave);
      if (rr <= rtol2 * rr0) break;
      const double beta = rz_new / rz;
      rz = rz_new;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        if (act[k]) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            p[k][i] = z[k][i] + beta * p[k][i];
            P[slot[k] * n + 3 * el[k] + i] = p[k][i];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    if (act[k] && slot[k] == 2) {
      double* out = corr + (long)blockIdx.x * n + 3 * el[k];
#pragma unroll
      for (int i = 0; i < 3; ++i) out[i] = x[k][i];
    }
  }
  if (tid == 0) {
    info[2 * blockIdx.x] = (double)iters;
    info[2 * blockIdx.x + 1] = bad ? -1.0 : (rr0 > 0.0 ? sqrt(rr / rr0) : 0.0);
  }
}

template <int EPT, int BT>
void launch_hood(lrbms_ctx* ctx, int Q, const QVecE& th, int nmark, const int* marked_dev, const double* A_diag,
                 const double* A_cpl, const double* D_corr, const double* b, double* corr, double rtol, int max_iter,
                 double* info_dev, hipStream_t st) {
  const size_t lds = sizeof(double) * (5 * (size_t)ctx->t.n + 3 * (BT / 64));
  hipLaunchKernelGGL((k_hood_pcg<EPT, BT>), dim3(nmark), dim3(BT), lds, st, ctx->t, ctx->S, ctx->nbr, Q, th, marked_dev, A_diag,
                     A_cpl, D_corr, b, corr, rtol * rtol, max_iter, info_dev);
}

}  // namespace

int64_t local_correction_work_size(lrbms_ctx* ctx, int nmark) {
  (void)ctx;
  return 3 * (int64_t)nmark + 8;
}

int launch_local_correction(lrbms_ctx* ctx, int Q, const double* theta, int nmark, const int32_t* marked, const double* A_diag,
                            const double* A_cpl, const double* D_corr, const double* b, double* work, double* corr, double rtol,
                            int max_iter, double* info, hipStream_t st) {
  if (nmark < 1 || Q < 1 || Q > 8 || max_iter < 1 || !(rtol > 0.0))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "local_correction_solve: bad nmark / Q / max_iter / rtol");
  for (int m = 0; m < nmark; ++m) {
    if (marked[m] < 0 || marked[m] >= ctx->S) return lrbms_fail(ctx, LRBMS_E_INVALID, "local_correction_solve: marked index out of range");
    // the operator blocks of the WHOLE neighbourhood must be on this rank (a sharded discretization runs its corrector
    // problems on a context whose local set is its local + halo subdomains)
    for (int slot = 0; slot < 5; ++slot)
      if (ctx->nbr_host[(size_t)marked[m] * 5 + slot] >= ctx->S)
        return lrbms_fail(ctx, LRBMS_E_INVALID, "local_correction_solve: the neighbourhood of a marked subdomain is not assembled on this rank");
  }
  const int nel = 5 * ctx->t.nT;
  if ((size_t)5 * ctx->t.n * sizeof(double) + 1024 > 160 * 1024)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "local_correction_solve: neighbourhood does not fit in LDS");
  QVecE th;
  for (int q = 0; q < 8; ++q) th.v[q] = q < Q ? theta[q] : 0.0;
  double* info_dev = work;
  int* marked_dev = reinterpret_cast<int*>(work + 2 * (size_t)nmark);
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(marked_dev, marked, sizeof(int) * (size_t)nmark, hipMemcpyHostToDevice, st));
  if (nel <= 640)
    launch_hood<1, 640>(ctx, Q, th, nmark, marked_dev, A_diag, A_cpl, D_corr, b, corr, rtol, max_iter, info_dev, st);
  else if (nel <= 1024)
    launch_hood<1, 1024>(ctx, Q, th, nmark, marked_dev, A_diag, A_cpl, D_corr, b, corr, rtol, max_iter, info_dev, st);
  else if (nel <= 2048)
    launch_hood<2, 1024>(ctx, Q, th, nmark, marked_dev, A_diag, A_cpl, D_corr, b, corr, rtol, max_iter, info_dev, st);
  else if (nel <= 4096)
    launch_hood<4, 1024>(ctx, Q, th, nmark, marked_dev, A_diag, A_cpl, D_corr, b, corr, rtol, max_iter, info_dev, st);
  else if (nel <= 8192)
    launch_hood<8, 1024>(ctx, Q, th, nmark, marked_dev, A_diag, A_cpl, D_corr, b, corr, rtol, max_iter, info_dev, st);
  else
    return lrbms_fail(ctx, LRBMS_E_INVALID, "local_correction_solve: template too large");
  LRBMS_LAUNCH_CHECK(ctx);
  std::vector<double> host(2 * (size_t)nmark);
  LRBMS_HIP_CHECK(ctx, hipMemcpyAsync(host.data(), info_dev, sizeof(double) * 2 * (size_t)nmark, hipMemcpyDeviceToHost, st));
  LRBMS_HIP_CHECK(ctx, hipStreamSynchronize(st));
  bool ok = true;
  for (int m = 0; m < nmark; ++m) {
    if (info) {
      info[2 * m] = host[2 * m];
      info[2 * m + 1] = host[2 * m + 1];
    }
    const double rel = host[2 * m + 1];
    if (!(rel >= 0.0) || rel > rtol) ok = false;
  }
  if (!ok) return lrbms_fail(ctx, LRBMS_E_NOT_CONVERGED, "local_correction_solve: CG did not reach rtol (or the operator is not SPD)");
  return LRBMS_OK;
}
