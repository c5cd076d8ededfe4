// Batched fp64 Gram / projection GEMM on the CDNA4 matrix cores:
//   G[b] (Mx x My) = alpha * X[b]^T diag(rowscale) Y[b],   X [K][ldx], Y [K][ldy] row-major ("K-major" operands).
//
// This is the dense half of the projections P1 / P2: V^T (A V), Wt^T (E Wt), Rt^T (B Rt), ...
// with K = n (DG DoFs), n_rt or n_T of ONE subdomain, so every problem is small (K <= a few hundred) and the batch
// (number of subdomains) supplies the parallelism.
//
// Mapping: v_mfma_f64_16x16x4_f64.  A operand lane l holds A[i = l&15][k = l>>4] = X[k][i], B operand lane l holds
// B[k = l>>4][j = l&15] = Y[k][j]: both are read from K-major LDS tiles with 16 consecutive lanes on 16 consecutive
// doubles.  The tile rows are padded to BM+16 doubles so that the four k-rows a wave reads at once fall into
// different halves of the 64 LDS banks (ds_read_b64 services 32 lanes per cycle).  C/D: lane l holds rows
// (l>>4) + 4 r, r = 0..3, of column l&15.
//
// Workgroup = 256 threads = 4 waves, output tile 64 x 64 (wave tile 32 x 32 = 2 x 2 MFMA tiles),
// K staged in chunks of 16 rows.  fp64 MFMA peak on gfx950 = 78.6 TFLOP/s (= one 16x16x4 per 64 cycles per SIMD).
#include "lrbms_dev.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 16, PAD = 16;

__global__ __launch_bounds__(256) void k_gemm_tn(int K, int Mx, int My, const double* __restrict__ X, long sx, int ldx,
                                                 const double* __restrict__ Y, long sy, int ldy, double* __restrict__ G,
                                                 long sg, int ldg, const double* __restrict__ rowscale, double alpha) {
  __shared__ double Xs[BK][BM + PAD];
  __shared__ double Ys[BK][BN + PAD];
  const int bz = blockIdx.z;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  X += (long)bz * sx;
  Y += (long)bz * sy;
  G += (long)bz * sg;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lk = lane >> 4;

  d4 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  // staging: thread -> (row = tid / 16, 4 consecutive columns)
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const int kr = k0 + srow;
    double xv[4] = {0, 0, 0, 0}, yv[4] = {0, 0, 0, 0};
    if (kr < K) {
      const double sc = rowscale ? rowscale[kr] : 1.0;
      const double* xp = X + (long)kr * ldx + m0 + scol;
      const double* yp = Y + (long)kr * ldy + n0 + scol;
      for (int c = 0; c < 4; ++c) {
        if (m0 + scol + c < Mx) xv[c] = xp[c];
        if (n0 + scol + c < My) yv[c] = sc * yp[c];
      }
    }
    __syncthreads();  // previous chunk fully consumed
    for (int c = 0; c < 4; ++c) {
      Xs[srow][scol + c] = xv[c];
      Ys[srow][scol + c] = yv[c];
    }
    __syncthreads();
    for (int kk = 0; kk < BK; kk += 4) {
      double a[2], b[2];
      for (int mi = 0; mi < 2; ++mi) a[mi] = Xs[kk + lk][wr * 32 + mi * 16 + li];
      for (int ni = 0; ni < 2; ++ni) b[ni] = Ys[kk + lk][wc * 32 + ni * 16 + li];
      for (int mi = 0; mi < 2; ++mi)
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  }
  for (int mi = 0; mi < 2; ++mi)
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wc * 32 + ni * 16 + li;
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + mi * 16 + lk + 4 * r;
        if (row < Mx && col < My) G[(long)row * ldg + col] = alpha * acc[mi][ni][r];
      }
    }
}

}  // namespace

int launch_gemm_tn(lrbms_ctx* ctx, int batch, int K, int Mx, int My, const double* X, long sx, int ldx, const double* Y,
                   long sy, int ldy, double* G, long sg, int ldg, const double* rowscale, double alpha, hipStream_t st) {
  if (batch <= 0 || K <= 0 || Mx <= 0 || My <= 0) return lrbms_fail(ctx, LRBMS_E_INVALID, "gemm_tn: empty problem");
  if (ldx < Mx || ldy < My || ldg < My) return lrbms_fail(ctx, LRBMS_E_INVALID, "gemm_tn: leading dimension too small");
  dim3 grid((My + BN - 1) / BN, (Mx + BM - 1) / BM, batch);
  if (grid.z > 65535) return lrbms_fail(ctx, LRBMS_E_INVALID, "gemm_tn: batch > 65535");
  hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, st, K, Mx, My, X, sx, ldx, Y, sy, ldy, G, sg, ldg, rowscale, alpha);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}
