// Microbenchmark (gfx950): does a wave's stream of v_mfma_f64_16x16x4_f64 overlap with another wave's work on the SAME
// SIMD?  One workgroup of 8 waves per CU; hardware wave w runs on SIMD w % 4, so waves w and w + 4 share a SIMD.
// Waves 4-7 ("M") issue independent f64 MFMAs; waves 0-3 ("P") run one of: f64 FMA chains, int32 VALU chains, LDS
// write+read, L2-resident global loads, v_readlane broadcasts.  Times: M alone, P alone, both.  If they overlap,
// both ~ max; if they serialise, both ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

enum { P_NONE = 0, P_FMA64 = 1, P_INT = 2, P_LDS = 3, P_GLOAD = 4, P_READLANE = 5, P_MFMA = 6 };

template <int PMODE, bool RUN_M>
__global__ __launch_bounds__(512) void k(double* out, const double* in, int iters_m, int iters_p, double a, double b) {
  __shared__ double lds[4 * 64 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double res = 0.0;
  if (wave >= 4) {
    if (RUN_M) {
      d4 acc[21];
#pragma unroll
      for (int i = 0; i < 21; ++i) acc[i] = (d4){0, 0, 0, 0};
      double av = a + lane, bv = b;
      for (int it = 0; it < iters_m; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int i = 0; i < 21; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 21; ++i) res += acc[i][0] + acc[i][3];
    }
  } else {
    if (PMODE == P_FMA64) {
      double x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane * 1e-3 + i;
      for (int it = 0; it < iters_p; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
      }
      for (int i = 0; i < 8; ++i) res += x[i];
    } else if (PMODE == P_INT) {
      unsigned x[8];
      for (int i = 0; i < 8; ++i) x[i] = lane + i;
      for (int it = 0; it < iters_p; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = x[i] * 1664525u + 1013904223u;
      }
      for (int i = 0; i < 8; ++i) res += x[i];
    } else if (PMODE == P_LDS) {
      double* my = lds + wave * 64 * 8;
      double x = lane;
      for (int it = 0; it < iters_p; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) my[r * 64 + lane] = x + r;
        double s = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += my[r * 64 + ((lane + 1) & 63)];
        x = s * 1e-3;
      }
      res = x;
    } else if (PMODE == P_GLOAD) {
      const double* p = in + (size_t)blockIdx.x * 4096 + wave * 1024 + lane;
      double s = 0;
      for (int it = 0; it < iters_p; ++it) {
        double v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = __builtin_nontemporal_load(p + ((it * 8 + r) & 15) * 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) s += v[r];
      }
      res = s;
    } else if (PMODE == P_READLANE) {
      double x = lane * 1.5;
      double s = 0;
      for (int it = 0; it < iters_p; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lo = __builtin_amdgcn_readlane(__double2loint(x), r), hi = __builtin_amdgcn_readlane(__double2hiint(x), r);
          s = __builtin_fma(__hiloint2double(hi, lo), a, s);
        }
        x = s * 1e-9 + lane;
      }
      res = s;
    } else if (PMODE == P_MFMA) {
      d4 acc[3];
      for (int i = 0; i < 3; ++i) acc[i] = (d4){0, 0, 0, 0};
      double av = a + lane, bv = b;
      for (int it = 0; it < iters_p; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 3; ++i) res += acc[i][0];
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = res;
}

template <typename K>
float run(K kern, int blocks, double* out, const double* in, int im, int ip) {
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, in, im, ip, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, in, im, ip, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

#define CASE(NAME, MODE, IP)                                                                             \
  {                                                                                                      \
    const float tp = run(k<MODE, false>, blocks, out, in, im, IP);                                       \
    const float tb = run(k<MODE, true>, blocks, out, in, im, IP);                                        \
    printf("%-34s P alone %7.3f ms | M alone %7.3f ms | both %7.3f ms | sum %7.3f  max %7.3f -> %s\n", NAME, tp, tm, tb, tp + tm, \
           tp > tm ? tp : tm, tb < 0.5 * (tp + tm + (tp > tm ? tp : tm)) ? "OVERLAP" : "SERIAL");      \
  }

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  double *out, *in;
  hipMalloc(&out, sizeof(double) * 512 * blocks);
  hipMalloc(&in, sizeof(double) * 4096 * blocks);
  hipMemset(in, 0, sizeof(double) * 4096 * blocks);
  const int im = 400;   // 400 x 63 MFMAs per wave
  const float tm = run(k<P_NONE, true>, blocks, out, in, im, 0);
  printf("blocks %d; M = %d x 63 v_mfma_f64_16x16x4 per wave: %.3f ms -> %.1f cycles per MFMA at 2.4 GHz\n", blocks, im, tm,
         tm * 1e-3 * 2.4e9 / (im * 63.0));
  CASE("f64 FMA (8 chains)", P_FMA64, 6000)
  CASE("int32 mad (8 chains)", P_INT, 6000)
  CASE("LDS write+read (8+8 per iter)", P_LDS, 3000)
  CASE("global loads (L2, 8 per iter)", P_GLOAD, 3000)
  CASE("v_readlane x2 + fma (16 per iter)", P_READLANE, 2500)
  CASE("f64 MFMA (3 acc: dependent chains)", P_MFMA, 1500)
  return 0;
}
