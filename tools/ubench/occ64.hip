// Microbenchmark (gfx950): producer / consumer workgroups with one barrier per chunk, as k_f1: 4 "producer" waves run a
// latency-bound sequence (dependent LDS round trips + a few FMAs), 4 "consumer" waves issue NM independent f64 MFMAs.
// Question: with f64 MFMA streams monopolising a SIMD's issue (overlap64.hip), does a SECOND workgroup per CU fill the
// gaps, i.e. is the time per chunk ~ (MFMA issue + producer issue) instead of (MFMA issue + producer latency)?
// usage: occ64 <lds_kb_per_wg> <blocks> : lds_kb >= 81 forces one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NACC>
__global__ __launch_bounds__(512) void k(double* out, int chunks, int pdepth, double a, double b) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double res = 0.0;
  if (wave >= 4) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double av = a + lane, bv = b;
    for (int c = 0; c < chunks; ++c) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int r = 0; r < NM / NACC; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i) res += acc[i][0] + acc[i][3];
  } else {
    double* my = lds + wave * 64;
    double x = lane;
    for (int c = 0; c < chunks; ++c) {
      for (int d = 0; d < pdepth; ++d) {          // dependent LDS round trips (write, read a neighbour's value, 2 FMAs)
        my[lane] = x;
        const double y = my[(lane + 1) & 63];
        x = __builtin_fma(y, a, b) * 0.5 + x * 0.25;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    res = x;
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = res;
}

template <typename K>
float run(K kern, int blocks, size_t lds, double* out, int chunks, int pdepth) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, out, chunks, pdepth, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, out, chunks, pdepth, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

int main(int argc, char** argv) {
  double* out;
  hipMalloc(&out, sizeof(double) * 512 * 2048);
  const int chunks = 256;
  printf("per chunk: consumers NM f64 MFMAs (NACC accumulators), producers `depth` dependent LDS round trips; cycles at 2.4 GHz\n");
  for (int depth : {0, 10, 20, 40}) {
    // one WG per CU (LDS 100 KB), 256 blocks: every CU works through `chunks` chunks
    const float t1 = run(k<63, 21>, 256, 100 * 1024, out, chunks, depth);
    // two WGs per CU (LDS 40 KB), 512 blocks, each with HALF the MFMAs per chunk (the column groups split over two workgroups):
    const float t2 = run(k<32, 16>, 512, 40 * 1024, out, chunks, depth);
    // two WGs per CU, full MFMA count each (twice the total work)
    const float t3 = run(k<63, 21>, 512, 40 * 1024, out, chunks, depth);
    printf("depth %2d: 1 WG/CU x 63 MFMA: %6.0f cyc/chunk | 2 WG/CU x 32 MFMA: %6.0f cyc/chunk (same total MFMA work) | 2 WG/CU x 63 MFMA: %6.0f\n",
           depth, t1 * 1e-3 * 2.4e9 / chunks, t2 * 1e-3 * 2.4e9 / chunks, t3 * 1e-3 * 2.4e9 / chunks);
  }
  return 0;
}
