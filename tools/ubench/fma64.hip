// Microbenchmark: issue cost / latency of v_fma_f64 and v_mfma_f64_16x16x4_f64 on gfx950 (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k_fma(double* out, unsigned long long* cyc, int iters, double a, double b) {
  double x[CH];
  for (int i = 0; i < CH; ++i) x[i] = threadIdx.x * 1e-3 + i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < CH; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k_mfma(double* out, unsigned long long* cyc, int iters, double a, double b) {
  d4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = (d4){0, 0, 0, 0};
  double av = a + threadIdx.x, bv = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, int threads, int blocks, int iters, int ops_per_iter) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * threads * blocks); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s threads/blk %4d blocks %5d: %7.2f memtime-ticks per instr per wave, kernel %.3f ms\n", name, threads, blocks,
         (double)h / ((double)iters * ops_per_iter), ms);
  hipFree(out); hipFree(cyc);
}
int main() {
  const int it = 2000;
  run("fma64 1 chain, 1 wave/SIMD", k_fma<1>, 256, 256, it, 8);
  run("fma64 4 chains, 1 wave/SIMD", k_fma<4>, 256, 256, it, 32);
  run("fma64 8 chains, 1 wave/SIMD", k_fma<8>, 256, 256, it, 64);
  run("fma64 8 chains, 2 wave/SIMD", k_fma<8>, 512, 256, it, 64);
  run("fma64 8 chains, 4 wave/SIMD", k_fma<8>, 1024, 256, it, 64);
  run("mfma64 1 acc, 1 wave/SIMD", k_mfma<1>, 256, 256, it, 8);
  run("mfma64 4 acc, 1 wave/SIMD", k_mfma<4>, 256, 256, it, 32);
  run("mfma64 4 acc, 2 wave/SIMD", k_mfma<4>, 512, 256, it, 32);
  return 0;
}
