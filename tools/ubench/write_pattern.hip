// Microbenchmark: HBM write bandwidth of 1 GiB as a function of HOW the workgroups cover it.
//   mode 0: grid-stride (every workgroup marches through the whole array, torch.fill_ style)
//   mode 1: one contiguous chunk of `chunk` bytes per workgroup (k_thin_rt style: 51 KB .. 256 KB per workgroup)
//   mode 2: like 1, but each workgroup's chunk is written as `parts` separate pieces `stride` bytes apart (4 blocks of a subdomain)
// build: hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_stride(double2* out, long n2, double v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) out[i] = make_double2(v, v);
}
__global__ __launch_bounds__(256) void k_chunk(double2* out, long chunk2, double v) {
  double2* o = out + (long)blockIdx.x * chunk2;
  for (long i = threadIdx.x; i < chunk2; i += 256) o[i] = make_double2(v, v);
}
// each thread computes a little before each store (dependent FMAs) to mimic a producer
__global__ __launch_bounds__(256) void k_chunk_work(double2* out, long chunk2, double v, int work) {
  double2* o = out + (long)blockIdx.x * chunk2;
  for (long i = threadIdx.x; i < chunk2; i += 256) {
    double a = v + i;
    for (int w = 0; w < work; ++w) a = __builtin_fma(a, 1.0000001, 0.5);
    o[i] = make_double2(a, v);
  }
}
// k_thin_rt's pattern: workgroup (side, s) writes blocks 1 + side and 5 + side (51 200 B each) of subdomain s in TWO arrays
// [S][9][6400] doubles; interleaved = one 16-byte store into each of the four blocks per item, sequential = block after block
__global__ __launch_bounds__(256) void k_thin(double2* A, double2* B, int interleaved, double v) {
  const int side = blockIdx.x, s = blockIdx.y;
  double2* dst[4] = {A + ((long)s * 9 + 5 + side) * 3200, B + ((long)s * 9 + 5 + side) * 3200, A + ((long)s * 9 + 1 + side) * 3200,
                     B + ((long)s * 9 + 1 + side) * 3200};
  if (interleaved) {
    for (int i = threadIdx.x; i < 3200; i += 256)
      for (int k = 0; k < 4; ++k) dst[k][i] = make_double2(v, v);
  } else {
    for (int k = 0; k < 4; ++k)
      for (int i = threadIdx.x; i < 3200; i += 256) dst[k][i] = make_double2(v, v);
  }
}
int main() {
  const long bytes = 1L << 30;
  double2* d;
  hipMalloc(&d, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.3f ms  %5.2f TB/s\n", name, ms / 10, bytes / (ms / 10) / 1e9);
  };
  const long n2 = bytes / 16;
  for (int g : {1024, 4096, 16384})
    time(("grid-stride, " + std::to_string(g) + " workgroups").c_str(), [&] { hipLaunchKernelGGL(k_stride, dim3(g), dim3(256), 0, 0, d, n2, 1.5); });
  for (long chunk : {16L << 10, 51200L, 64L << 10, 256L << 10, 1L << 20}) {
    const long nb = bytes / chunk;
    time(("chunk " + std::to_string(chunk) + " B per workgroup, " + std::to_string(nb) + " wgs").c_str(),
         [&] { hipLaunchKernelGGL(k_chunk, dim3(nb), dim3(256), 0, 0, d, chunk / 16, 1.5); });
  }
  for (int work : {8, 32, 128})
    time(("chunk 256 KB + " + std::to_string(work) + " dependent FMAs per store").c_str(),
         [&] { hipLaunchKernelGGL(k_chunk_work, dim3(bytes / (256L << 10)), dim3(256), 0, 0, d, (256L << 10) / 16, 1.5, work); });
  {
    double2 *A, *B;
    const long per = 1024L * 9 * 3200 * 16;
    hipMalloc(&A, per);
    hipMalloc(&B, per);
    const double gb = 4096.0 * 4 * 51200;
    for (int mode : {1, 0}) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_thin, dim3(4, 1024), dim3(256), 0, 0, A, B, mode, 1.5);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_thin, dim3(4, 1024), dim3(256), 0, 0, A, B, mode, 1.5);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("k_thin_rt pattern (4 blocks / workgroup), %-12s %7.3f ms  %5.2f TB/s\n", mode ? "interleaved" : "sequential", ms / 10, gb / (ms / 10) / 1e9);
    }
  }
  return 0;
}
