// Microbenchmark (gfx950): what does it cost a workgroup of 8 waves to issue its vector-memory loads in ONE BURST (all waves at
// the same time, as the staging phase of k_f1u does: the loads queue in the CU's address unit and the issuing waves stall) against
// issuing the same loads ONE AT A TIME BETWEEN its f64 MFMAs (the address unit works while the matrix pipe does)?
// Every wave: per chunk NL 8-byte-per-lane loads of L2-resident rows + NM v_mfma_f64_16x16x4_f64; the loaded values are consumed
// one chunk later (register prefetch).  build: hipcc --offload-arch=gfx950 -O3 -o interleave64 interleave64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline double gload(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// MODE 0: no loads; 1: burst at the start of the chunk; 2: one load after every (NM / NL)-th MFMA; 3: burst in the middle of the MFMAs
template <int MODE, int NL, int NM>
__global__ __launch_bounds__(512) void k(double* out, const double* in, int chunks, double a, double b) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NA = 21;
  d4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = (d4){0, 0, 0, 0};
  double av = a + lane, bv = b;
  const double* p = in + (size_t)blockIdx.x * 65536 + wave * 4096 + lane;
  double v[NL > 0 ? NL : 1];
#pragma unroll
  for (int r = 0; r < NL; ++r) v[r] = 0.0;
  double s = 0.0;
  for (int c = 0; c < chunks; ++c) {
    const double* pc = p + (c & 7) * 512;
    if (MODE != 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < NL; ++r) {
        asm volatile("" : "+v"(v[r]));
        s += v[r];
      }
      bv = b + s * 1e-30;
    }
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < NL; ++r) v[r] = gload(pc + r * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int STEP = NL > 0 ? NM / NL : NM;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      acc[i % NA] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i % NA], 0, 0, 0);
      if (MODE == 2 && NL > 0 && i % STEP == STEP - 1 && i / STEP < NL) {
        __builtin_amdgcn_sched_barrier(0);
        v[i / STEP] = gload(pc + (i / STEP) * 64);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 3 && i == NM / 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NL; ++r) v[r] = gload(pc + r * 64);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double res = s;
#pragma unroll
  for (int r = 0; r < NL; ++r) {
    asm volatile("" : "+v"(v[r]));
    res += v[r];
  }
  for (int i = 0; i < NA; ++i) res += acc[i][0] + acc[i][3];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = res;
}

template <typename K>
float run(K kern, int blocks, double* out, const double* in, int chunks) {
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, in, chunks, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, in, chunks, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

template <int NL, int NM>
void sweep(int blocks, double* out, const double* in, int chunks) {
  const float t0 = run(k<0, NL, NM>, blocks, out, in, chunks), t1 = run(k<1, NL, NM>, blocks, out, in, chunks),
              t2 = run(k<2, NL, NM>, blocks, out, in, chunks), t3 = run(k<3, NL, NM>, blocks, out, in, chunks);
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / chunks; };
  printf("NL %2d loads + NM %2d MFMAs per wave and chunk: cycles per chunk  no loads %6.0f | burst at start %6.0f (+%5.0f) | "
         "interleaved %6.0f (+%5.0f) | burst mid-MFMA %6.0f (+%5.0f)\n",
         NL, NM, cyc(t0), cyc(t1), cyc(t1) - cyc(t0), cyc(t2), cyc(t2) - cyc(t0), cyc(t3), cyc(t3) - cyc(t0));
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  const int chunks = 512;
  double *out, *in;
  hipMalloc(&out, sizeof(double) * 512 * blocks);
  hipMalloc(&in, sizeof(double) * 65536 * (size_t)blocks);
  hipMemset(in, 0, sizeof(double) * 65536 * (size_t)blocks);
  sweep<4, 32>(blocks, out, in, chunks);
  sweep<8, 32>(blocks, out, in, chunks);
  sweep<12, 36>(blocks, out, in, chunks);
  sweep<16, 32>(blocks, out, in, chunks);
  sweep<16, 64>(blocks, out, in, chunks);
  return 0;
}
