// Micro-benchmark 4: v_mfma_f32_32x32x16_f16 stream (the K-split z-slide pattern: per pair 2 x ds_read_b128 + 3 MFMAs, 14 weight
// pairs in registers) with NF independent filler VALU instructions per pair placed between the MFMAs, one wave per SIMD
// (256 threads) or two (512 threads, both running the loop).  How many fillers hide in the MFMA shadows?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

template <int WAVES, int NF, int TRANS, int SPLIT>  // SPLIT: fillers split evenly behind each of the 3 MFMAs instead of after the triple
__global__ void __launch_bounds__(64 * WAVES, 1) k(float* out, const u32x4* wsrc, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  u32x4 w1[14], w2[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) { w1[j] = wsrc[(j * 2) * 64 + lane]; w2[j] = wsrc[(j * 2 + 1) * 64 + lane]; }
  const int base = (lane & 31) * 144 + (lane >> 5) * 16 + (wave & 3) * 64 + 512;
  f32x16 A[2] = {}, B[2] = {};
  constexpr int PD = 3;
  u32x4 fa[PD + 1][2];
  float f[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) f[i] = seed + i;
  constexpr int NP = 27;
  int fq = 0;
  auto filler = [&](int n) {
#pragma unroll
    for (int q = 0; q < n; ++q) {
      if (q < TRANS) f[(fq + q) % 12] = __builtin_amdgcn_exp2f(f[(fq + q) % 12]);
      else f[(fq + q) % 12] = f[(fq + q) % 12] * 1.0001f + 0.5f;
    }
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < PD; ++i) { const char* p = lds + base + i * 144 * 8; fa[i][0] = *(const u32x4*)p; fa[i][1] = *(const u32x4*)(p + 32); }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i + PD < NP) {
        const char* p = lds + base + ((i + PD) % 24) * 144 * 8;
        fa[(i + PD) % (PD + 1)][0] = *(const u32x4*)p;
        fa[(i + PD) % (PD + 1)][1] = *(const u32x4*)(p + 32);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int j = i % 14, t = i >= 14;
      A[t] = MF(fa[i % (PD + 1)][0], w1[j], A[t]);
      if (SPLIT) { filler(NF / 3); __builtin_amdgcn_sched_barrier(0); }
      B[t] = MF(fa[i % (PD + 1)][0], w2[j], B[t]);
      if (SPLIT) { filler(NF / 3); __builtin_amdgcn_sched_barrier(0); }
      B[t] = MF(fa[i % (PD + 1)][1], w1[j], B[t]);
      filler(SPLIT ? NF - 2 * (NF / 3) : NF);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += A[0][r] + B[0][r] + A[1][r] + B[1][r];
  for (int i = 0; i < 12; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES, int NF, int TRANS, int SPLIT>
void run(float* out, const u32x4* w) {
  const int iters = 300;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<WAVES, NF, TRANS, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k<WAVES, NF, TRANS, SPLIT>), dim3(256), dim3(64 * WAVES), 100 * 1024, 0, out, w, 10, 1.f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WAVES, NF, TRANS, SPLIT>), dim3(256), dim3(64 * WAVES), 100 * 1024, 0, out, w, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD=%d fillers/pair=%2d (trans %d) split=%d  %.3f ms  %.1f ns per pair per wave\n", WAVES / 4, NF, TRANS, SPLIT, ms,
         ms * 1e6 / (27.0 * iters));
}
int main() {
  float* out; (void)hipMalloc(&out, 4 * 512 * 1024);
  u32x4* w; (void)hipMalloc(&w, 64 * 28 * 16); (void)hipMemset(w, 0x3c, 64 * 28 * 16);
  run<4, 0, 0, 1>(out, w);
  run<4, 6, 0, 1>(out, w);
  run<4, 9, 0, 1>(out, w);
  run<4, 12, 0, 1>(out, w);
  run<4, 15, 0, 1>(out, w);
  run<4, 18, 0, 1>(out, w);
  run<4, 15, 0, 0>(out, w);
  run<4, 15, 3, 1>(out, w);
  run<8, 0, 0, 1>(out, w);
  run<8, 6, 0, 1>(out, w);
  run<8, 9, 0, 1>(out, w);
  return 0;
}
