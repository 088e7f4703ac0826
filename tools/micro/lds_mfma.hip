// Micro-benchmark: can ds_read_b128 A-fragment traffic hide behind v_mfma_f32_32x32x16_f16 on gfx950?
// Each wave runs "pairs": 2 x ds_read_b128 (prefetched PD pairs ahead) + 3 MFMAs, the z-slide conv's inner pattern.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_mfma tools/micro/lds_mfma.hip && tools/micro/lds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

template <int MODE, int PD>  // MODE 0: reads + MFMAs, 1: MFMAs only, 2: reads only
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  const char* base = lds + (lane & 31) * 144 + (lane >> 5) * 16 + wave * 64;  // conflict-free record addressing
  u32x4 w1 = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, w2 = w1;
  f32x16 A = {}, B = {};
  u32x4 fa[PD + 1][2];
  constexpr int NP = 27;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      if (MODE != 1) { fa[i][0] = *(const u32x4*)(base + i * 144 * 8); fa[i][1] = *(const u32x4*)(base + i * 144 * 8 + 32); }
      else { fa[i][0] = w1; fa[i][1] = w2; }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i + PD < NP) {
        if (MODE != 1) {
          fa[(i + PD) % (PD + 1)][0] = *(const u32x4*)(base + ((i + PD) % 24) * 144 * 8);
          fa[(i + PD) % (PD + 1)][1] = *(const u32x4*)(base + ((i + PD) % 24) * 144 * 8 + 32);
        } else { fa[(i + PD) % (PD + 1)][0] = w1; fa[(i + PD) % (PD + 1)][1] = w2; }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2) {
        A = MF(fa[i % (PD + 1)][0], w1, A);
        B = MF(fa[i % (PD + 1)][0], w2, B);
        B = MF(fa[i % (PD + 1)][1], w1, B);
      } else {
        asm volatile("" ::"v"(fa[i % (PD + 1)][0]), "v"(fa[i % (PD + 1)][1]));
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += A[r] + B[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int PD>
void run(const char* name, float* out, int waves) {
  const int iters = 400;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<MODE, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((k<MODE, PD>), dim3(256), dim3(64 * waves), 40 * 1024, 0, out, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, PD>), dim3(256), dim3(64 * waves), 40 * 1024, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double pairs = 27.0 * iters;
  printf("%-34s waves/WG=%d PD=%d  %.3f ms  %.1f ns per pair per wave (MFMA-only ideal 48.7 ns @1.97GHz)\n", name, waves, PD, ms, ms * 1e6 / pairs);
}
int main() {
  float* out; (void)hipMalloc(&out, 4 * 256 * 1024);
  run<1, 3>("MFMAs only", out, 4);
  run<2, 3>("LDS reads only", out, 4);
  run<0, 3>("reads + MFMAs", out, 4);
  run<0, 5>("reads + MFMAs", out, 4);
  run<1, 3>("MFMAs only", out, 8);
  run<2, 3>("LDS reads only", out, 8);
  run<0, 3>("reads + MFMAs (2 waves/SIMD)", out, 8);
  return 0;
}
