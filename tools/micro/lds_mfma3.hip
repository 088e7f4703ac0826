// Micro-benchmark 3: one wave per SIMD (256-thread workgroup, up to 512 registers), v_mfma_f32_16x16x32_f16, the pattern of the
// single-wave z-slide design: per (tile, tap) "pair" 2 x ds_read_b128 (A fragments x1, x2') + 3 MFMAs (16 cycles each) with 27
// weight-fragment pairs in registers (216), plus NF independent filler VALU instructions per pair (the support work that has to
// hide in the MFMA shadows).  How many fillers are free?  Does the LDS keep up with 4 waves x 2 reads per 48 cycles?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_mfma3 tools/micro/lds_mfma3.hip && tools/micro/lds_mfma3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

template <int MODE, int PD, int NF, int TRANS>  // MODE 0: reads + MFMAs, 1: MFMAs only, 2: reads only; NF fillers per pair, TRANS of them v_exp
__global__ void __launch_bounds__(256, 1) k(float* out, const u32x4* wsrc, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 98304 / 4; i += blockDim.x) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  u32x4 w1[27], w2[27];
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    w1[j] = wsrc[(j * 2) * 64 + lane];
    w2[j] = wsrc[(j * 2 + 1) * 64 + lane];
  }
  // planar A image: [term][kgroup][voxel][16 B]; lane = (voxel l&15, kgroup l>>4); two waves share a voxel tile (cout halves)
  const int base = (lane >> 4) * 2304 + ((lane & 15) + (wave >> 1) * 16) * 16 + 1536;
  f32x4 A[2] = {}, B[2] = {};
  u32x4 fa[PD + 1][2];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = seed + i;
  constexpr int NP = 54;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      if (MODE != 1) { const char* p = lds + base + i * 16; fa[i][0] = *(const u32x4*)p; fa[i][1] = *(const u32x4*)(p + 9216); }
      else { fa[i][0] = w1[0]; fa[i][1] = w2[0]; }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i + PD < NP) {
        if (MODE != 1) {
          const char* p = lds + base + ((i + PD) % 27) * 16 + ((i + PD) / 27) * 18432;
          fa[(i + PD) % (PD + 1)][0] = *(const u32x4*)p;
          fa[(i + PD) % (PD + 1)][1] = *(const u32x4*)(p + 9216);
        } else { fa[(i + PD) % (PD + 1)][0] = w1[0]; fa[(i + PD) % (PD + 1)][1] = w2[0]; }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int j = i % 27, t = i / 27;
      if (MODE != 2) {
        A[t] = MF(w1[j], fa[i % (PD + 1)][0], A[t]);
        B[t] = MF(w2[j], fa[i % (PD + 1)][0], B[t]);
        B[t] = MF(w1[j], fa[i % (PD + 1)][1], B[t]);
      } else {
        asm volatile("" ::"v"(fa[i % (PD + 1)][0]), "v"(fa[i % (PD + 1)][1]));
      }
#pragma unroll
      for (int q = 0; q < NF; ++q) {  // independent filler chains
        if (q < TRANS) f[q % 8] = __builtin_amdgcn_exp2f(f[q % 8]);
        else f[q % 8] = f[q % 8] * 1.0001f + 0.5f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int r = 0; r < 4; ++r) s += A[0][r] + B[0][r] + A[1][r] + B[1][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int PD, int NF, int TRANS>
void run(const char* name, float* out, const u32x4* w) {
  const int iters = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<MODE, PD, NF, TRANS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k<MODE, PD, NF, TRANS>), dim3(256), dim3(256), 100 * 1024, 0, out, w, 10, 1.f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, PD, NF, TRANS>), dim3(256), dim3(256), 100 * 1024, 0, out, w, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("NF=%2d trans=%d PD=%d %-14s %.3f ms  %.1f ns per pair (3 MFMAs of 16 cycles: 20.9 ns at 2.3 GHz)\n", NF, TRANS, PD, name, ms,
         ms * 1e6 / (54.0 * iters));
}
int main() {
  float* out; (void)hipMalloc(&out, 4 * 256 * 1024);
  u32x4* w; (void)hipMalloc(&w, 64 * 54 * 16); (void)hipMemset(w, 0x3c, 64 * 54 * 16);
  run<1, 3, 0, 0>("MFMAs only", out, w);
  run<2, 3, 0, 0>("reads only", out, w);
  run<0, 3, 0, 0>("reads + MFMAs", out, w);
  run<0, 4, 0, 0>("reads + MFMAs", out, w);
  run<0, 3, 2, 0>("reads + MFMAs", out, w);
  run<0, 3, 4, 0>("reads + MFMAs", out, w);
  run<0, 3, 6, 0>("reads + MFMAs", out, w);
  run<0, 3, 8, 0>("reads + MFMAs", out, w);
  run<0, 3, 6, 2>("reads + MFMAs", out, w);
  run<1, 3, 6, 0>("MFMAs only", out, w);
  return 0;
}
