// Micro-benchmark 2: which feature of the z-slide conv's matrix loop breaks the overlap of ds_read_b128 A-fragment reads with
// v_mfma_f32_32x32x16_f16?  (lds_mfma.hip: with constant operands the two overlap almost perfectly; in the kernel they add.)
//   FEAT bits: 1 = 14 weight-fragment pairs in registers (112 VGPRs), cycled through, instead of one constant pair
//              2 = per-pair address arithmetic like the kernel's (add, and, cndmask on a per-lane mask)
//              4 = two accumulator sets (tile 0 / tile 1) as in the kernel
//              8 = 8 waves per workgroup, only waves 0-3 run the loop, 4-7 idle at a barrier (the kernel's occupancy: 256 regs)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_mfma2 tools/micro/lds_mfma2.hip && tools/micro/lds_mfma2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

template <int MODE, int PD, int FEAT>  // MODE 0: reads + MFMAs, 1: MFMAs only, 2: reads only
__global__ void __launch_bounds__(512, 1) k(float* out, const u32x4* wsrc, int iters, int mask_in) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  if ((FEAT & 8) && wave >= 4) return;
  constexpr int NW = (FEAT & 1) ? 14 : 1;
  u32x4 w1[NW], w2[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    w1[j] = wsrc[(j * 2) * 64 + lane];
    w2[j] = wsrc[(j * 2 + 1) * 64 + lane];
  }
  const int base = (lane & 31) * 144 + (lane >> 5) * 16 + (wave & 3) * 64 + 512;  // conflict-free record addressing
  const bool edge = ((lane + mask_in) & 7) == 0;  // per-lane mask as the kernel's first-column test
  f32x16 A[2] = {}, B[2] = {};
  u32x4 fa[PD + 1][2];
  constexpr int NP = 27;
  auto addr = [&](int i) {
    int a = base + (i % 24) * 144 * 8;
    if (FEAT & 2) {
      int b = a + mask_in * 144;            // (runtime zero: keeps the add)
      a = edge ? (b & 255) : b;             // redirect into the zero area, same bank quad
    }
    return a;
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      if (MODE != 1) { const char* p = lds + addr(i); fa[i][0] = *(const u32x4*)p; fa[i][1] = *(const u32x4*)(p + 32); }
      else { fa[i][0] = w1[0]; fa[i][1] = w2[0]; }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i + PD < NP) {
        if (MODE != 1) {
          const char* p = lds + addr(i + PD);
          fa[(i + PD) % (PD + 1)][0] = *(const u32x4*)p;
          fa[(i + PD) % (PD + 1)][1] = *(const u32x4*)(p + 32);
        } else { fa[(i + PD) % (PD + 1)][0] = w1[0]; fa[(i + PD) % (PD + 1)][1] = w2[0]; }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int j = (FEAT & 1) ? i % 14 : 0;
      const int t = (FEAT & 4) ? (i >= 14) : 0;
      if (MODE != 2) {
        A[t] = MF(fa[i % (PD + 1)][0], w1[j], A[t]);
        B[t] = MF(fa[i % (PD + 1)][0], w2[j], B[t]);
        B[t] = MF(fa[i % (PD + 1)][1], w1[j], B[t]);
      } else {
        asm volatile("" ::"v"(fa[i % (PD + 1)][0]), "v"(fa[i % (PD + 1)][1]));
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += A[0][r] + B[0][r] + A[1][r] + B[1][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int PD, int FEAT>
void run(const char* name, float* out, const u32x4* w) {
  const int iters = 400;
  const int threads = (FEAT & 8) ? 512 : 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<MODE, PD, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k<MODE, PD, FEAT>), dim3(256), dim3(threads), 100 * 1024, 0, out, w, 10, 0);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, PD, FEAT>), dim3(256), dim3(threads), 100 * 1024, 0, out, w, iters, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("feat=%2d PD=%d %-14s %.3f ms  %.1f ns per pair per wave\n", FEAT, PD, name, ms, ms * 1e6 / (27.0 * iters));
}
template <int PD, int FEAT>
void trio(float* out, const u32x4* w) {
  run<1, PD, FEAT>("MFMAs only", out, w);
  run<2, PD, FEAT>("reads only", out, w);
  run<0, PD, FEAT>("reads + MFMAs", out, w);
}
int main() {
  float* out; (void)hipMalloc(&out, 4 * 256 * 1024);
  u32x4* w; (void)hipMalloc(&w, 64 * 28 * 16); (void)hipMemset(w, 0x3c, 64 * 28 * 16);
  trio<3, 0>(out, w);
  trio<3, 1>(out, w);
  trio<3, 2>(out, w);
  trio<3, 4>(out, w);
  trio<3, 7>(out, w);
  trio<2, 7>(out, w);
  trio<3, 15>(out, w);
  trio<3, 8>(out, w);
  return 0;
}
