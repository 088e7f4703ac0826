// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 under different accumulator dependency patterns (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/micro/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f16x8 a0, a1, w0, w1;
  for (int e = 0; e < 8; ++e) { a0[e] = (_Float16)(threadIdx.x * 0.001f + e); a1[e] = (_Float16)(e * 0.5f); w0[e] = (_Float16)0.25f; w1[e] = (_Float16)0.125f; }
  f32x16 A0 = {}, B0 = {}, A1 = {}, B1 = {}, C0 = {}, C1 = {};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {        // current kernel: A, B, B  (B depends on the MFMA right before it)
      A0 = MF(a0, w0, A0); B0 = MF(a0, w1, B0); B0 = MF(a1, w0, B0);
      A0 = MF(a1, w0, A0); B0 = MF(a1, w1, B0); B0 = MF(a0, w0, B0);
    } else if (MODE == 1) { // two tiles interleaved
      A0 = MF(a0, w0, A0); A1 = MF(a1, w0, A1); B0 = MF(a0, w1, B0); B1 = MF(a1, w1, B1); B0 = MF(a1, w0, B0); B1 = MF(a0, w0, B1);
    } else if (MODE == 2) { // six independent accumulators
      A0 = MF(a0, w0, A0); A1 = MF(a1, w0, A1); B0 = MF(a0, w1, B0); B1 = MF(a1, w1, B1); C0 = MF(a1, w0, C0); C1 = MF(a0, w0, C1);
    } else {                // one accumulator, fully serial
      A0 = MF(a0, w0, A0); A0 = MF(a0, w1, A0); A0 = MF(a1, w0, A0); A0 = MF(a1, w0, A0); A0 = MF(a1, w1, A0); A0 = MF(a0, w0, A0);
    }
    asm volatile("" : "+v"(a0), "+v"(a1));
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += A0[r] + B0[r] + A1[r] + B1[r] + C0[r] + C1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = 6.0 * iters;  // per wave
  const double tf = mf * 32768.0 * 4 * wgs / (ms * 1e-3) / 1e12;
  printf("%-28s wgs=%d  %.3f ms  %.1f ns/MFMA/wave  %.0f TFLOP/s (%.1f cycles @2.4GHz)\n", name, wgs, ms, ms * 1e6 / mf, tf, ms * 1e6 / mf * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 1024);
  for (int wgs : {1, 256}) {
    run<0>("A,B,B (current)", out, wgs);
    run<1>("2 tiles interleaved", out, wgs);
    run<2>("6 independent", out, wgs);
    run<3>("1 accumulator serial", out, wgs);
  }
  return 0;
}
