// Do device-scope atomics from workgroups on DIFFERENT XCDs combine on ordinary hipMalloc memory (the max-|dy| word of the training
// step, the range-flag words, the partial-statistics slots all rely on it)?  512 workgroups: atomicMax(blockIdx + 1), atomicAdd(1),
// atomicOr(1 << (blockIdx % 8 = XCD)) into three words; expected after the kernel: 512, 512, 0xff.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_atomics xcd_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(unsigned* w) {
  if (threadIdx.x == 0) {
    atomicMax(w, blockIdx.x + 1);
    atomicAdd(w + 1, 1u);
    atomicOr(w + 2, 1u << (blockIdx.x & 7));
    atomicMax(w + 3 + 16 * (blockIdx.x & 7), blockIdx.x + 1);  // one word per XCD, 64 B apart
  }
}
int main() {
  unsigned* w; CK(hipMalloc(&w, 4096));
  int bad = 0;
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipMemset(w, 0, 4096));
    hipLaunchKernelGGL(k, dim3(512), dim3(64), 0, 0, w);
    CK(hipDeviceSynchronize());
    unsigned h[256]; CK(hipMemcpy(h, w, sizeof h, hipMemcpyDeviceToHost));
    if (h[0] != 512 || h[1] != 512 || h[2] != 0xff) { ++bad; std::printf("rep %d: max %u add %u or 0x%x\n", rep, h[0], h[1], h[2]); }
  }
  std::printf("%d of 20 launches combined wrongly%s\n", bad, bad ? "" : ": device-scope atomics on hipMalloc memory combine across XCDs");
  return 0;
}
