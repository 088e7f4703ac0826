// How much of a denoise step is launch overhead?  A chain of N dependent launches of a kernel that does (almost) nothing, replayed as
// one hipGraph and launched eagerly: microseconds per launch.   hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.f;
}
int main() {
  const int N = 130, reps = 200;
  float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int grid : {1, 256, 1024}) {
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, buf, grid * 256);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, buf, grid * 256);
    CK(hipStreamSynchronize(s));
    const double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, buf, grid * 256);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * N);
    std::printf("grid %4d x 256 threads: %.2f us per launch eager, %.2f us per launch in a %d-node graph\n", grid, eager, graph, N);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
