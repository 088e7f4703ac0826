// What does a barrier among the P workgroups of ONE sample cost inside a launch (the multi-workgroup form of the deepest level /
// the small attention that VERDICT r03 asks for)?  256 workgroups in groups of P exchange a 1 KB payload NB times:
//   method 0: plain stores + __threadfence() (agent-scope release: L2 write-back on a multi-XCD part) + atomic counter, spin,
//             __threadfence() (acquire: invalidate), plain loads
//   method 1: payload through agent-scope relaxed atomic stores / loads (write-through, no fence), s_waitcnt vmcnt(0) before
//             the arrival, relaxed atomics for the counter
// with the group's workgroups on ONE XCD (blockIdx % 8 equal: the hardware deals workgroups round-robin over the 8 XCDs) or on
// consecutive blockIdx (P different XCDs), and with `dirty` KB of unrelated global stores per workgroup in front of every barrier
// (what a release fence then has to write back).  Every spin is bounded: a barrier that does not complete sets a flag and falls
// through, so the grid always drains.
//   hipcc --offload-arch=gfx950 -O3 -o group_barrier group_barrier.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Args {
  unsigned* count;  // [groups] arrivals of the current barrier
  unsigned* gen;    // [groups] generation: bumped by the last arriver
  float* payload;   // [groups][P][256]
  float* scratch;   // [wgs][dirty floats]
  float* out;       // [wgs]
  int* flag;
  int P, NB, same_xcd, method, dirty_floats, groups;
};

__device__ __forceinline__ bool group_barrier(unsigned* count, unsigned* gen, int P, int method, int* flag) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (method == 0) __threadfence();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == (unsigned)P - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { ok = false; atomicOr(flag, 1); break; }
      }
    }
    if (method == 0) __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ void __launch_bounds__(256) kern(Args a) {
  const int w = blockIdx.x;
  int group, part;
  if (a.same_xcd) {
    const int xcd = w & 7, slot = w >> 3;
    group = (slot / a.P) * 8 + xcd;
    part = slot % a.P;
  } else {
    group = w / a.P;
    part = w % a.P;
  }
  if (group >= a.groups) return;
  float acc = 0.f;
  float* mine = a.payload + ((size_t)group * a.P + part) * 256;
  float* sc = a.scratch + (size_t)w * a.dirty_floats;
  for (int it = 0; it < a.NB; ++it) {
    for (int i = threadIdx.x; i < a.dirty_floats; i += 256) sc[i] = acc + i;
    const float v = acc + it + part;
    if (a.method == 0) mine[threadIdx.x] = v;
    else __hip_atomic_store(mine + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!group_barrier(a.count + group, a.gen + group, a.P, a.method, a.flag)) break;
    for (int p = 0; p < a.P; ++p) {
      const float* theirs = a.payload + ((size_t)group * a.P + p) * 256;
      acc += a.method == 0 ? theirs[threadIdx.x] : __hip_atomic_load(theirs + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // second barrier of the round: nobody overwrites its payload before everybody has read it
    if (!group_barrier(a.count + group, a.gen + group, a.P, a.method, a.flag)) break;
  }
  if (threadIdx.x == 0) a.out[w] = acc;
}

int main() {
  const int WGS = 256, NB = 100;
  Args a{};
  CK(hipMalloc(&a.count, 4 * WGS)); CK(hipMalloc(&a.gen, 4 * WGS)); CK(hipMalloc(&a.payload, 4 * WGS * 256));
  CK(hipMalloc(&a.scratch, (size_t)4 * WGS * 65536)); CK(hipMalloc(&a.out, 4 * WGS)); CK(hipMalloc(&a.flag, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int P : {2, 3, 4})
    for (int same : {1, 0})
      for (int method : {0, 1})
        for (int dirty_kb : {0, 64}) {
          a.P = P; a.NB = NB; a.same_xcd = same; a.method = method; a.dirty_floats = dirty_kb * 256; a.groups = WGS / P / 8 * 8;
          const int grid = a.groups * P;
          CK(hipMemset(a.count, 0, 4 * WGS)); CK(hipMemset(a.gen, 0, 4 * WGS)); CK(hipMemset(a.flag, 0, 4));
          hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, a);
          CK(hipStreamSynchronize(s));
          const int reps = 5;
          auto t0 = std::chrono::steady_clock::now();
          for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, a);
          CK(hipStreamSynchronize(s));
          const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
          // the same launch without barriers costs the payload / scratch traffic only: measure it with P = 1 semantics? (reported
          // as is: two barriers + one exchange per round)
          int flag = 0; CK(hipMemcpy(&flag, a.flag, 4, hipMemcpyDeviceToHost));
          float o = 0; CK(hipMemcpy(&o, a.out, 4, hipMemcpyDeviceToHost));
          std::printf("P %d  %-9s  method %d  dirty %2d KB/round: %7.1f us per launch = %5.2f us per round (2 barriers + exchange)%s  [check %.0f]\n",
                      P, same ? "same XCD" : "adjacent", method, dirty_kb, us, us / NB, flag ? "  TIMED OUT" : "", o);
        }
  return 0;
}
