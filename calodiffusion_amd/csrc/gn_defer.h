// Consumer-side GroupNorm finalisation.
//
// A kernel that consumes GroupNorm coefficients {gamma*rstd, beta - mean*gamma*rstd, add, 0} per (sample, channel) can fold
// them itself from the producer's channel partials {sum, sum of squares} in its prologue, instead of reading the output of a
// separate gn_finalize launch (44 dependent ~5 us launches per denoise step).  Every workgroup of the consumer repeats the
// small reduction for its sample: units x C pairs from L2, fp64, FIXED order (8 interleaved slices per channel combined in a
// fixed tree) => deterministic and identical in all workgroups.
#pragma once
#include "cd_common.h"

namespace cd {

// LDS needed next to the [C][4] float coefficient table
__host__ __device__ inline int gn_defer_scratch_bytes(int C) { return C * 16 + 64 * 8; }

// Call from EVERY thread of the workgroup (blockDim.x a multiple of 64).  coef_lds: [C][4] floats; scratch:
// gn_defer_scratch_bytes(C) bytes, 8-byte aligned.  Returns after a barrier: coef_lds is ready to read.
__device__ __forceinline__ void gn_defer_to_lds(const GnDefer& g, int b, float* coef_lds, void* scratch) {
  double* sC = (double*)scratch;                          // [C][2]
  float* sMR = (float*)((char*)scratch + g.C * 16);       // [G][2]
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int C = g.C, G = g.groups;
  // the affine parameters do not depend on the statistics: their loads go out first and land under the reduction (every consumer
  // kernel starts with this fold, so a global-load latency here is a latency of the whole launch)
  float pg = 0.f, pb = 0.f, pa = 0.f;
  if (tid < C) {
    pg = g.gamma[tid];
    pb = g.beta[tid];
    pa = g.add ? g.add[(size_t)b * g.add_ld + tid] : 0.f;
  }
  for (int i = tid; i < C * 8; i += nthreads) {
    const int c = i >> 3, j = i & 7;
    const float* p = g.part + ((size_t)b * g.units * C + c) * 2;
    double a1 = 0.0, a2 = 0.0;
    // four partials per trip, their loads issued together (one L2 latency per trip instead of one per partial); same order of
    // additions as a plain loop
    for (int u = j; u < g.units; u += 32) {
      float2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = u + 8 * k < g.units ? *(const float2*)(p + (size_t)(u + 8 * k) * C * 2) : float2{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (u + 8 * k < g.units) {
          a1 += (double)v[k].x;
          a2 += (double)v[k].y;
        }
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      a1 += __shfl_xor(a1, m, 64);
      a2 += __shfl_xor(a2, m, 64);
    }
    if (j == 0) {
      sC[c * 2] = a1;
      sC[c * 2 + 1] = a2;
    }
  }
  __syncthreads();
  const int cpg = C / G;
  if (tid < G) {
    double a1 = 0.0, a2 = 0.0;
    for (int c = 0; c < cpg; ++c) {
      a1 += sC[(tid * cpg + c) * 2];
      a2 += sC[(tid * cpg + c) * 2 + 1];
    }
    const double cnt = (double)g.vox * cpg;
    const double mu = a1 / cnt;
    double var = a2 / cnt - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    sMR[tid * 2] = (float)mu;
    sMR[tid * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    if (g.stat_out) {
      g.stat_out[((size_t)b * G + tid) * 2] = sMR[tid * 2];
      g.stat_out[((size_t)b * G + tid) * 2 + 1] = sMR[tid * 2 + 1];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += nthreads) {
    const int gi = c / cpg;
    const bool pre = c == tid;  // (C <= blockDim.x in every caller: the preloaded registers; otherwise read here)
    const float sc = sMR[gi * 2 + 1] * (pre ? pg : g.gamma[c]);
    f32x4 o;
    o[0] = sc;
    o[1] = (pre ? pb : g.beta[c]) - sMR[gi * 2] * sc;
    o[2] = pre ? pa : (g.add ? g.add[(size_t)b * g.add_ld + c] : 0.f);
    o[3] = 0.f;
    *(f32x4*)(coef_lds + c * 4) = o;
    if (g.coef_out) *(f32x4*)(g.coef_out + ((size_t)b * C + c) * 4) = o;
  }
  __syncthreads();
}

}  // namespace cd
