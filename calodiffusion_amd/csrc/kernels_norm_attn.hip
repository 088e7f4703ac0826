// GroupNorm and linear-attention kernels (HBM-bound passes) for channels-last fp32 activations on gfx950.
// Reference semantics: nn.GroupNorm(eps=1e-5, biased variance) as used by Block / PreNorm / LinearAttention.to_out
// (calodiffusion/models/models.py:155,293,325) and LinearAttention.forward (models.py:301-318).
#include "cd_common.h"
#include "gn_defer.h"
#include <cstdio>
#include <cstdlib>

namespace cd {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------------------
// GroupNorm, split so that every piece fuses into a neighbour:
//   statistics   per-(sample, unit, channel) {sum, sum of squares} partials in fp32.  Producers: the conv / pointwise
//                epilogues, the apply kernel below (for a following PreNorm), or ch_stats_kernel for a tensor that
//                comes from a kernel without a stats epilogue.  Partials are reduced in a fixed order in fp64
//                => bitwise-deterministic statistics.
//   finalize     folds mean / rstd / gamma / beta (+ the per-sample embedding that follows the activation) into
//                per-(sample, channel) coefficients {scale, shift, add, 0}:  y = act(scale*x + shift) + add
//   apply        either inside the consuming conv's LDS staging (ConvFusion::coef) or gn_apply_kernel.
// ------------------------------------------------------------------------------------------------------------
int gn_nsplit_for(int64_t vox, int batch) {
  // enough blocks to fill 256 CUs a few times over, but >= 256 voxels per block
  static const int target = getenv("CD_GN_NSPLIT_TARGET") ? atoi(getenv("CD_GN_NSPLIT_TARGET")) : 2048;
  int64_t want = (target + batch - 1) / batch;
  int64_t cap = (vox + 255) / 256;
  int64_t n = want < cap ? want : cap;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

__global__ void __launch_bounds__(256) ch_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int channels,
                                                       int64_t vox, int nsplit) {
  __shared__ double sP[256][2];
  const int tid = threadIdx.x;
  const int split = blockIdx.x, b = blockIdx.y;
  const int cols = channels >> 2;
  const int rows = 256 / cols;
  const int64_t per = (vox + nsplit - 1) / nsplit;
  const int64_t v0 = split * per;
  const int64_t v1 = (v0 + per < vox) ? v0 + per : vox;
  const int colid = tid % cols, row = tid / cols;
  // four channels per thread, kept separate
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    const float* base = x + (size_t)b * vox * channels + colid * 4;
    for (int64_t v = v0 + row; v < v1; v += rows) {
      const f32x4 t = *(const f32x4*)(base + (size_t)v * channels);
      s1 += t;
      s2 += t * t;
    }
  }
  float* dst = part + (((size_t)b * nsplit + split) * channels) * 2;
  for (int e = 0; e < 4; ++e) {
    sP[tid][0] = (double)s1[e];
    sP[tid][1] = (double)s2[e];
    __syncthreads();
    if (tid < cols) {
      double a1 = 0.0, a2 = 0.0;
      for (int r = 0; r < rows; ++r) {
        a1 += sP[r * cols + tid][0];
        a2 += sP[r * cols + tid][1];
      }
      dst[(tid * 4 + e) * 2] = (float)a1;
      dst[(tid * 4 + e) * 2 + 1] = (float)a2;
    }
    __syncthreads();
  }
}

void launch_ch_stats(const float* x, float* part, int batch, int channels, int64_t vox, int nsplit, hipStream_t s) {
  CD_REQUIRE(channels % 4 == 0 && channels <= 256, "channel stats: channels must be a multiple of 4 and <= 256");
  char cat[64];
  std::snprintf(cat, sizeof cat, "ch_stats C%d n%ld", channels, (long)vox);
  prof::Scope scope(cat, s, 0, 4.0 * batch * (double)vox * channels);
  hipLaunchKernelGGL(ch_stats_kernel, dim3(nsplit, batch), dim3(256), 0, s, x, part, channels, vox, nsplit);
  CD_HIP(hipGetLastError());
}

// one block per sample; thread c < channels writes coef[b][c]
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ part, int units, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ add_bc,
                                                          int add_ld, float* __restrict__ coef, int channels, int groups,
                                                          int64_t vox, float* __restrict__ stat_out) {
  __shared__ double sS[256][2];
  __shared__ double sC[256][2];
  __shared__ float sMean[64], sRstd[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  // 256 threads = channels x slices; each slice sums every nsl-th unit, slices are then added in a fixed order
  const int nsl = 256 / channels;
  const int c = tid % channels, sl = tid / channels;
  if (sl < nsl) {
    const float* p = part + ((size_t)b * units * channels + c) * 2;
    double a1 = 0.0, a2 = 0.0;
    for (int u = sl; u < units; u += nsl) {
      const float2 v = *(const float2*)(p + (size_t)u * channels * 2);
      a1 += (double)v.x;
      a2 += (double)v.y;
    }
    sS[tid][0] = a1;
    sS[tid][1] = a2;
  }
  __syncthreads();
  if (tid < channels) {
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < nsl; ++k) {
      a1 += sS[k * channels + tid][0];
      a2 += sS[k * channels + tid][1];
    }
    sC[tid][0] = a1;
    sC[tid][1] = a2;
  }
  __syncthreads();
  const int cpg = channels / groups;
  if (tid < groups) {
    double a1 = 0.0, a2 = 0.0;
    for (int c = 0; c < cpg; ++c) {
      a1 += sC[tid * cpg + c][0];
      a2 += sC[tid * cpg + c][1];
    }
    const double cnt = (double)vox * cpg;
    const double mu = a1 / cnt;
    double var = a2 / cnt - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    sMean[tid] = (float)mu;
    sRstd[tid] = (float)(1.0 / sqrt(var + 1e-5));
    if (stat_out) {
      stat_out[((size_t)b * groups + tid) * 2] = sMean[tid];
      stat_out[((size_t)b * groups + tid) * 2 + 1] = sRstd[tid];
    }
  }
  __syncthreads();
  if (tid < channels) {
    const int g = tid / cpg;
    const float sc = sRstd[g] * gamma[tid];
    f32x4 o;
    o[0] = sc;
    o[1] = beta[tid] - sMean[g] * sc;
    o[2] = add_bc ? add_bc[(size_t)b * add_ld + tid] : 0.f;
    o[3] = 0.f;
    *(f32x4*)(coef + ((size_t)b * channels + tid) * 4) = o;
  }
}

void launch_gn_finalize(const float* part, int units, const float* gamma, const float* beta, const float* add_bc, int add_ld,
                        float* coef, int batch, int channels, int groups, int64_t vox, hipStream_t s, float* stat_out) {
  CD_REQUIRE(channels <= 256 && groups <= 64 && channels % groups == 0, "group norm: <= 256 channels, <= 64 groups");
  prof::Scope scope("gn_finalize", s, 0, 0);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch), dim3(256), 0, s, part, units, gamma, beta, add_bc, add_ld, coef, channels,
                     groups, vox, stat_out);
  CD_HIP(hipGetLastError());
}

// y = act(scale*x + shift) + add [+ residual]; optionally emits the channel partials of y (for a following PreNorm)
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ coef, int channels, int64_t vox, int silu,
                                                       const float* __restrict__ residual,
                                                       const float* __restrict__ residual1, int res_c0,
                                                       int blocks_per_sample, float* __restrict__ part_out, GnDefer defer) {
  __shared__ double sP[256][2];
  __shared__ __attribute__((aligned(16))) float sCoef[256 * 4];
  __shared__ __attribute__((aligned(16))) char sDefer[256 * 16 + 64 * 8];
  if (defer.part) gn_defer_to_lds(defer, blockIdx.x / blocks_per_sample, sCoef, sDefer);
  const int tid = threadIdx.x;
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  const int cols = channels >> 2;
  const int rows = 256 / cols;
  const int64_t vper = (vox + blocks_per_sample - 1) / blocks_per_sample;
  const int64_t v0 = blk * vper;
  const int64_t v1 = (v0 + vper < vox) ? v0 + vper : vox;
  const int colid = tid % cols, row = tid / cols;
  const int c = colid * 4;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    f32x4 cf[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      cf[e] = defer.part ? *(const f32x4*)(sCoef + (c + e) * 4) : *(const f32x4*)(coef + ((size_t)b * channels + c + e) * 4);
    const size_t sbase = (size_t)b * vox * channels;
    const int res_c1 = channels - res_c0;
    // SiLU on the transcendental unit (v_exp / v_rcp, ~3 ulp: the conv kernels' fused form, zs_silu): libm's expf was ~25 VALU
    // instructions per element on a kernel that should only wait for HBM
    auto act = [&](f32x4 t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = cf[e][0] * t[e] + cf[e][1];
        if (silu) u = u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
        t[e] = u + cf[e][2];
      }
      return t;
    };
    auto res_at = [&](int64_t v) {
      if (residual1)  // identity shortcut of a block whose input is a (never materialised) channel concat of two tensors
        return c < res_c0 ? *(const f32x4*)(residual + ((size_t)b * vox + v) * res_c0 + c)
                          : *(const f32x4*)(residual1 + ((size_t)b * vox + v) * res_c1 + (c - res_c0));
      return *(const f32x4*)(residual + sbase + (size_t)v * channels + c);
    };
    int64_t v = v0 + row;
    // four voxels per trip: all eight loads are in flight before the first result is needed
    for (; v + 3 * rows < v1; v += 4 * rows) {
      f32x4 t[4], r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *(const f32x4*)(x + sbase + (size_t)(v + u * rows) * channels + c);
      if (residual) {
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = res_at(v + u * rows);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 o = act(t[u]);
        if (residual) o += r[u];
        *(f32x4*)(y + sbase + (size_t)(v + u * rows) * channels + c) = o;
        s1 += o;
        s2 += o * o;
      }
    }
    for (; v < v1; v += rows) {
      f32x4 o = act(*(const f32x4*)(x + sbase + (size_t)v * channels + c));
      if (residual) o += res_at(v);
      *(f32x4*)(y + sbase + (size_t)v * channels + c) = o;
      s1 += o;
      s2 += o * o;
    }
  }
  if (part_out) {
    float* dst = part_out + (((size_t)b * blocks_per_sample + blk) * channels) * 2;
    for (int e = 0; e < 4; ++e) {
      sP[tid][0] = (double)s1[e];
      sP[tid][1] = (double)s2[e];
      __syncthreads();
      if (tid < cols) {
        double a1 = 0.0, a2 = 0.0;
        for (int r = 0; r < rows; ++r) {
          a1 += sP[r * cols + tid][0];
          a2 += sP[r * cols + tid][1];
        }
        dst[(tid * 4 + e) * 2] = (float)a1;
        dst[(tid * 4 + e) * 2 + 1] = (float)a2;
      }
      __syncthreads();
    }
  }
}

int gn_apply_blocks_per_sample(int batch, int channels, int64_t vox) {
  const int rows = 256 / (channels / 4);
  int64_t bps = (vox + (int64_t)rows * 8 - 1) / ((int64_t)rows * 8);  // >= 8 float4 per thread
  static const int target = getenv("CD_GN_APPLY_WGS") ? atoi(getenv("CD_GN_APPLY_WGS")) : 1024;  // fewer, longer-lived workgroups: each repeats the GroupNorm fold (0.386 -> 0.368 ms per step against 4096)
  const int64_t want = (target + batch - 1) / batch;
  if (bps > want) bps = want;
  if (bps < 1) bps = 1;
  return (int)bps;
}

void launch_gn_apply(const float* x, float* y, const float* coef, int batch, int channels, int64_t vox, int silu,
                     const float* residual, const float* residual1, int res_c0, float* part_out, hipStream_t s,
                     const GnDefer* defer) {
  CD_REQUIRE(channels % 4 == 0 && channels <= 256, "group norm: channels must be a multiple of 4 and <= 256");
  CD_REQUIRE(!defer || !defer->part || (defer->C == channels && defer->groups <= 64), "gn_apply: bad deferred normalisation");
  const int bps = gn_apply_blocks_per_sample(batch, channels, vox);
  char cat[64];
  std::snprintf(cat, sizeof cat, "gn_apply C%d n%ld", channels, (long)vox);
  prof::Scope scope(cat, s, 0, 4.0 * batch * (double)vox * channels * (2 + (residual ? 1 : 0)));
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)(batch * bps)), dim3(256), 0, s, x, y, coef, channels, vox, silu, residual,
                     residual1, res_c0, bps, part_out, defer ? *defer : GnDefer());
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Linear attention (heads = 1, dim_head = 32).  qkv is channels-last (B, n, 96): q = [0,32), k = [32,64), v = [64,96).
//   k <- softmax over the n voxels (per channel d);  context[d][e] = sum_n k[d,n] v[e,n]      (models.py:309,312)
//   q <- softmax over the 32 channels (per voxel) * 32^-1/2;  out[e,n] = sum_d context[d][e] q[d,n]   (:308,311,314)
// Pass 1 (this kernel): per (sample, voxel range) a local max m[d], local sum s[d] = sum exp(k - m) and the
// un-normalised context partial ctx[d][e] = sum exp(k[n][d] - m[d]) v[n][e] on the matrix cores (K = voxels).
// Pass 2 (attn_combine_kernel): merge the partials log-sum-exp style and fold the result into the to_out 1x1 conv:
//   W'[c][d] = scale * sum_e W_out[c][e] context[d][e], written in packed MFMA layout as per-sample weights.
// Pass 3 is the pointwise kernel with the softmax-32 A prologue and per-sample weights.
// ------------------------------------------------------------------------------------------------------------
int attn_nsplit_for(int64_t vox, int batch) {
  int64_t want = (1024 + batch - 1) / batch;
  int64_t cap = (vox + 511) / 512;  // >= 128 voxels per wave
  int64_t n = want < cap ? want : cap;
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  return (int)n;
}
size_t attn_partial_floats(int batch, int nsplit) { return (size_t)batch * nsplit * (64 + 1024); }

__global__ void __launch_bounds__(256) attn_context_kernel(const float* __restrict__ qkv, float* __restrict__ partials,
                                                           int64_t vox, int nsplit) {
  __shared__ float sMax[4][32];
  __shared__ float sSum[8][32];
  __shared__ float sCtx[4][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int split = blockIdx.x, b = blockIdx.y;
  const int64_t per = (((vox + nsplit - 1) / nsplit) + 1) & ~(int64_t)1;  // even
  const int64_t v0 = split * per;
  const int64_t v1 = (v0 + per < vox) ? v0 + per : vox;
  const float* base = qkv + (size_t)b * vox * 96;

  // local max of k over this block's voxels, per channel.  Both sweeps below fetch eight voxels per trip (round 4: one voxel per
  // trip was a memory round trip per 256 bytes and wave -- 40 us at Dataset-2's level 0 for 53 MB, with the training step's seven
  // attention blocks 0.17 ms of pure latency); indices past the block are clamped and their values masked.
  float m = -3.0e38f;
  for (int64_t n = v0 + 2 * wave + half; n < v1; n += 64) {
    float kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t nu = n + 8 * u;
      kv[u] = base[(size_t)(nu < v1 ? nu : v1 - 1) * 96 + 32 + col];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, kv[u]);  // (a clamped index repeats a voxel of the block: harmless under max)
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (half == 0) sMax[wave][col] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sMax[0][col], sMax[1][col]), fmaxf(sMax[2][col], sMax[3][col]));

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ssum = 0.f;
  // A[i = d][k = voxel half] = exp(k[n][d] - m[d]),  B[k][j = e] = v[n][e]
  // (the loop bound is wave-uniform: per is even and a wave's two halves take voxels n, n + 1 of a pair, both below v0 + per or
  // both not -- every lane of a wave runs the same MFMAs, in the same order as the one-voxel-per-trip form: bit-identical sums)
  for (int64_t n = v0 + 2 * wave + half; n < v0 + per; n += 64) {
    float kk[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t nu = n + 8 * u;
      const size_t o = (size_t)(nu < v1 ? nu : v1 - 1) * 96;
      kk[u] = base[o + 32 + col];
      vv[u] = base[o + 64 + col];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t nu = n + 8 * u;
      if (nu - half < v0 + per) {  // (wave-uniform: the pair's first voxel)
        const bool in = nu < v1;
        const float av = in ? expf(kk[u] - m) : 0.f;
        const float bv = in ? vv[u] : 0.f;
        ssum += av;
        acc = MFMA32(av, bv, acc);
      }
    }
  }
  sSum[wave * 2 + half][col] = ssum;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * half;  // row = A index = channel d; column = e
    sCtx[wave][d * 32 + col] = acc[r];
  }
  __syncthreads();
  float* out = partials + ((size_t)b * nsplit + split) * (64 + 1024);
  if (tid < 32) {
    out[tid] = m;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += sSum[i][tid];
    out[32 + tid] = t;
  }
  for (int i = tid; i < 1024; i += 256) out[64 + i] = (sCtx[0][i] + sCtx[1][i]) + (sCtx[2][i] + sCtx[3][i]);
}

void launch_attn_context(const float* qkv, float* partials, int batch, int64_t vox, int nsplit, hipStream_t s) {
  prof::Scope scope("attn_context", s, 2.0 * 32 * 32 * (double)vox * batch, 4.0 * batch * (double)vox * 96);
  hipLaunchKernelGGL(attn_context_kernel, dim3(nsplit, batch), dim3(256), 0, s, qkv, partials, vox, nsplit);
  CD_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) attn_combine_kernel(const float* __restrict__ partials, int nsplit,
                                                           const float* __restrict__ w_out, int cout,
                                                           float* __restrict__ wpk_b, float scale, float* __restrict__ ctx_out,
                                                           float* __restrict__ kstat_out, int layout_T) {
  __shared__ float sM[32], sInv[32];
  __shared__ float sCtx[1024];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* p = partials + (size_t)b * nsplit * (64 + 1024);
  if (tid < 32) {
    float M = -3.0e38f;
    for (int i = 0; i < nsplit; ++i) M = fmaxf(M, p[(size_t)i * 1088 + tid]);
    float S = 0.f;
    for (int i = 0; i < nsplit; ++i) S += p[(size_t)i * 1088 + 32 + tid] * expf(p[(size_t)i * 1088 + tid] - M);
    sM[tid] = M;
    sInv[tid] = scale / S;
    if (kstat_out) {
      kstat_out[((size_t)b * 32 + tid) * 2] = M;
      kstat_out[((size_t)b * 32 + tid) * 2 + 1] = 1.f / S;
    }
  }
  __syncthreads();
  // rescaling factors of the partials, once per (partial, channel) instead of once per context entry (32 x fewer exponentials and
  // loads of the maxima in the loop below; same expression, same summation order: bit-identical)
  __shared__ float sF[128 * 32];  // nsplit <= 128 (attn_nsplit_for)
  for (int i = tid; i < nsplit * 32; i += 256) sF[i] = expf(p[(size_t)(i >> 5) * 1088 + (i & 31)] - sM[i & 31]);
  __syncthreads();
  for (int i = tid; i < 1024; i += 256) {
    const int d = i >> 5;
    float c = 0.f;
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {  // four partials' loads in flight
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(k + u) * 1088 + 64 + i];
#pragma unroll
      for (int u = 0; u < 4; ++u) c += v[u] * sF[(k + u) * 32 + d];
    }
    for (; k < nsplit; ++k) c += p[(size_t)k * 1088 + 64 + i] * sF[k * 32 + d];
    sCtx[i] = c * sInv[d];
    if (ctx_out) ctx_out[(size_t)b * 1024 + i] = sCtx[i] / scale;
  }
  __syncthreads();
  // W'[c][d] = sum_e W_out[c][e] * ctx[d][e]; packed: ((ct*4+q)*64 + h*32+j)*4+e4 with c = ct*32+j, d = h*16 + 4q + e4
  const int CT = (cout + 31) / 32;
  float* wo = wpk_b + (size_t)b * CT * 1024;
  for (int i = tid; i < CT * 1024; i += 256) {
    const int e4 = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3, ct = i >> 10;
    // layout_T (attn_out_kernel): the k-slot order of an accumulator-register operand, d = row(r = 4q + e4, half)
    const int c = ct * 32 + (lane & 31);
    const int d = layout_T ? (e4 + 8 * q + 4 * (lane >> 5)) : (lane >> 5) * 16 + q * 4 + e4;
    float acc = 0.f;
    if (c < cout)
      for (int e = 0; e < 32; ++e) acc = fmaf(w_out[c * 32 + e], sCtx[d * 32 + e], acc);
    wo[i] = acc;
  }
}

void launch_attn_combine(const float* partials, int nsplit, const float* w_out, int cout, float* wpk_b, int batch,
                         float scale, hipStream_t s, float* ctx_out, float* kstat_out, bool layout_T) {
  prof::Scope scope("attn_combine", s, 0, 0);
  hipLaunchKernelGGL(attn_combine_kernel, dim3(batch), dim3(256), 0, s, partials, nsplit, w_out, cout, wpk_b, scale, ctx_out,
                     kstat_out, layout_T ? 1 : 0);
  CD_HIP(hipGetLastError());
}

}  // namespace cd
