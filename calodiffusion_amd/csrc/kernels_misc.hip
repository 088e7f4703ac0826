// Small kernels around the U-Net: conditioning MLPs, EDM pre-conditioning head, sampler update, Philox noise,
// loss reduction and layout transposes.
#include "cd_common.h"
#include "gn_defer.h"

#include <cmath>

namespace cd {

// ------------------------------------------------------------------------------------------------------------
// Conditioning: time MLP, cond MLP (exact-erf GELU), concat, and every ResnetBlock's SiLU->Linear(128, C) projection
// in ONE launch (20 nn.Linear calls per forward in the reference: models.py:176-180, 575-608, 704-707).
// Also derives the EDM scalings of Loss.get_scaling (loss.py:29-41) and the time embedding input
// (calodiffusion.py:144-152) from sigma.  One 128-thread block per sample.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// out[j] = bias[j] + sum_k w[j][k] * in[k]: one wave per output row, lanes stride k (coalesced 256-B reads of the
// row-major torch weight), butterfly reduction.
__device__ __forceinline__ float wave_dot(const float* __restrict__ wr, const float* in, int nin, int lane) {
  float acc = 0.f;
  for (int k = lane; k < nin; k += 64) acc = fmaf(wr[k], in[k], acc);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  return acc;
}

// A wave owns 8 output rows at a time: their 8 weight-row loads are in flight together (one row per trip made every layer of
// the two small MLPs a chain of 8 L2 round trips per wave: 46 us for the whole kernel).
__device__ void dense(const float* __restrict__ w, const float* __restrict__ bias, const float* in, float* out, int nin,
                      int nout, bool gelu) {
  constexpr int R = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int j0 = wave * R; j0 < nout; j0 += nw * R) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int k = lane; k < nin; k += 64) {
      const float xv = in[k];
      float wv[R];
#pragma unroll
      for (int r = 0; r < R; ++r) wv[r] = w[(size_t)min(j0 + r, nout - 1) * nin + k];  // clamped: rows past the end are dropped
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = fmaf(wv[r], xv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
    }
    if (lane < R && j0 + lane < nout) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane == r) v = acc[r];
      v += bias[j0 + lane];
      out[j0 + lane] = gelu ? gelu_erf(v) : v;
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(512) embed_kernel(EmbedArgs a) {
  __shared__ float bufA[256], bufB[256], cat[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int bc = a.cond_rows ? b % a.cond_rows : b;  // row of `cond`
  const float tv = a.part == 2 ? 1.f
                   : a.part == 1 ? a.time_or_sigma[(size_t)b * a.time_stride]
                   : a.cond_rows ? a.time_or_sigma[(size_t)(b / a.cond_rows) * a.time_stride] : a.time_or_sigma[b];
  float t_in = tv;
  if (a.time_kind == 0) t_in = 0.5f * logf(tv);
  else if (a.time_kind == 1) t_in = tv / sqrtf(1.f + tv * tv);
  if (a.scal && a.part != 2 && tid == 0 && blockIdx.y == 0) {
    const float sd = a.sigma_data;
    const float s2 = tv * tv + sd * sd;
    a.scal[b * 4 + 0] = 1.f / sqrtf(s2);           // c_in
    a.scal[b * 4 + 1] = sd * sd / s2;              // c_skip
    a.scal[b * 4 + 2] = tv * sd / sqrtf(s2);       // c_out
    a.scal[b * 4 + 3] = tv;
  }
  const int half = a.half, q = half / 2;
  // SinusoidalPositionEmbeddings(q) (models.py:132-144): [sin(v f_i), cos(v f_i)], f_i = exp(-i ln(1e4) / (q/2 - 1))
  auto sinusoidal = [&](float v, float* dst) {
    const int hd = q / 2;
    const float step = (float)(9.210340371976184 / (double)(hd - 1));  // np.log(10000) / (half_dim - 1): a double, rounded to
                                                                       // fp32 when it multiplies the arange tensor
    for (int i = tid; i < hd; i += blockDim.x) {
      const float ang = v * expf((float)i * -step);
      dst[i] = sinf(ang);
      dst[hd + i] = cosf(ang);
    }
    __syncthreads();
  };
  // time branch: Linear(1, half/2) GELU | sinusoidal(half/2);  Linear(half/2, half) GELU Linear(half, half)
  if (a.part == 2) {  // condition row: the time half of `conditions` contributes nothing (SiLU(0) = 0)
    for (int i = tid; i < half; i += blockDim.x) cat[i] = 0.f;
    __syncthreads();
  } else {
    if (a.time_sin) {
      sinusoidal(t_in, bufB);
    } else {
      if (tid == 0) bufA[0] = t_in;
      __syncthreads();
      dense(a.tw1, a.tb1, bufA, bufB, 1, q, true);
    }
    dense(a.tw2, a.tb2, bufB, bufA, q, half, true);
    dense(a.tw3, a.tb3, bufA, cat, half, half, false);
  }
  // cond branch: Linear(cond_size, hidden) GELU | sinusoidal(half/2) of the scalar condition;  Linear(hidden, half) GELU
  // Linear(half, half)
  if (a.part == 1) {  // time row: no condition half
    for (int i = tid; i < half; i += blockDim.x) cat[half + i] = 0.f;
    __syncthreads();
  } else {
    if (a.cond_sin) {
      sinusoidal(a.cond[bc], bufB);
    } else {
      for (int i = tid; i < a.cond_size; i += blockDim.x) bufA[i] = a.cond[(size_t)bc * a.cond_size + i];
      __syncthreads();
      dense(a.cw1, a.cb1, bufA, bufB, a.cond_size, a.cond_hidden, true);
    }
    dense(a.cw2, a.cb2, bufB, bufA, a.cond_hidden, half, true);
    dense(a.cw3, a.cb3, bufA, cat + half, half, half, false);
  }
  // SiLU of conditions = cat(t, c)  (models.py:707; ResnetBlock.mlp[0])
  for (int i = tid; i < 2 * half; i += blockDim.x) {
    const float v = cat[i];
    bufA[i] = v / (1.f + expf(-v));
  }
  __syncthreads();
  // all ResnetBlock projections: a wave owns 4 output rows at a time so that 4 independent weight-row loads are in flight
  // gridDim.y workgroups per sample share the projection layers (each repeats the two small MLPs above: the projections'
  // ~700 weight rows are the latency chain of this kernel)
  const int nin = 2 * half;
  for (int l = blockIdx.y; l < a.n_layers; l += gridDim.y) {
    const EmbedLayer L = a.layers[l];
    for (int j0 = wave * 4; j0 < L.cout; j0 += nw * 4) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = lane; k < nin; k += 64) {
        const float xv = bufA[k];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < L.cout) acc[r] = fmaf(L.w[(size_t)(j0 + r) * nin + k], xv, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
        if (lane == 0 && j0 + r < L.cout) a.emb[(size_t)b * a.emb_ld + L.offset + j0 + r] = a.part == 2 ? acc[r] : acc[r] + L.b[j0 + r];
      }
    }
  }
}

void launch_embed(const EmbedArgs& a, hipStream_t s) {
  CD_REQUIRE(a.half * 2 <= 256 && a.cond_hidden <= 256 && a.cond_size <= 256, "embedding widths above 256 unsupported");
  prof::Scope scope("embed", s, 0, 0);
  const int groups = a.n_layers >= 4 ? 4 : 1;
  CD_REQUIRE(!a.cond_rows || a.batch % a.cond_rows == 0, "internal: embedding chunk must be whole steps");
  hipLaunchKernelGGL(embed_kernel, dim3(a.batch, groups), dim3(512), 0, s, a);
  CD_HIP(hipGetLastError());
}

// out[b][j] = bias[j] + sum_k w[j][k] * silu(cond[b][k])   (ResnetBlock.mlp, models.py:176-180; block-level tests only)
__global__ void silu_linear_kernel(const float* __restrict__ cond, const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ out, int nin, int nout) {
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < nout; j += blockDim.x) {
    float acc = bias[j];
    for (int k = 0; k < nin; ++k) {
      const float v = cond[(size_t)b * nin + k];
      acc = fmaf(w[(size_t)j * nin + k], v / (1.f + expf(-v)), acc);
    }
    out[(size_t)b * nout + j] = acc;
  }
}
void launch_silu_linear(const float* cond, const float* w, const float* bias, float* out, int batch, int nin, int nout,
                        hipStream_t s) {
  hipLaunchKernelGGL(silu_linear_kernel, dim3(batch), dim3(128), 0, s, cond, w, bias, out, nin, nout);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Head: final 1x1x1 conv 32 -> 1 (models.py:696) fused with the EDM output scaling (calodiffusion.py:161-167).
// 8 lanes per voxel, each a float4 of the 128-B channel vector.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_kernel(HeadArgs a) {
  const int64_t total = (int64_t)a.batch * a.vox;
  const int sub = threadIdx.x & 7;
  const f32x4 w = *(const f32x4*)(a.w + sub * 4);
  const float bias = a.bias[0];
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; i < total; i += ((int64_t)gridDim.x * 256) >> 3) {
    const f32x4 h = *(const f32x4*)(a.h + (size_t)i * 32 + sub * 4);
    float p = (h[0] * w[0] + h[1] * w[1]) + (h[2] * w[2] + h[3] * w[3]);
    p += __shfl_xor(p, 1, 64);
    p += __shfl_xor(p, 2, 64);
    p += __shfl_xor(p, 4, 64);
    if (sub == 0) {
      float pred = p + bias;
      if (a.scal) {
        const int b = (int)(i / a.vox);
        const float xv = a.x[i];
        if (a.objective == 0) pred = a.scal[b * 4 + 1] * xv + a.scal[b * 4 + 2] * pred;
        else if (a.objective == 1) pred = xv - a.scal[b * 4 + 3] * pred;
      }
      a.out[i] = pred;
      if (a.upd_stepvals) {  // (ddim_update_kernel's arithmetic)
        const float sigma = a.upd_stepvals[0], sprev = a.upd_stepvals[1], dsig = a.upd_stepvals[2], denom = a.upd_stepvals[3];
        const float eps = (a.x[i] - pred) / sigma;
        float r = pred + sprev * eps;
        if (a.upd_noise) r += dsig * a.upd_noise[i] / denom;
        a.upd_x_next[i] = r;
        if (a.upd_xs) a.upd_xs[i] = r;
        if (a.upd_x0s) a.upd_x0s[i] = pred;
      }
    }
  }
}

// the same with the final ResnetBlock's GroupNorm(8) + SiLU + shortcut folded in: workgroups belong to one sample (blockIdx.y)
__global__ void __launch_bounds__(256) head_gn_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float sCoef[32 * 4];
  __shared__ __attribute__((aligned(16))) char sDefer[32 * 16 + 64 * 8];
  const int b = blockIdx.y;
  gn_defer_to_lds(a.defer, b, sCoef, sDefer);
  const int sub = threadIdx.x & 7;
  const f32x4 w = *(const f32x4*)(a.w + sub * 4);
  const float bias = a.bias[0];
  f32x4 cf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(sCoef + (sub * 4 + e) * 4);
  const int64_t per = (a.vox + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)blockIdx.x * per, v1 = v0 + per < a.vox ? v0 + per : a.vox;
  // four voxels per trip: eight 16-byte loads in flight per thread (one voxel per trip streamed Dataset-3's 332 MB at 3.5 TB/s)
  for (int64_t vb = v0 + (threadIdx.x >> 3); vb < v1; vb += 128) {
    f32x4 h4[4], r4[4];
    float xv4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t v = vb + 32 * k < v1 ? vb + 32 * k : vb;  // (past the end: a repeat of the first, not stored)
      const int64_t i = (int64_t)b * a.vox + v;
      h4[k] = *(const f32x4*)(a.h + (size_t)i * 32 + sub * 4);
      r4[k] = *(const f32x4*)(a.res + (size_t)i * 32 + sub * 4);
      xv4[k] = (a.scal && sub == 0) ? a.x[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t v = vb + 32 * k;
      const int64_t i = (int64_t)b * a.vox + v;
      f32x4 h = h4[k];
      const f32x4 r = r4[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = cf[e][0] * h[e] + cf[e][1];
        u = u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));  // (gn_apply's SiLU)
        h[e] = u + cf[e][2] + r[e];
      }
      float p = (h[0] * w[0] + h[1] * w[1]) + (h[2] * w[2] + h[3] * w[3]);
      p += __shfl_xor(p, 1, 64);
      p += __shfl_xor(p, 2, 64);
      p += __shfl_xor(p, 4, 64);
      if (sub == 0 && v < v1) {
        float pred = p + bias;
        const float xv = xv4[k];
        if (a.scal) {
          if (a.objective == 0) pred = a.scal[b * 4 + 1] * xv + a.scal[b * 4 + 2] * pred;
          else if (a.objective == 1) pred = xv - a.scal[b * 4 + 3] * pred;
        }
        a.out[i] = pred;
        if (a.upd_stepvals) {  // (ddim_update_kernel's arithmetic)
          const float sigma = a.upd_stepvals[0], sprev = a.upd_stepvals[1], dsig = a.upd_stepvals[2], denom = a.upd_stepvals[3];
          const float eps = (a.x[i] - pred) / sigma;
          float r = pred + sprev * eps;
          if (a.upd_noise) r += dsig * a.upd_noise[i] / denom;
          a.upd_x_next[i] = r;
          if (a.upd_xs) a.upd_xs[i] = r;
          if (a.upd_x0s) a.upd_x0s[i] = pred;
        }
      }
    }
  }
}

void launch_head(const HeadArgs& a, hipStream_t s) {
  CD_REQUIRE(!a.upd_stepvals || (a.x && a.scal && a.upd_x_next), "head: the fused sampler update needs x, the scalings and x_next");
  if (a.defer.part) {
    CD_REQUIRE(a.defer.C == 32 && a.res, "head: the fused final block is 32 channels wide with an identity shortcut");
    prof::Scope scope("head_gn", s, 64.0 * a.batch * a.vox, 4.0 * a.batch * a.vox * 66);
    int per_sample = (int)((1024 + a.batch - 1) / a.batch);  // one round of ~1024 workgroups, each folds the GroupNorm once
    const int cap = (int)((a.vox + 255) / 256);
    if (per_sample > cap) per_sample = cap;
    if (per_sample < 1) per_sample = 1;
    hipLaunchKernelGGL(head_gn_kernel, dim3((unsigned)per_sample, (unsigned)a.batch), dim3(256), 0, s, a);
    CD_HIP(hipGetLastError());
    return;
  }
  const int64_t total = (int64_t)a.batch * a.vox;
  int64_t blocks = (total * 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  prof::Scope scope("head", s, 64.0 * total, 4.0 * total * 34);
  hipLaunchKernelGGL(head_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Sampler loop helpers (DDim.__call__, models/sample.py:72-107).  Per-step scalars live in device memory so that one
// captured step graph can be replayed for every iteration.
//   stepvals = {sigma, sigma_prev*[t>0], ddim_sigma, denom}
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) load_step_kernel(const float* __restrict__ table, int* counter, float* stepvals, float* sigma_b,
                                                         int batch, StepChunk ch) {
  const int step = *counter;
  const float* row = table + (size_t)step * 4;
  const int tid = threadIdx.x;
  if (tid < 4) stepvals[tid] = row[tid];
  const float sg = row[0];
  for (int i = tid; i < batch; i += blockDim.x) sigma_b[i] = sg;
  if (ch.chunk_steps) {  // this step's embeddings and scalings out of the chunk computed ahead (one workgroup: ~150 KB from L2)
    const int slot = step % ch.chunk_steps;
    const f32x4* es = (const f32x4*)(ch.emb_src + (size_t)slot * ch.emb_floats);
    f32x4* ed = (f32x4*)ch.emb_dst;
    const f32x4* ss = (const f32x4*)(ch.scal_src + (size_t)slot * ch.scal_floats);
    f32x4* sd = (f32x4*)ch.scal_dst;
    if (ch.emb_cond) {  // separable form: the step's time row + every sample's condition row
      const int rq = ch.emb_floats / 4;
      const f32x4* ec = (const f32x4*)ch.emb_cond;
      for (int i = tid; i < batch * rq; i += blockDim.x) ed[i] = es[i % rq] + ec[i];
      for (int i = tid; i < batch; i += blockDim.x) sd[i] = ss[0];
    } else {
      for (int i = tid; i < ch.emb_floats / 4; i += blockDim.x) ed[i] = es[i];
      for (int i = tid; i < ch.scal_floats / 4; i += blockDim.x) sd[i] = ss[i];
    }
  }
  __syncthreads();
  if (tid == 0) *counter = step + 1;
}
void launch_load_step(const float* table, int* counter, float* stepvals, float* sigma_b, int batch, hipStream_t s,
                      const StepChunk* chunk) {
  StepChunk ch;
  if (chunk) {
    ch = *chunk;
    CD_REQUIRE(ch.emb_floats % 4 == 0 && ch.scal_floats % 4 == 0, "internal: step chunk rows must be whole float4s");
  }
  hipLaunchKernelGGL(load_step_kernel, dim3(1), dim3(chunk ? 1024 : 256), 0, s, table, counter, stepvals, sigma_b, batch, ch);
  CD_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) ddim_update_kernel(const float* x, const float* __restrict__ x0,
                                                          const float* __restrict__ noise, const float* __restrict__ sv,
                                                          float* x_next, float* __restrict__ xs,
                                                          float* __restrict__ x0s, int64_t n) {
  const float sigma = sv[0], sprev = sv[1], dsig = sv[2], denom = sv[3];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i], x0v = x0[i];
    const float eps = (xv - x0v) / sigma;            // noise_pred (sample.py:90)
    float r = x0v + sprev * eps;                     // mask * sigma_prev * noise_pred (sample.py:104)
    if (noise) r += dsig * noise[i] / denom;
    x_next[i] = r;
    if (xs) xs[i] = r;
    if (x0s) x0s[i] = x0v;
  }
}
void launch_ddim_update(const float* x, const float* x0, const float* noise, const float* stepvals, float* x_next,
                        float* xs_slot, float* x0s_slot, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  prof::Scope scope("ddim_update", s, 0, 12.0 * n);
  hipLaunchKernelGGL(ddim_update_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, x0, noise, stepvals, x_next, xs_slot,
                     x0s_slot, n);
  CD_HIP(hipGetLastError());
}

// y = x * (*scale)   (x = start * sigma_start, sample.py:66)
__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ sc, int64_t n) {
  const float f = sc[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * f;
}
void launch_scale(const float* x, float* y, const float* sc, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, sc, n);
  CD_HIP(hipGetLastError());
}

__global__ void scale_imm_kernel(const float* __restrict__ x, float* __restrict__ y, float f, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * f;
}
void launch_scale_imm(const float* x, float* y, float scale, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_imm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, scale, n);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller unit normals.  Element i of a stream is a pure function of (seed, offset + i):
// counter = (offset+i)/4, lane = (offset+i)%4, so batch shards on different GPUs draw disjoint slices of one stream.
// (The reference's torch.randn CPU stream (mt19937) cannot be reproduced on device; parity tests pass noise in.)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// dev (optional): {seed, base offset, stride} in device memory and the sampler's step counter -- the stream position then is
// base + ((step - 1) * per_step + index) * stride, so that one captured step graph serves every step and every trajectory of a
// stochastic sampler
__global__ void __launch_bounds__(256) randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset,
                                                    const uint64_t* __restrict__ dev, const int* __restrict__ step_counter,
                                                    int per_step, int index) {
  if (dev) {
    seed = dev[0];
    offset = dev[1] + ((uint64_t)(*step_counter - 1) * (uint64_t)per_step + (uint64_t)index) * dev[2];
  }
  const uint64_t first = offset >> 2, last = (offset + (uint64_t)n + 3) >> 2;  // counter range [first, last)
  for (uint64_t ctr = first + (uint64_t)blockIdx.x * 256 + threadIdx.x; ctr < last; ctr += (uint64_t)gridDim.x * 256) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float u1 = ((float)(c[2 * p] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
      const float u2 = ((float)(c[2 * p + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float rad = sqrtf(-2.f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * p] = rad * cs;
      z[2 * p + 1] = rad * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint64_t g = ctr * 4 + e;
      if (g >= offset && g < offset + (uint64_t)n) out[g - offset] = z[e];
    }
  }
}
void launch_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (n <= 0) return;
  prof::Scope scope("randn", s, 0, 4.0 * n);  // (a one-workgroup draw doubles as the profiler's own per-launch overhead: bench.py)
  int64_t blocks = (n / 4 + 256) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, seed, offset, (const uint64_t*)nullptr,
                     (const int*)nullptr, 1, 0);
  CD_HIP(hipGetLastError());
}
void launch_randn_step(float* out, int64_t n, const uint64_t* seed_offset_stride_dev, const int* step_counter, hipStream_t s,
                       int per_step, int index) {
  if (n <= 0) return;
  int64_t blocks = (n / 4 + 256) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, (uint64_t)0, (uint64_t)0,
                     seed_offset_stride_dev, step_counter, per_step, index);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Generic sampler programs (cd_sampler_run): every per-step scalar is a column of row (*counter - 1) of a device table, so a
// step whose op list does not change is one captured graph replayed for the whole trajectory.
// ------------------------------------------------------------------------------------------------------------
__global__ void or_word_kernel(int* word, int bits) { atomicOr(word, bits); }
void launch_or_word(int* word, int bits, hipStream_t s) {
  hipLaunchKernelGGL(or_word_kernel, dim3(1), dim3(1), 0, s, word, bits);
  CD_HIP(hipGetLastError());
}
__global__ void step_advance_kernel(int* counter) { *counter = *counter + 1; }
void launch_step_advance(int* counter, hipStream_t s) {
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, counter);
  CD_HIP(hipGetLastError());
}
__global__ void fill_from_table_kernel(float* __restrict__ dst, int count, const float* __restrict__ table, int ncol, int col,
                                       const int* __restrict__ counter) {
  const float v = table[(size_t)(*counter - 1) * ncol + col];
  for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = v;
}
void launch_fill_from_table(float* dst, int count, const float* table, int ncol, int col, const int* step_counter, hipStream_t s) {
  hipLaunchKernelGGL(fill_from_table_kernel, dim3(1), dim3(256), 0, s, dst, count, table, ncol, col, step_counter);
  CD_HIP(hipGetLastError());
}
struct LincombSrc { const float* p[6]; };
__global__ void __launch_bounds__(256) lincomb_kernel(float* out, LincombSrc src, int nsrc, const float* __restrict__ table,
                                                      int ncol, int col, const int* __restrict__ counter, int64_t n) {
  float c[6];
  const float* row = table + (size_t)(*counter - 1) * ncol + col;
#pragma unroll
  for (int k = 0; k < 6; ++k) c[k] = k < nsrc ? row[k] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float acc = c[0] * src.p[0][i];
#pragma unroll
    for (int k = 1; k < 6; ++k)
      if (k < nsrc) acc += c[k] * src.p[k][i];
    out[i] = acc;
  }
}
void launch_lincomb(float* out, const float* const* src, int nsrc, const float* table, int ncol, int col, const int* step_counter,
                    int64_t n, hipStream_t s) {
  CD_REQUIRE(nsrc >= 1 && nsrc <= 6, "lincomb: 1..6 terms");
  LincombSrc ls{};
  for (int k = 0; k < 6; ++k) ls.p[k] = src[k < nsrc ? k : 0];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, ls, nsrc, table, ncol, col, step_counter, n);
  CD_HIP(hipGetLastError());
}
// The same in the operation order of a chain of torch elementwise kernels (CD_SOP_LINDIV): every product and every sum is
// rounded to fp32 on its own (no contraction into fused multiply-adds), the terms are added left to right and the result is
// divided (IEEE, correctly rounded) by the coefficient that follows the terms.  DPM-Solver's second- and third-order steps
// amplify the rounding of their intermediate states by sigma_max / sigma_mid (utils/sampling.py:419-456), so matching the
// reference there means rounding where it rounds.
__global__ void __launch_bounds__(256) lincomb_div_kernel(float* out, LincombSrc src, int nsrc, const float* __restrict__ table,
                                                          int ncol, int col, const int* __restrict__ counter, int64_t n) {
#pragma clang fp contract(off)
  float c[6];
  const float* row = table + (size_t)(*counter - 1) * ncol + col;
#pragma unroll
  for (int k = 0; k < 6; ++k) c[k] = k < nsrc ? row[k] : 0.f;
  const float div = row[nsrc];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // (plain operators: the pragma above governs them, while the __f*_rn helpers of the HIP headers carry the contraction
    // flags of the translation unit and come out as v_fmac)
    float acc = c[0] * src.p[0][i];
#pragma unroll
    for (int k = 1; k < 6; ++k)
      if (k < nsrc) {
        const float prod = c[k] * src.p[k][i];
        acc = acc + prod;
      }
    out[i] = acc / div;
  }
}
void launch_lincomb_div(float* out, const float* const* src, int nsrc, const float* table, int ncol, int col,
                        const int* step_counter, int64_t n, hipStream_t s) {
  CD_REQUIRE(nsrc >= 1 && nsrc <= 6, "lincomb: 1..6 terms");
  LincombSrc ls{};
  for (int k = 0; k < 6; ++k) ls.p[k] = src[k < nsrc ? k : 0];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lincomb_div_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, ls, nsrc, table, ncol, col, step_counter, n);
  CD_HIP(hipGetLastError());
}
__global__ void __launch_bounds__(256) record_step_kernel(float* __restrict__ traj, const float* __restrict__ src,
                                                          const int* __restrict__ counter, int64_t n) {
  float* dst = traj + (size_t)(*counter - 1) * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
void launch_record_step(float* traj, const float* src, const int* step_counter, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(record_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, traj, src, step_counter, n);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Training forward: x_noisy = data + sigma*noise (loss.py:169) and the weighted L2 reduction (loss.py:103-104,176)
// ------------------------------------------------------------------------------------------------------------
__global__ void axpy_sigma_kernel(const float* __restrict__ data, const float* __restrict__ noise,
                                  const float* __restrict__ sigma_b, float* __restrict__ out, int64_t per) {
  const int b = blockIdx.y;
  const float sg = sigma_b[b];
  const size_t base = (size_t)b * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256)
    out[base + i] = data[base + i] + sg * noise[base + i];
}
void launch_axpy_sigma(const float* data, const float* noise, const float* sigma_b, float* out, int batch, int64_t per,
                       hipStream_t s) {
  int64_t bx = (per + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(axpy_sigma_kernel, dim3((unsigned)bx, batch), dim3(256), 0, s, data, noise, sigma_b, out, per);
  CD_HIP(hipGetLastError());
}

// per-sample sum of the element loss of Loss._loss (models/loss.py:97-116) in fp64, one block per sample: 0 'l2' and 2 'mse':
// d^2; 1 'l1': |d|; 3 'huber' = torch smooth_l1_loss, beta 1: d^2 / 2 below |d| = 1, |d| - 1/2 above
__global__ void __launch_bounds__(256) loss_partial_kernel(const float* __restrict__ x0, const float* __restrict__ data,
                                                           const float* __restrict__ noise, const float* __restrict__ sigma_b,
                                                           double* __restrict__ partial, int64_t per, int loss_type, int objective) {
  __shared__ double sh[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)b * per;
  const float sg = sigma_b[b];
  double acc = 0.0;
  for (int64_t i = tid; i < per; i += 256) {
    const float d = objective_residual(objective, x0[base + i], data[base + i], objective == 1 ? noise[base + i] : 0.f, sg);
    const float ad = fabsf(d);
    const float e = loss_type == 1 ? ad : (loss_type == 3 ? (ad < 1.f ? 0.5f * d * d : ad - 0.5f) : d * d);
    acc += (double)e;
  }
  sh[tid] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) sh[tid] += sh[tid + st];
    __syncthreads();
  }
  if (tid == 0) partial[b] = sh[0];
}
void launch_loss_partial(const float* x0, const float* data, const float* noise, const float* sigma_b, double* partial, int batch,
                         int64_t per, hipStream_t s, int loss_type, int objective) {
  hipLaunchKernelGGL(loss_partial_kernel, dim3(batch), dim3(256), 0, s, x0, data, noise, sigma_b, partial, per, loss_type, objective);
  CD_HIP(hipGetLastError());
}
// loss = sum_b w_b * partial[b] / (mean_b(w_b) * B * per)
// (only 'l2' carries the weight; the torch.nn.functional losses of the other types are plain means, loss.py:106-111)
__global__ void loss_final_kernel(const double* __restrict__ partial, const float* __restrict__ sigma_b, double* loss,
                                  int batch, int64_t per, int loss_type, int objective) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double num = 0.0, wsum = 0.0;
    for (int b = 0; b < batch; ++b) {
      const float w = objective_weight(objective, loss_type, sigma_b[b]);
      num += (double)w * partial[b];
      wsum += (double)w;
    }
    loss[0] = num / ((wsum / batch) * (double)batch * (double)per);
  }
}
void launch_loss_final(const double* partial, const float* sigma_b, double* loss, int batch, int64_t per, hipStream_t s,
                       int loss_type, int objective) {
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, partial, sigma_b, loss, batch, per, loss_type, objective);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// layout transposes (tests / generic unet_forward input only; the denoise path never transposes: its I/O has C = 1)
// ------------------------------------------------------------------------------------------------------------
__global__ void to_cl_kernel(const float* __restrict__ src, float* __restrict__ dst, int channels, int64_t vox) {
  const int b = blockIdx.y;
  const int64_t total = vox * channels;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % channels);
    const int64_t v = i / channels;
    dst[(size_t)b * total + i] = src[(size_t)b * total + (size_t)c * vox + v];
  }
}
__global__ void to_planar_kernel(const float* __restrict__ src, float* __restrict__ dst, int channels, int64_t vox) {
  const int b = blockIdx.y;
  const int64_t total = vox * channels;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i % vox;
    const int c = (int)(i / vox);
    dst[(size_t)b * total + i] = src[(size_t)b * total + (size_t)v * channels + c];
  }
}
void launch_transpose_to_cl(const float* ncdhw, float* ndhwc, int batch, int channels, int64_t vox, hipStream_t s) {
  int64_t bx = (vox * channels + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(to_cl_kernel, dim3((unsigned)bx, batch), dim3(256), 0, s, ncdhw, ndhwc, channels, vox);
  CD_HIP(hipGetLastError());
}
void launch_transpose_to_planar(const float* ndhwc, float* ncdhw, int batch, int channels, int64_t vox, hipStream_t s) {
  int64_t bx = (vox * channels + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(to_planar_kernel, dim3((unsigned)bx, batch), dim3(256), 0, s, ndhwc, ncdhw, channels, vox);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Adam step over many parameter tensors in one launch per 48 tensors (torch.optim.Adam as the reference's training loop
// builds it, train/train.py:144: no amsgrad, optional L2 weight decay), same element-wise formulas and operation order as
// torch's implementation:  m <- m + (1-b1)(g - m);  v <- b2 v + (1-b2) g g;  p <- p - (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(AdamChunk c, float w1, float beta2, float w2, float eps, float weight_decay,
                                                   float step_size, float bc2_sqrt) {
  const int t = blockIdx.y;
  float* __restrict__ p = c.p[t];
  const float* __restrict__ g = c.g[t];
  float* __restrict__ m = c.m[t];
  float* __restrict__ v = c.v[t];
  const int64_t n = c.n[t];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i];
    const float pi = p[i];
    if (weight_decay != 0.f) gi = gi + weight_decay * pi;
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * beta2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

void launch_adam(const AdamChunk& c, int ntensors, int64_t max_numel, double lr, double beta1, double beta2, float eps,
                 float weight_decay, int step, hipStream_t s) {
  const double bc1 = 1.0 - std::pow(beta1, step), bc2 = 1.0 - std::pow(beta2, step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)std::sqrt(bc2);
  int64_t bx = (max_numel + 256 * 4 - 1) / (256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  // 1 - beta in double like torch (python floats), then rounded once to fp32
  const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)bx, (unsigned)ntensors), dim3(256), 0, s, c, w1, (float)beta2, w2, eps, weight_decay, step_size,
                     bc2_sqrt);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Inverse pre-processing of generated showers (ReverseNormCaloChall, calodiffusion/utils/utils.py:446-573, for the regular
// grids: dataset_num 2 / 3, showerMap 'layer-logit-norm' / 'logit-norm'): un-normalise, inverse logit, (layer mode) clamp
// negatives and rescale every calorimeter layer to the layer energy given by the conditioning vector, scale to the incident
// energy, apply the read-out threshold.  One workgroup per (sample, layer z): the layer sum is a workgroup reduction.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rev_logit(float x, float alpha) {  // utils.py:233-237 (alpha 1e-6); HGCal_utils.py:13-17 (alpha 1e-8): always the caller's
  const float ex = expf(x);
  const float o = ex / (1.f + ex);
  return (o - alpha) / (1.f - 2.f * alpha);
}

__global__ void __launch_bounds__(256) reverse_norm_kernel(ReverseNormArgs a) {
  __shared__ float red[256];
  __shared__ float s_layer;
  const int b = blockIdx.y, z = blockIdx.x, tid = threadIdx.x;
  const int PV = a.H * a.W;
  float layer_e = 0.f;
  if (a.layer_mode) {
    // this sample's layer energies: reverse transform, normalise to the total deposited energy (utils.py:519-528)
    const float* le = a.layerE + (size_t)b * (a.D + 1);
    float part = 0.f;
    for (int i = tid; i < a.D; i += 256) part += rev_logit(le[1 + i] * a.layers_std + a.layers_mean, a.alpha);
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) {
      const float total = le[0] * a.totalE_std + a.totalE_mean;
      s_layer = rev_logit(le[1 + z] * a.layers_std + a.layers_mean, a.alpha) / red[0] * total;
    }
    __syncthreads();
    layer_e = s_layer;
    __syncthreads();
  }
  const float* v = a.voxels + ((size_t)b * a.D + z) * PV;
  float* out = a.out + ((size_t)b * a.D + z) * PV;
  if (a.stage == 1) {  // un-normalise + inverse logit only: what the HGCal / Dataset-1 variants do BEFORE their geometry decode
    for (int i = tid; i < PV; i += 256) out[i] = rev_logit(v[i] * a.logit_std + a.logit_mean, a.alpha);
    return;
  }
  float part = 0.f;
  for (int i = tid; i < PV; i += 256) {
    // stage 2: the input is already in deposited-energy-fraction space (the decoded showers)
    float d = a.stage == 2 ? v[i] : rev_logit(v[i] * a.logit_std + a.logit_mean, a.alpha);
    if (a.layer_mode) d = d < 0.f ? 0.f : d;
    out[i] = d;
    part += d;
  }
  float fac = 1.f;
  if (a.layer_mode) {
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const float prev = red[0];
    fac = layer_e / (prev + 1e-10f);
    if (layer_e < a.layer_eps || prev < a.layer_eps) fac = 1.f;
  }
  const float en = a.energy[b];
  for (int i = tid; i < PV; i += 256) {
    float d = out[i] * fac * a.max_deposit * en;
    if (a.ecut > 0.f && d < a.ecut) d = 0.f;
    out[i] = d;
  }
}

void launch_reverse_norm(const ReverseNormArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(reverse_norm_kernel, dim3((unsigned)a.D, (unsigned)a.batch), dim3(256), 0, s, a);
  CD_HIP(hipGetLastError());
}

}  // namespace cd
