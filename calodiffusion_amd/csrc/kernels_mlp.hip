// Layer-energy model of LayerDiffusion: the conditional residual MLP ("ResNet", reference calodiffusion/models/models.py:373-457)
// with its EDM pre-conditioning (calodiffusion.py:154-169) and the whole DDim/DDPM/Euler trajectory (models/sample.py:40-110) in
// ONE launch.  A (B, D+1) vector per shower is tiny and the samples are independent, so one workgroup owns one sample and walks
// all sampler steps with the state in LDS: no per-step launches, no intermediate in HBM.  The ~0.7 M weights (2.7 MB) stream
// from L2 every step; a wave owns 8 output rows at a time so 8 independent 1-KiB row reads are in flight per wave
// (16 rows x 16 waves measured 3x slower: register pressure).
#include "cd_common.h"

#include <cmath>

namespace cd {

namespace {

__device__ __forceinline__ float mlp_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

constexpr int MLP_ROWS = 8;

// out[j] = act(bias[j] + sum_k w[j][k] in[k]) + (post ? post[j] : 0).  `out` may alias `post`, never `in`.
__device__ void mlp_dense(const float* __restrict__ w, const float* __restrict__ bias, const float* in, float* out,
                          const float* post, int nin, int nout, bool gelu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool vec = (nin & 3) == 0;
  for (int j0 = wave * MLP_ROWS; j0 < nout; j0 += nw * MLP_ROWS) {
    float acc[MLP_ROWS];
#pragma unroll
    for (int r = 0; r < MLP_ROWS; ++r) acc[r] = 0.f;
    if (vec) {
      for (int k4 = lane; k4 * 4 < nin; k4 += 64) {
        const f32x4 xv = *(const f32x4*)(in + k4 * 4);
        f32x4 wv[MLP_ROWS];
#pragma unroll
        for (int r = 0; r < MLP_ROWS; ++r) {
          const int j = j0 + r < nout ? j0 + r : nout - 1;  // clamped: the load stays unconditional, the row is dropped below
          wv[r] = *(const f32x4*)(w + (size_t)j * nin + k4 * 4);
        }
#pragma unroll
        for (int r = 0; r < MLP_ROWS; ++r)
          acc[r] = fmaf(wv[r][3], xv[3], fmaf(wv[r][2], xv[2], fmaf(wv[r][1], xv[1], fmaf(wv[r][0], xv[0], acc[r]))));
      }
    } else {
      for (int k = lane; k < nin; k += 64) {
        const float xv = in[k];
#pragma unroll
        for (int r = 0; r < MLP_ROWS; ++r) {
          const int j = j0 + r < nout ? j0 + r : nout - 1;
          acc[r] = fmaf(w[(size_t)j * nin + k], xv, acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < MLP_ROWS; ++r) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
    }
    if (lane < MLP_ROWS && j0 + lane < nout) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < MLP_ROWS; ++r)
        if (lane == r) v = acc[r];
      v += bias[j0 + lane];
      if (gelu) v = mlp_gelu(v);
      if (post) v += post[j0 + lane];
      out[j0 + lane] = v;
    }
  }
  __syncthreads();
}

constexpr int MLP_MAXV = 256;  // dim_in, cond_emb
constexpr int MLP_MAXH = 512;  // hidden

}  // namespace

__global__ void __launch_bounds__(512) layer_mlp_kernel(LayerMlpArgs a) {
  __shared__ __attribute__((aligned(16))) float xs_[MLP_MAXV], xin[MLP_MAXV], cat[MLP_MAXV], gcat[MLP_MAXV], pred[MLP_MAXV];
  __shared__ __attribute__((aligned(16))) float h0[MLP_MAXH], h1[MLP_MAXH], emb[MLP_MAXH], tb0[MLP_MAXV], tb1[MLP_MAXV];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int dim = a.dim_in, half = a.cond_emb / 2, q = half / 2;
  const float* const* W = a.w;
  // weight order = the reference module's state_dict order: time_mlp (3 Linear), cond_mlp (3), in_lay, per block
  // {embeder, dense1, dense2}, out_lay; each as (weight, bias)
  const float sd = a.sigma_data;

  float scale0 = 1.f;
  if (a.mode == 2) scale0 = a.table[0];  // x = start * sigma_start (sample.py:66)
  for (int i = tid; i < dim; i += blockDim.x) xs_[i] = a.x[(size_t)b * dim + i] * scale0;
  for (int i = tid; i < a.cond_size; i += blockDim.x) tb0[i] = a.cond[(size_t)b * a.cond_size + i];
  __syncthreads();
  // cond branch (constant over the trajectory): Linear(cond_size, q) GELU Linear(q, half) GELU Linear(half, half) -> cat[0:half]
  mlp_dense(W[6], W[7], tb0, tb1, nullptr, a.cond_size, q, true);
  mlp_dense(W[8], W[9], tb1, tb0, nullptr, q, half, true);
  mlp_dense(W[10], W[11], tb0, cat, nullptr, half, half, false);

  const int n_steps = a.mode == 2 ? a.n_steps : 1;
  for (int step = 0; step < n_steps; ++step) {
    float sigma = 0.f, t_in, c_in = 1.f, c_skip = 0.f, c_out = 1.f;
    if (a.mode == 0) {
      t_in = a.tsig[b];  // raw forward: the caller has applied the time embedding (ResNet.forward, models.py:444)
    } else {
      sigma = a.mode == 2 ? a.table[(size_t)step * 4] : a.tsig[b];
      t_in = a.time_kind == 0 ? 0.5f * logf(sigma) : a.time_kind == 1 ? sigma / sqrtf(1.f + sigma * sigma) : sigma;
      const float s2 = sigma * sigma + sd * sd;
      c_in = 1.f / sqrtf(s2);
      c_skip = sd * sd / s2;
      c_out = sigma * sd / sqrtf(s2);
    }
    if (tid == 0) tb0[0] = t_in;
    for (int i = tid; i < dim; i += blockDim.x) xin[i] = xs_[i] * c_in;
    __syncthreads();
    // time branch: Unflatten, Linear(1, q) GELU Linear(q, half) GELU Linear(half, half) -> cat[half:2 half]
    mlp_dense(W[0], W[1], tb0, tb1, nullptr, 1, q, true);
    mlp_dense(W[2], W[3], tb1, tb0, nullptr, q, half, true);
    mlp_dense(W[4], W[5], tb0, cat + half, nullptr, half, half, false);
    for (int i = tid; i < 2 * half; i += blockDim.x) gcat[i] = mlp_gelu(cat[i]);  // ResDense.embeder[0]
    mlp_dense(W[12], W[13], xin, h0, nullptr, dim, a.hidden, false);  // in_lay (its barrier also publishes gcat)
    for (int r = 0; r < a.n_res; ++r) {
      const float* const* L = W + 14 + 6 * r;
      mlp_dense(L[0], L[1], gcat, emb, nullptr, 2 * half, a.hidden, false);  // embed = Linear(GELU(cond))
      mlp_dense(L[2], L[3], h0, h1, emb, a.hidden, a.hidden, true);          // h = GELU(dense1(x)) + embed
      mlp_dense(L[4], L[5], h1, h0, h0, a.hidden, a.hidden, true);           // x = GELU(dense2(h)) + x
    }
    mlp_dense(W[14 + 6 * a.n_res], W[15 + 6 * a.n_res], h0, pred, nullptr, a.hidden, dim, false);  // out_lay
    if (a.mode == 0) {
      for (int i = tid; i < dim; i += blockDim.x) a.out[(size_t)b * dim + i] = pred[i];
      return;
    }
    float sprev = 0.f, dsig = 0.f, denom = 1.f;
    if (a.mode == 2) {
      sprev = a.table[(size_t)step * 4 + 1];
      dsig = a.table[(size_t)step * 4 + 2];
      denom = a.table[(size_t)step * 4 + 3];
    }
    for (int i = tid; i < dim; i += blockDim.x) {
      const float xv = xs_[i], p = pred[i];
      float x0 = p;  // mean_pred
      if (a.objective == 0) x0 = c_skip * xv + c_out * p;
      else if (a.objective == 1) x0 = xv - sigma * p;
      if (a.mode == 1) {
        a.out[(size_t)b * dim + i] = x0;
      } else {
        const float eps = (xv - x0) / sigma;  // sample.py:90
        float r = x0 + sprev * eps;           // sample.py:104 (sigma_prev already carries the t > 0 mask)
        const size_t o = ((size_t)step * a.batch + b) * dim + i;
        if (a.noise) r += dsig * a.noise[o] / denom;
        xs_[i] = r;
        if (a.xs) a.xs[o] = r;
        if (a.x0s) a.x0s[o] = x0;
        if (step == n_steps - 1) a.out[(size_t)b * dim + i] = r;
      }
    }
    __syncthreads();
  }
}

void launch_layer_mlp(const LayerMlpArgs& a, hipStream_t s) {
  CD_REQUIRE(a.dim_in >= 1 && a.dim_in <= MLP_MAXV && a.cond_emb >= 4 && a.cond_emb <= MLP_MAXV && (a.cond_emb & 3) == 0 &&
                 a.hidden >= 1 && a.hidden <= MLP_MAXH && a.cond_size >= 1 && a.cond_size <= MLP_MAXV && a.n_res >= 0 &&
                 a.n_res <= 8,
             "layer MLP: dim_in / cond_emb / cond_size up to 256, hidden up to 512, at most 8 residual blocks");
  // algorithmic work per sample and step: the dense layers' multiply-adds; the weights are the traffic (L2-resident)
  const double half = a.cond_emb / 2, q = half / 2;
  const double macs = q + q * half + half * half + (double)a.dim_in * a.hidden * 2 +
                      a.n_res * ((double)a.cond_emb * a.hidden + 2.0 * a.hidden * a.hidden);
  const int n_steps = a.mode == 2 ? a.n_steps : 1;
  prof::Scope scope("layer_mlp", s, 2.0 * macs * a.batch * n_steps, 4.0 * macs);
  hipLaunchKernelGGL(layer_mlp_kernel, dim3(a.batch), dim3(512), 0, s, a);
  CD_HIP(hipGetLastError());
}

}  // namespace cd
