// Training step of LayerDiffusion's layer-energy model (the conditional residual MLP "ResNet", reference
// calodiffusion/models/models.py:373-457, trained through LayerDiffusion.compute_loss in the layer state,
// models/layerdiffusion.py:52-57, with the hybrid_weight / l2 loss, models/loss.py:103-104,118-142,163-179):
//   x = data + sigma * noise;  x0 = c_skip x + c_out F(c_in x, cond, t(sigma));  L = sum_b w_b sum_i (x0 - data)^2 / (mean(w) B D)
// One workgroup per sample runs the forward with every activation in LDS, then the explicit chain rule back to the per-layer
// output deltas; inputs and deltas of the 20 Linear layers go to a per-sample tape in HBM and linear_wgrad_kernel
// (kernels_bwd.hip) forms all weight / bias gradients (sums over the batch, fixed order => deterministic) in one launch.
#include "cd_common.h"

#include <cmath>
#include <vector>

namespace cd {

namespace {

__device__ __forceinline__ float tm_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float tm_gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// pre[j] = bias[j] + sum_k w[j][k] in[k]; a wave owns 8 rows at a time (8 weight rows in flight)
__device__ void tm_dense(const float* __restrict__ w, const float* __restrict__ bias, const float* in, float* pre, int nin,
                         int nout) {
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
      for (int r = 0; r < R; ++r) wv[r] = w[(size_t)min(j0 + r, nout - 1) * nin + k];
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
      pre[j0 + lane] = v + bias[j0 + lane];
    }
  }
  __syncthreads();
}
// din[k] = sum_j w[j][k] dout[j]   (thread per k: coalesced rows, 8 in flight)
__device__ void tm_dense_T(const float* __restrict__ w, const float* dout, float* din, int nin, int nout) {
  for (int k = threadIdx.x; k < nin; k += blockDim.x) {
    float acc = 0.f;
    int j = 0;
    for (; j + 8 <= nout; j += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(j + u) * nin + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(wv[u], dout[j + u], acc);
    }
    for (; j < nout; ++j) acc = fmaf(w[(size_t)j * nin + k], dout[j], acc);
    din[k] = acc;
  }
  __syncthreads();
}

constexpr int TM_MAXV = 256, TM_MAXH = 512, TM_MAXR = 8;

}  // namespace

// tape row (floats) of one sample: inputs and output deltas of every Linear layer
LayerTapeLayout layer_tape_layout(int dim, int hidden, int cond_emb, int cond_size, int n_res) {
  LayerTapeLayout L;
  const int half = cond_emb / 2, q = half / 2;
  int o = 0;
  auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
  L.xin = take(dim); L.t_in = take(1); L.a1t = take(q); L.a2t = take(half); L.cin = take(cond_size); L.a1c = take(q);
  L.a2c = take(half); L.g = take(2 * half);
  for (int r = 0; r < n_res; ++r) { L.hprev[r] = take(hidden); L.h1[r] = take(hidden); }
  L.hfin = take(hidden);
  L.dpred = take(dim);
  for (int r = 0; r < n_res; ++r) { L.dv[r] = take(hidden); L.du[r] = take(hidden); L.de[r] = take(hidden); }
  L.dh0 = take(hidden);
  L.d3t = take(half); L.d2t = take(half); L.d1t = take(q); L.d3c = take(half); L.d2c = take(half); L.d1c = take(q);
  L.total = o;
  return L;
}

__global__ void __launch_bounds__(512) layer_mlp_train_kernel(LayerMlpTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float xs_[TM_MAXV], x0s[TM_MAXV], vecA[TM_MAXV], vecB[TM_MAXV], cat[TM_MAXV], gcat[TM_MAXV];
  __shared__ __attribute__((aligned(16))) float p1t[TM_MAXV], p2t[TM_MAXV], p1c[TM_MAXV], p2c[TM_MAXV], dg[TM_MAXV], dvec[TM_MAXV];
  __shared__ __attribute__((aligned(16))) float h[TM_MAXH], pu[TM_MAXR][TM_MAXH], pv[TM_MAXR][TM_MAXH], emb[TM_MAXH], tmp[TM_MAXH];
  __shared__ __attribute__((aligned(16))) float dh[TM_MAXH], dtmp[TM_MAXH];
  __shared__ double sred[8];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int dim = a.dim_in, Hd = a.hidden, half = a.cond_emb / 2, q = half / 2, R = a.n_res;
  const float* const* W = a.w;
  const LayerTapeLayout L = a.layout;
  float* T = a.tape + (size_t)b * L.total;

  // ---- loss weights: w_b = 1 + sigma^-2, mean over the batch in a fixed order (every workgroup repeats it) ----------------
  const float sigma = a.sigma[b], sd = a.sigma_data;
  double wsum = 0.0;
  for (int n = 0; n < a.batch; ++n) {
    const float sg = a.sigma[n];
    wsum += (double)(1.0f + 1.0f / (sg * sg));
  }
  const float wmean = (float)(wsum / a.batch);
  const float wb = 1.0f + 1.0f / (sigma * sigma);
  const float s2 = sigma * sigma + sd * sd;
  const float c_in = 1.f / sqrtf(s2), c_skip = sd * sd / s2, c_out = sigma * sd / sqrtf(s2);
  const float t_in = a.time_kind == 0 ? 0.5f * logf(sigma) : a.time_kind == 1 ? sigma / sqrtf(1.f + sigma * sigma) : sigma;

  // ---- forward ------------------------------------------------------------------------------------------------------------
  for (int i = tid; i < dim; i += nt) {
    const float x = a.data[(size_t)b * dim + i] + sigma * a.noise[(size_t)b * dim + i];
    xs_[i] = x;
    vecA[i] = x * c_in;
    T[L.xin + i] = x * c_in;
  }
  __syncthreads();
  tm_dense(W[12], W[13], vecA, h, dim, Hd);  // in_lay
  // time branch
  if (tid == 0) { vecA[0] = t_in; T[L.t_in] = t_in; }
  __syncthreads();
  tm_dense(W[0], W[1], vecA, p1t, 1, q);
  for (int i = tid; i < q; i += nt) { vecB[i] = tm_gelu(p1t[i]); T[L.a1t + i] = vecB[i]; }
  __syncthreads();
  tm_dense(W[2], W[3], vecB, p2t, q, half);
  for (int i = tid; i < half; i += nt) { vecA[i] = tm_gelu(p2t[i]); T[L.a2t + i] = vecA[i]; }
  __syncthreads();
  tm_dense(W[4], W[5], vecA, cat + half, half, half);
  // cond branch
  for (int i = tid; i < a.cond_size; i += nt) { vecA[i] = a.cond[(size_t)b * a.cond_size + i]; T[L.cin + i] = vecA[i]; }
  __syncthreads();
  tm_dense(W[6], W[7], vecA, p1c, a.cond_size, q);
  for (int i = tid; i < q; i += nt) { vecB[i] = tm_gelu(p1c[i]); T[L.a1c + i] = vecB[i]; }
  __syncthreads();
  tm_dense(W[8], W[9], vecB, p2c, q, half);
  for (int i = tid; i < half; i += nt) { vecA[i] = tm_gelu(p2c[i]); T[L.a2c + i] = vecA[i]; }
  __syncthreads();
  tm_dense(W[10], W[11], vecA, cat, half, half);
  for (int i = tid; i < 2 * half; i += nt) { gcat[i] = tm_gelu(cat[i]); T[L.g + i] = gcat[i]; }
  __syncthreads();
  for (int r = 0; r < R; ++r) {
    const float* const* Lw = W + 14 + 6 * r;
    for (int i = tid; i < Hd; i += nt) T[L.hprev[r] + i] = h[i];
    tm_dense(Lw[0], Lw[1], gcat, emb, 2 * half, Hd);  // embed = Linear(GELU(cond))
    tm_dense(Lw[2], Lw[3], h, pu[r], Hd, Hd);          // u = dense1(h)
    for (int i = tid; i < Hd; i += nt) { tmp[i] = tm_gelu(pu[r][i]) + emb[i]; T[L.h1[r] + i] = tmp[i]; }
    __syncthreads();
    tm_dense(Lw[4], Lw[5], tmp, pv[r], Hd, Hd);        // v = dense2(h1)
    for (int i = tid; i < Hd; i += nt) h[i] = tm_gelu(pv[r][i]) + h[i];
    __syncthreads();
  }
  for (int i = tid; i < Hd; i += nt) T[L.hfin + i] = h[i];
  tm_dense(W[14 + 6 * R], W[15 + 6 * R], h, x0s, Hd, dim);  // pred (out_lay)

  // ---- loss and d pred ----------------------------------------------------------------------------------------------------
  // the element losses of Loss._loss (models/loss.py:97-116) as in head_loss_bwd_kernel: only 'l2' carries the hybrid weight; the
  // torch.nn.functional losses are plain means -- d loss / d x0 = gscale f'(d): l2 2 w d / (mean(w) N), mse 2 d / N,
  // l1 sign(d) / N, huber (smooth_l1, beta 1) clamp(d, -1, 1) / N
  const int lt = a.loss_type;
  const float gscale = lt == 0 ? 2.f * wb / (wmean * (float)a.batch * (float)dim) : (lt == 2 ? 2.f : 1.f) / ((float)a.batch * (float)dim);
  double lacc = 0.0;
  for (int i = tid; i < dim; i += nt) {
    const float x0 = c_skip * xs_[i] + c_out * x0s[i];
    const float d = x0 - a.data[(size_t)b * dim + i];
    const float ad = fabsf(d);
    lacc += (double)(lt == 1 ? ad : (lt == 3 ? (ad < 1.f ? 0.5f * d * d : ad - 0.5f) : d * d));
    const float fp = lt == 1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : (lt == 3 ? fminf(fmaxf(d, -1.f), 1.f) : d);
    dvec[i] = gscale * fp * c_out;
    T[L.dpred + i] = dvec[i];
  }
  for (int o = 32; o > 0; o >>= 1) lacc += __shfl_xor(lacc, o, 64);
  if ((tid & 63) == 0) sred[tid >> 6] = lacc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < (nt >> 6); ++i) s += sred[i];
    a.loss_part[b] = lt == 0 ? s * (double)wb : s;
  }

  // ---- backward -----------------------------------------------------------------------------------------------------------
  tm_dense_T(W[14 + 6 * R], dvec, dh, Hd, dim);  // d h_final
  for (int i = tid; i < 2 * half; i += nt) dg[i] = 0.f;
  __syncthreads();
  for (int r = R - 1; r >= 0; --r) {
    const float* const* Lw = W + 14 + 6 * r;
    for (int i = tid; i < Hd; i += nt) { dtmp[i] = dh[i] * tm_gelu_grad(pv[r][i]); T[L.dv[r] + i] = dtmp[i]; }  // dv
    __syncthreads();
    tm_dense_T(Lw[4], dtmp, tmp, Hd, Hd);  // d h1 (= d embed)
    for (int i = tid; i < Hd; i += nt) {
      T[L.de[r] + i] = tmp[i];
      dtmp[i] = tmp[i] * tm_gelu_grad(pu[r][i]);  // du
      T[L.du[r] + i] = dtmp[i];
    }
    __syncthreads();
    tm_dense_T(Lw[0], tmp, emb, 2 * half, Hd);  // d g += We^T de
    for (int i = tid; i < 2 * half; i += nt) dg[i] += emb[i];
    tm_dense_T(Lw[2], dtmp, tmp, Hd, Hd);       // W1^T du
    for (int i = tid; i < Hd; i += nt) dh[i] += tmp[i];  // residual + dense1 path
    __syncthreads();
  }
  for (int i = tid; i < Hd; i += nt) T[L.dh0 + i] = dh[i];
  // d cat = d g * gelu'(cat): first half = cond branch output, second half = time branch output
  for (int i = tid; i < 2 * half; i += nt) dg[i] *= tm_gelu_grad(cat[i]);
  __syncthreads();
  for (int i = tid; i < half; i += nt) { T[L.d3c + i] = dg[i]; T[L.d3t + i] = dg[half + i]; }
  tm_dense_T(W[4], dg + half, vecA, half, half);  // time branch
  for (int i = tid; i < half; i += nt) { vecA[i] *= tm_gelu_grad(p2t[i]); T[L.d2t + i] = vecA[i]; }
  __syncthreads();
  tm_dense_T(W[2], vecA, vecB, q, half);
  for (int i = tid; i < q; i += nt) T[L.d1t + i] = vecB[i] * tm_gelu_grad(p1t[i]);
  __syncthreads();
  tm_dense_T(W[10], dg, vecA, half, half);  // cond branch
  for (int i = tid; i < half; i += nt) { vecA[i] *= tm_gelu_grad(p2c[i]); T[L.d2c + i] = vecA[i]; }
  __syncthreads();
  tm_dense_T(W[8], vecA, vecB, q, half);
  for (int i = tid; i < q; i += nt) T[L.d1c + i] = vecB[i] * tm_gelu_grad(p1c[i]);
}

__global__ void layer_loss_final_kernel(const double* __restrict__ part, const float* __restrict__ sigma, int batch, int dim,
                                        double* __restrict__ loss, int loss_type) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0, w = 0.0;
  for (int n = 0; n < batch; ++n) {
    s += part[n];
    const float sg = sigma[n];
    w += (double)(1.0f + 1.0f / (sg * sg));
  }
  const float wmean = loss_type == 0 ? (float)(w / batch) : 1.0f;
  *loss = s / ((double)wmean * (double)batch * (double)dim);
}

size_t layer_train_workspace_bytes(const LayerMlpTrainArgs& a) {
  return ((size_t)a.batch * a.layout.total * 4 + 255) / 256 * 256 + (size_t)a.batch * 8 + 256 + 64 * sizeof(LinearWgradJob) + 256;
}

// grads: flat buffer with the parameters' numel in state_dict order (weight, bias, weight, bias, ...)
void launch_layer_mlp_train(LayerMlpTrainArgs a, float* grads, double* loss_out, void* workspace, hipStream_t s) {
  CD_REQUIRE(a.dim_in >= 1 && a.dim_in <= TM_MAXV && a.cond_emb >= 4 && a.cond_emb <= TM_MAXV && (a.cond_emb & 3) == 0 &&
                 a.hidden >= 1 && a.hidden <= TM_MAXH && a.cond_size >= 1 && a.cond_size <= TM_MAXV && a.n_res >= 0 &&
                 a.n_res <= TM_MAXR,
             "layer MLP training: dim_in / cond_emb / cond_size up to 256, hidden up to 512, at most 8 residual blocks");
  const LayerTapeLayout L = a.layout;
  char* ws = (char*)workspace;
  a.tape = (float*)ws;
  ws += ((size_t)a.batch * L.total * 4 + 255) / 256 * 256;
  a.loss_part = (double*)ws;
  ws += ((size_t)a.batch * 8 + 255) / 256 * 256;
  LinearWgradJob* jobs_dev = (LinearWgradJob*)ws;
  hipLaunchKernelGGL(layer_mlp_train_kernel, dim3(a.batch), dim3(512), 0, s, a);
  CD_HIP(hipGetLastError());
  hipLaunchKernelGGL(layer_loss_final_kernel, dim3(1), dim3(64), 0, s, a.loss_part, a.sigma, a.batch, a.dim_in, loss_out, a.loss_type);
  CD_HIP(hipGetLastError());
  // weight / bias gradients of the 20 Linear layers: one launch over a job table
  const int half = a.cond_emb / 2, q = half / 2, Hd = a.hidden, D = a.dim_in;
  std::vector<LinearWgradJob> jobs;
  size_t goff = 0;
  int max_elems = 1;
  auto add = [&](int delta_off, int in_off, int nout, int nin) {
    LinearWgradJob j;
    j.delta = a.tape + delta_off; j.in = a.tape + in_off; j.dw = grads + goff; j.db = grads + goff + (size_t)nout * nin;
    j.nout = nout; j.nin = nin; j.delta_ld = L.total; j.in_ld = L.total;
    goff += (size_t)nout * nin + nout;
    if (nout * nin > max_elems) max_elems = nout * nin;
    jobs.push_back(j);
  };
  add(L.d1t, L.t_in, q, 1); add(L.d2t, L.a1t, half, q); add(L.d3t, L.a2t, half, half);          // time_mlp.{1,3,5}
  add(L.d1c, L.cin, q, a.cond_size); add(L.d2c, L.a1c, half, q); add(L.d3c, L.a2c, half, half);  // cond_mlp.{0,2,4}
  add(L.dh0, L.xin, Hd, D);                                                                      // in_lay
  for (int r = 0; r < a.n_res; ++r) {
    add(L.de[r], L.g, Hd, 2 * half);     // embeder.1
    add(L.du[r], L.hprev[r], Hd, Hd);    // dense1.0
    add(L.dv[r], L.h1[r], Hd, Hd);       // dense2.0
  }
  add(L.dpred, L.hfin, D, Hd);           // out_lay
  CD_REQUIRE(jobs.size() <= 64, "too many linear layers");
  CD_HIP(hipMemcpyAsync(jobs_dev, jobs.data(), sizeof(LinearWgradJob) * jobs.size(), hipMemcpyHostToDevice, s));
  CD_HIP(hipStreamSynchronize(s));  // `jobs` lives on this stack frame
  launch_linear_wgrad(jobs_dev, (int)jobs.size(), max_elems, a.batch, s);
}

}  // namespace cd
