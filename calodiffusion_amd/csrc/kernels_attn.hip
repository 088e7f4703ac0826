// Fused linear attention for the sampling path (LinearAttention, calodiffusion/models/models.py:281-329, heads = 1,
// dim_head = 32, wrapped in PreNorm(GroupNorm(1)) :111-117): the (B, n, 96) qkv tensor is never written.
//
//   pass 1  attn_kv_context_kernel   x -> xn = GroupNorm(1) affine (folded coefficients) -> k, v = W_k xn, W_v xn on the matrix
//                                    cores -> exp(k - max) and the un-normalised context  ctx[d][e] = sum_n e^{k[n][d]-m[d]} v[n][e]
//                                    straight from the accumulator registers (the projection's C layout IS the A/B layout of
//                                    the context MFMA); per-workgroup {max, sum, ctx} partials, merged log-sum-exp style by
//                                    attn_combine_kernel (kernels_norm_attn.hip), which also folds W_out into per-sample weights
//   pass 2  attn_out_kernel          x -> xn -> q^T = W_q xn^T (channels in registers => the 32-way softmax over channels is
//                                    lane-local) -> y = softmax(q) W'^T + b with the q^T registers as the A operand -> y and its
//                                    channel statistics (for the GroupNorm(1) that follows)
//
// HBM traffic per attention block: x twice + y once (3 x B n C floats) instead of x + 2 x qkv + q + y (= 10 x for C = 32).
// The q / k / v projections run on the fp16 matrix pipe with the convs' f16x2 split (split16.h: fp32-grade rounding, 3 MFMAs
// per 16 channels instead of 16 f32 MFMAs at half the rate); the context and output products take their operands from
// accumulator registers and stay on v_mfma_f32_32x32x2_f32.
#include "cd_common.h"
#include "gn_defer.h"
#include "split16.h"

#include <cstdio>
#include <cstdlib>

namespace cd {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

struct AttnArgs {
  const float* x;     // (B, vox, C) channels-last
  const float* coef;  // [B][C][4] = {scale, shift, -, -} of the PreNorm
  const u32x4* wqkv;  // f16x2 image of to_qkv [k-step][ct (0 = q, 1 = k, 2 = v)][term][lane] (launch_pack_weights_f16x2, taps = 1)
  int C;
  int64_t vox;
  int tiles_per_wg;   // 32-voxel tiles per workgroup (multiple of 8)
  // pass 1
  float* partials;    // [B][nsplit][64 + 1024]
  int nsplit;
  // pass 2
  const float* wT;    // [B][CT][4][64][4]: per-sample folded output weights, transposed-K layout (attn_combine, layout_T),
                      // or null: every workgroup folds them itself from `partials` (fold_*)
  const float* fold_wout;  // to_out weight, torch (C, 32)
  float fold_scale;        // dim_head^-1/2
  const float* bias;  // (C)
  float* y;           // (B, vox, C)
  float* ch_part;     // [B][units][C][2]
  GnDefer defer;      // PreNorm coefficients folded in the prologue instead of read from `coef`
  const float *out_gamma, *out_beta;  // attn_small_kernel: affine parameters of the closing GroupNorm(1, C)
  int* status;        // bit 0: an operand of the fp16-pipe products (normalised x, v, the folded output weights) left the fp16 range
  // attn_coop_kernel: the nsplit workgroups of a sample meet at two barriers inside the launch
  unsigned* sync;     // [B][2] = {arrivals, generation}: zero once, self-resetting
  int coop;           // the partials / channel sums read below were written by OTHER workgroups of this launch
  // moment form (MOM instances): pass 1 also emits, per partial, s'[d] = sum_n (softmax(q)[n][d] - 1/32) and the 32 x 32 matrix
  // S'[d][d'] = sum_n (softmax(q)[n][d] - 1/32)(softmax(q)[n][d'] - 1/32), from which pass 2 knows the statistics of its own
  // output before it computes it
  float* mom;         // [B][nsplit][32 + 1024]
};
constexpr int ATTN_MOM = 32 + 1024;
// value written by another workgroup of the SAME launch (same XCD, see attn_coop_kernel): an agent-scope load, which does not
// linger in this CU's vector cache
__device__ __forceinline__ float attn_peer_load(const float* p, int coop) {
  return coop ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// A fragments of one 32-voxel tile for v_mfma_f32_32x32x16_f16: lane (voxel n0 + col, half) holds, per 16-channel k-step ks,
// channels ks*16 + half*8 + 0..7 of the normalised input as two fp16 terms (f16x2).  NKS = C / 16 k-steps.
template <int NCH>
struct RawTile {
  f32x4 v[NCH * 2][2];
};
// global loads of one tile (issued ahead of their use; tiles past the end are clamped to the last one and never consumed)
template <int NCH>
__device__ __forceinline__ void load_raw(const AttnArgs& a, int b, int64_t t, int64_t tlast, int col, int half, RawTile<NCH>& raw) {
  const int64_t tc = t < tlast ? t : tlast;
  int64_t n = tc * 32 + col;
  n = n < a.vox ? n : a.vox - 1;
  const float* src = a.x + ((size_t)b * a.vox + n) * a.C + half * 8;
#pragma unroll
  for (int ks = 0; ks < NCH * 2; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) raw.v[ks][j] = *(const f32x4*)(src + ks * 16 + j * 4);
}
// cf_lds: the coefficients are re-read per tile from an LDS copy ([ch][8] quads of this half-wave) instead of living in 32 NCH
// registers (the moment form of pass 1 has none to spare)
template <int NCH, bool CFL = false>
__device__ __forceinline__ void norm_split(const AttnArgs& a, int64_t n0, int col, const f32x4 (&cf)[NCH][8], const RawTile<NCH>& raw,
                                           u32x4 (&x1)[NCH * 2], u32x4 (&x2)[NCH * 2], float& amax, const f32x4* cf_lds = nullptr) {
  const bool valid = n0 + col < a.vox;
#pragma unroll
  for (int ks = 0; ks < NCH * 2; ++ks) {
    u32x2 t1[2], t2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 v = raw.v[ks][j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // cf[ks >> 1][(ks & 1) * 4 + j * 2 + (e >> 1)] = {scale, shift} pairs of channels ks*16 + half*8 + j*4 + (e & ~1), +1
        const f32x4 c2 = CFL ? cf_lds[(ks >> 1) * 8 + (ks & 1) * 4 + j * 2 + (e >> 1)] : cf[ks >> 1][(ks & 1) * 4 + j * 2 + (e >> 1)];
        v[e] = valid ? c2[(e & 1) * 2] * v[e] + c2[(e & 1) * 2 + 1] : 0.f;
      }
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      split2(v, t1[j], t2[j]);
    }
    x1[ks] = u32x4{t1[0][0], t1[0][1], t1[1][0], t1[1][1]};
    x2[ks] = u32x4{t2[0][0], t2[0][1], t2[1][0], t2[1][1]};
  }
}
// Tiles whose loads are in flight ahead of their use, per wave (0 = load and consume in place).  Measured on MI355X with the
// one-round grids below: depth 2-3 makes pass 1 slower (0.146 -> 0.155 ms per denoise step, register pressure) and leaves
// pass 2 unchanged -- these kernels are bound by their VALU work (split, exp) and the f32 context MFMAs, not by load latency.
#ifndef ATTN_OUT_D1
#define ATTN_OUT_D1 0
#endif
template <int NCH>
struct AttnDepthOut {
  static constexpr int value = NCH == 1 ? ATTN_OUT_D1 : 0;
};
// e^x for x <= 0 on the transcendental unit: 2^(x log2 e) with the rounding error of the product carried into a first-order
// correction (relative error ~1e-7, against ~|x| 6e-8 for the bare product; libm's expf costs ~25 VALU instructions and
// these kernels are VALU-bound: 16 exponentials per lane and tile).
__device__ __forceinline__ float attn_exp(float x) {
  const float L2E = 1.4426950408889634f, L2E_LO = 1.925963033500519e-8f;  // log2 e = L2E + L2E_LO
  const float t = x * L2E;
  const float lo = __builtin_fmaf(x, L2E, -t) + x * L2E_LO;  // exact remainder of the product + low part of the constant
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, lo * 0.6931471805599453f, e);
}
__device__ __forceinline__ void attn_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// must be called by every thread of the workgroup (the deferred form contains barriers)
template <int NCH>
__device__ __forceinline__ void load_coef(const AttnArgs& a, int b, int half, f32x4 (&cf)[NCH][8]) {
  __shared__ __attribute__((aligned(16))) float sCoef[NCH * 32 * 4];
  __shared__ __attribute__((aligned(16))) char sDefer[NCH * 32 * 16 + 64 * 8];
  const float* base = a.defer.part ? nullptr : a.coef + (size_t)b * a.C * 4;
  if (a.defer.part) {
    gn_defer_to_lds(a.defer, b, sCoef, sDefer);
    base = sCoef;
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* p = base + (ch * 32 + (i >> 2) * 16 + half * 8 + 2 * (i & 3)) * 4;
      cf[ch][i] = f32x4{p[0], p[1], p[4], p[5]};
    }
}

// Whether the passes re-read the coefficients per tile from an LDS table instead of holding them in 32 NCH registers: the moment
// form of pass 1 has none to spare, and from 64 channels on both passes spilled (33 / 116 / 194 registers at 64 / 96 / 128 channels)
template <int NCH, bool MOM>
struct AttnCfl {
  static constexpr bool value = MOM || NCH >= 2;
};
// the table: [half][ch][8] quads; returns this lane's half.  Call from every thread (a barrier follows the fill).
template <int NCH>
__device__ __forceinline__ const f32x4* attn_cf_table(const f32x4 (&cf)[NCH][8], f32x4* table) {
  const int lane = threadIdx.x & 63, half = lane >> 5;
  if (threadIdx.x < 64 && (lane & 31) == 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int i = 0; i < 8; ++i) table[(half * NCH + ch) * 8 + i] = cf[ch][i];
  }
  __syncthreads();
  return table + half * NCH * 8;
}

// pass 1 of sample b over tiles [t0, t1): partial `split` of the sample
template <int NCH, bool MOM = false>
__device__ __forceinline__ void attn_pass1(const AttnArgs& a, int b, int split, int64_t t0, int64_t t1, const f32x4 (&cf)[NCH][8],
                                           const f32x4* cf_lds = nullptr /* AttnCfl: the table of attn_cf_table */) {
  __shared__ float sMax[8][32];
  __shared__ float sSum[16][32];
  __shared__ float sCtx[8][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  constexpr int NKS = NCH * 2;
  u32x4 wk1[NKS], wk2[NKS], wv1[NKS], wv2[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    wk1[ks] = a.wqkv[(size_t)(ks * 3 + 1) * 128 + lane];
    wk2[ks] = a.wqkv[(size_t)(ks * 3 + 1) * 128 + 64 + lane];
    wv1[ks] = a.wqkv[(size_t)(ks * 3 + 2) * 128 + lane];
    wv2[ks] = a.wqkv[(size_t)(ks * 3 + 2) * 128 + 64 + lane];
  }
  // moment form: the q projection weights, the per-wave tile on its way from the lane-per-voxel layout of the softmax to the
  // lane-per-channel layout of an MFMA operand, and the accumulators of s' and S'
  u32x4 wq1[MOM ? NKS : 1], wq2[MOM ? NKS : 1];
  __shared__ __attribute__((aligned(16))) float sQt[MOM ? 8 * 32 * 36 : 4];
  f32x16 mS, mSB;
  float msum = 0.f;
  if constexpr (MOM) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      wq1[ks] = a.wqkv[(size_t)(ks * 3 + 0) * 128 + lane];
      wq2[ks] = a.wqkv[(size_t)(ks * 3 + 0) * 128 + 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mS[r] = mSB[r] = 0.f;
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // y = x W^T on the fp16 pipe: A += x1 w1, B += x1 w2' + x2' w1, y = A + B / 2048
  auto project = [&](const u32x4 (&x1)[NKS], const u32x4 (&x2)[NKS], const u32x4 (&w1)[NKS], const u32x4 (&w2)[NKS]) {
    f32x16 pa = zero16, pb = zero16;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      pa = MFMA_F16(x1[ks], w1[ks], pa);
      pb = MFMA_F16(x1[ks], w2[ks], pb);
      pb = MFMA_F16(x2[ks], w1[ks], pb);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pa[r] += pb[r] * (1.f / 2048.f);
    return pa;
  };

  // ONE sweep with a running maximum per wave (the two-sweep form -- maxima first, then exponentials -- projected k twice and read
  // x twice): e = exp(k - m) against the wave's maximum so far; when a tile raises it, what the wave has accumulated is rescaled
  // by exp(m_old - m_new) first -- rare after a wave's first tiles, so the rescale sits behind a wave-uniform branch.  The eight
  // waves' {m, sum, ctx} are merged the same way at the end.
  // ctx[d][e'] += e^T v on the matrix cores: the k / v accumulator registers are the A / B operands of the context MFMA (k-slot =
  // the lane's half).  Row d of ctx lives in register r of the lanes of half h with d = (r & 3) + 8 (r >> 2) + 4 h, while the
  // factor of channel d is computed in lane d: it reaches the rows through a 32-float line of LDS per wave.
  __shared__ __attribute__((aligned(16))) float sFac[8][32];
  RawTile<NCH> raw;
  const int64_t tlast = t1 - 1;
  float m = -3.0e38f;  // (sentinel: never enters an exponential)
  bool any = false;
  float amax = 0.f;  // largest |operand| of the fp16-pipe products of this thread (normalised x, v)
  // The context product itself runs on the fp16 pipe with the convs' two-term split (split16.h): e in (0, 1] and v as f16x2, the
  // registers 8s .. 8s+7 of a lane being the 8 k-slots of k-step s (the same voxels in e and in v), 3 MFMAs per k-step into two
  // accumulators -- 6 x 32 cycles per tile against 16 x 64 for v_mfma_f32_32x32x2_f32, for ~110 more vector instructions.
  f32x16 ctx, ctxB;  // ctx += e1 v1;  ctxB += e1 v2' + e2' v1;  context = ctx + ctxB / 2048
#pragma unroll
  for (int r = 0; r < 16; ++r) ctx[r] = ctxB[r] = 0.f;
  float ssum = 0.f;
  auto scale_rows = [&](float fac) {  // ctx[d][:] *= fac[d], fac given in lane d
    sFac[wave][col] = fac;  // (both halves write the same value)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its own writes are visible to its reads in order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 f4 = *(const f32x4*)&sFac[wave][8 * q + 4 * half];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ctx[4 * q + e] *= f4[e];
        ctxB[4 * q + e] *= f4[e];
      }
    }
  };
  auto split8 = [](const float (&x)[16], int s8, u32x4& hi, u32x4& lo) {  // registers 8 s8 .. 8 s8 + 7 -> one f16x2 operand pair
    u32x2 h0, l0, h1, l1;
    split2(f32x4{x[8 * s8], x[8 * s8 + 1], x[8 * s8 + 2], x[8 * s8 + 3]}, h0, l0);
    split2(f32x4{x[8 * s8 + 4], x[8 * s8 + 5], x[8 * s8 + 6], x[8 * s8 + 7]}, h1, l1);
    hi = u32x4{h0[0], h0[1], h1[0], h1[1]};
    lo = u32x4{l0[0], l0[1], l1[0], l1[1]};
  };
  for (int64_t tt = t0 + wave; tt < t1; tt += 8) {
    u32x4 x1[NKS], x2[NKS];
    load_raw<NCH>(a, b, tt, tlast, col, half, raw);
    norm_split<NCH, AttnCfl<NCH, MOM>::value>(a, tt * 32, col, cf, raw, x1, x2, amax, cf_lds);
    const f32x16 k = project(x1, x2, wk1, wk2);
    const f32x16 v = project(x1, x2, wv1, wv2);
    const bool full = tt * 32 + 32 <= a.vox;  // (only a sample's last tile can be partial)
    float tm = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (full || tt * 32 + row < a.vox) tm = fmaxf(tm, k[r]);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
    if (!any) {
      m = tm;  // first tile of this wave: nothing accumulated yet
      any = true;
    } else if (__builtin_amdgcn_ballot_w64(tm > m) != 0) {
      const float mn = fmaxf(m, tm);
      const float fac = attn_exp(m - mn);
      ssum *= fac;
      scale_rows(fac);
      m = mn;
    }
    float ex[16], vv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      ex[r] = (full || tt * 32 + row < a.vox) ? attn_exp(k[r] - m) : 0.f;
      ssum += ex[r];
      vv[r] = v[r];
      amax = fmaxf(amax, fabsf(v[r]));
    }
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) {
      u32x4 e1, e2, v1, v2;
      split8(ex, s8, e1, e2);
      split8(vv, s8, v1, v2);
      ctx = MFMA_F16(e1, v1, ctx);
      ctxB = MFMA_F16(e1, v2, ctxB);
      ctxB = MFMA_F16(e2, v1, ctxB);
    }
    if constexpr (MOM) {
      // softmax(q) of the tile as pass 2 computes it (q^T: a lane holds 16 of its voxel's 32 channels, the other half-wave the
      // rest), centred on 1/32 -- with near-uniform softmaxes S would be a large constant plus a small signal and the quadratic
      // form W' S W'^T of the fold a difference of large numbers; rows past the end of the sample contribute nothing
      f32x16 q = zero16, qb = zero16;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        q = MFMA_F16(wq1[ks], x1[ks], q);
        qb = MFMA_F16(wq2[ks], x1[ks], qb);
        qb = MFMA_F16(wq1[ks], x2[ks], qb);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] += qb[r] * (1.f / 2048.f);
      float mx = q[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, q[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // (the bare 2^(x log2 e): ~|x| 6e-8 relative, ample for statistics -- pass 2's own softmax keeps attn_exp)
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        q[r] = __builtin_amdgcn_exp2f((q[r] - mx) * 1.4426950408889634f);
        ss += q[r];
      }
      ss += __shfl_xor(ss, 32, 64);
      const bool valid = tt * 32 + col < a.vox;
      const float inv = valid ? 1.f / ss : 0.f, cen = valid ? (1.f / 32.f) : 0.f;
      // [voxel][channel] through the wave's LDS tile: written as quads (d = 8 g + 4 half + 0..3), read back with a lane per channel
      float* tq = sQt + wave * (32 * 36);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(tq + col * 36 + 8 * g + 4 * half) = f32x4{q[4 * g] * inv - cen, q[4 * g + 1] * inv - cen, q[4 * g + 2] * inv - cen,
                                                            q[4 * g + 3] * inv - cen};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes are visible to its reads in order)
      float dq[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dq[r] = tq[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + col];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) msum += dq[r];
#pragma unroll
      for (int s8 = 0; s8 < 2; ++s8) {
        u32x4 d1, d2;
        split8(dq, s8, d1, d2);
        mS = MFMA_F16(d1, d1, mS);
        mSB = MFMA_F16(d1, d2, mSB);  // P = h^T l; the other cross term is its transpose, added in the merge below
      }
    }
  }
  if (a.status && !(amax <= 65504.f)) atomicOr(a.status, 1);  // (also catches NaN)
  // merge the waves: common maximum, every wave's sums and context rescaled to it
  if (half == 0) sMax[wave][col] = m;
  attn_barrier_lds();
  float M = sMax[0][col];
#pragma unroll
  for (int w = 1; w < 8; ++w) M = fmaxf(M, sMax[w][col]);
  {
    const float fac = any ? attn_exp(m - M) : 0.f;  // (a workgroup has at least one tile, so M is a real maximum)
    ssum *= fac;
    scale_rows(fac);
    m = M;
  }
  sSum[wave * 2 + half][col] = ssum;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * half;  // row = channel d; column = e
    sCtx[wave][d * 32 + col] = ctx[r] + ctxB[r] * (1.f / 2048.f);
  }
  __syncthreads();
  float* out = a.partials + ((size_t)b * a.nsplit + split) * (64 + 1024);
  if (tid < 32) {
    out[tid] = m;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sSum[i][tid];
    out[32 + tid] = t;
  }
  for (int i = tid; i < 1024; i += 512) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += sCtx[w][i];
    out[64 + i] = t;
  }
  if constexpr (MOM) {
    __syncthreads();  // the context slices have been read
    sSum[wave * 2 + half][col] = msum;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
      sCtx[wave][d * 32 + col] = mS[r];
      sQt[wave * 1024 + d * 32 + col] = mSB[r];  // (the transposition tiles are done with: 8 x 1152 floats)
    }
    __syncthreads();
    float* mo = a.mom + ((size_t)b * a.nsplit + split) * ATTN_MOM;
    if (tid < 32) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += sSum[i][tid];
      mo[tid] = t;
    }
    for (int i = tid; i < 1024; i += 512) {
      const int it = (i & 31) * 32 + (i >> 5);
      float t = 0.f, p = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        t += sCtx[w][i];
        p += sQt[w * 1024 + i] + sQt[w * 1024 + it];
      }
      mo[32 + i] = t + p * (1.f / 2048.f);
    }
  }
}

template <int NCH, bool MOM>
__global__ void __launch_bounds__(512) attn_kv_context_kernel(AttnArgs a) {
  const int half = (threadIdx.x & 63) >> 5;
  const int split = blockIdx.x, b = blockIdx.y;
  const int64_t T = (a.vox + 31) / 32;
  const int64_t t0 = (int64_t)split * a.tiles_per_wg;
  const int64_t t1 = (t0 + a.tiles_per_wg < T) ? t0 + a.tiles_per_wg : T;
  f32x4 cf[NCH][8];
  load_coef<NCH>(a, b, half, cf);
  __shared__ __attribute__((aligned(16))) f32x4 sCfT[2 * NCH * 8];
  const f32x4* cfl = nullptr;
  if constexpr (AttnCfl<NCH, MOM>::value) cfl = attn_cf_table<NCH>(cf, sCfT);
  attn_pass1<NCH, MOM>(a, b, split, t0, t1, cf, cfl);
}

// Merge of the pass-1 partials and fold of the context into the output projection, by every workgroup of pass 2 for its own
// sample (same arithmetic, in the same order, as attn_combine_kernel, kernels_norm_attn.hip: the separate launch was 11 us of
// pure latency per attention block):  W'[c][d] = scale * sum_e W_out[c][e] ctx[d][e] / sum[d]  in the k-slot order of an
// accumulator-register A operand, into sW[NCH][4][64][4].
// MOM: also the closing GroupNorm(1, C) of the block, whose statistics follow from the fold and the pass-1 moments without y ever
// existing: with u = 1/32, m_c = b_c + u sum_d W'[c][d] and delta_n = softmax(q_n) - u,
//   y[n][c] = m_c + W'_c . delta_n,   sum_n y = N m_c + W'_c . s',   sum_n y^2 = N m_c^2 + 2 m_c W'_c . s' + W'_c S' W'_c^T
// (fixed-order fp64).  sOut[c] = {scale, shift} with  out = scale * (y - b_c) + shift + x.
template <int NCH, bool MOM = false>
__device__ __forceinline__ void attn_fold_weights(const AttnArgs& a, int b, float* sW, float (*sOut)[2] = nullptr) {
  __shared__ float sM[32], sInv[32];
  __shared__ float sCtx[1024];
  __shared__ float sWout[NCH * 32 * 33];  // W_out rows padded to 33 floats: the 32 lanes of a half-wave read 32 different rows
  const int tid = threadIdx.x;
  const float* p = a.partials + (size_t)b * a.nsplit * (64 + 1024);
  // (this prologue is pure latency for every workgroup of pass 2: independent loads are issued together, W_out is staged while
  // the maxima are merged)
  for (int i = tid; i < NCH * 1024; i += 512) sWout[(i >> 5) * 33 + (i & 31)] = a.fold_wout[i];
  float mreg[3] = {0.f, 0.f, 0.f};  // moment form: entries tid, tid + 512, tid + 1024 of the merged {s', S'}
  if constexpr (MOM) {
    const float* mo = a.mom + (size_t)b * a.nsplit * ATTN_MOM;
#pragma unroll 4
    for (int k = 0; k < a.nsplit; ++k) {
      mreg[0] += attn_peer_load(mo + (size_t)k * ATTN_MOM + tid, a.coop);
      mreg[1] += attn_peer_load(mo + (size_t)k * ATTN_MOM + 512 + tid, a.coop);
      if (tid < 32) mreg[2] += attn_peer_load(mo + (size_t)k * ATTN_MOM + 1024 + tid, a.coop);
    }
  }
  if (tid < 32) {
    float M = -3.0e38f;
#pragma unroll 4
    for (int i = 0; i < a.nsplit; ++i) M = fmaxf(M, attn_peer_load(p + (size_t)i * 1088 + tid, a.coop));
    float S = 0.f;
#pragma unroll 4
    for (int i = 0; i < a.nsplit; ++i)
      S += attn_peer_load(p + (size_t)i * 1088 + 32 + tid, a.coop) * expf(attn_peer_load(p + (size_t)i * 1088 + tid, a.coop) - M);
    sM[tid] = M;
    sInv[tid] = a.fold_scale / S;
  }
  __syncthreads();
  for (int i = tid; i < 1024; i += 512) {
    const int d = i >> 5;
    float c = 0.f;
#pragma unroll 4
    for (int k = 0; k < a.nsplit; ++k)
      c += attn_peer_load(p + (size_t)k * 1088 + 64 + i, a.coop) * expf(attn_peer_load(p + (size_t)k * 1088 + d, a.coop) - sM[d]);
    sCtx[i] = c * sInv[d];
  }
  __syncthreads();
  for (int i = tid; i < NCH * 1024; i += 512) {
    const int e4 = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3, ct = i >> 10;
    const int c = ct * 32 + (lane & 31);
    const int d = e4 + 8 * q + 4 * (lane >> 5);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) acc = fmaf(sWout[c * 33 + e], sCtx[d * 32 + e], acc);
    sW[i] = acc;
  }
  __syncthreads();
  if constexpr (MOM) {
    static_assert(!MOM || NCH == 1, "moment form: 32 channels");
    __shared__ float sS[1024 + 32];
    __shared__ double sAB[32][2];
    // the merged moments (their loads were requested at the top of the function)
    // (entry i of a partial: i < 32 is s'[i], kept behind the matrix at sS[1024 + i]; the rest is S' flat at sS[i - 32])
    sS[tid < 32 ? 1024 + tid : tid - 32] = mreg[0];
    sS[480 + tid] = mreg[1];
    if (tid < 32) sS[992 + tid] = mreg[2];
    __syncthreads();
    // W'[c][d] out of the operand-order image
    auto wp = [&](int c, int d) { return sW[(((d >> 3) * 64) + c + 32 * ((d >> 2) & 1)) * 4 + (d & 3)]; };
    // thread (c, d), two channels per thread: W'[c][d] * sum_e S'[e][d] W'[c][e]  (S' is symmetric: row e, column d), W'[c][d],
    // W'[c][d] s'[d]; then the sums over d inside the half-wave (fixed xor tree)
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int c = (tid >> 5) + 16 * rep, d = tid & 31;
      double t = 0.0;
#pragma unroll 8
      for (int e = 0; e < 32; ++e) t += (double)sS[e * 32 + d] * (double)wp(c, e);
      const double w = (double)wp(c, d);
      double quad = t * w, rows = w, lin = w * (double)sS[1024 + d];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) {
        quad += __shfl_xor(quad, m, 64);
        rows += __shfl_xor(rows, m, 64);
        lin += __shfl_xor(lin, m, 64);
      }
      if (d == 0) {
        const double N = (double)a.vox;
        const double mc = (a.bias ? (double)a.bias[c] : 0.0) + rows * (1.0 / 32.0);
        sAB[c][0] = N * mc + lin;
        sAB[c][1] = N * mc * mc + 2.0 * mc * lin + quad;
      }
    }
    __syncthreads();
    if (tid < 32) {
      double a1 = sAB[tid][0], a2 = sAB[tid][1];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) {
        a1 += __shfl_xor(a1, m, 64);
        a2 += __shfl_xor(a2, m, 64);
      }
      const double cnt = (double)a.vox * 32.0;
      const double mu = a1 / cnt;
      double var = a2 / cnt - mu * mu;
      var = var < 0.0 ? 0.0 : var;
      const double rstd = 1.0 / sqrt(var + 1e-5);
      const float sc = (float)rstd * a.out_gamma[tid];
      const float bc = a.bias ? a.bias[tid] : 0.f;
      sOut[tid][0] = sc;
      sOut[tid][1] = a.out_beta[tid] + (float)((double)bc - mu) * sc;
    }
    __syncthreads();
  }
}

// pass 2 of sample b over tiles [t0, t1): y and, as unit `unit` of `nunits`, its channel partials
// MOM: the output is the block's (closing GroupNorm from the fold's closed form, residual added), no channel sums
template <int NCH, bool MOM = false>
__device__ __forceinline__ void attn_pass2(const AttnArgs& a, int b, int unit, int nunits, int64_t t0, int64_t t1,
                                           const f32x4 (&cf)[NCH][8], float* trbuf = nullptr, const f32x4* cf_lds = nullptr) {
  __shared__ float sRed[MOM ? 1 : 8][NCH * 32][2];
  __shared__ float sOut[MOM ? NCH * 32 : 1][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  constexpr int NKS = NCH * 2;
  u32x4 wq1[NKS], wq2[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    wq1[ks] = a.wqkv[(size_t)(ks * 3 + 0) * 128 + lane];
    wq2[ks] = a.wqkv[(size_t)(ks * 3 + 0) * 128 + 64 + lane];
  }
  f32x4 wt[NCH][4];
  if (a.wT) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int q = 0; q < 4; ++q) wt[ch][q] = ((const f32x4*)a.wT)[(((size_t)b * NCH + ch) * 4 + q) * 64 + lane];  // output channel tile ch
  } else {
    __shared__ __attribute__((aligned(16))) float sW[NCH * 1024];
    attn_fold_weights<NCH, MOM>(a, b, sW, sOut);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int q = 0; q < 4; ++q) wt[ch][q] = ((const f32x4*)sW)[(ch * 4 + q) * 64 + lane];
  }
  // the folded weights as f16x2 B operands of the output product (fp16 pipe, as the context product of pass 1): k-step s = the
  // registers 8s .. 8s+7 of wt (the d-slots of q's registers 8s .. 8s+7)
  u32x4 wb1[NCH][2], wb2[NCH][2];
  float amax = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(wt[ch][q][0]), fabsf(wt[ch][q][1])), fmaxf(fabsf(wt[ch][q][2]), fabsf(wt[ch][q][3]))));
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) {
      u32x2 h0, l0, h1, l1;
      split2(wt[ch][2 * s8], h0, l0);
      split2(wt[ch][2 * s8 + 1], h1, l1);
      wb1[ch][s8] = u32x4{h0[0], h0[1], h1[0], h1[1]};
      wb2[ch][s8] = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bv[NCH], s1[NCH], s2[NCH], osc[NCH];
#pragma unroll
  for (int ct = 0; ct < NCH; ++ct) {
    bv[ct] = a.bias ? a.bias[ct * 32 + col] : 0.f;
    s1[ct] = s2[ct] = 0.f;
    osc[ct] = 1.f;
    if constexpr (MOM) {  // out = osc * o + bv + x  (the bias is inside the shift)
      osc[ct] = sOut[ct * 32 + col][0];
      bv[ct] = sOut[ct * 32 + col][1];
    }
  }
  float* const yb = a.y + (size_t)b * a.vox * a.C;
  const float* const xb = a.x + (size_t)b * a.vox * a.C;

  constexpr int PRE = AttnDepthOut<NCH>::value, DEPTH = PRE ? PRE : 1;
  RawTile<NCH> raw[DEPTH];
  const int64_t tlast = t1 - 1;
  if (PRE) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_raw<NCH>(a, b, t0 + wave + 8 * d, tlast, col, half, raw[d]);
  }
  for (int64_t tg = t0 + wave; tg < t1; tg += 8 * DEPTH) {
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int64_t t = tg + 8 * d;
    if (t >= t1) break;
    u32x4 x1[NKS], x2[NKS];
    if (!PRE) load_raw<NCH>(a, b, t, tlast, col, half, raw[d]);
    norm_split<NCH, AttnCfl<NCH, MOM>::value>(a, t * 32, col, cf, raw[d], x1, x2, amax, cf_lds);
    if (PRE && t + 8 * DEPTH < t1) load_raw<NCH>(a, b, t + 8 * DEPTH, tlast, col, half, raw[d]);
    // q^T[d][n]: A = W_q (row d), B = xn^T (column n); the registers of a lane are 16 channels d of its voxel n = col
    f32x16 q = zero16, qb = zero16;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      q = MFMA_F16(wq1[ks], x1[ks], q);
      qb = MFMA_F16(wq2[ks], x1[ks], qb);
      qb = MFMA_F16(wq1[ks], x2[ks], qb);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r] += qb[r] * (1.f / 2048.f);
    float mx = q[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, q[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      q[r] = attn_exp(q[r] - mx);
      ss += q[r];
    }
    ss += __shfl_xor(ss, 32, 64);
    const float inv = 1.f / ss;
    // y[n][co] = sum_d softmax(q)[n][d] W'[co][d]: A = q^T registers (row n = col, k-slots = the lane's registers <-> d = row(r, half))
    u32x4 p1[2], p2[2];
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) {
      u32x2 h0, l0, h1, l1;
      split2(f32x4{q[8 * s8] * inv, q[8 * s8 + 1] * inv, q[8 * s8 + 2] * inv, q[8 * s8 + 3] * inv}, h0, l0);
      split2(f32x4{q[8 * s8 + 4] * inv, q[8 * s8 + 5] * inv, q[8 * s8 + 6] * inv, q[8 * s8 + 7] * inv}, h1, l1);
      p1[s8] = u32x4{h0[0], h0[1], h1[0], h1[1]};
      p2[s8] = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
#pragma unroll
    for (int ct = 0; ct < NCH; ++ct) {
      f32x16 o = zero16, ob = zero16;
#pragma unroll
      for (int s8 = 0; s8 < 2; ++s8) {
        o = MFMA_F16(p1[s8], wb1[ct][s8], o);
        ob = MFMA_F16(p1[s8], wb2[ct][s8], ob);
        ob = MFMA_F16(p2[s8], wb1[ct][s8], ob);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] += ob[r] * (1.f / 2048.f);
      if (trbuf && t * 32 + 32 <= a.vox) {
        // whole tile: out as 16-byte quads (row 8 k + (lane >> 3), channels 4 (lane & 7) .. + 3) after a transpose through the wave's
        // LDS tile -- 16 scalar row stores per lane in accumulator layout ran at a fraction of the HBM rate
        float* tr = trbuf + wave * (32 * 36);
        const size_t qoff = (size_t)t * 32 * a.C + ct * 32 + (lane & 7) * 4;
        f32x4 xres[4];
        if constexpr (MOM) {  // the residual rows, in the layout of the stores (this wave read the tile a moment ago: cache hits)
#pragma unroll
          for (int k = 0; k < 4; ++k) xres[k] = *(const f32x4*)(xb + qoff + (size_t)(8 * k + (lane >> 3)) * a.C);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = MOM ? __builtin_fmaf(o[r], osc[ct], bv[ct]) : o[r] + bv[ct];
          tr[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + col] = v;
          if (!MOM) {
            s1[ct] += v;
            s2[ct] += v * v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes are visible to its reads in order)
        float* yt = yb + qoff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = 8 * k + (lane >> 3);
          f32x4 v4 = *(const f32x4*)(tr + row * 36 + (lane & 7) * 4);
          if constexpr (MOM) v4 += xres[k];
          *(f32x4*)(yt + (size_t)row * a.C) = v4;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the tile buffer is reused)
      } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t n = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < a.vox) {
          const size_t at = (size_t)n * a.C + ct * 32 + col;
          const float v = MOM ? __builtin_fmaf(o[r], osc[ct], bv[ct]) + xb[at] : o[r] + bv[ct];
          yb[at] = v;
          s1[ct] += v;
          s2[ct] += v * v;
        }
      }
      }
    }
  }
  }
  if (a.status && !(amax <= 65504.f)) atomicOr(a.status, 1);
  if constexpr (!MOM) if (a.ch_part) {
#pragma unroll
    for (int ct = 0; ct < NCH; ++ct) {
      const float t1s = s1[ct] + __shfl_xor(s1[ct], 32, 64), t2s = s2[ct] + __shfl_xor(s2[ct], 32, 64);
      if (half == 0) {
        sRed[wave][ct * 32 + col][0] = t1s;
        sRed[wave][ct * 32 + col][1] = t2s;
      }
    }
    __syncthreads();
    if (tid < NCH * 32) {
      float r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        r1 += sRed[w][tid][0];
        r2 += sRed[w][tid][1];
      }
      float* dst = a.ch_part + (((size_t)b * nunits + unit) * a.C + tid) * 2;
      dst[0] = r1;
      dst[1] = r2;
    }
  }
}

template <int NCH, bool MOM>
__global__ void __launch_bounds__(512) attn_out_kernel(AttnArgs a) {
  const int half = (threadIdx.x & 63) >> 5;
  const int unit = blockIdx.x, b = blockIdx.y;
  const int64_t T = (a.vox + 31) / 32;
  const int64_t t0 = (int64_t)unit * a.tiles_per_wg;
  const int64_t t1 = (t0 + a.tiles_per_wg < T) ? t0 + a.tiles_per_wg : T;
  f32x4 cf[NCH][8];
  load_coef<NCH>(a, b, half, cf);
  __shared__ __attribute__((aligned(16))) float sTr[8 * 32 * 36];  // per-wave output tile on its way to row-major quads
  __shared__ __attribute__((aligned(16))) f32x4 sCfT[2 * NCH * 8];
  const f32x4* cfl = nullptr;
  if constexpr (AttnCfl<NCH, MOM>::value) cfl = attn_cf_table<NCH>(cf, sCfT);
  attn_pass2<NCH, MOM>(a, b, unit, (int)gridDim.x, t0, t1, cf, sTr, cfl);
}

// The whole Residual(PreNorm(LinearAttention)) of one sample in ONE workgroup, for grids of a few hundred voxels (Dataset-2 below
// level 0: 736 and 96 voxels), where the three launches of the general path -- pass 1, pass 2, GroupNorm + residual -- are three
// times a launch and a prologue for microseconds of work: pass 1 over all the sample's tiles, its {max, sum, context} through the
// (L2-resident) partial buffer into the fold of pass 2, pass 2 with y written un-normalised, then the closing GroupNorm(1, C)
// from the workgroup's own channel sums (the arithmetic of gn_defer.h) and  out = gn(y) + x  in place.
template <int NCH>
__global__ void __launch_bounds__(512) attn_small_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sOut[NCH * 32][2];  // {scale, shift} of the closing GroupNorm per channel
  __shared__ double sTot[2];
  const int tid = threadIdx.x, half = (tid & 63) >> 5;
  const int b = blockIdx.x;
  const int64_t T = (a.vox + 31) / 32;
  f32x4 cf[NCH][8];
  load_coef<NCH>(a, b, half, cf);
  __shared__ __attribute__((aligned(16))) f32x4 sCfT[2 * NCH * 8];
  const f32x4* cfl = nullptr;
  if constexpr (AttnCfl<NCH, false>::value) cfl = attn_cf_table<NCH>(cf, sCfT);
  attn_pass1<NCH>(a, b, 0, 0, T, cf, cfl);
  __threadfence_block();
  __syncthreads();  // the partial of this sample is in memory (written and read by this workgroup only)
  attn_pass2<NCH>(a, b, 0, 1, 0, T, cf, nullptr, cfl);
  __threadfence_block();
  __syncthreads();  // y and its channel sums are in memory
  const int C = NCH * 32;
  const float* cp = a.ch_part + (size_t)b * C * 2;
  if (tid < 64) {
    // one wave: lane l takes channels l, l + 64, then a fixed fp64 xor tree (one thread summing C pairs was C dependent trips to L2:
    // ~10 us of this 30-50 us kernel)
    double a1 = 0.0, a2 = 0.0;
    for (int c = tid; c < C; c += 64) {
      const float2 v = *(const float2*)(cp + c * 2);
      a1 += (double)v.x;
      a2 += (double)v.y;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      a1 += __shfl_xor(a1, m, 64);
      a2 += __shfl_xor(a2, m, 64);
    }
    if (tid == 0) {
      const double cnt = (double)a.vox * C;
      const double mu = a1 / cnt;
      double var = a2 / cnt - mu * mu;
      var = var < 0.0 ? 0.0 : var;
      sTot[0] = mu;
      sTot[1] = 1.0 / sqrt(var + 1e-5);
    }
  }
  __syncthreads();
  if (tid < C) {
    const float sc = (float)sTot[1] * a.out_gamma[tid];
    sOut[tid][0] = sc;
    sOut[tid][1] = a.out_beta[tid] - (float)sTot[0] * sc;
  }
  __syncthreads();
  float* const yb = a.y + (size_t)b * a.vox * C;
  const float* const xb = a.x + (size_t)b * a.vox * C;
  const int64_t n4 = a.vox * C / 4;
  for (int64_t i = tid; i < n4; i += 512) {
    const int c = (int)((i * 4) % C);
    const f32x4 y = ((const f32x4*)yb)[i], x = ((const f32x4*)xb)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = sOut[c + e][0] * y[e] + sOut[c + e][1] + x[e];
    ((f32x4*)yb)[i] = o;
  }
}

// Barrier among the P workgroups of one sample inside a launch (sense-reversing: the last arriver resets the count and bumps the
// generation the others poll).  Measured on MI355X (tools/micro/group_barrier.hip, profiles/r04_group_barrier.txt): 3.3 us per
// barrier when the P workgroups sit on ONE XCD -- their L2 is the point of coherence, agent-scope relaxed atomics and s_waitcnt
// suffice, no cache write-back -- 11 us with release / acquire fences, and it never completes across XCDs with relaxed polls (each
// XCD's L2 keeps its own copy of the polled line): hence attn_coop_kernel's blockIdx -> (sample, part) mapping.  The spin is
// BOUNDED: a barrier that does not complete raises the range flag, which makes the caller re-run the call on the full-range
// (single-workgroup, unfused) path, and falls through -- the grid always drains.
__device__ __forceinline__ void attn_group_barrier(unsigned* sync, int P, int* status) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned arrived = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == (unsigned)P - 1) {
      __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 21)) {  // ~0.1 s: the peers are not coming (not co-resident, another launch on the same words)
          if (status) atomicOr(status, 1);
          break;
        }
      }
    }
  }
  __syncthreads();
}

// attn_small_kernel with the voxels of a sample dealt to P co-operating workgroups: pass 1 over the part's
// tiles -> {max, sum, context} partial -> barrier -> pass 2 over the part's tiles (its fold merges the P partials exactly as the
// two-launch form merges its splits) -> barrier -> closing GroupNorm(1, C) from all parts' channel sums and  out = gn(y) + x
// over the part's voxels.  Same arithmetic as the level-0 three-launch path with nsplit = P (bit-identical to it, not to the
// single-workgroup form, whose partial is one piece).  blockIdx -> (sample, part): workgroups are dealt round-robin over the 8
// XCDs, so the parts of sample b = 8 j + x are blockIdx = 8 (P j + part) + x: one XCD, consecutive in its dispatch order.
template <int NCH>
__global__ void __launch_bounds__(512) attn_coop_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sOut[NCH * 32][2];
  __shared__ double sTot[2];
  const int tid = threadIdx.x, half = (tid & 63) >> 5;
  const int P = a.nsplit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / P) * 8 + xcd, part = slot % P;
  const int64_t T = (a.vox + 31) / 32;
  const int64_t t0 = (int64_t)part * a.tiles_per_wg;
  const int64_t t1 = (t0 + a.tiles_per_wg < T) ? t0 + a.tiles_per_wg : T;
  unsigned* const sync = a.sync + (size_t)b * 2;
  f32x4 cf[NCH][8];
  load_coef<NCH>(a, b, half, cf);
  __shared__ __attribute__((aligned(16))) f32x4 sCfT[2 * NCH * 8];
  const f32x4* cfl = nullptr;
  if constexpr (AttnCfl<NCH, false>::value) cfl = attn_cf_table<NCH>(cf, sCfT);
  attn_pass1<NCH>(a, b, part, t0, t1, cf, cfl);
  attn_group_barrier(sync, P, a.status);
  attn_pass2<NCH>(a, b, part, P, t0, t1, cf, nullptr, cfl);
  attn_group_barrier(sync, P, a.status);
  const int C = NCH * 32;
  if (tid < 64) {
    double a1 = 0.0, a2 = 0.0;
    for (int u = 0; u < P; ++u) {
      const float* cp = a.ch_part + ((size_t)b * P + u) * C * 2;
      for (int c = tid; c < C; c += 64) {
        a1 += (double)attn_peer_load(cp + c * 2, 1);
        a2 += (double)attn_peer_load(cp + c * 2 + 1, 1);
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      a1 += __shfl_xor(a1, m, 64);
      a2 += __shfl_xor(a2, m, 64);
    }
    if (tid == 0) {
      const double cnt = (double)a.vox * C;
      const double mu = a1 / cnt;
      double var = a2 / cnt - mu * mu;
      var = var < 0.0 ? 0.0 : var;
      sTot[0] = mu;
      sTot[1] = 1.0 / sqrt(var + 1e-5);
    }
  }
  __syncthreads();
  if (tid < C) {
    const float sc = (float)sTot[1] * a.out_gamma[tid];
    sOut[tid][0] = sc;
    sOut[tid][1] = a.out_beta[tid] - (float)sTot[0] * sc;
  }
  __syncthreads();
  float* const yb = a.y + (size_t)b * a.vox * C;
  const float* const xb = a.x + (size_t)b * a.vox * C;
  const int64_t v0 = t0 * 32, v1 = t1 * 32 < a.vox ? t1 * 32 : a.vox;
  for (int64_t i = v0 * C / 4 + tid; i < v1 * C / 4; i += 512) {
    const int c = (int)((i * 4) % C);
    const f32x4 y = ((const f32x4*)yb)[i], x = ((const f32x4*)xb)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = sOut[c + e][0] * y[e] + sOut[c + e][1] + x[e];
    ((f32x4*)yb)[i] = o;
  }
}

int tiles_per_wg_for(int64_t vox, int nsplit) {
  const int64_t T = (vox + 31) / 32;
  int64_t per = (T + nsplit - 1) / nsplit;
  per = (per + 7) / 8 * 8;
  return (int)per;
}

}  // namespace

// workgroups per sample for both passes: one round of the 256 CUs (a 512-thread workgroup of these kernels owns a CU: 8 waves
// x 256 VGPRs), at least 8 tiles (one per wave) each.  Two WGs per CU's worth (448 for Dataset-2's level 0 at batch 64) ran
// as two rounds, each paying the prologue again: 66 -> 4x us per launch.
int attn_fused_nsplit_for(int64_t vox, int batch) {
  const int64_t T = (vox + 31) / 32;
  static const int target = getenv("CD_ATTN_WGS") ? atoi(getenv("CD_ATTN_WGS")) : 256;
  int64_t want = (target + batch - 1) / batch;
  const int64_t cap = (T + 15) / 16;
  int64_t n = want < cap ? want : cap;
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  // tiles_per_wg is rounded up to a multiple of 8: drop workgroups that would start past the end
  const int per = tiles_per_wg_for(vox, (int)n);
  n = (T + per - 1) / per;
  return (int)n;
}

bool attn_moments_eligible(int C) { return C == 32; }  // (64 channels: pass 1 would spill ~130 registers)
size_t attn_moment_floats(int batch, int nsplit) { return (size_t)batch * nsplit * ATTN_MOM; }

void launch_attn_kv_context(const float* x, int C, const float* coef, const void* wqkv_f16x2, float* partials, int batch,
                            int64_t vox, int nsplit, hipStream_t s, const GnDefer* defer, int* status, float* moments) {
  CD_REQUIRE(C == 32 || C == 64 || C == 96 || C == 128, "fused attention: 32..128 input channels");
  CD_REQUIRE(!moments || attn_moments_eligible(C), "fused attention: the moment form takes 32 channels");
  AttnArgs a{};
  a.mom = moments;
  a.status = status;
  a.x = x; a.coef = coef; a.wqkv = (const u32x4*)wqkv_f16x2; a.C = C; a.vox = vox; a.tiles_per_wg = tiles_per_wg_for(vox, nsplit);
  a.partials = partials; a.nsplit = nsplit;
  if (defer) a.defer = *defer;
  char cat[64];
  std::snprintf(cat, sizeof cat, "attn_kv_context C%d n%ld", C, (long)vox);
  prof::Scope scope(cat, s, 2.0 * ((moments ? 3.0 : 2.0) * C + (moments ? 64 : 32)) * 32 * (double)vox * batch,
                    4.0 * batch * (double)vox * C);
  const dim3 grid((unsigned)nsplit, (unsigned)batch);
  if (moments) {
    hipLaunchKernelGGL((attn_kv_context_kernel<1, true>), grid, dim3(512), 0, s, a);
    CD_HIP(hipGetLastError());
    return;
  }
  switch (C / 32) {
    case 1: hipLaunchKernelGGL((attn_kv_context_kernel<1, false>), grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((attn_kv_context_kernel<2, false>), grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL((attn_kv_context_kernel<3, false>), grid, dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((attn_kv_context_kernel<4, false>), grid, dim3(512), 0, s, a); break;
  }
  CD_HIP(hipGetLastError());
}

void launch_attn_out(const float* x, int C, const float* coef, const void* wqkv_f16x2, const float* wT_b, const float* bias,
                     float* y, float* ch_part, int batch, int64_t vox, int nsplit, hipStream_t s, const GnDefer* defer,
                     const float* partials, const float* w_out, float scale, int* status, const float* moments,
                     const float* out_gamma, const float* out_beta) {
  CD_REQUIRE(C == 32 || C == 64 || C == 96 || C == 128, "fused attention: 32..128 channels");
  CD_REQUIRE(!moments || (attn_moments_eligible(C) && !wT_b && out_gamma && out_beta && (vox * C) % 4 == 0),
             "fused attention: the moment form folds the weights itself and needs the closing GroupNorm's parameters");
  AttnArgs a{};
  a.mom = const_cast<float*>(moments); a.out_gamma = out_gamma; a.out_beta = out_beta;
  a.status = status;
  a.x = x; a.coef = coef; a.wqkv = (const u32x4*)wqkv_f16x2; a.C = C; a.vox = vox; a.tiles_per_wg = tiles_per_wg_for(vox, nsplit);
  a.wT = wT_b; a.bias = bias; a.y = y; a.ch_part = ch_part;
  CD_REQUIRE(wT_b || (partials && w_out), "attn_out: folded weights or the pass-1 partials to fold them from");
  a.partials = const_cast<float*>(partials); a.nsplit = nsplit; a.fold_wout = w_out; a.fold_scale = scale;
  if (defer) a.defer = *defer;
  char cat[64];
  std::snprintf(cat, sizeof cat, "attn_out C%d n%ld", C, (long)vox);
  prof::Scope scope(cat, s, 2.0 * (2.0 * C) * 32 * (double)vox * batch, 4.0 * batch * (double)vox * C * 2);
  const dim3 grid((unsigned)nsplit, (unsigned)batch);
  if (moments) {
    hipLaunchKernelGGL((attn_out_kernel<1, true>), grid, dim3(512), 0, s, a);
    CD_HIP(hipGetLastError());
    return;
  }
  switch (C / 32) {
    case 1: hipLaunchKernelGGL((attn_out_kernel<1, false>), grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((attn_out_kernel<2, false>), grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL((attn_out_kernel<3, false>), grid, dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((attn_out_kernel<4, false>), grid, dim3(512), 0, s, a); break;
  }
  CD_HIP(hipGetLastError());
}

// One launch for the whole block (see attn_small_kernel).  partials: [B][64 + 1024]; ch_part: [B][C][2]; y: (B, vox, C), the block
// output.  Eligible: attn_small_eligible(vox).
bool attn_small_eligible(int64_t vox) {
  static const int max_vox = getenv("CD_ATTN_SMALL_MAX") ? atoi(getenv("CD_ATTN_SMALL_MAX")) : 1024;  // <= 4 tiles per wave
  return vox <= max_vox;
}
void launch_attn_small(const float* x, int C, const float* coef, const void* wqkv_f16x2, float* partials, const float* w_out,
                       float scale, const float* bias, const float* out_gamma, const float* out_beta, float* y, float* ch_part,
                       int batch, int64_t vox, hipStream_t s, const GnDefer* defer, int* status, int max_parts) {
  CD_REQUIRE(C == 32 || C == 64 || C == 96 || C == 128, "fused attention: 32..128 channels");
  CD_REQUIRE((vox * C) % 4 == 0, "fused attention: whole float4 rows");
  AttnArgs a{};
  a.status = status;
  a.x = x; a.coef = coef; a.wqkv = (const u32x4*)wqkv_f16x2; a.C = C; a.vox = vox; a.tiles_per_wg = tiles_per_wg_for(vox, 1);
  a.partials = partials; a.nsplit = 1; a.wT = nullptr; a.fold_wout = w_out; a.fold_scale = scale; a.bias = bias;
  a.y = y; a.ch_part = ch_part; a.out_gamma = out_gamma; a.out_beta = out_beta;
  if (defer) a.defer = *defer;
  char cat[64];
  std::snprintf(cat, sizeof cat, "attn_small C%d n%ld", C, (long)vox);
  prof::Scope scope(cat, s, 2.0 * (4.0 * C + 32) * 32 * (double)vox * batch, 4.0 * batch * (double)vox * C * 3);
  // several co-operating workgroups per sample (attn_coop_kernel): opt-in through CD_ATTN_COOP=<parts> (2 .. max_parts) while it
  // is being measured.  Needs whole XCD groups (batch a multiple of 8), every part at least one tile per wave-pair, and all
  // batch x parts workgroups co-resident (one 512-thread workgroup per CU).
  static const int coop_env = getenv("CD_ATTN_COOP") ? atoi(getenv("CD_ATTN_COOP")) : 0;
  const int64_t T = (vox + 31) / 32;
  int P = coop_env < max_parts ? coop_env : max_parts;
  while (P > 1 && ((int64_t)batch * P > 256 || T < 4 * P)) --P;
  if (P > 1 && batch % 8 == 0 && status) {
    int per = (int)((T + P - 1) / P);
    while (P > 1 && (int64_t)(P - 1) * per >= T) { --P; per = (int)((T + P - 1) / P); }  // (every part has a tile)
    if (P > 1) {
      static unsigned* sync = nullptr;
      static const int kSyncSamples = 8192;
      if (!sync) {
        CD_HIP(hipMalloc((void**)&sync, sizeof(unsigned) * 2 * kSyncSamples));
        CD_HIP(hipMemset(sync, 0, sizeof(unsigned) * 2 * kSyncSamples));
      }
      CD_REQUIRE(batch <= kSyncSamples, "attention: batch too large for the co-operative form");
      a.sync = sync; a.coop = 1; a.nsplit = P; a.tiles_per_wg = per;
      const dim3 cgrid((unsigned)(batch * P));
      switch (C / 32) {
        case 1: hipLaunchKernelGGL(attn_coop_kernel<1>, cgrid, dim3(512), 0, s, a); break;
        case 2: hipLaunchKernelGGL(attn_coop_kernel<2>, cgrid, dim3(512), 0, s, a); break;
        case 3: hipLaunchKernelGGL(attn_coop_kernel<3>, cgrid, dim3(512), 0, s, a); break;
        default: hipLaunchKernelGGL(attn_coop_kernel<4>, cgrid, dim3(512), 0, s, a); break;
      }
      CD_HIP(hipGetLastError());
      return;
    }
  }
  const dim3 grid((unsigned)batch);
  switch (C / 32) {
    case 1: hipLaunchKernelGGL(attn_small_kernel<1>, grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL(attn_small_kernel<2>, grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL(attn_small_kernel<3>, grid, dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL(attn_small_kernel<4>, grid, dim3(512), 0, s, a); break;
  }
  CD_HIP(hipGetLastError());
}

}  // namespace cd
