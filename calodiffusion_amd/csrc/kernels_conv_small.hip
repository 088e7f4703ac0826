// 3x3x3 stride-1 phi-periodic convolution for the deepest U-Net levels, where a whole sample is at most 128 voxels
// (Dataset-2 level 2: 12 x 4 x 2 = 96).  At that size the general flat kernel is pure latency: tens of tiny workgroups, each
// looping over sub-chunks and taps with a barrier and an L2 round trip per step (17-28 us per launch for ~2 us of MFMA work).
//
// Here one workgroup (4 waves) owns one (sample, 32-output-channel tile):
//  * the sample's input -- up to 64 channels at a time, optionally normalised on the fly (fused GroupNorm + SiLU + embedding) --
//    is split to f16x2 (split16.h) and laid out in LDS as a (D+2) x (H+2) x (W+1) block of voxel records: zero planes in
//    front and behind, the phi rows -1 and H holding copies of rows H-1 and 0, one zero record closing every r row.  Every
//    tap of every voxel is then "record + constant": no wrap or validity logic in the loop;
//  * the 27 x (cin/16) (tap, k-step) pairs are dealt round-robin to the 4 waves; a wave streams its weight fragments from L2
//    through a 6-deep register ring and applies each to all (<= 4) row tiles of the sample: 3 MFMAs per tile and pair;
//  * the four K-slices of every tile are exchanged through LDS, wave t sums tile t, adds the bias, stores and accumulates the
//    channel statistics; the workgroup has seen the whole sample, so it emits ONE partial per (sample, channel).
#include "cd_common.h"
#include "split16.h"
#include "gn_defer.h"

#include <cstdio>
#include <cstdlib>

namespace cd {

namespace {

struct ConvSmallArgs {
  const float* in0;
  const float* in1;
  int c0, c1;
  const float* coef;  // [B][c0+c1][4] or null
  int act;
  const u32x4* wpk;   // f16x2 image [ks][tap][ct][term][lane]
  int CTtot;
  const float* bias;
  float* out;         // (B, vox, cout)
  int cout;
  float* ch_part;     // [B][1][cout][2] or null
  int D, H, W;
  int VB;             // bytes per voxel record = min(cin, CB) * 4 + 16
  int CB;             // input channels staged per block (64, or 96 / 128 where the image still fits LDS: fewer stage / barrier phases)
  int* status;        // bit 0: a staged value exceeded the fp16 range
  GnDefer defer;      // input normalisation folded in the prologue (table at cs_lds + coef_lds_off) instead of `coef`
  int coef_lds_off;
  const unsigned* in_absmax;  // power-of-two input rescaling (ConvFusion::in_absmax) or null
  const float* add_src;       // plain output only: out = conv + add_src (ConvFusion::add_src) or null
  // closing GroupNorm + SiLU + shortcut of a ResnetBlock in the epilogue (ConvFusion::GnOut), gamma == null: plain conv output
  const float *gn_gamma, *gn_beta;
  int gn_cpg;                 // channels per group (divides 32: a group never straddles two channel tiles)
  const float *res0, *res1;
  int res_c0;
  float* part_out;
  // whole ResnetBlock (cout == 32: one workgroup owns all channels of the sample): wpk2 != null.  The launch runs
  //   conv1 (wpk, bias) -> GroupNorm(gn1) + SiLU + embedding -> conv2 (wpk2, bias2) -> the closing GroupNorm + SiLU + shortcut above;
  // conv1's raw output goes through `h1` ((B, vox, 32), L2-resident: written and re-read by this workgroup only).
  const u32x4* wpk2;
  const float* bias2;
  const float *gn1_gamma, *gn1_beta, *emb;  // emb: (B, emb_ld) slice of this block or null
  int emb_ld;
  float* h1;
};

// weight fragments requested this many pairs ahead: 6 with four waves (512 registers each); 2 with eight (256 each: at 6 the kernel
// spilled 78 registers -- HGCal 57.6 showers/s at 6, 59.3 at 3, 59.8 at 2, same box -- and the second wave of a SIMD covers what the
// shallower ring exposes).  Tile groups -- waves 0-3 on tiles 0-1, waves 4-7 on tiles 2-3, 64 accumulator registers instead of 128, no
// spills with the six-deep ring -- were built and measured too: 57.7 against 59.3 showers/s, the doubled weight-fragment traffic
// from L2 costs more than the spills.
#ifndef CS_PD8
#define CS_PD8 2
#endif
template <int NW>
struct CsPd {
  static constexpr int value = NW == 8 ? CS_PD8 : 6;
};
// waves per workgroup: the (tap, k-step) pairs of a conv are dealt round-robin to them (K split).  4 where the step is power-bound
// (Dataset-2's deepest level: 8 waves measured the same step time for twice the partial exchange); 8 on four-tile samples (HGCal's 7 x 3 x 5
// grid at batch 16 lights 48 CUs, nothing is power-bound there: 1.466 -> 1.442 ms per denoise step, same-box A/B, round 4).  The choice
// depends on the geometry only, never on the batch: a shower's result must not change with the batch it is sampled in.
template <int NT, int CS_NW>  // NT: row tiles (32 voxels each) of the sample: ceil(vox / 32) <= 4
__global__ void __launch_bounds__(CS_NW * 64, 1) conv_small_f16x2_kernel(ConvSmallArgs a) {
  constexpr int CS_THREADS = CS_NW * 64;
  constexpr int CS_PD = CsPd<CS_NW>::value;
  extern __shared__ __attribute__((aligned(16))) char cs_lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
  // (wave-uniform by construction: telling the compiler so moves the pair -> (k-step, tap) decode, the tap displacement and the weight
  // fragment addresses to the scalar unit and its registers)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, ct = blockIdx.y;
  const int D = a.D, H = a.H, W = a.W, PV = H * W, vox = D * PV;
  const int pitch = W + 1, prow = (H + 2) * pitch;  // records per row / per plane
  const int nrec = (D + 2) * prow;
  const int VB = a.VB;
  char* const part = cs_lds;  // the partial exchange re-uses the image after the last MFMA

  // record (z, h, w) of the lane's voxel in each tile: address of its (kz = kh = 1, kw = 1) tap
  int rec[NT];
  bool okv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int v = t * 32 + col;
    okv[t] = v < vox;
    const int vv = okv[t] ? v : 0;
    const int z = vv / PV, p = vv - z * PV, h = p / W, w = p - h * W;
    rec[t] = (((z + 1) * (H + 2) + (h + 1)) * pitch + w) * VB + half * 16;
  }

  f32x16 accA[NT], accB[NT];
  float amax = 0.f;
  float gscale = 1.f, ginv = 1.f;
  if (a.in_absmax) pow2_scale_for(*a.in_absmax, &gscale, &ginv);
  if (a.defer.part) gn_defer_to_lds(a.defer, b, (float*)(cs_lds + a.coef_lds_off), cs_lds + a.coef_lds_off + a.defer.C * 16);

  // one convolution of the sample: input (in0 | in1), optionally normalised (coefficients from the LDS table or `coef`), into accA / accB
  auto conv_pass = [&](const float* in0, const float* in1, int c0, int c1, const float* coef_g, bool lds_table, int act,
                       const u32x4* wpk_pass) {
  const int cin = c0 + c1;
  const bool normed = coef_g || lds_table;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[t][r] = accB[t][r] = 0.f;
  for (int cb = 0; cb < cin; cb += a.CB) {  // input channels in blocks of <= CB
    const int cn = min(a.CB, cin - cb);     // channels in this block (multiple of 16)
    const int nq = cn >> 2;               // channel quads per voxel
    __syncthreads();                      // previous block's MFMAs have finished reading the image
    // ---- zero the image, then stage the sample (interior records; phi halo rows are copies) ----------------------
    for (int i = tid; i < nrec * VB / 16; i += CS_THREADS) ((u32x4*)cs_lds)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    // this wave's first weight fragments are requested now: their L2 latency hides behind the staging
    const int npairs = 27 * (cn >> 4);
    const int mine = (npairs - wave + CS_NW - 1) / CS_NW;
    const u32x4* wbase = wpk_pass + lane;
    auto wptr = [&](int i) {  // pair index i of this wave -> weight fragment pointer
      const int p = wave + CS_NW * i;
      const int ks = p / 27, tap = p - ks * 27;
      return wbase + ((size_t)(((cb >> 4) + ks) * 27 + tap) * a.CTtot + ct) * 128;
    };
    u32x4 wr[CS_PD][2];
#pragma unroll
    for (int i = 0; i < CS_PD; ++i)
      if (i < mine) {
        const u32x4* wp = wptr(i);
        wr[i][0] = wp[0];
        wr[i][1] = wp[64];
      }
    // staging: all of a thread's global loads are issued before the first conversion (4 items per batch).  Index arithmetic by
    // reciprocal ((i + 0.5) / d is never within float error of an integer for i < 2^20): run-time integer divisions were five per item,
    // ~200 instructions where the conversion itself is ~30
    const float inv_nq = 1.f / (float)nq, inv_pv = 1.f / (float)PV, inv_w = 1.f / (float)W;
    auto fdiv = [](int x, float inv) { return (int)(((float)x + 0.5f) * inv); };
    for (int i0 = tid; i0 < vox * nq; i0 += 4 * CS_THREADS) {
      f32x4 xs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = min(i0 + k * CS_THREADS, vox * nq - 1);
        const int v = fdiv(i, inv_nq), q = i - v * nq;
        const int c = cb + q * 4;
        const float* src = c < c0 ? in0 + ((size_t)b * vox + v) * c0 + c : in1 + ((size_t)b * vox + v) * c1 + (c - c0);
        xs[k] = *(const f32x4*)src;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * CS_THREADS;
        if (i >= vox * nq) break;
        const int v = fdiv(i, inv_nq), q = i - v * nq;
        const int c = cb + q * 4;
        f32x4 x = xs[k] * gscale;
        if (normed) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 cf = lds_table ? *(const f32x4*)(cs_lds + a.coef_lds_off + (c + e) * 16)
                                       : *(const f32x4*)(coef_g + ((size_t)b * cin + c + e) * 4);
            float t = cf[0] * x[e] + cf[1];
            if (act) t = cd_fast_silu(t);
            x[e] = t + cf[2];
          }
        }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))));
        u32x2 t1, t2;
        split2(x, t1, t2);
        const int z = fdiv(v, inv_pv), p = v - z * PV, h = fdiv(p, inv_w), w = p - h * W;
        // record layout: [k-step][term][16 ch] => quad q sits at (q >> 2) * 64 + term * 32 + (q & 3) * 8
        const int off = (q >> 2) * 64 + (q & 3) * 8;
        char* d = cs_lds + (((z + 1) * (H + 2) + (h + 1)) * pitch + w) * VB + off;
        *(u32x2*)d = t1;
        *(u32x2*)(d + 32) = t2;
        if (h == 0) {  // copy into the phi halo row H
          char* d2 = d + H * pitch * VB;
          *(u32x2*)d2 = t1;
          *(u32x2*)(d2 + 32) = t2;
        }
        if (h == H - 1) {  // copy into the phi halo row -1
          char* d2 = d - H * pitch * VB;
          *(u32x2*)d2 = t1;
          *(u32x2*)(d2 + 32) = t2;
        }
      }
    }
    __syncthreads();

    // ---- this wave's (tap, k-step) pairs of the block: p = wave, wave + 4, ... ---------------------------------------
    for (int i0 = 0; i0 < mine; i0 += CS_PD) {
#pragma unroll
      for (int k = 0; k < CS_PD; ++k) {
        const int i = i0 + k;
        if (i < mine) {
          const int p = wave + CS_NW * i;
          const int ks = p / 27, tap = p - ks * 27;
          const int kz = tap / 9, kh = (tap - kz * 9) / 3, kw = tap - kz * 9 - kh * 3;
          const int toff = ((kz - 1) * prow + (kh - 1) * pitch + (kw - 1)) * VB + ks * 64;
          const u32x4 w1 = wr[k][0], w2 = wr[k][1];
          if (i + CS_PD < mine) {  // refill this ring slot for pair i + CS_PD
            const u32x4* wp = wptr(i + CS_PD);
            wr[k][0] = wp[0];
            wr[k][1] = wp[64];
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const char* ap = cs_lds + rec[t] + toff;
            const u32x4 a1 = *(const u32x4*)ap, a2 = *(const u32x4*)(ap + 32);
            accA[t] = MFMA_F16(a1, w1, accA[t]);
            accB[t] = MFMA_F16(a1, w2, accB[t]);
            accB[t] = MFMA_F16(a2, w1, accB[t]);
          }
        }
      }
    }
  }
  };  // conv_pass

  // ---- exchange the K-slices: wave t sums tile t -> its 16 rows x (channel = col), bias added ------------------------
  float hold[16];
  auto reduce_tiles = [&](const float* bias_pass) {
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    char* d = part + ((t * CS_NW + wave) * 4) * 1024 + lane * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(f32x4*)(d + g * 1024) = f32x4{accA[t][4 * g] + accB[t][4 * g] * (1.f / 2048.f), accA[t][4 * g + 1] + accB[t][4 * g + 1] * (1.f / 2048.f),
                                      accA[t][4 * g + 2] + accB[t][4 * g + 2] * (1.f / 2048.f),
                                      accA[t][4 * g + 3] + accB[t][4 * g + 3] * (1.f / 2048.f)};
  }
  __syncthreads();
  if (wave < NT) {
    const float bv = bias_pass ? bias_pass[ct * 32 + col] : 0.f;
    f32x16 sum;
#pragma unroll
    for (int w = 0; w < CS_NW; ++w) {
      const char* d = part + ((wave * CS_NW + w) * 4) * 1024 + lane * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 x = *(const f32x4*)(d + g * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[4 * g + e] = w == 0 ? x[e] : sum[4 * g + e] + x[e];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) hold[r] = sum[r] * ginv + bv;
  }
  };  // reduce_tiles
  // per-channel sums of the held tile rows; store (optional) of the raw conv output
  auto tile_stats = [&](float* store_to, int ld, float& s1, float& s2) {
    s1 = s2 = 0.f;
    if (wave < NT) {
      float* o = store_to ? store_to + ((size_t)b * vox + wave * 32) * ld + ct * 32 + col : nullptr;
      const float* ad = (a.add_src && store_to == a.out) ? a.add_src + ((size_t)b * vox + wave * 32) * ld + ct * 32 + col : nullptr;
      if (ad) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (wave * 32 + row < vox) hold[r] += ad[(size_t)row * ld];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (wave * 32 + row < vox) {
          const float v = hold[r];
          if (o) o[(size_t)row * ld] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
  };
  // GroupNorm of the held (sample, 32-channel tile) from its own sums: {scale, shift} of channel col into out2[col * stride ...]
  // -- the arithmetic of gn_defer.h (fp64 group sums of the per-channel float sums).  Ends with a barrier.
  auto group_norm_coef = [&](float s1, float s2, const float* gamma, const float* beta, int cpg, float* red, float* dst, int stride,
                             const float* add, int add_ld) {
    __syncthreads();  // previous users of the head of the LDS block are done
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) {
      red[(wave * 32 + col) * 2] = s1;
      red[(wave * 32 + col) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 32) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < CS_NW; ++w) {
        t1 += red[(w * 32 + tid) * 2];
        t2 += red[(w * 32 + tid) * 2 + 1];
      }
      // group sums over the tile's lanes: the cpg channels of a group are cpg consecutive lanes
      double a1 = 0.0, a2 = 0.0;
      const int g0 = tid - tid % cpg;
      for (int c = 0; c < cpg; ++c) {
        a1 += (double)__shfl(t1, g0 + c, 64);
        a2 += (double)__shfl(t2, g0 + c, 64);
      }
      const double cnt = (double)vox * cpg;
      const double mu = a1 / cnt;
      double var = a2 / cnt - mu * mu;
      var = var < 0.0 ? 0.0 : var;
      const float sc = (float)(1.0 / sqrt(var + 1e-5)) * gamma[ct * 32 + tid];
      dst[tid * stride] = sc;
      dst[tid * stride + 1] = beta[ct * 32 + tid] - (float)mu * sc;
      if (stride == 4) {
        dst[tid * 4 + 2] = add ? add[(size_t)b * add_ld + ct * 32 + tid] : 0.f;
        dst[tid * 4 + 3] = 0.f;
      }
    }
    __syncthreads();
  };

  // one pass = one convolution; a whole ResnetBlock (wpk2) is two, conv1's GroupNorm + SiLU + embedding being folded into the staging
  // of conv2 through the coefficient table in LDS.  (One loop body instead of two call sites of the lambdas: each is inlined once.)
  float s1 = 0.f, s2 = 0.f;
  float* const red = (float*)cs_lds;                        // [CS_NW][32][2] per-wave channel sums (head of the image region)
  float* const table = (float*)(cs_lds + a.coef_lds_off);   // [32][4] {scale, shift, add, 0} of the GroupNorm this workgroup computes
  const int npass = a.wpk2 ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const bool second = pass == 1, last = pass == npass - 1;
    conv_pass(second ? a.h1 : a.in0, second ? nullptr : a.in1, second ? 32 : a.c0, second ? 0 : a.c1, second ? nullptr : a.coef,
              second ? true : a.defer.part != nullptr, second ? 1 : a.act, second ? a.wpk2 : a.wpk);
    reduce_tiles(second ? a.bias2 : a.bias);
    tile_stats(last ? (a.gn_gamma ? nullptr : a.out) : a.h1, last ? a.cout : 32, s1, s2);
    if (!last || a.gn_gamma)
      group_norm_coef(s1, s2, last ? a.gn_gamma : a.gn1_gamma, last ? a.gn_beta : a.gn1_beta, a.gn_cpg, red, table, 4,
                      last ? nullptr : a.emb, a.emb_ld);
    if (!last) {
      __threadfence_block();
      __syncthreads();  // h1 is in memory
    }
  }
  if (a.status && amax > 65504.f) atomicOr(a.status, 1);
  if (a.gn_gamma) {
    // ---- the workgroup holds the whole (sample, channel tile): GroupNorm + SiLU, shortcut, store -- what gn_apply does in a launch of
    // its own, with the same arithmetic
    float y1 = 0.f, y2 = 0.f;
    if (wave < NT) {
      const float sc = table[col * 4], sh = table[col * 4 + 1];
      const int c = ct * 32 + col;
      float* o = a.out + ((size_t)b * vox + wave * 32) * a.cout + c;
      const float* rp;
      int rld;
      if (a.res1 && c >= a.res_c0) {
        rld = a.cout - a.res_c0;
        rp = a.res1 + ((size_t)b * vox + wave * 32) * rld + (c - a.res_c0);
      } else {
        rld = a.res1 ? a.res_c0 : a.cout;
        rp = a.res0 + ((size_t)b * vox + wave * 32) * rld + c;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (wave * 32 + row < vox) {
          float u = sc * hold[r] + sh;
          u = u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));  // (gn_apply's SiLU)
          const float y = u + rp[(size_t)row * rld];
          o[(size_t)row * a.cout] = y;
          y1 += y;
          y2 += y * y;
        }
      }
    }
    if (a.part_out) {
      __syncthreads();
      y1 += __shfl_xor(y1, 32, 64);
      y2 += __shfl_xor(y2, 32, 64);
      if (half == 0) {
        red[(wave * 32 + col) * 2] = y1;
        red[(wave * 32 + col) * 2 + 1] = y2;
      }
      __syncthreads();
      if (tid < 32) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < CS_NW; ++w) {
          t1 += red[(w * 32 + tid) * 2];
          t2 += red[(w * 32 + tid) * 2 + 1];
        }
        float* dst = a.part_out + ((size_t)b * a.cout + ct * 32 + tid) * 2;
        dst[0] = t1;
        dst[1] = t2;
      }
    }
    return;
  }
  if (a.ch_part) {
    __syncthreads();  // partial reads done: re-use the head of the LDS block for the statistics
    float* red = (float*)cs_lds;
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) {
      red[(wave * 32 + col) * 2] = s1;
      red[(wave * 32 + col) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 32) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < CS_NW; ++w) {
        t1 += red[(w * 32 + tid) * 2];
        t2 += red[(w * 32 + tid) * 2 + 1];
      }
      float* dst = a.ch_part + ((size_t)b * a.cout + ct * 32 + tid) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

// Eight waves from this many 32-voxel tiles per sample on (four below).  Four-tile samples (HGCal's deepest level): +1.7 % on its
// sampling step; three-tile samples (Dataset-2's 12 x 4 x 2 level, which the TRAINING step runs through this kernel 24 times --
// sampling takes deep_level_kernel there): 6.773 / 6.782 -> 6.626 / 6.680 ms per training step, same box (tools/lib_ab.sh cs3).
#ifndef CS_NW8_FROM
#define CS_NW8_FROM 3
#endif
constexpr int cs_waves_for(int NT) { return NT >= CS_NW8_FROM ? 8 : 4; }
template <int NT>
void launch_small_inst(const ConvSmallArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  constexpr int NW = cs_waves_for(NT);
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv_small_f16x2_kernel<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_small_f16x2_kernel<NT, NW>), grid, dim3(NW * 64), lds, s, a);
  CD_HIP(hipGetLastError());
}

}  // namespace

// Eligible: 3x3x3 stride 1, whole sample <= 128 voxels.  Returns false otherwise.
bool try_launch_conv_small(const float* in0, int c0, const float* in1, int c1, const void* wpk_f16x2, const float* bias, float* out,
                           int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu) {
  static const bool off = getenv("CD_NO_CONV_SMALL") != nullptr;
  if (off) return false;
  if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sz == 1 && g.sh == 1 && g.sw == 1)) return false;
  const int64_t vox = g.in.vox();
  if (vox > 128 || vox < 1) return false;
  const int cin = c0 + c1;
  if (cin % 16 || cout % 32 || c0 % 4 || c1 % 4) return false;
  // a 64-channel block must not straddle the two sources unless the split is at a multiple of 4 (quads never straddle)
  const int NT = (int)((vox + 31) / 32);
  const size_t partial = (size_t)NT * cs_waves_for(NT) * 4096;
  // coefficient table: of the deferred input normalisation and / or of the GroupNorm the kernel computes itself ([32][4] floats)
  const size_t table = fu.defer.part ? (size_t)fu.defer.C * 16 + gn_defer_scratch_bytes(fu.defer.C) : (fu.gn_out.gamma ? 512 : 0);
  // channels per staged block: all of them up to 128 if the image then still fits (HGCal's 96-channel level: one stage / barrier /
  // MFMA phase instead of 64 + 32), else 64.  A property of the geometry, never of the batch.
  static const int cb_max = getenv("CD_CONV_SMALL_CB") ? atoi(getenv("CD_CONV_SMALL_CB")) : 128;
  int CB = cin < cb_max ? cin : cb_max;
  if (CB < 64) CB = cin < 64 ? cin : 64;
  size_t lds = 0, coef_off = 0;
  int VB = 0;
  for (;; CB = 64) {
    VB = (cin < CB ? cin : CB) * 4 + 16;
    const size_t image = (size_t)(g.in.d + 2) * (g.in.h + 2) * (g.in.w + 1) * VB;
    lds = image > partial ? image : partial;
    lds = (lds + 255) & ~(size_t)255;
    coef_off = lds;
    lds += table;
    if (lds <= 160 * 1024 || CB <= 64) break;
  }
  if (lds > 160 * 1024) return false;
  ConvSmallArgs a;
  a.in0 = in0; a.in1 = in1; a.c0 = c0; a.c1 = c1; a.coef = fu.coef; a.act = fu.act;
  a.wpk = (const u32x4*)wpk_f16x2; a.CTtot = cout / 32; a.bias = bias; a.out = out; a.cout = cout; a.ch_part = fu.ch_part;
  a.D = g.in.d; a.H = g.in.h; a.W = g.in.w; a.VB = VB; a.CB = CB; a.status = fu.status;
  a.defer = fu.defer; a.coef_lds_off = (int)coef_off; a.in_absmax = fu.in_absmax;
  a.add_src = !bias ? fu.add_src : nullptr;
  a.gn_gamma = nullptr; a.gn_beta = nullptr; a.gn_cpg = 0; a.res0 = a.res1 = nullptr; a.res_c0 = 0; a.part_out = nullptr;
  a.wpk2 = nullptr; a.bias2 = nullptr; a.gn1_gamma = a.gn1_beta = a.emb = nullptr; a.emb_ld = 0; a.h1 = nullptr;
  const ConvFusion::GnOut& go = fu.gn_out;
  if (go.gamma && go.done && go.groups > 0 && cout % go.groups == 0 && 32 % (cout / go.groups) == 0 && go.res0 && !fu.in_absmax) {
    a.gn_gamma = go.gamma; a.gn_beta = go.beta; a.gn_cpg = cout / go.groups;
    a.res0 = go.res0; a.res1 = go.res1; a.res_c0 = go.res_c0; a.part_out = go.part_out;
    *go.done = 1;
  }
  const dim3 grid((unsigned)batch, (unsigned)(cout / 32));
  switch (NT) {
    case 1: launch_small_inst<1>(a, grid, lds, s); break;
    case 2: launch_small_inst<2>(a, grid, lds, s); break;
    case 3: launch_small_inst<3>(a, grid, lds, s); break;
    default: launch_small_inst<4>(a, grid, lds, s); break;
  }
  if (fu.units) *fu.units = (fu.ch_part && !a.gn_gamma) ? 1 : 0;
  if (a.add_src && !a.gn_gamma && fu.add_done) *fu.add_done = 1;
  return true;
}

// Whole ResnetBlock in one launch (see ConvSmallArgs::wpk2): cout == 32, the sample <= 128 voxels, f16x2 arithmetic.
//   out = silu(gn2(conv2(silu(gn1(conv1(x0 | x1))) + emb))) + shortcut;  h1: (B, vox, 32) scratch.  Returns false if not eligible.
bool try_launch_res_block_small(const float* x0, int c0, const float* x1, int c1, const void* w1_f16x2, const float* b1,
                                const float* gn1_gamma, const float* gn1_beta, const float* emb, int emb_ld, const void* w2_f16x2,
                                const float* b2, const float* gn2_gamma, const float* gn2_beta, int groups, const float* res0,
                                const float* res1, int res_c0, float* h1, float* out, float* part_out, int batch, int cout,
                                Dims3 dims, int* status, hipStream_t s) {
  static const bool off = getenv("CD_NO_CONV_SMALL") != nullptr || getenv("CD_NO_BLOCK_SMALL") != nullptr;
  if (off) return false;
  const int64_t vox = dims.vox();
  const int cin = c0 + c1;
  if (vox > 128 || vox < 1 || cout != 32 || cin % 16 || c0 % 4 || c1 % 4 || groups <= 0 || 32 % groups || !res0) return false;
  const int VB = (cin < 64 ? cin : 64) * 4 + 16;  // (conv2's 32 input channels need no more)
  const int NT = (int)((vox + 31) / 32);
  const size_t image = (size_t)(dims.d + 2) * (dims.h + 2) * (dims.w + 1) * VB;
  const size_t partial = (size_t)NT * cs_waves_for(NT) * 4096;
  size_t lds = image > partial ? image : partial;
  lds = (lds + 255) & ~(size_t)255;
  const size_t coef_off = lds;
  lds += 512;
  if (lds > 160 * 1024) return false;
  ConvSmallArgs a;
  a.in0 = x0; a.in1 = x1; a.c0 = c0; a.c1 = c1; a.coef = nullptr; a.act = 0;
  a.wpk = (const u32x4*)w1_f16x2; a.CTtot = 1; a.bias = b1; a.out = out; a.cout = 32; a.ch_part = nullptr;
  a.D = dims.d; a.H = dims.h; a.W = dims.w; a.VB = VB; a.CB = 64; a.status = status;
  a.defer = GnDefer(); a.coef_lds_off = (int)coef_off; a.in_absmax = nullptr; a.add_src = nullptr;
  a.gn_gamma = gn2_gamma; a.gn_beta = gn2_beta; a.gn_cpg = 32 / groups; a.res0 = res0; a.res1 = res1; a.res_c0 = res_c0;
  a.part_out = part_out;
  a.wpk2 = (const u32x4*)w2_f16x2; a.bias2 = b2; a.gn1_gamma = gn1_gamma; a.gn1_beta = gn1_beta; a.emb = emb; a.emb_ld = emb_ld;
  a.h1 = h1;
  char cat[96];
  std::snprintf(cat, sizeof cat, "resblock_small C%d->32 @%dx%dx%d", cin, dims.d, dims.h, dims.w);
  prof::Scope scope(cat, s, 2.0 * 27 * (cin + 32) * 32 * (double)vox * batch, 4.0 * batch * (double)vox * (cin + 64));
  const dim3 grid((unsigned)batch, 1u);
  switch (NT) {
    case 1: launch_small_inst<1>(a, grid, lds, s); break;
    case 2: launch_small_inst<2>(a, grid, lds, s); break;
    case 3: launch_small_inst<3>(a, grid, lds, s); break;
    default: launch_small_inst<4>(a, grid, lds, s); break;
  }
  return true;
}

}  // namespace cd
