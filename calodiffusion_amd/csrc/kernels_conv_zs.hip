// "z-slide" 3x3x3 stride-1 phi-periodic convolution for the full-resolution level of the U-Net (CylindricalConv,
// calodiffusion/models/models.py:65-96 as used by Block.proj :153) -- the kernel that dominates a denoise step.
//
// Arithmetic ("f16x2"): gfx950's f32-input MFMA runs at 1/16 of the 16-bit rate.  Every fp32 operand is split into two
// fp16 terms, x = x1 + 2^-11 x2' with x1 = f16(x), x2' = f16((x - x1) * 2^11) (22 significant bits; the 2^11 keeps the
// second term out of the fp16 subnormals), and the product is formed from three MFMAs into two fp32 accumulators,
//     A += x1*w1          B += x1*w2' + x2'*w1          result = A + 2^-11 B           (dropped: x2*w2 <= 2^-22 relative)
// fp16 x fp16 products are exact in fp32 and v_mfma_f32_32x32x16_f16 accumulates in fp32: the result carries fp32
// rounding-level error (measured 3.7e-7 relative on K = 864 against 3.7e-7 for an fp32 dot product) at 3/16 of the
// matrix-pipe time of the f32 MFMA.  Values beyond the fp16 range (|x| > 65504) turn into inf/NaN in the output and raise
// the plan's range flag; the bf16x3 / f32 kernels (CD_CONV_PRECISION) have the full fp32 range.
//
// Structure (one workgroup = 4 waves = one CU, one contiguous chunk of one sample's flattened (z, phi, r) voxels):
//  * the 27 taps x 2 sixteen-channel k-steps are SPLIT OVER THE 4 WAVES and each wave keeps its 13-14 (tap, k-step)
//    weight fragments in registers for the whole chunk (112 VGPRs): no weight traffic at all inside the loop;
//  * input planes live in a 5-slot LDS ring (144 B per voxel: 2 k-steps x 2 terms x 16 fp16 + 16 B pad => conflict-free
//    ds_read_b128 A-fragments); every input plane is fetched, normalised (fused GroupNorm + SiLU + embedding), split and
//    written ONCE per chunk, one plane per step, its global loads issued before the step's MFMAs and converted after;
//  * a step = 128 output voxels = 4 row tiles; every wave runs its K-slice over all 4 tiles, the partial 32x32 tiles
//    are exchanged through LDS (48 KiB) and wave t sums, adds bias, stores and accumulates the channel statistics of
//    tile t.  Fixed summation order => deterministic.
#include "cd_common.h"
#include "split16.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cd {

// packed f16x2 weights: [k-step = ci/16][tap][ct = co/32][term][lane = h*32+j][8 fp16] = W_term[co = ct*32+j][ci = ks*16+8h+0..7]
__global__ void pack_weights_f16x2_kernel(const float* __restrict__ w, u32x4* __restrict__ wpk, int cout, int cin, int taps,
                                          size_t total, int transposed, int flip) {
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;  // one thread per (ks, tap, ct, lane)
  if (idx >= total) return;
  const int lane = idx & 63;
  size_t rest = idx >> 6;
  const int CT = (cout + 31) / 32;
  const int ct = rest % CT;
  rest /= CT;
  const int tap = rest % taps;
  const int ks = rest / taps;
  const int h = lane >> 5, j = lane & 31;
  const int co = ct * 32 + j;
  f32x4 v[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ks * 16 + h * 8 + e;
    const int st = flip ? taps - 1 - tap : tap;
    const size_t src = transposed ? ((size_t)ci * cout + co) * taps + st : ((size_t)co * cin + ci) * taps + st;
    v[e >> 2][e & 3] = (co < cout && ci < cin) ? w[src] : 0.f;
  }
  u32x2 a1, a2, b1, b2;
  split2(v[0], a1, a2);
  split2(v[1], b1, b2);
  u32x4* dst = wpk + (((size_t)(ks * taps + tap) * CT + ct) * 2) * 64 + lane;
  dst[0] = u32x4{a1[0], a1[1], b1[0], b1[1]};
  dst[64] = u32x4{a2[0], a2[1], b2[0], b2[1]};
}

void launch_pack_weights_f16x2(const float* w_torch, void* wpk, int cout, int cin, int taps, hipStream_t s, bool transposed,
                               bool flip) {
  CD_REQUIRE(cin % 16 == 0, "f16x2 convolution needs input channels in multiples of 16");
  const size_t total = (size_t)(cin / 16) * taps * ((cout + 31) / 32) * 64;
  hipLaunchKernelGGL(pack_weights_f16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_torch, (u32x4*)wpk, cout,
                     cin, taps, total, transposed ? 1 : 0, flip ? 1 : 0);
  CD_HIP(hipGetLastError());
}

namespace {

constexpr int ZS_VB = 144;     // bytes per voxel record in the ring
constexpr int ZS_RING = 5;     // planes in the ring
constexpr int ZS_NSL = 5;      // staging slots per thread per plane (plane <= 160 voxels)
constexpr int ZS_PART = 3 * 4 * 4096;  // partial-tile exchange: 4 tiles x 3 foreign waves x 4 KiB

struct ConvZsArgs {
  const float* in;   // (B, vox, ldc) channels-last, already offset to the first of the 32 input channels
  int ldc;
  const float* coef; // [B][coef_c][4] already offset to the same first channel, or null
  int coef_c, act;
  const u32x4* wpk;  // f16x2 image, already offset to the first k-step of these 32 input channels
  int CTtot;
  const float* bias; // null for a continuation launch
  float* out;        // (B, vox, cout)
  int cout;
  float* ch_part;    // [B][nchunk*4][cout][2] or null
  int D, H, W;
  int nchunk, CV;    // voxels per chunk (multiple of 128)
  int* status;       // bit 0: a staged value exceeded the fp16 range
  int dbg;           // timing experiments (CD_ZS_DBG): 1 = no conversion, 2 = no MFMA loop, 4 = no reduce/store
};

template <int WV, bool ACC>
__device__ __forceinline__ void zs_wave(const ConvZsArgs& a, char* lds) {
  constexpr int KSTEP = WV >> 1, TB = (WV & 1) * 14, NP = (WV & 1) ? 13 : 14;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int chunk = blockIdx.x, b = blockIdx.y, ct = blockIdx.z;
  const int H = a.H, W = a.W, PV = H * W, vox = a.D * PV;
  const int PVB = PV * ZS_VB;
  const int ZADDR = ZS_RING * PVB;                                  // all-zero voxel record
  char* const part = lds + ((ZADDR + ZS_VB + 255) & ~255);

  // ---- this wave's weight fragments: registers for the whole chunk -------------------------------------------
  u32x4 w1[NP], w2[NP];
  {
    const u32x4* wq = a.wpk + ((size_t)(KSTEP * 27 + TB) * a.CTtot + ct) * 128 + lane;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      w1[j] = wq[(size_t)j * a.CTtot * 128];
      w2[j] = wq[(size_t)j * a.CTtot * 128 + 64];
    }
  }

  const int v0 = chunk * a.CV;
  const int cend = min(v0 + a.CV, vox);
  const int nsteps = (cend - v0 + 127) >> 7;

  // ---- staging role: thread = (channel quad q, voxel p0 + 32k) --------------------------------------------------
  const int q = tid & 7, p0 = tid >> 3;
  f32x4 cf[4];
  if (a.coef) {
#pragma unroll
    for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(a.coef + ((size_t)b * a.coef_c + q * 4 + e) * 4);
  }
  const float bv = a.bias ? a.bias[ct * 32 + col] : 0.f;
  const float* src_b = a.in + (size_t)b * vox * a.ldc + q * 4;
  const int st_off = (q >> 2) * 64 + (q & 3) * 8;
  float amax = 0.f;
  f32x4 ld[ZS_NSL];
  auto ring_slot = [&](int z) { return (z + ZS_RING) % ZS_RING; };  // z >= -1
  // Loads are unconditional (plane and voxel indices clamped into range; an out-of-range plane is zero-filled by
  // convert()): a predicated load would be sunk by the compiler into convert()'s matching branch, behind the MFMAs.
  auto issue = [&](int z) {
    const int zc = min(max(z, 0), a.D - 1);
    const float* src = src_b + (size_t)zc * PV * a.ldc;
#pragma unroll
    for (int k = 0; k < ZS_NSL; ++k) {
      const int p = min(p0 + 32 * k, PV - 1);
      ld[k] = *(const f32x4*)(src + (size_t)p * a.ldc);
    }
  };
  auto convert = [&](int z) {
    char* dst = lds + ring_slot(z) * PVB + st_off;
    const bool zero = z < 0 || z >= a.D;
#pragma unroll
    for (int k = 0; k < ZS_NSL; ++k) {
      const int p = p0 + 32 * k;
      if (p < PV) {
        u32x2 t1 = {0u, 0u}, t2 = {0u, 0u};
        if (!zero && !(a.dbg & 1)) {
          f32x4 v = ld[k];
          if (a.coef) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = cf[e][0] * v[e] + cf[e][1];
              if (a.act) t = t / (1.f + expf(-t));
              v[e] = t + cf[e][2];
            }
          }
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          split2(v, t1, t2);
        }
        *(u32x2*)(dst + p * ZS_VB) = t1;
        *(u32x2*)(dst + p * ZS_VB + 32) = t2;
      }
    }
  };

  // ---- prologue: zero record, planes needed by step 0 -----------------------------------------------------------
  if (tid < ZS_VB / 4) ((float*)(lds + ZADDR))[tid] = 0.f;
  const int zfirst = v0 / PV;
  int zstaged = (min(v0 + 127, cend - 1)) / PV + 1;
  for (int z = zfirst - 1; z <= zstaged; ++z) {
    issue(z);
    convert(z);
  }

  // ---- per-lane geometry of its row (voxel) in the current tile, advanced by 32 voxels per tile -------------------
  int gz, gh, gw, grs;  // plane, phi row, r column, ring slot of plane gz-1
  {
    const int v = v0 + col;
    gz = v / PV;
    const int p = v - gz * PV;
    gh = p / W;
    gw = p - gh * W;
    grs = (gz + ZS_RING - 1) % ZS_RING;
  }
  const int adv_h = 32 / W, adv_w = 32 - adv_h * W;
  const int WB = W * ZS_VB;
  const int kconst = KSTEP * 64 + half * 16;
  float s1 = 0.f, s2 = 0.f;
  float* const out_b = a.out + (size_t)b * vox * a.cout + ct * 32 + col;

  // every pre-loop load (weights, coefficients, bias) has landed: no vmcnt wait may be needed inside the step loop
  // other than the one on the incoming plane (a conservative vmcnt(0) there would serialise the output stores)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    const int vs = v0 + s * 128;
    int zin = -2;
    if (s + 1 < nsteps) {
      const int need = min(vs + 255, cend - 1) / PV + 1;
      if (zstaged < need) zin = zstaged + 1;
    }
    issue(zin);
    __builtin_amdgcn_sched_barrier(0);

    // geometry of the 4 tiles of this step
    int base[4][3], ro0[4], ro2[4];
    bool okl[4], okr[4], okv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pb = (gh * W + gw) * ZS_VB + kconst;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        int sl = grs + kz;
        sl = sl >= ZS_RING ? sl - ZS_RING : sl;
        base[t][kz] = sl * PVB + pb;
      }
      ro0[t] = gh > 0 ? -WB : (H - 1) * WB;
      ro2[t] = gh < H - 1 ? WB : -(H - 1) * WB;
      okv[t] = vs + t * 32 + col < cend;
      okl[t] = okv[t] && gw > 0;
      okr[t] = okv[t] && gw < W - 1;
      // advance to the next tile
      gw += adv_w;
      gh += adv_h;
      if (gw >= W) { gw -= W; gh += 1; }
      if (gh >= H) { gh -= H; gz += 1; grs = grs == ZS_RING - 1 ? 0 : grs + 1; }
    }
    auto frag_addr = [&](int t, int j) -> int {
      const int tap = TB + j;
      const int kz = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int n = base[t][kz] + (kh == 0 ? ro0[t] : (kh == 2 ? ro2[t] : 0)) + (kw - 1) * ZS_VB;
      const bool ok = kw == 0 ? okl[t] : (kw == 2 ? okr[t] : okv[t]);
      return ok ? n : ZADDR + kconst;
    };

    f32x16 own;
    f32x16 accA, accB;
    u32x4 fa[3][2];
    constexpr int NI = 4 * NP;
    auto load_frag = [&](int i) {
      const int t = i / NP, j = i % NP;
      const char* p = lds + frag_addr(t, j);
      fa[i % 3][0] = *(const u32x4*)p;
      fa[i % 3][1] = *(const u32x4*)(p + 32);
    };
    load_frag(0);
    load_frag(1);
    if (!(a.dbg & 2))
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = i / NP, j = i % NP;
      if (i + 2 < NI && !(a.dbg & 8)) load_frag(i + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.f;
      }
      accA = MFMA_F16(fa[i % 3][0], w1[j], accA);
      accB = MFMA_F16(fa[i % 3][0], w2[j], accB);
      accB = MFMA_F16(fa[i % 3][1], w1[j], accB);
      if (j == NP - 1) {
        f32x16 pt;
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[r] = accA[r] + accB[r] * (1.f / 2048.f);
        if (t == WV) {
          own = pt;
        } else if (!(a.dbg & 16)) {
          const int slot = WV < t ? WV : WV - 1;
          char* d = part + ((t * 3 + slot) * 4) * 1024 + lane * 16;
#pragma unroll
          for (int g = 0; g < 4; ++g) *(f32x4*)(d + g * 1024) = f32x4{pt[4 * g], pt[4 * g + 1], pt[4 * g + 2], pt[4 * g + 3]};
        }
      }
    }

    if (zin != -2) {
      convert(zin);
      zstaged = zin;
    }
    __syncthreads();

    // ---- tile WV: sum the four K-slices in wave order, bias, store, statistics --------------------------------
    if (!(a.dbg & 4)) {
      f32x16 sum;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        f32x16 v;
        if (w == WV) {
          v = own;
        } else {
          const int slot = w < WV ? w : w - 1;
          const char* d = part + ((WV * 3 + slot) * 4) * 1024 + lane * 16;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 x = *(const f32x4*)(d + g * 1024);
            v[4 * g] = x[0]; v[4 * g + 1] = x[1]; v[4 * g + 2] = x[2]; v[4 * g + 3] = x[3];
          }
        }
        if (w == 0) sum = v;
        else
#pragma unroll
          for (int r = 0; r < 16; ++r) sum[r] += v[r];
      }
      const int vt = vs + WV * 32;
      float* o = out_b + (size_t)vt * a.cout;
      if (ACC) {  // continuation launch of a wider-K conv: add to what the previous launch stored
        f32x16 prev;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          prev[r] = vt + row < cend ? o[(size_t)row * a.cout] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] += prev[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (vt + row < cend) {
          const float v = sum[r] + bv;
          o[(size_t)row * a.cout] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    __syncthreads();
  }

  if (a.ch_part) {
    const float t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
    if (half == 0) {
      float* dst = a.ch_part + ((((size_t)b * a.nchunk + chunk) * 4 + WV) * a.cout + ct * 32 + col) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
  if (a.status && amax > 65504.f) atomicOr(a.status, 1);
}

template <bool ACC>
__global__ void __launch_bounds__(256, 1) conv_zslide_f16x2_kernel(ConvZsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char zs_lds[];
  switch (threadIdx.x >> 6) {
    case 0: zs_wave<0, ACC>(a, zs_lds); break;
    case 1: zs_wave<1, ACC>(a, zs_lds); break;
    case 2: zs_wave<2, ACC>(a, zs_lds); break;
    default: zs_wave<3, ACC>(a, zs_lds); break;
  }
}

}  // namespace

// Eligible: 3x3x3 stride 1, planes of 128..158 voxels (Dataset-2's 16x9), 32-channel input blocks.  Returns false otherwise.
bool try_launch_conv_zslide(const float* in0, int c0, const float* in1, int c1, const void* wpk_f16x2, const float* bias, float* out,
                            int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu) {
  if (getenv("CD_NO_ZSLIDE")) return false;
  if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sz == 1 && g.sh == 1 && g.sw == 1)) return false;
  const int PV = g.in.h * g.in.w;
  const int64_t vox = g.in.vox();
  if (PV < 128 || PV * 8 > ZS_NSL * 256) return false;
  const size_t lds = (((size_t)ZS_RING * PV * ZS_VB + ZS_VB + 255) & ~(size_t)255) + ZS_PART;
  if (lds > 160 * 1024) return false;
  if (vox < 256 || cout % 32 || c0 % 32 || c1 % 32) return false;
  const int CTtot = cout / 32;
  // chunks per sample: fill the 256 CUs (one workgroup each) with as few rounds and as little halo restaging as possible
  int best = 1;
  double best_eff = 0.0;
  const int max_chunks = (int)(vox / 256);
  for (int n = 1; n <= max_chunks && n <= 64; ++n) {
    const int64_t cv = ((vox + n - 1) / n + 127) / 128 * 128;
    const int nc = (int)((vox + cv - 1) / cv);
    if (nc != n) continue;
    const int64_t total = (int64_t)batch * nc * CTtot;
    const int64_t rounds = (total + 255) / 256;
    const double planes = (double)cv / PV;
    const double eff = (double)total / (rounds * 256.0) * planes / (planes + 2.5);  // halo planes + prologue
    if (eff > best_eff * 1.0001) { best_eff = eff; best = n; }
  }
  const int nchunk = best;
  const int CV = (int)(((vox + nchunk - 1) / nchunk + 127) / 128 * 128);
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv_zslide_f16x2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CD_HIP(hipFuncSetAttribute((const void*)conv_zslide_f16x2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int nblk = (c0 + c1) / 32;
  for (int kb = 0; kb < nblk; ++kb) {
    ConvZsArgs a;
    const int ch = kb * 32;
    if (ch < c0) { a.in = in0 + ch; a.ldc = c0; }
    else { a.in = in1 + (ch - c0); a.ldc = c1; }
    a.coef = fu.coef ? fu.coef + (size_t)ch * 4 : nullptr;
    a.coef_c = c0 + c1;
    a.act = fu.act;
    a.wpk = (const u32x4*)wpk_f16x2 + (size_t)(kb * 2) * 27 * CTtot * 128;
    a.CTtot = CTtot;
    a.bias = kb == 0 ? bias : nullptr;
    a.out = out;
    a.cout = cout;
    a.ch_part = kb == nblk - 1 ? fu.ch_part : nullptr;
    a.D = g.in.d; a.H = g.in.h; a.W = g.in.w;
    a.nchunk = nchunk; a.CV = CV;
    a.status = fu.status;
    a.dbg = getenv("CD_ZS_DBG") ? atoi(getenv("CD_ZS_DBG")) : 0;
    const dim3 grid((unsigned)nchunk, (unsigned)batch, (unsigned)CTtot);
    if (kb == 0) hipLaunchKernelGGL(conv_zslide_f16x2_kernel<false>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv_zslide_f16x2_kernel<true>, grid, dim3(256), lds, s, a);
    CD_HIP(hipGetLastError());
  }
  if (fu.units) *fu.units = nchunk * 4;
  return true;
}

}  // namespace cd
