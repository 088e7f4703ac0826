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
#include "gn_defer.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cd {

// packed f16x2 weights: [k-step = ci/16][tap][ct = co/32][term][lane = h*32+j][8 fp16] = W_term[co = ct*32+j][ci = ks*16+8h+0..7]
__device__ __forceinline__ void pack_weights_f16x2_elem(size_t idx, const float* __restrict__ w, u32x4* __restrict__ wpk, int cout,
                                                        int cin, int taps, int transposed, int flip) {
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
__global__ void pack_weights_f16x2_kernel(const float* __restrict__ w, u32x4* __restrict__ wpk, int cout, int cin, int taps,
                                          size_t total, int transposed, int flip) {
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;  // one thread per (ks, tap, ct, lane)
  if (idx >= total) return;
  pack_weights_f16x2_elem(idx, w, wpk, cout, cin, taps, transposed, flip);
}
// the f16x2 images of every tensor of a plan in one launch (PackJob, cd_common.h): blockIdx.y = job
__global__ void __launch_bounds__(256) pack_jobs_f16x2_kernel(const PackJob* __restrict__ jobs) {
  const PackJob j = jobs[blockIdx.y];
  if (!j.f16) return;
  const size_t t0 = blockIdx.x * (size_t)256 + threadIdx.x, stride = gridDim.x * (size_t)256;
  for (size_t i = t0; i < j.n_f16; i += stride)
    pack_weights_f16x2_elem(i, j.src, (u32x4*)j.f16, j.cout, j.cin, j.taps, j.kind == 2 || j.tr, j.flip);
}
void launch_pack_jobs_f16x2(const PackJob* d_jobs, int njobs, hipStream_t s) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(pack_jobs_f16x2_kernel, dim3(48, (unsigned)njobs), dim3(256), 0, s, d_jobs);
  CD_HIP(hipGetLastError());
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

constexpr int ZS_VB = 144;     // bytes per voxel record: 2 k-steps x 2 terms x 16 fp16 + 16 B pad (odd number of 16-B slots
                               // => conflict-free ds_read_b128 over consecutive records)
// planes in the LDS ring (ConvZsArgs::NR): a 64-voxel step reads 3-4 planes and one more is staged for the next step.
// With planes of >= 128 voxels two consecutive steps cross at most one plane boundary and 4 slots suffice; smaller
// (strip-)planes need 5.
constexpr int ZS_ZERO = 1024;  // zero area in front of the ring: the r - 1 / r + 1 neighbours of the edge columns read it, at the lane's own
                               // offset mod 256 plus the tap's constant (<= 2 records + 48 B): 255 + 288 + 48 < 1024
constexpr int ZS_NSL = 5;      // staging slots per helper thread per plane (plane <= 160 voxels)
constexpr int ZS_TILES = 2;    // 32-voxel row tiles per step
constexpr int ZS_STEP = 32 * ZS_TILES;
constexpr int ZS_PART = 2 * ZS_TILES * 4 * 4096;  // partial-tile exchange, double buffered: 2 x tiles x 4 K-slices x 4 KiB

struct ConvZsArgs {
  const float* in;   // (B, vox, ldc) channels-last, already offset to the first of the 32 input channels
  int ldc;
  const float* coef; // [B][coef_c][4] already offset to the same first channel, or null
  int coef_c, act;
  const u32x4* wpk;  // f16x2 image, already offset to the first k-step of these 32 input channels
  int CTtot;
  const float* bias; // null for a continuation launch
  long long acc_delta = 0;  // continuation instances read what they add to at out + acc_delta bytes (ConvFusion::add_src; 0: out itself)
  float* out;        // (B, vox, cout)
  int cout;
  float* ch_part;    // [B][nchunk*4][cout][2] or null
  int D, H, W;
  int NR;            // planes in the LDS ring (4 or 5, see above)
  int HS;            // phi rows per strip: H (whole planes, phi wrap by address select) or a divisor of H (strips with halo rows)
  int nchunk, CV;    // chunks per strip; voxels per chunk (multiple of ZS_STEP)
  int* status;       // bit 0: a staged value exceeded the fp16 range
  GnDefer defer;     // input normalisation folded in the prologue (table of all defer.C channels in LDS) instead of `coef`
  int choff;         // first of this launch's 32 input channels in that table
  const unsigned* in_absmax;  // input rescaling by a power of two (ConvFusion::in_absmax) or null
  int dbg;           // timing experiments (builds with -DCD_ZS_EXPERIMENTS, CD_ZS_DBG): 2 = no plane loads / conversion,
                     // 4 = no reduce/store, 16 = no MFMAs or fragment reads, 32 = no fragment reads, 64 = no MFMAs
};

// LDS image: [512 B of zeros][ring: NR planes][partials].  A plane is H rows of W records, planes 256-byte aligned: the
// records of consecutive voxels are 144 B apart everywhere -- across row ends, across planes (the slot stride is a multiple of
// 256 B) and across the phi wrap of a whole plane when H*W is a multiple of 16 -- so the 16 lanes of a ds_read_b128 phase always
// hit 16 different bank quads.  (A zero pad record per row, as in the first version, shifts every row by 144 B mod 256 and
// made two lanes of most phases collide.)  The r - 1 neighbour of column 0 and the r + 1 neighbour of column W - 1 are read
// from the zero area instead, at the lane's own offset mod 256 so that the redirected lanes keep their bank quads.
// Tap addresses are (row base of (kz, kh)) + constant: the matrix waves spend ~1-2 VALU instructions per MFMA triple on
// addressing.  That matters: one vector issue port per SIMD serves the matrix wave's MFMAs (8 of every 32 cycles) AND every
// VALU instruction of both resident waves.
struct ZsGeo {
  int PV, vox;            // plane / sample size in voxels
  int SPV, halo, rows;    // strip-plane voxels (HS * W); strips carry one phi halo row on either side; image rows per plane
  int h0;                 // first phi row of this workgroup's strip
  int pitch, PLB, RB, ZPART;
  int chunk, strip;
  int v0, cend, nsteps, zfirst;  // chunk = voxels [v0, cend) of the strip's own flattened (z, phi-in-strip, r) index space
};
__device__ __forceinline__ ZsGeo zs_geo(const ConvZsArgs& a) {
  ZsGeo g;
  g.PV = a.H * a.W;
  g.vox = a.D * g.PV;
  g.SPV = a.HS * a.W;
  g.halo = a.HS < a.H;
  g.rows = a.HS + 2 * g.halo;
  g.strip = blockIdx.x / a.nchunk;
  g.chunk = blockIdx.x - g.strip * a.nchunk;
  g.h0 = g.strip * a.HS;
  g.pitch = a.W;
  g.PLB = (g.rows * g.pitch * ZS_VB + 255) & ~255;
  g.RB = ZS_ZERO;  // ring starts after the zero area
  g.ZPART = g.RB + a.NR * g.PLB;
  g.v0 = g.chunk * a.CV;
  g.cend = min(g.v0 + a.CV, a.D * g.SPV);
  g.nsteps = (g.cend - g.v0 + ZS_STEP - 1) / ZS_STEP;
  g.zfirst = g.v0 / g.SPV;
  return g;
}

// Workgroup barriers that wait for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, which would tie the
// helper waves' global loads (issued a step ahead) and output stores to every barrier.
__device__ __forceinline__ void zs_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void zs_barrier_bare() { asm volatile("s_barrier" ::: "memory"); }
// SiLU on the transcendental unit: t * rcp(1 + exp2(-t * log2 e)), ~3 ulp
__device__ __forceinline__ float zs_silu(float t) {
  return t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
}

// ---- matrix waves 0..3: K-slice WV of every tile ------------------------------------------------------------
// K split: wave WV owns k-step WV >> 1; its two waves share the 27 taps, [0, Ne) and [Ne, 27) with Ne = 14 on even tiles and
// 13 on odd tiles, so that every wave runs 27 (tap, k-step) pairs per two-tile step (both hold tap 13's fragments).
template <int WV, int DBG>
__device__ __forceinline__ void zs_matrix_wave(const ConvZsArgs& a, char* lds) {
  constexpr int KSTEP = WV >> 1, ODD = WV & 1, T0 = ODD ? 13 : 0;  // weights held: taps T0 .. T0+13
  static_assert(ZS_TILES == 2, "the tap split alternates over the two tiles of a step");
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  const int ct = blockIdx.z;
  const ZsGeo G = zs_geo(a);
  const int H = a.HS, W = a.W;  // rows of the strip-plane
  char* const part = lds + G.ZPART;

  u32x4 w1[14], w2[14];
  {
    const u32x4* wq = a.wpk + ((size_t)(KSTEP * 27 + T0) * a.CTtot + ct) * 128 + lane;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      w1[j] = wq[(size_t)j * a.CTtot * 128];
      w2[j] = wq[(size_t)j * a.CTtot * 128 + 64];
    }
  }

  // per-lane geometry of its row (voxel) in the current tile, advanced by 32 voxels per tile
  int gh, gw, grs;  // phi row, r column, ring slot of plane z-1
  {
    const int v = G.v0 + col;
    const int gz = v / G.SPV;
    const int p = v - gz * G.SPV;
    gh = p / W;
    gw = p - gh * W;
    grs = (gz + a.NR - 1) % a.NR;
  }
  const int adv_h = 32 / W, adv_w = 32 - adv_h * W;
  const int RWB = G.pitch * ZS_VB;  // bytes per row of records
  // constant part of every fragment address: ring base, this wave's k-step, this lane's channel half, and the -1 record of
  // the kw = 0 tap (so that the per-tap constants kw * ZS_VB are non-negative immediates)
  const int kconst = G.RB + KSTEP * 64 + half * 16 - ZS_VB;
  const int W1 = W - 1;

  if (a.defer.part) gn_defer_to_lds(a.defer, blockIdx.y, (float*)(lds + G.ZPART), lds + G.ZPART + a.defer.C * 16);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // weights have landed: no vmcnt wait inside the loop
  zs_barrier_lds();                    // P: prologue planes staged by the helper waves

  // pair i of a step: tile and tap (27 pairs: tile 0 then tile 1)
  constexpr int N0 = ODD ? 13 : 14;  // pairs of tile 0 (even tile): even wave [0,14), odd wave [14,27)
  constexpr int NI = 27;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < G.nsteps; ++s) {
    int rb[ZS_TILES][3][3];  // address of the kw = 0 tap's fragment for every (kz, kh)
    bool eL[ZS_TILES], eR[ZS_TILES];  // this lane's voxel is in the first / last column: its kw = 0 / kw = 2 taps read zeros
#pragma unroll
    for (int t = 0; t < ZS_TILES; ++t) {
      eL[t] = gw == 0;
      eR[t] = gw == W1;
      const int pb = ((gh + G.halo) * G.pitch + gw) * ZS_VB + kconst;
      // phi neighbours: strips carry halo rows; whole planes wrap around
      const int ro0 = (G.halo || gh > 0) ? -RWB : (H - 1) * RWB;
      const int ro2 = (G.halo || gh < H - 1) ? RWB : -(H - 1) * RWB;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        int sl = grs + kz;
        sl = sl >= a.NR ? sl - a.NR : sl;
        const int bz = sl * G.PLB + pb;
        rb[t][kz][0] = bz + ro0;
        rb[t][kz][1] = bz;
        rb[t][kz][2] = bz + ro2;
      }
      gw += adv_w;
      gh += adv_h;
      if (gw >= W) { gw -= W; gh += 1; }
      if (gh >= H) { gh -= H; grs = grs == a.NR - 1 ? 0 : grs + 1; }
    }
    constexpr int PD = 3;  // fragments are requested PD pairs ahead of their MFMAs
    u32x4 fa[PD + 1][2];
    auto pair_tile = [](int i) { return i < N0 ? 0 : 1; };
    auto pair_tap = [](int i) {  // absolute tap of pair i
      if (i < N0) return ODD ? 14 + i : i;            // tile 0: even wave 0..13, odd wave 14..26
      const int j = i - N0;
      return ODD ? 13 + j : j;                        // tile 1: even wave 0..12, odd wave 13..26
    };
    auto load_frag = [&](int i) {
      if (DBG & 32) return;
      const int t = pair_tile(i), tap = pair_tap(i);
      const int kz = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      int base = rb[t][kz][kh] + kw * ZS_VB;
      if (kw == 0) base = eL[t] ? (base & 255) : base;  // into the zero area, same bank quad
      if (kw == 2) base = eR[t] ? (base & 255) : base;
      const char* p = lds + base;
      fa[i % (PD + 1)][0] = *(const u32x4*)p;
      fa[i % (PD + 1)][1] = *(const u32x4*)(p + 32);
    };
    auto write_partial = [&](int t, const f32x16& A, const f32x16& B) {
      f32x16 pt;
#pragma unroll
      for (int r = 0; r < 16; ++r) pt[r] = A[r] + B[r] * (1.f / 2048.f);
      char* d = part + (((s & 1) * ZS_TILES * 4 + t * 4 + WV) * 4) * 1024 + lane * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(d + g * 1024) = f32x4{pt[4 * g], pt[4 * g + 1], pt[4 * g + 2], pt[4 * g + 3]};
    };
    f32x16 accA[2], accB[2];  // one accumulator pair per tile: tile 0's is folded and written under tile 1's MFMAs
    if (DBG & 16) {  // timing experiment: no fragment reads, no MFMAs
      zs_barrier_lds();
      continue;
    }
#pragma unroll
    for (int i = 0; i < PD; ++i) load_frag(i);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = pair_tile(i), j = pair_tap(i) - T0;
      const bool first = i == 0 || i == N0;
      if (i + PD < NI) load_frag(i + PD);
      __builtin_amdgcn_sched_barrier(0);
      if (DBG & 64) {  // timing experiment: fragments are read but not multiplied
        asm volatile("" ::"v"(fa[i % (PD + 1)][0]), "v"(fa[i % (PD + 1)][1]));
        if (first) { accA[t] = zero16; accB[t] = zero16; }
      } else {
        accA[t] = MFMA_F16(fa[i % (PD + 1)][0], w1[j], first ? zero16 : accA[t]);
        accB[t] = MFMA_F16(fa[i % (PD + 1)][0], w2[j], first ? zero16 : accB[t]);
        accB[t] = MFMA_F16(fa[i % (PD + 1)][1], w1[j], accB[t]);
      }
      if (i == N0 + 3) write_partial(0, accA[0], accB[0]);
    }
    write_partial(1, accA[1], accB[1]);
    zs_barrier_lds();  // A: this step's partials are complete (buffer s & 1); the next step's planes are staged
  }
}

// Sink for the output rows a helper lane does not own (chunk tail, the dummy epilogues).  Every epilogue issues exactly 8
// stores on every path, so the compiler can count the vector-memory operations between a plane's loads and their use
// (s_waitcnt vmcnt(8 + k) instead of vmcnt(k): in-order retirement would otherwise make every conversion wait for the previous
// step's output stores as well).
__device__ float zs_sink[8 * 256];  // one cell per (row, lane): eight stores to ONE address would be merged by the compiler

// ---- helper waves 4..7: stage incoming planes; sum / store half a tile of the previous step ----------------------------
template <bool ACC, int DBG>
__device__ __forceinline__ void zs_helper_wave(const ConvZsArgs& a, char* lds, const int h) {
  const int tid = threadIdx.x - 256, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y, ct = blockIdx.z;
  const ZsGeo G = zs_geo(a);
  const int PV = G.PV, SPV = G.SPV;
  const int NIMG = G.rows * a.W;  // voxels staged per plane: the strip's rows and, for strips, one halo row on either side
  const int chunk = G.chunk + G.strip * a.nchunk;
  char* const part = lds + G.ZPART;

  // staging role: thread = (channel quad q, image voxel p0 + 32k)
  const int q = tid & 7, p0 = tid >> 3;
  f32x4 cf[4];
  const bool normed = a.coef || a.defer.part;
  const float bv = a.bias ? a.bias[ct * 32 + col] : 0.f;
  const float* src_b = a.in + (size_t)b * G.vox * a.ldc + q * 4;
  int rec[ZS_NSL];   // byte offset of this thread's record k inside a plane image (+ its quad's place in the record)
  int srcv[ZS_NSL];  // ... and the voxel of the global plane it is filled from (phi halo rows wrap around)
#pragma unroll
  for (int k = 0; k < ZS_NSL; ++k) {
    const int p = min(p0 + 32 * k, NIMG - 1);
    const int ph = p / a.W, pw = p - ph * a.W;
    rec[k] = G.RB + (ph * G.pitch + pw) * ZS_VB + (q >> 2) * 64 + (q & 3) * 8;
    int sh = G.h0 + ph - G.halo;
    sh = sh < 0 ? sh + a.H : (sh >= a.H ? sh - a.H : sh);
    srcv[k] = sh * a.W + pw;
  }
  float amax = 0.f;
  float gscale = 1.f, ginv = 1.f;
  if (a.in_absmax) pow2_scale_for(*a.in_absmax, &gscale, &ginv);
  f32x4 ld[ZS_NSL];
  static_assert(ZS_NSL == 5, "landed() names the five staging registers");
  // Plane loads are issued and awaited by hand: vmcnt retires in order, and between a plane's loads (issued in interval s-1) and
  // their conversion (start of interval s) the wave issues exactly NYOUNG vector-memory operations -- the 8 output stores of
  // epilogue(s-2), themselves asm statements -- so `s_waitcnt vmcnt(NYOUNG)` waits for the loads and not for those stores.  (Left to the compiler the wait is vmcnt(4..0): conditional paths make it assume no younger operation,
  // and every conversion then also waits a store round trip.)  Out-of-range planes are clamped and zero-filled by convert().
  // (a continuation launch also reads its rows back: it simply waits for everything; so does the timing experiment that
  // drops the epilogue and with it the eight stores)
  constexpr int NYOUNG = (ACC || (DBG & 4)) ? 0 : 8;
  auto issue_to = [&](f32x4 (&dst)[ZS_NSL], int z) {
    const int zc = min(max(z, 0), a.D - 1);
    const float* src = src_b + (size_t)zc * PV * a.ldc;
#pragma unroll
    for (int k = 0; k < ZS_NSL; ++k) {
      const float* p = src + (size_t)srcv[k] * a.ldc;
      asm volatile("global_load_dwordx4 %0, %1, off ; zs_plane_load" : "=v"(dst[k]) : "v"(p) : "memory");
    }
  };
  auto issue = [&](int z) { issue_to(ld, z); };
#define ZS_LANDED(younger)                                                                                   \
  asm volatile("s_waitcnt vmcnt(%5) ; zs_landed"                                                             \
               : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4])                             \
               : "n"(younger)                                                                                \
               : "memory")
  auto convert_from = [&](const f32x4 (&src)[ZS_NSL], int z, int k0, int k1) {  // slots [k0, k1) of plane z
    const int slot = (z + a.NR) % a.NR;  // z >= -1
    const bool zero = z < 0 || z >= a.D;
    char* dst = lds + slot * G.PLB;
#pragma unroll
    for (int k = 0; k < ZS_NSL; ++k) {
      if (k < k0 || k >= k1) continue;
      if (p0 + 32 * k < NIMG) {
        u32x2 t1 = {0u, 0u}, t2 = {0u, 0u};
        if (!zero && !(DBG & 1)) {
          f32x4 v = src[k] * gscale;
          if (normed) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = cf[e][0] * v[e] + cf[e][1];
              if (a.act) t = zs_silu(t);
              v[e] = t + cf[e][2];
            }
          }
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          split2(v, t1, t2);
        }
        *(u32x2*)(dst + rec[k]) = t1;
        *(u32x2*)(dst + rec[k] + 32) = t2;
      }
    }
  };

  auto convert = [&](int z, int k0, int k1) { convert_from(ld, z, k0, k1); };

  // prologue: the zero area, planes needed by step 0 (z-1 .. z+1 of its first voxel, one more if the step crosses a plane
  // boundary: at most 4 since a plane holds at least one step).  Their loads all go out first, under the construction of the
  // GroupNorm table, so the prologue pays one memory latency instead of one per plane.
  if (tid < ZS_ZERO / 4) ((float*)lds)[tid] = 0.f;
  auto need = [&](int k) {  // highest plane that step k reads
    k = min(k, G.nsteps - 1);
    return min(G.v0 + k * ZS_STEP + ZS_STEP - 1, G.cend - 1) / SPV + 1;
  };
  int zstaged = need(0);
  f32x4 ldp[4][ZS_NSL];
  const int zp0 = G.zfirst - 1, npro = zstaged - zp0 + 1;  // 3 or 4
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < npro) issue_to(ldp[i], zp0 + i);
  if (a.defer.part) {  // table built by the whole workgroup in the (still unused) partial-exchange region
    gn_defer_to_lds(a.defer, b, (float*)part, part + a.defer.C * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(part + (a.choff + q * 4 + e) * 16);
  } else if (a.coef) {
#pragma unroll
    for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(a.coef + ((size_t)b * a.coef_c + q * 4 + e) * 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ldp[i][0]), "+v"(ldp[i][1]), "+v"(ldp[i][2]), "+v"(ldp[i][3]), "+v"(ldp[i][4])
                 :
                 : "memory");
    if (i < npro) convert_from(ldp[i], zp0 + i, 0, ZS_NSL);
  }
  for (int z = zp0 + 4; z <= zstaged; ++z) {  // (not reached for planes of >= 64 voxels)
    issue(z);
    ZS_LANDED(0);
    convert(z, 0, ZS_NSL);
  }
  float s1 = 0.f, s2 = 0.f;
  // a strip-space voxel v = z * SPV + p lives at global voxel z * PV + h0 * W + p
  float* const out_b = a.out + ((size_t)b * G.vox + (size_t)G.h0 * a.W) * a.cout + ct * 32 + col;
  const float inv_spv = 1.f / (float)SPV;
  auto gvox = [&](int v) {  // exact: (v + 0.5) / SPV is never within float error of an integer
    const int z = (int)(((float)v + 0.5f) * inv_spv);
    return z * PV + (v - z * SPV);
  };
  // Incoming planes (at most one per step: a plane is >= 64 voxels): converted while the matrix waves run the step BEFORE
  // the one that first reads the plane -- only then is its ring slot (plane - 4) free -- from loads issued a step earlier.
  int zpend = zstaged < need(1) ? zstaged + 1 : -2;
  if (zpend != -2) {
    issue(zpend);
    ZS_LANDED(0);  // no stores follow these loads: the count of the loop's wait does not hold for them
  }
  zs_barrier_lds();  // P

  const int th = h >> 1, rh = h & 1;  // this wave sums rows 16*rh .. 16*rh+15 (accumulator registers 8*rh .. 8*rh+7) of tile th
  float sum[8];
  auto read_partials = [&](int s) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const char* d = part + (((s & 1) * ZS_TILES * 4 + th * 4 + w) * 4 + 2 * rh) * 1024 + lane * 16;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 x = *(const f32x4*)(d + g * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[4 * g + e] = w == 0 ? x[e] : sum[4 * g + e] + x[e];
      }
    }
  };
  float* const sink = zs_sink + (threadIdx.x & 255);
  auto epilogue = [&](int s, bool live) {  // bias, store, statistics of this wave's 16 rows of step s
    const int vt = G.v0 + s * ZS_STEP + th * 32;
    float* dst[8];  // this lane's 8 rows
    bool ok[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = (r & 3) + 8 * ((r >> 2) + 2 * rh) + 4 * half;
      ok[r] = live && vt + row < G.cend;
      dst[r] = ok[r] ? out_b + (size_t)gvox(vt + row) * a.cout : sink + r * 256;
    }
    if (ACC) {  // continuation launch of a wider-K conv: add to what the previous launch stored
      float prev[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) prev[r] = ok[r] ? *(const float*)((const char*)dst[r] + a.acc_delta) : 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum[r] = sum[r] * ginv + prev[r];
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) sum[r] *= ginv;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float v = sum[r] + bv;
      // exactly one store instruction per row on every path (ZS_LANDED counts them): not left to the compiler, which
      // merged the eight sink stores of the peeled first interval into one and let a plane be converted before it landed
      asm volatile("global_store_dword %0, %1, off ; zs_row_store" ::"v"(dst[r]), "v"(v) : "memory");
      const float m = ok[r] ? v : 0.f;
      s1 += m;
      s2 += m * m;
    }
  };

  // interval s = the time the matrix waves spend in step s (between barriers A(s-1) and A(s)): conversion of the pending plane,
  // loads of the next one, then the 8 output stores of step s-1 (always 8: landed() counts them).
  for (int s = 0; s < G.nsteps; ++s) {
    if (zpend != -2) {
      if (!(DBG & 2)) {
        ZS_LANDED(NYOUNG);
        convert(zpend, 0, ZS_NSL);  // read first by step s+1
      }
      zstaged = zpend;
    }
    zpend = (s + 2 < G.nsteps && zstaged < need(s + 2)) ? zstaged + 1 : -2;
    if (zpend != -2 && !(DBG & 2)) issue(zpend);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) {
      read_partials(s - 1);  // s = 0: nothing to sum yet, the stores go to the sink
      epilogue(s - 1, s > 0);
    }
    zs_barrier_lds();  // A(s)
  }
  read_partials(G.nsteps - 1);
  epilogue(G.nsteps - 1, true);

  if (a.ch_part) {
    const float t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
    if (half == 0) {
      float* dst = a.ch_part + ((((size_t)b * gridDim.x + chunk) * 4 + h) * a.cout + ct * 32 + col) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
  if (a.status && amax > 65504.f) atomicOr(a.status, 1);
}

template <bool ACC, int DBG = 0>
__global__ void __launch_bounds__(512, 1) conv_zslide_f16x2_kernel(ConvZsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char zs_lds[];
  // (the GroupNorm coefficient table -- gn_defer_to_lds, a whole-workgroup call with barriers -- is built inside the two roles:
  // the helper waves first put the loads of the chunk's first planes in flight)
  switch (threadIdx.x >> 6) {
    case 0: zs_matrix_wave<0, DBG>(a, zs_lds); break;
    case 1: zs_matrix_wave<1, DBG>(a, zs_lds); break;
    case 2: zs_matrix_wave<2, DBG>(a, zs_lds); break;
    case 3: zs_matrix_wave<3, DBG>(a, zs_lds); break;
    default: zs_helper_wave<ACC, DBG>(a, zs_lds, (int)(threadIdx.x >> 6) - 4); break;
  }
}


// ============================================================================================================
// One-wave-per-SIMD form (the default).
//
// What the kernel above is bound by (round 2: in-kernel s_memtime stamps, PMC instruction counts, tools/micro/lds_mfma{2,3,4}):
// NOT the matrix pipe (SQ_VALU_MFMA_BUSY 43 %) but the vector instructions around it.  Beside a wave's OWN stream of
// v_mfma_f32_32x32x16_f16 every other vector instruction costs ~3 cycles of wall time (none hide: lds_mfma4), and the same
// instruction issued by the OTHER wave of the SIMD -- the helper wave here, the support phase of the ping-pong form tried in
// between -- proceeds at one per 10-17 cycles: ~330 helper instructions per 64-voxel step take as long as the step's 81 MFMAs
// (2,965 cycles with their LDS fragment reads), and the two roles wait for each other at every step barrier.
//
// So: 4 waves, one per SIMD, 512 registers each, every wave doing BOTH jobs in one instruction stream -- its K-slice of the
// MFMAs (as zs_matrix_wave) and a quarter of the staging / reduction work (as zs_helper_wave): the vector instructions then cost
// their ~3 cycles each and nothing waits on a partner.  Per 64-voxel step and wave:
//   convert the plane the NEXT step is the first to read (loads issued a step earlier, in registers: there is room now) ->
//   issue the loads of the plane after that -> sum the four K-slices of the PREVIOUS step's rows this wave owns (its own slice
//   never left its registers), bias, store, statistics -> tap addresses -> 81 MFMAs over the two row tiles, partial tiles to
//   the exchange buffer (double buffered by step parity) -> one barrier.
// LDS: [zeros 512][coef, bias, flag 1024][ring NR planes][exchange 2 x 24 KB]  (Dataset-2: 130.5 KB)
// ============================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));  // pairs: v_pk_add_f32 / v_pk_fma_f32
#ifndef Z3_PAD
#define Z3_PAD 0  // 1: plane images with a zero record per row (no per-lane select for the r +- 1 taps of the edge columns, but
                  // the shifted records break the conflict-free ds_read_b128 pattern: PMC showed 190 k bank-conflict cycles per
                  // shader engine and launch against 16 k for the linear image, whose edge lanes are redirected into the zero area)
#endif
constexpr int Z3_PD = 3;                  // fragment pairs requested ahead of their MFMAs
constexpr int Z3_COEF = 1024;             // [32][4] floats: the GroupNorm coefficients of this launch's 32 input channels;
                                          // +512: bias[32]; +640: range flag word
constexpr int Z3_XCH = 4 * 2048;          // exchange buffer of one step: 4 reducers x the partner's K-slice (8 rows x 64 lanes x 4 B)

// LDS access by byte address: an address_space(3) pointer made from the integer -- through the generic `lds + offset` form
// every access pays a `v_add_u32 addr, 0, offset` for the (zero) base of the dynamic LDS symbol
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* z3_lds(int byte_addr) {
  return (__attribute__((address_space(3))) T*)(uintptr_t)(unsigned)byte_addr;
}

// diagnostic build only (-DCD_ZS_EXPERIMENTS, CD_ZS_DBG=2048): per-wave cycle sums of the parts of a step
__device__ unsigned long long z3_stamp_buf[256 * 4 * 12];
__device__ __forceinline__ unsigned long long z3_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

struct Z3Geo : ZsGeo {
  int XCH;
};
__device__ __forceinline__ Z3Geo z3_geo(const ConvZsArgs& a) {
  Z3Geo g;
  (ZsGeo&)g = zs_geo(a);
  // plane image with one zero record before every row and one after the last: the kw = 0 / kw = 2 taps of the first / last column
  // read a pad instead of selecting the zero area per lane
  if (Z3_PAD) {
    g.pitch = a.W + 1;
    g.PLB = ((g.rows * g.pitch + 1) * ZS_VB + 255) & ~255;
  }
  g.RB = ZS_ZERO + Z3_COEF;
  g.XCH = g.RB + a.NR * g.PLB;
  g.ZPART = g.XCH;  // (prologue scratch of gn_defer_to_lds: the exchange region is still unused then)
  return g;
}

// MODE (compile-time specialisation: every run-time switch costs select / branch instructions in all 256 threads):
//   1 = NORMED (input = GroupNorm + SiLU + embedding of the tensor read), 2 = HALO (phi strips with halo rows instead of whole
//   planes), 4 = SCALED (input rescaled by a power of two from its max: the training gradients)
template <int WV, bool ACC, int MODE, int DBG = 0, int NSL = ZS_NSL>
__device__ __forceinline__ void z3_wave(const ConvZsArgs& a, char* lds) {
  constexpr bool NORMED = (MODE & 1) != 0, HALO = (MODE & 2) != 0, SCALED = (MODE & 4) != 0;
  // matrix role: all 27 taps of row tile TILE of the step, for the 16 input channels of k-step KSTEP.  (The K split used to be
  // four ways -- k-step x half the taps, both tiles per wave: every wave then prepared the addresses of two tiles, handed three
  // quarters of two partial tiles to the other waves and summed three foreign slices per row, ~350 vector / LDS instructions per
  // step that are additive to the wave's 81 MFMAs.  Two ways: one tile's addresses, half a tile handed over, one foreign slice.)
  constexpr int KSTEP = WV & 1, TILE = WV >> 1;
  constexpr int TH = WV >> 1, RH = WV & 1;  // reduction role: rows 16*RH .. 16*RH+15 (accumulator registers 8*RH..8*RH+7) of tile TH
  static_assert(ZS_TILES == 2 && TH == TILE, "a wave reduces half of the tile it multiplies; its partner (the other k-step) the other half");
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  const int tid = threadIdx.x;  // 256 threads: staging and reduction roles
  const int b = blockIdx.y, ct = blockIdx.z;
  unsigned long long t_begin = 0, t_loop = 0;
  if (DBG) t_begin = z3_stamp();
  const Z3Geo G = z3_geo(a);
  const int H = a.HS, W = a.W, PV = G.PV, SPV = G.SPV;
  const int NIMG = G.rows * W;  // voxels staged per plane (strip rows + halo rows)
  const int nsteps = G.nsteps;
  const int chunk = G.chunk + G.strip * a.nchunk;
  float* const coef_lds = (float*)(lds + ZS_ZERO);
  float* const bias_lds = (float*)(lds + ZS_ZERO + 512);
  int* const flag_lds = (int*)(lds + ZS_ZERO + 640);

  // ---- weights of this wave's K-slice: registers for the whole chunk --------------------------------------------
  // (requested in the prologue AFTER the planes of step 0, which the prologue's converts wait for: the 216 KB of weights per
  // workgroup then stream in from L2 under those converts instead of in front of the plane loads)
  u32x4 w1[27], w2[27];

  // ---- staging role (thread = channel quad q = tid & 7 of image voxels p0 + 32k, p0 = tid >> 3, k < 5) -------------
  // Pieces beyond the image (p0 + 32k >= NIMG) repeat its last voxel: the same bytes written twice instead of a branch per piece.
  const int q = tid & 7, p0 = tid >> 3;
  const float* const src_b = a.in + (size_t)b * G.vox * a.ldc;  // wave-uniform: the loads take it as their scalar base
  const int src_off = (G.h0 - G.halo) * W;  // image voxel p comes from plane voxel p + src_off (mod PV: phi halo rows wrap)
  float gscale = 1.f, ginv = 1.f;
  if (SCALED) pow2_scale_for(*a.in_absmax, &gscale, &ginv);
  int srco[NSL];  // byte offset in a plane of the quad this thread's piece k is filled from
  int dsto[NSL];  // byte offset in a plane image of the record quad it fills
#pragma unroll
  for (int k = 0; k < NSL; ++k) {
    const int pi = min(p0 + 32 * k, NIMG - 1);
    int v = pi + src_off;
    v = v < 0 ? v + PV : (v >= PV ? v - PV : v);
    srco[k] = (v * a.ldc + q * 4) * 4;
    dsto[k] = (Z3_PAD ? pi + pi / W + 1 : pi) * ZS_VB + (q >> 2) * 64 + (q & 3) * 8 + G.RB;
  }
  f32x4 ld[NSL];
  static_assert(NSL == 5 || NSL == 7 || NSL == 8, "Z3_LANDED names five, seven or eight staging registers");
  // Plane loads are issued and awaited by hand, but -- unlike zs_helper_wave -- with nothing to count: they are the LAST
  // vector-memory operations of a step (after the reduction's row stores), so the wait one step later is a plain vmcnt(0); the
  // stores it also covers were issued a whole matrix phase before the loads and have long been acknowledged.
  constexpr int NYOUNG = 0;
  auto uniform_ptr = [](const void* p) {  // the pointer as a scalar register pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
  };
  auto issue = [&](int z) {
    const int zc = min(max(z, 0), a.D - 1);
    const float* src = uniform_ptr(src_b + (size_t)zc * PV * a.ldc);
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      asm volatile("global_load_dwordx4 %0, %1, %2 ; zs_plane_load" : "=v"(ld[k]) : "v"(srco[k]), "s"(src) : "memory");
  };
#define Z3_LANDED(younger)                                                                                   \
  do {                                                                                                       \
    asm volatile("s_waitcnt vmcnt(%8) ; zs_landed"                                                           \
                 : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[NSL > 5 ? 5 : 0]),  \
                   "+v"(ld[NSL > 5 ? 6 : 1]), "+v"(ld[NSL > 7 ? 7 : 2])                                       \
                 : "n"(younger)                                                                              \
                 : "memory");                                                                                \
  } while (0)
  // normalise + split the five pieces in v[] (this thread's quad of image voxels p0 + 32k) into the ring slot of plane z
  auto convert = [&](f32x4 (&v)[NSL], int z) {
    float amax = 0.f;
    const int zz = z + a.NR;  // ring slot (z + NR) mod NR, z >= -1, NR = 4 or 5, without a division
    const int slot = a.NR == 4 ? (zz & 3) : zz - 5 * ((zz * 205) >> 10);
    const bool zero = z < 0 || z >= a.D;
    const int sbase = slot * G.PLB;
    f32x4 cf[4], cn[4];  // {scale, shift, add, -} per channel; cn = {scale, shift} * -log2(e) for the sigmoid's exponent
    if (NORMED) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cf[e] = *z3_lds<const f32x4>(ZS_ZERO + (q * 4 + e) * 16);
        cn[e][0] = cf[e][0] * -1.4426950408889634f;
        cn[e][1] = cf[e][1] * -1.4426950408889634f;
      }
    }
    if (zero) {  // a plane outside the volume (wave-uniform: one branch, not one per piece)
#pragma unroll
      for (int k = 0; k < NSL; ++k) {
        int d = dsto[k] + sbase;
        asm volatile("" : "+v"(d));  // (one address register, the second write through the offset field)
        *z3_lds<u32x2>(d) = u32x2{0u, 0u};
        *z3_lds<u32x2>(d + 32) = u32x2{0u, 0u};
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
      u32x2 t1, t2;
      f32x4 x = v[k];
      if (SCALED) x = x * gscale;  // (training: input gradients rescaled by a power of two)
      if (NORMED) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // SiLU(t) + add, t = scale x + shift: t / (1 + 2^(-t log2 e)) + add on the transcendental unit -- fma, exp, add,
          // rcp, fma, fma
          const float t = cf[e][0] * x[e] + cf[e][1];
          const float ex = __builtin_amdgcn_exp2f(cn[e][0] * x[e] + cn[e][1]);
          x[e] = t * __builtin_amdgcn_rcpf(1.f + ex) + cf[e][2];
        }
      }
      amax = fmaxf(fmaxf(amax, fabsf(x[0])), fabsf(x[1]));
      amax = fmaxf(fmaxf(amax, fabsf(x[2])), fabsf(x[3]));
      split2(x, t1, t2);
      int d = dsto[k] + sbase;
      asm volatile("" : "+v"(d));  // (one address register, the second write through the offset field)
      *z3_lds<u32x2>(d) = t1;
      *z3_lds<u32x2>(d + 32) = t2;
    }
    if (amax > 65504.f) *flag_lds = 1;  // (rare; flushed to a.status at the end)
  };

  // ---- which step is the first to read which plane: scalar, incremental (no divisions in the loop) ----------------
  auto need = [&](int k) {  // highest plane that step k reads
    k = min(k, nsteps - 1);
    return min(G.v0 + k * ZS_STEP + ZS_STEP - 1, G.cend - 1) / SPV + 1;
  };
  struct { int k, vend, zlim, need; } trk;
  auto trk_init = [&](int k) {  // state of step k (k < nsteps)
    trk.k = k;
    trk.vend = min(G.v0 + k * ZS_STEP + ZS_STEP - 1, G.cend - 1);
    const int zv = trk.vend / SPV;
    trk.zlim = (zv + 1) * SPV;
    trk.need = zv + 1;
  };
  auto trk_next = [&]() {  // advance to the next step; returns the plane it is the first to read, or -2
    trk.k += 1;
    if (trk.k >= nsteps) return -2;
    trk.vend = min(trk.vend + ZS_STEP, G.cend - 1);
    if (trk.vend < trk.zlim) return -2;
    trk.zlim += SPV;  // (a plane holds at least one step: at most one new plane per step)
    trk.need += 1;
    return trk.need;
  };

  // ---- prologue: zero area, GroupNorm table, bias, the planes of step 0 (3 or 4) ------------------------------------
  if (tid < ZS_ZERO / 4) ((float*)lds)[tid] = 0.f;
  if (Z3_PAD)
    for (int i = tid * 16; i < a.NR * G.PLB; i += 256 * 16) *z3_lds<f32x4>(G.RB + i) = f32x4{0.f, 0.f, 0.f, 0.f};  // (the pad records)
  if (tid < 32) bias_lds[tid] = a.bias ? a.bias[ct * 32 + tid] : 0.f;
  if (tid == 32) *flag_lds = 0;
  const int zstaged0 = need(0);
  const int zp0 = G.zfirst - 1;
  // the plane step 1 is the first to read: its loads go out first (converted during step 0), then -- all at once, one memory
  // latency for the lot -- the three or four planes of step 0
  trk_init(0);
  int zpend = trk_next();
  if (zpend != -2) issue(zpend);
  {
    f32x4 ldp[4][NSL];
    auto fetch = [&](f32x4 (&dst)[NSL], int z) {
      const int zc = min(max(z, 0), a.D - 1);
      const char* src = (const char*)(src_b + (size_t)zc * PV * a.ldc);
#pragma unroll
      for (int k = 0; k < NSL; ++k) dst[k] = *(const f32x4*)(src + srco[k]);
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) fetch(ldp[i], zp0 + i);
    if (zp0 + 3 <= zstaged0) fetch(ldp[3], zp0 + 3);
    if (NORMED) {
      if (a.defer.part) {  // table of all defer.C channels built by the whole workgroup in the (still unused) exchange region
        char* scratch = lds + G.ZPART;
        gn_defer_to_lds(a.defer, b, (float*)scratch, scratch + a.defer.C * 16);
        if (tid < 32) *(f32x4*)(coef_lds + tid * 4) = *(const f32x4*)(scratch + (a.choff + tid) * 16);
      } else {
        if (tid < 32) *(f32x4*)(coef_lds + tid * 4) = *(const f32x4*)(a.coef + ((size_t)b * a.coef_c + tid) * 4);
      }
    }
    {
      const u32x4* wq = a.wpk + ((size_t)(KSTEP * 27) * a.CTtot + ct) * 128 + lane;
#pragma unroll
      for (int j = 0; j < 27; ++j) {
        w1[j] = wq[(size_t)j * a.CTtot * 128];
        w2[j] = wq[(size_t)j * a.CTtot * 128 + 64];
      }
    }
    __syncthreads();  // the table is complete, the ring zeroed
#pragma unroll
    for (int i = 0; i < 3; ++i) convert(ldp[i], zp0 + i);
    if (zp0 + 3 <= zstaged0) convert(ldp[3], zp0 + 3);
  }
  // the weights in accumulation registers (MFMA reads its A/B operands from either file): the 216 of them leave the vector
  // registers to the accumulators, whose hand-over then needs no v_accvgpr_read
#pragma unroll
  for (int j = 0; j < 27; ++j) asm volatile("" : "+a"(w1[j]), "+a"(w2[j]));

  // ---- matrix role: per-lane position of its row (voxel) in the next tile, advanced tile by tile: phi row gh and r column
  // gw in the strip-plane, po = byte offset of its record in a plane image (+ this wave's constants), sb = byte offset of the ring
  // slot of plane z - 1
  int gh, gw, po, sb;
  const int RWB = G.pitch * ZS_VB;  // bytes per image row (with its pad record)
  const int ring_bytes = a.NR * G.PLB;
  {
    const int v = G.v0 + TILE * 32 + col;
    const int gz = v / SPV;
    const int p = v - gz * SPV;
    gh = p / W;
    gw = p - gh * W;
    // (record of (row, column) = row * pitch + column + 1; the kw = 0 tap is one record back)
    po = ((gh + (HALO ? 1 : 0)) * G.pitch + gw) * ZS_VB + G.RB + KSTEP * 64 + half * 16 - (Z3_PAD ? 0 : ZS_VB);
    sb = ((gz + a.NR - 1) % a.NR) * G.PLB;
  }
  // one step on: 64 voxels (a strip-plane holds at least one step, so at most one plane boundary is crossed)
  const int adv_h = ZS_STEP / W, adv_w = ZS_STEP - adv_h * W;
  const int adv_po = (adv_h * G.pitch + adv_w) * ZS_VB, plane_po = H * RWB;
  auto advance_step = [&]() {
    gw += adv_w;
    gh += adv_h;
    po += adv_po;
    if (gw >= W) { gw -= W; gh += 1; po += (G.pitch - W) * ZS_VB; }
    if (gh >= H) {  // into the next plane
      gh -= H;
      po -= plane_po;
      sb += G.PLB;
      sb = sb == ring_bytes ? 0 : sb;
    }
  };
  // tap addresses of the wave's tile: fragment of tap (kz, kh, kw) = (row base of (kz, kh)) + kw * ZS_VB, the constant folded into
  // the ds_read's offset field.  Per (kz, kh): rb = the row base; aL / aR = the base the kw = 0 / kw = 2 tap reads from -- rb, or
  // for the lanes of the first / last column the zero area at rb's offset mod 256 (the lane keeps its bank quad): ~45 vector
  // instructions per step here instead of ~3.5 per fragment pair in the MFMA loop (where every instruction of the wave's own
  // stream costs ~4 cycles next to the MFMAs).
  int rb[1][3][3], aL[1][3][3], aR[1][3][3];
  const int W1 = W - 1;
  auto prepare = [&]() {
    {
      constexpr int t = 0;
      // edge lanes keep only the low byte of the address (= the zero area at the same offset mod 256): one v_and per base
      const int mL = gw == 0 ? 255 : -1, mR = gw == W1 ? 255 : -1;
      int ro0, ro2;
      if (HALO) {  // strips carry their phi neighbours as halo rows
        ro0 = -RWB;
        ro2 = RWB;
      } else {     // whole planes wrap around
        ro0 = gh > 0 ? -RWB : (H - 1) * RWB;
        ro2 = gh < H - 1 ? RWB : -(H - 1) * RWB;
      }
      int bz[3];
      bz[0] = sb + po;
#pragma unroll
      for (int kz = 1; kz < 3; ++kz) {
        const unsigned x = (unsigned)(sb + kz * G.PLB);
        bz[kz] = (int)min(x, x - (unsigned)ring_bytes) + po;  // slot wrap: x < ring ? x : x - ring
      }
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int r = bz[kz] + (kh == 0 ? ro0 : (kh == 2 ? ro2 : 0));
          rb[t][kz][kh] = r;
          if (Z3_PAD) {
            aL[t][kz][kh] = aR[t][kz][kh] = r;
          } else {
            aL[t][kz][kh] = r & mL;
            aR[t][kz][kh] = r & mR;
          }
        }
      advance_step();
    }
  };

  constexpr int NI = 27;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x2 own[4];  // this wave's own K-slice of the rows it reduces: stays in registers until the next step's reduction

  // ---- the 81 MFMAs of step s; the partner's half of the partial tile to the exchange buffer of parity s & 1 ---------------
  auto matrix = [&](int s) {
    constexpr int PD = Z3_PD;
    const int xch = G.XCH + (s & 1) * Z3_XCH;
    u32x4 fa[PD + 1][2];
    auto load_frag = [&](int tap) {
      const int kz = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int base = (kw == 0 ? aL[0][kz][kh] : (kw == 2 ? aR[0][kz][kh] : rb[0][kz][kh])) + kw * ZS_VB;
      fa[tap % (PD + 1)][0] = *z3_lds<const u32x4>(base);
      fa[tap % (PD + 1)][1] = *z3_lds<const u32x4>(base + 32);
    };
    f32x16 accA, accB;
#pragma unroll
    for (int i = 0; i < PD; ++i) load_frag(i);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i + PD < NI) load_frag(i + PD);
      __builtin_amdgcn_sched_barrier(0);
      accA = MFMA_F16(fa[i % (PD + 1)][0], w1[i], i == 0 ? zero16 : accA);
      accB = MFMA_F16(fa[i % (PD + 1)][0], w2[i], i == 0 ? zero16 : accB);
      accB = MFMA_F16(fa[i % (PD + 1)][1], w1[i], accB);
    }
    // K-slice partial of the tile: the rows this wave reduces stay in registers, the partner's go to its exchange region
    f32x2 pt[8];
    const f32x2 lo = {1.f / 2048.f, 1.f / 2048.f};
#pragma unroll
    for (int r = 0; r < 8; ++r) pt[r] = f32x2{accA[2 * r], accA[2 * r + 1]} + f32x2{accB[2 * r], accB[2 * r + 1]} * lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) own[r] = pt[4 * RH + r];
    constexpr int PH = 1 - RH;  // the partner's row half
    const int d = xch + (WV ^ 1) * 2048 + lane * 16;
    *z3_lds<f32x4>(d) = f32x4{pt[4 * PH][0], pt[4 * PH][1], pt[4 * PH + 1][0], pt[4 * PH + 1][1]};
    *z3_lds<f32x4>(d + 1024) = f32x4{pt[4 * PH + 2][0], pt[4 * PH + 2][1], pt[4 * PH + 3][0], pt[4 * PH + 3][1]};
  };

  // ---- reduction of step s: this wave's 16 rows of tile TH ------------------------------------------------------------
  f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
  float* const out_sb = a.out + ((size_t)b * G.vox + (size_t)G.h0 * W) * a.cout + ct * 32;  // wave-uniform
  // accumulator register r of this wave's half tile = row (r & 3) + 8 (r >> 2) + 4 half + 16 RH; byte offsets of its 8 rows from
  // the tile's first output row (whole planes: output row = strip voxel, so the tile base is a scalar)
  int rowo[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) rowo[r] = (((r & 3) + 8 * (r >> 2) + 4 * half + 16 * RH) * a.cout + col) * 4;
  // strips: a tile's rows are strip voxels vt + row = (plane rz, in-plane rp + row); called for s = 0, 1, 2, ... in order, so (rz, rp)
  // of the wave's tile advance by one step per call (scalar; a strip-plane holds at least one step)
  int rz = 0, rp = 0;
  if (HALO) {
    const int vt0 = G.v0 + TH * 32;
    rz = vt0 / SPV;
    rp = vt0 - rz * SPV;
  }
  int rrow8[8];  // row of accumulator register r in the tile
#pragma unroll
  for (int r = 0; r < 8; ++r) rrow8[r] = (r & 3) + 8 * (r >> 2) + 4 * half + 16 * RH;
  auto reduce_store = [&](int s) {
    const int xch = G.XCH + (s & 1) * Z3_XCH;
    const int vt = G.v0 + s * ZS_STEP + TH * 32;
    const bool full = vt + 32 <= G.cend;  // whole tile inside the chunk (all but the last step)
    // byte offsets of the 8 rows from a scalar base: whole planes -- output row = strip voxel, base = the tile's first row; strips --
    // base = the strip's first row of plane 0, a row's plane found from (rz, rp)
    int off[8];
    const float* tb;
    if (!HALO) {
      tb = uniform_ptr(out_sb + (size_t)vt * a.cout);
#pragma unroll
      for (int r = 0; r < 8; ++r) off[r] = rowo[r];
    } else {
      tb = uniform_ptr(out_sb);
      const int gbase = rz * PV + rp;  // (scalar) strip-relative output voxel of the tile's first row
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int pr = rp + rrow8[r];
        const int gv = gbase + rrow8[r] + (pr >= SPV ? PV - SPV : 0);  // rows past the plane's end continue in the next plane
        off[r] = (gv * a.cout + col) * 4;
      }
      rp += ZS_STEP;
      if (rp >= SPV) { rp -= SPV; rz += 1; }
    }
    f32x2 prev[4];
    if (ACC && full) {
      const char* tbs = (const char*)tb + a.acc_delta;
#pragma unroll
      for (int r = 0; r < 8; ++r) prev[r >> 1][r & 1] = *(const float*)(tbs + (unsigned)off[r]);
    }
    f32x2 sum[4];  // k-step 0's slice + k-step 1's (fp32 addition commutes: the same sum whichever wave reduces)
    {
      const int d = xch + WV * 2048 + lane * 16;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 x = *z3_lds<const f32x4>(d + g * 1024);
        sum[2 * g] = own[2 * g] + f32x2{x[0], x[1]};
        sum[2 * g + 1] = own[2 * g + 1] + f32x2{x[2], x[3]};
      }
    }
    const float bv1 = *z3_lds<const float>(ZS_ZERO + 512 + col * 4);
    const f32x2 bvv = {bv1, bv1};
    if (SCALED) {
      const f32x2 gi = {ginv, ginv};
#pragma unroll
      for (int r = 0; r < 4; ++r) sum[r] = sum[r] * gi;
    }
    if (full) {
      if (ACC) {  // continuation launch of a wider-K conv: add to what the previous launch stored
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = sum[r] + prev[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sum[r] = sum[r] + bvv;
        s1 = s1 + sum[r];
        s2 = s2 + sum[r] * sum[r];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
        asm volatile("global_store_dword %0, %1, %2" ::"v"(off[r]), "v"(sum[r >> 1][r & 1]), "s"(tb) : "memory");
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (vt + rrow8[r] < G.cend) {  // (chunk tail)
          float* dst = (float*)((char*)tb + (unsigned)off[r]);
          float v = sum[r >> 1][r & 1];
          if (ACC) v += *(const float*)((const char*)dst + a.acc_delta);  // continuation launch: add to what the previous launch stored
          v += bv1;
          *dst = v;
          s1[r & 1] += v;
          s2[r & 1] += v * v;
        }
      }
    }
  };

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), as a builtin so that the compiler knows it: the weights (and the plane of
                                       // step 1) have landed, no compiler-inserted vmcnt wait inside the loop
  zs_barrier_lds();  // P: the planes of step 0 are staged

  // one step; `first`: the pending plane's loads were awaited by the full drain above
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;  // (DBG) wait, convert, reduce, prepare, matrix, barrier, #converts
  auto step = [&](int s, bool first) {
    if (DBG) t0 = z3_stamp();
    if (zpend != -2) {
      if (!first) Z3_LANDED(NYOUNG);  // (first: the loads were awaited by the full drain before the loop)
      if (DBG) { t1 = z3_stamp(); st[0] += t1 - t0; t0 = t1; }
      convert(ld, zpend);  // read first by step s + 1
      if (DBG) { t1 = z3_stamp(); st[1] += t1 - t0; t0 = t1; st[6] += 1; }
    }
    if (s >= 1) reduce_store(s - 1);
    __builtin_amdgcn_sched_barrier(0);
    zpend = trk_next();  // the plane step s + 2 is the first to read
    if (zpend != -2) issue(zpend);
    __builtin_amdgcn_sched_barrier(0);
    if (DBG) { t1 = z3_stamp(); st[2] += t1 - t0; t0 = t1; }
    prepare();
    if (DBG) { t1 = z3_stamp(); st[3] += t1 - t0; t0 = t1; }
    matrix(s);
    if (DBG) { t1 = z3_stamp(); st[4] += t1 - t0; t0 = t1; }
    zs_barrier_lds();  // the partial tiles of step s are complete; the plane of step s + 1 is staged
    if (DBG) { t1 = z3_stamp(); st[5] += t1 - t0; }
  };
  if (DBG) t_loop = z3_stamp();
  step(0, true);
  for (int s = 1; s < nsteps; ++s) step(s, false);
  reduce_store(nsteps - 1);
  if (DBG && lane == 0) {
    unsigned long long* d = z3_stamp_buf + ((size_t)((blockIdx.y * gridDim.x + blockIdx.x) & 255) * 4 + WV) * 12;
    for (int i = 0; i < 7; ++i) d[i] = st[i];
    d[7] = nsteps;
    d[8] = t_loop - t_begin;
    d[9] = z3_stamp() - t_begin;
  }

  if (a.ch_part) {
    const float u1 = s1[0] + s1[1], u2 = s2[0] + s2[1];
    const float t1 = u1 + __shfl_xor(u1, 32, 64), t2 = u2 + __shfl_xor(u2, 32, 64);
    if (half == 0) {
      float* dst = a.ch_part + ((((size_t)b * gridDim.x + chunk) * 4 + WV) * a.cout + ct * 32 + col) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
  if (a.status && tid == 0 && *flag_lds) atomicOr(a.status, 1);  // (the loop ends with a barrier)
}

// NSL: staging pieces per thread and plane = plane images of up to 32 NSL voxels (5: 160 -- Dataset-2's whole planes; 7: 224 --
// Dataset-3's strips of 10 phi rows + 2 halo rows of 18 voxels instead of 5 + 2, HGCal's of 6 + 2 rows of 21 instead of 4 + 2:
// less halo restaged per output row; 8: 256 -- Dataset-3's level-1 planes of 25 x 9 voxels whole)
template <bool ACC, int MODE, int DBG = 0, int NSL = ZS_NSL>
__global__ void __launch_bounds__(256, 1) conv_zslide_sw_f16x2_kernel(ConvZsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char zs_lds[];
  switch (threadIdx.x >> 6) {
    case 0: z3_wave<0, ACC, MODE, DBG, NSL>(a, zs_lds); break;
    case 1: z3_wave<1, ACC, MODE, DBG, NSL>(a, zs_lds); break;
    case 2: z3_wave<2, ACC, MODE, DBG, NSL>(a, zs_lds); break;
    default: z3_wave<3, ACC, MODE, DBG, NSL>(a, zs_lds); break;
  }
}

// launch of one K-block: the specialisation for (continuation, normed input, strips, rescaled input)
void z3_launch(const ConvZsArgs& a, bool acc, dim3 grid, size_t lds, hipStream_t s) {
  const int mode = ((a.coef || a.defer.part) ? 1 : 0) | (a.HS < a.H ? 2 : 0) | (a.in_absmax ? 4 : 0);
  const int staged = (a.HS + (a.HS < a.H ? 2 : 0)) * a.W;  // voxels of a plane image
  const int nsl = staged > 7 * 32 ? 8 : (staged > ZS_NSL * 32 ? 7 : ZS_NSL);
#define Z3_CASE(ACCV, M, N)                                                                                                           \
  if (acc == ACCV && mode == M && nsl == N) {                                                                                         \
    static bool attr = false;                                                                                                         \
    if (!attr) {                                                                                                                      \
      CD_HIP(hipFuncSetAttribute((const void*)conv_zslide_sw_f16x2_kernel<ACCV, M, 0, N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024));                                                                                        \
      attr = true;                                                                                                                    \
    }                                                                                                                                 \
    hipLaunchKernelGGL((conv_zslide_sw_f16x2_kernel<ACCV, M, 0, N>), grid, dim3(256), lds, s, a);                                      \
    return;                                                                                                                           \
  }
  Z3_CASE(false, 0, 5) Z3_CASE(false, 1, 5) Z3_CASE(false, 2, 5) Z3_CASE(false, 3, 5) Z3_CASE(false, 4, 5) Z3_CASE(false, 6, 5)
  Z3_CASE(true, 0, 5) Z3_CASE(true, 2, 5) Z3_CASE(true, 4, 5) Z3_CASE(true, 6, 5)
  Z3_CASE(false, 0, 7) Z3_CASE(false, 1, 7) Z3_CASE(false, 2, 7) Z3_CASE(false, 3, 7) Z3_CASE(false, 4, 7) Z3_CASE(false, 6, 7)
  Z3_CASE(true, 0, 7) Z3_CASE(true, 2, 7) Z3_CASE(true, 4, 7) Z3_CASE(true, 6, 7)
  Z3_CASE(false, 0, 8) Z3_CASE(false, 1, 8) Z3_CASE(false, 2, 8) Z3_CASE(false, 3, 8) Z3_CASE(false, 4, 8) Z3_CASE(false, 6, 8)
  Z3_CASE(true, 0, 8) Z3_CASE(true, 2, 8) Z3_CASE(true, 4, 8) Z3_CASE(true, 6, 8)
#undef Z3_CASE
  CD_REQUIRE(false, "z-slide conv: no kernel instance for this combination of continuation / normalised / strip / rescaled input");
}

}  // namespace

// Eligible: 3x3x3 stride 1 on grids whose planes -- or phi strips of them (HS rows, HS | H, with one halo row either side) --
// hold 64..160 voxels and fit the LDS ring: Dataset-2's 16x9 planes whole, Dataset-3's 50x18 in 10 strips of 5 rows, HGCal's
// 12x21 in 3 strips of 4.  Returns false otherwise.
bool try_launch_conv_zslide(const float* in0, int c0, const float* in1, int c1, const void* wpk_f16x2, const float* bias, float* out,
                            int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu) {
  // (environment switches of this launcher are read once: it sits on the eager hot path, one call per K-block)
  static const bool no_zslide = getenv("CD_NO_ZSLIDE") != nullptr;
  static const int strip_env = getenv("CD_ZS_STRIP") ? atoi(getenv("CD_ZS_STRIP")) : 0;  // testing: force a strip height
  static const int dbg_env = getenv("CD_ZS_DBG") ? atoi(getenv("CD_ZS_DBG")) : 0;
  if (no_zslide) return false;
  if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sz == 1 && g.sh == 1 && g.sw == 1)) return false;
  if (cout % 32 || c0 % 32 || c1 % 32) return false;
  const int H = g.in.h, W = g.in.w;
  auto ring_for = [&](int hs) { return hs * W >= 2 * ZS_STEP ? 4 : 5; };
  // the one-wave-per-SIMD form is specialised at compile time; anything outside its instances (a normalised input without the
  // activation, a normalised AND rescaled input) takes the matrix-wave / helper-wave form, as does CD_ZS_V1=1 (A/B)
  static const bool v1_env = getenv("CD_ZS_V1") != nullptr;
  const bool normed_in = fu.coef || fu.defer.part;
  const bool v1 = v1_env || (normed_in && !fu.act) || (normed_in && fu.in_absmax) || (normed_in && (c0 + c1) > 32);
  auto lds_for = [&](int hs) {
    const int rows = hs + (hs < H ? 2 : 0);
    const size_t ring = (size_t)ring_for(hs) * (((size_t)rows * W * ZS_VB + 255) & ~(size_t)255);
    const size_t ring3 = Z3_PAD ? (size_t)ring_for(hs) * ((((size_t)rows * (W + 1) + 1) * ZS_VB + 255) & ~(size_t)255) : ring;
    return v1 ? (size_t)ZS_ZERO + ring + ZS_PART : (size_t)ZS_ZERO + Z3_COEF + ring3 + 2 * Z3_XCH;
  };
  // voxels of a plane image the staging threads cover: 5 pieces of 32; the one-wave-per-SIMD form also has a 7-piece instance for
  // strips (CD_ZS_NSL5=1: without it, A/B)
  static const bool nsl5_env = getenv("CD_ZS_NSL5") != nullptr;
  static const int nsl_env = getenv("CD_ZS_NSL") ? atoi(getenv("CD_ZS_NSL")) : 8;  // (A/B: cap the pieces at 5, 7 or 8)
  const int max_staged = (v1 || nsl5_env) ? ZS_NSL * 32 : std::min(8, std::max(5, nsl_env)) * 32;
  int HS = 0;
  for (int hs = H; hs >= 1; --hs) {  // the largest strip that fits: least halo restaging
    if (H % hs) continue;
    const int rows = hs + (hs < H ? 2 : 0);
    if (hs * W < ZS_STEP || rows * W > max_staged || lds_for(hs) > 160 * 1024) continue;
    HS = hs;
    break;
  }
  if (!HS) return false;
  if (strip_env) {
    const int hs = strip_env;
    const int rows = hs + (hs < H ? 2 : 0);
    if (hs >= 1 && H % hs == 0 && hs * W >= ZS_STEP && rows * W <= max_staged && lds_for(hs) <= 160 * 1024) HS = hs;
  }
  const int nstrip = H / HS;
  const int SPV = HS * W;
  const int64_t svox = (int64_t)g.in.d * SPV;  // voxels per strip
  if (svox < 2 * ZS_STEP) return false;
  const size_t lds = lds_for(HS);
  // what the kernels assume about this launch, checked where it is cheap: the GroupNorm fold of the prologue builds its table of
  // all defer.C channels plus its scratch (gn_defer_scratch_bytes) inside the exchange region, before the first exchange
  if (fu.defer.part) {
    CD_REQUIRE(fu.defer.C >= c0 + c1 && fu.defer.C % 4 == 0, "z-slide conv: the deferred GroupNorm must cover the input channels");
    CD_REQUIRE((size_t)fu.defer.C * 16 + (size_t)gn_defer_scratch_bytes(fu.defer.C) <= (size_t)(v1 ? ZS_PART : 2 * Z3_XCH),
               "z-slide conv: the GroupNorm fold's scratch does not fit the exchange region");
  }
  CD_REQUIRE(lds <= 160 * 1024 && (int64_t)batch * (H / HS) <= 65535 * 64, "z-slide conv: launch geometry out of range");
  const int CTtot = cout / 32;
  // chunks per strip: fill the 256 CUs (one workgroup each) with as few rounds and as little halo restaging as possible
  int best = 1;
  double best_eff = 0.0;
  const int max_chunks = (int)(svox / (2 * ZS_STEP));
  for (int n = 1; n <= max_chunks && n <= 64; ++n) {
    const int64_t cv = ((svox + n - 1) / n + ZS_STEP - 1) / ZS_STEP * ZS_STEP;
    const int nc = (int)((svox + cv - 1) / cv);
    if (nc != n) continue;
    const int64_t total = (int64_t)batch * nstrip * nc * CTtot;
    const int64_t rounds = (total + 255) / 256;
    const double planes = (double)cv / SPV;
    const double eff = (double)total / (rounds * 256.0) * planes / (planes + 2.5);  // halo planes + prologue
    if (eff > best_eff * 1.0001) { best_eff = eff; best = n; }
  }
  int nchunk = best;
#ifdef CD_ZS_EXPERIMENTS
  if (getenv("CD_ZS_NCHUNK")) nchunk = atoi(getenv("CD_ZS_NCHUNK"));  // (tools/zs_power.sh: the same workgroup program on fewer CUs)
#endif
  const int CV = (int)(((svox + nchunk - 1) / nchunk + ZS_STEP - 1) / ZS_STEP * ZS_STEP);
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
    a.defer = fu.defer;
    a.choff = ch;
    a.in_absmax = fu.in_absmax;
    a.act = fu.act;
    a.wpk = (const u32x4*)wpk_f16x2 + (size_t)(kb * 2) * 27 * CTtot * 128;
    a.CTtot = CTtot;
    a.bias = kb == 0 ? bias : nullptr;
    const bool add0 = kb == 0 && fu.add_src && !bias;  // out = conv + add_src: the first K-block runs as a continuation of add_src
    a.acc_delta = add0 ? (long long)((const char*)fu.add_src - (const char*)out) : 0;
    a.out = out;
    a.cout = cout;
    a.ch_part = kb == nblk - 1 ? fu.ch_part : nullptr;
    a.D = g.in.d; a.H = g.in.h; a.W = g.in.w; a.HS = HS; a.NR = ring_for(HS);
    a.nchunk = nchunk; a.CV = CV;
    a.status = fu.status;
    a.dbg = dbg_env;
    const dim3 grid((unsigned)(nstrip * nchunk), (unsigned)batch, (unsigned)CTtot);
#ifdef CD_ZS_EXPERIMENTS
    if (kb == 0 && a.dbg == 2048 && !v1) {  // stamps: print the per-wave cycle sums of one launch
      const bool normed = a.coef || a.defer.part;
      CD_REQUIRE(!a.in_absmax && a.HS == a.H, "CD_ZS_DBG=2048: whole planes, unscaled input");
      if (normed) {
        CD_HIP(hipFuncSetAttribute((const void*)conv_zslide_sw_f16x2_kernel<false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((conv_zslide_sw_f16x2_kernel<false, 1, 1>), grid, dim3(256), lds, s, a);
      } else {
        CD_HIP(hipFuncSetAttribute((const void*)conv_zslide_sw_f16x2_kernel<false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((conv_zslide_sw_f16x2_kernel<false, 0, 1>), grid, dim3(256), lds, s, a);
      }
      CD_HIP(hipGetLastError());
      static int nlaunch = 0;
      if (++nlaunch == 10) {
        CD_HIP(hipDeviceSynchronize());
        static unsigned long long h[256 * 4 * 12];
        CD_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(z3_stamp_buf), sizeof h));
        for (int wg : {0, 100, 255})
          for (int w = 0; w < 4; ++w) {
            const unsigned long long* d = h + ((size_t)wg * 4 + w) * 12;
            const double n = (double)d[7], nc = d[6] ? (double)d[6] : 1.;
            std::fprintf(stderr, "[z3 stamps] wg %3d wave %d: per step: load wait %5.0f (per convert %5.0f)  convert %5.0f (per convert %5.0f)  issue+reduce %5.0f  prepare %5.0f  matrix %5.0f  barrier %5.0f   (%llu converts in %llu steps)  prologue %llu  kernel %llu cycles\n",
                         wg, w, d[0] / n, d[0] / nc, d[1] / n, d[1] / nc, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6], d[7], d[8], d[9]);
          }
      }
      continue;
    }
#define ZS_DBG_CASE(D)                                                                                                     \
  case D:                                                                                                                  \
    CD_HIP(hipFuncSetAttribute((const void*)conv_zslide_f16x2_kernel<false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               160 * 1024));                                                                               \
    hipLaunchKernelGGL((conv_zslide_f16x2_kernel<false, D>), grid, dim3(512), lds, s, a);                                   \
    break;
    if (kb == 0 && a.dbg && v1) {
      switch (a.dbg) {
        ZS_DBG_CASE(2) ZS_DBG_CASE(4) ZS_DBG_CASE(6) ZS_DBG_CASE(22) ZS_DBG_CASE(38) ZS_DBG_CASE(70)
        default: CD_REQUIRE(false, "CD_ZS_DBG: not an instantiated experiment");
      }
      CD_HIP(hipGetLastError());
      continue;
    }
#endif
    if (v1) {
      if (kb == 0 && !add0) hipLaunchKernelGGL(conv_zslide_f16x2_kernel<false>, grid, dim3(512), lds, s, a);
      else hipLaunchKernelGGL(conv_zslide_f16x2_kernel<true>, grid, dim3(512), lds, s, a);
    } else {
      z3_launch(a, kb != 0 || add0, grid, lds, s);
    }
    CD_HIP(hipGetLastError());
  }
  if (fu.add_src && !bias && fu.add_done) *fu.add_done = 1;
  if (fu.units) *fu.units = nstrip * nchunk * 4;
  return true;
}

}  // namespace cd
