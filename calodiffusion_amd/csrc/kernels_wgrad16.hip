// Weight gradient of the stride-1 3x3x3 phi-periodic conv on the fp16 matrix pipe (f16x2 split, split16.h):
//     dW[co][ci][tap] = sum_{b, n} dy[b][n][co] * x[b][n + tap][ci]
// a contraction over VOXELS.  v_mfma_f32_32x32x16_f16 wants 8 consecutive K (= voxel) values of one channel per lane, the
// transpose of the channels-last tensors: both operands are read from [voxel][channel] LDS images with the gfx950
// transposing load ds_read_b64_tr_b16 (4 voxels x 16 channels per 16-lane group, delivered channel-major), so no transposed
// copy is ever built and the tap shift is a per-lane record address.
//
// A persistent workgroup (8 waves) loops over units = (sample, NZ consecutive z-planes):
//  * dy rows of the unit and the x planes they touch (+1 halo plane each side; phi halo rows; a zero record closing every r
//    row: every tap is "record + constant") are staged as f16x2 records.  dy is scaled by 2^s first, s from the tensor's
//    max |dy| (a one-word atomic-max pre-pass): gradients of O(1e-6) would sit in the fp16 subnormals otherwise;
//  * the 27 taps are dealt to the 8 waves (4,4,4,3,3,3,3,3: 7,7,7,6 per SIMD); a wave keeps its taps' 32x32 accumulator
//    pairs in registers across ALL its units; per 16-voxel K step it reads the dy fragments once (4 transposed loads) and
//    per tap the shifted x fragments (4 loads) for 3 MFMAs;
//  * one partial [27][32][32] per workgroup, summed in a fixed order by wgrad_reduce_kernel (deterministic).
#include "cd_common.h"
#include "split16.h"

#include <cstdio>
#include <cstdlib>

namespace cd {

typedef __fp16 fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __fp16 fh8 __attribute__((__vector_size__(8 * sizeof(__fp16))));

// max |x| as a bit pattern (non-negative floats order like their bit patterns).  Eight loads in flight per thread and ONE
// atomic per workgroup (one load per trip and one atomic per wave made this HBM-sized pass 50 us: 4096 same-address atomics).
__global__ void __launch_bounds__(256) absmax_bits_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ out) {
  __shared__ float sm[4];
  float m = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ((const f32x4*)x)[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = ((const f32x4*)x)[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}

namespace {

constexpr int WG_VB = 144;  // record: [k-step (16 channels)][term][16 fp16] + 16 B pad
constexpr int WG_NZ = 2;    // z-planes per unit

struct Wgrad16Args {
  const float* g;   // dy (B, vox, A), channel tile ta
  const float* x;   // (B, vox, xld) read at channel offset xoff, channel tile tb
  int A, xld, xoff;
  int D, H, W;
  int units_per_sample, total_units;
  float* partial;   // [gridDim.x][tilesA][tilesB][27][32][32]
  int tilesB;
  const unsigned* gmax_bits;  // max |dy| of the whole tensor (bit pattern)
  // x is read through a GroupNorm + SiLU + embedding: x_eff = silu(coef[0] x + coef[1]) + coef[2] per (sample, channel) -- the
  // second conv of a ResnetBlock saw its input that way in the forward pass (only ever formed in that conv's staging), and
  // recomputing it here saves the backward a gn_apply pass and a tensor per block
  const float* xcoef;         // [B][xld][4] or null
};

__device__ __forceinline__ fh8 cat8(fh4 a, fh4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

template <int NTAP>
__device__ __forceinline__ void k_step(const char* xL, const int (&xr)[2], const int (&toff)[4], int chan_off, fh8 G0, fh8 G1,
                                       f32x16 (&accA)[4], f32x16 (&accB)[4]) {
  fh8 X0[NTAP], X1[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    fh4 x0[2], x1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* xp = xL + xr[j] + toff[t] + chan_off;
      x0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp));
      x1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp + 32));
    }
    X0[t] = cat8(x0[0], x0[1]);
    X1[t] = cat8(x1[0], x1[1]);
  }
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    accA[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X0[t], accA[t], 0, 0, 0);
    accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X1[t], accB[t], 0, 0, 0);
    accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G1, X0[t], accB[t], 0, 0, 0);
  }
}

__global__ void __launch_bounds__(512, 1) wgrad_f16x2_kernel(Wgrad16Args a) {
  extern __shared__ __attribute__((aligned(16))) char wl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int ta = blockIdx.y / a.tilesB, tb = blockIdx.y % a.tilesB;
  const int H = a.H, W = a.W, PV = H * W, vox = a.D * PV;
  const int pitch = W + 1, prow = (H + 2) * pitch;
  const int R = WG_NZ * PV;                 // voxels per unit
  const int RP = (R + 15) & ~15;            // padded to whole 16-voxel K steps (extra dy records are zero)
  char* const gL = wl;                      // dy image: [RP][WG_VB]
  char* const xL = wl + (size_t)(RP + 1) * WG_VB;  // x image: one zero record (the r-1 neighbour of the first column of the
                                                   // first row), then [(NZ+2)][H+2][W+1] records
  const int nxrec = (WG_NZ + 2) * prow;

  // dy scale 2^s: bring the tensor's max |dy| to ~2^10 (exact power of two; undone when the partial is written)
  float gscale, ginv;
  pow2_scale_for(*a.gmax_bits, &gscale, &ginv);

  // zero both images once: pad records, out-of-range planes and tail rows are never written afterwards (or rewritten as 0)
  for (int i = tid; i < ((RP + 1) * WG_VB + nxrec * WG_VB) / 16; i += 512) ((u32x4*)wl)[i] = u32x4{0u, 0u, 0u, 0u};

  // this wave's taps: wave, wave + 8, wave + 16, wave + 24
  const int ntap = wave < 3 ? 4 : 3;
  int toff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tap = min(wave + 8 * t, 26);
    const int kz = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    toff[t] = ((kz - 1) * prow + (kh - 1) * pitch + (kw - 1)) * WG_VB;
  }
  f32x16 accA[4], accB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[t][r] = accB[t][r] = 0.f;

  // transposing-load roles: 16-lane group g = lane >> 4 reads voxel rows 8*(g>>1) + 4j + q (q = (lane&15)>>2), channel block
  // g & 1 (= k-step of the record), 8 bytes p = lane & 3 of that block's 32-byte term row
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int chan_off = (g4 & 1) * 64 + p4 * 8;
  const int vrow0 = 8 * (g4 >> 1) + q4;  // + 4j + 16 * chunk
  const int adv_h = 16 / W, adv_w = 16 - adv_h * W;
  // exact small-integer division by reciprocal: (v + 0.5) / d is never within float error of an integer for v < 2^20
  const float inv_pv = 1.f / (float)PV, inv_w = 1.f / (float)W;
  auto div_pv = [&](int v) { return (int)(((float)v + 0.5f) * inv_pv); };
  auto div_w = [&](int v) { return (int)(((float)v + 0.5f) * inv_w); };

  for (int u = blockIdx.x; u < a.total_units; u += gridDim.x) {
    const int n = u / a.units_per_sample, uz = u - n * a.units_per_sample;
    const int z0 = uz * WG_NZ;
    __syncthreads();  // previous unit fully consumed
    // ---- stage the unit: dy rows (scaled) and x planes z0-1 .. z0+NZ (zero outside the sample; interior rows + phi halo copies).
    // One item = a 16-byte channel quad of a voxel: items [0, R*8) are dy's, the rest x's.  Six items per thread and trip (eight spill: the accumulators of four taps
    // are live across the unit loop), their loads all issued before the first is converted (round 4: dy and x ran as separate loops of four -- five dependent memory round
    // trips per unit, ~10 us of a 14 us unit at level 0, with the matrix cores idle: one workgroup owns the CU's LDS).
    {
      const float* gs = a.g + ((size_t)n * vox + (size_t)z0 * PV) * a.A + ta * 32;
      const int nvalid = min(R, vox - z0 * PV);
      const float* xs = a.x + (size_t)n * vox * a.xld + a.xoff + tb * 32;
      const int ndy = R * 8, nst = (WG_NZ + 2) * PV * 8, nall = ndy + nst;
      f32x4 xc[4];  // (a thread stages the same channel quad of every voxel: item & 7 == tid & 7)
      if (a.xcoef) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xc[e] = *(const f32x4*)(a.xcoef + ((size_t)n * a.xld + a.xoff + tb * 32 + (tid & 7) * 4 + e) * 4);
      }
      for (int i0 = tid; i0 < nall; i0 += 6 * 512) {
        f32x4 val[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int it = i0 + k * 512;
          val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (it < ndy) {
            if ((it >> 3) < nvalid) val[k] = *(const f32x4*)(gs + (size_t)(it >> 3) * a.A + (it & 7) * 4);
          } else if (it < nall) {
            const int i = it - ndy, v = i >> 3;
            const int zl = div_pv(v), z = z0 - 1 + zl;
            if (z >= 0 && z < a.D) val[k] = *(const f32x4*)(xs + ((size_t)z * PV + (v - zl * PV)) * a.xld + (i & 7) * 4);
          }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int it = i0 + k * 512;
          if (it < ndy) {
            const int v = it >> 3, qd = it & 7;
            u32x2 t1, t2;
            split2(val[k] * gscale, t1, t2);
            char* d = gL + v * WG_VB + (qd >> 2) * 64 + (qd & 3) * 8;
            *(u32x2*)d = t1;
            *(u32x2*)(d + 32) = t2;
          } else if (it < nall) {
            const int i = it - ndy, v = i >> 3, qd = i & 7;
            const int zl = div_pv(v), p = v - zl * PV, h = div_w(p), w = p - h * W;
            const int z = z0 - 1 + zl;
            if (a.xcoef && z >= 0 && z < a.D) {
#pragma unroll
              for (int e = 0; e < 4; ++e) val[k][e] = cd_fast_silu(xc[e][0] * val[k][e] + xc[e][1]) + xc[e][2];
            }
            u32x2 t1, t2;
            split2(val[k], t1, t2);
            char* d = xL + ((zl * (H + 2) + (h + 1)) * pitch + w) * WG_VB + (qd >> 2) * 64 + (qd & 3) * 8;
            *(u32x2*)d = t1;
            *(u32x2*)(d + 32) = t2;
            if (h == 0) {
              char* d2 = d + H * pitch * WG_VB;
              *(u32x2*)d2 = t1;
              *(u32x2*)(d2 + 32) = t2;
            }
            if (h == H - 1) {
              char* d2 = d - H * pitch * WG_VB;
              *(u32x2*)d2 = t1;
              *(u32x2*)(d2 + 32) = t2;
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- K loop: 16 voxels per step -------------------------------------------------------------------------------
    // per-lane voxel of its two transposing loads (j = 0, 1): unit-local index and (plane, phi row, r column), advanced by 16
    // voxels per K step without divisions
    int vloc[2], vz[2], vh[2], vw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = vrow0 + 4 * j;
      vloc[j] = v;
      const int vv = min(v, R - 1);
      vz[j] = vv / PV;
      const int p = vv - vz[j] * PV;
      vh[j] = p / W;
      vw[j] = p - vh[j] * W;
    }
    for (int c = 0; c < RP / 16; ++c) {
      int xr[2];  // byte offset of the x record of (plane + 1, row + 1, column); past the unit: any valid record (dy is 0 there)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool in = vloc[j] < R;
        xr[j] = (((in ? vz[j] + 1 : 1) * (H + 2) + (in ? vh[j] + 1 : 1)) * pitch + (in ? vw[j] : 0)) * WG_VB;
      }
      // dy fragments (A operand: M = co): term 0 / term 1
      fh4 g0[2], g1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* gp = gL + (size_t)min(vloc[j], RP - 1) * WG_VB + chan_off;
        g0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp));
        g1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp + 32));
      }
      const fh8 G0 = cat8(g0[0], g0[1]), G1 = cat8(g1[0], g1[1]);
      // all of the step's x fragments are requested before its first MFMA (one LDS round trip per step, not per tap)
      if (ntap == 4) k_step<4>(xL, xr, toff, chan_off, G0, G1, accA, accB);
      else k_step<3>(xL, xr, toff, chan_off, G0, G1, accA, accB);
      // advance both voxels by 16
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        vloc[j] += 16;
        vw[j] += adv_w;
        vh[j] += adv_h;
        if (vw[j] >= W) { vw[j] -= W; vh[j] += 1; }
        while (vh[j] >= H) { vh[j] -= H; vz[j] += 1; }
      }
    }
  }

  float* pbase = a.partial + (((size_t)blockIdx.x * (a.A / 32) + ta) * a.tilesB + tb) * (size_t)27 * 1024;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntap) {
      const int tap = wave + 8 * t;
      float* pp = pbase + (size_t)tap * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = (accA[t][r] + accB[t][r] * (1.f / 2048.f)) * ginv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same contraction, sliding along z (round 4, second session).  wgrad_f16x2_kernel above restages four x planes for every two
// planes of dy (each x plane is fetched and split twice) and its staging and matrix phases alternate: a unit's loads are requested,
// awaited, converted, and only then do the MFMAs start -- 58 us per level-0 launch for 17 us of MFMAs.  Here a workgroup owns a
// CHUNK of consecutive units (NZ z-planes each) of one sample:
//  * the x planes live in a RING of S = 2 NZ + 2 plane slots (plane p in slot (p + 1) mod S): the unit in flight reads NZ + 2 of
//    them, the NZ new planes of the next unit go into the others -- every x plane is fetched and split once per chunk (+ two halo
//    planes per chunk), and dy alternates between two images;
//  * the next unit's global loads are issued BEFORE the K loop of the current one and converted after it: the memory round trip
//    runs under the MFMAs, one barrier per unit;
//  * taps -> waves so that a wave's first three taps are kz = 0, 1, 2 of ONE (kh, kw) (wave = 3 kh + kw < 8; the ninth (kh, kw)
//    goes to waves 0..2 as a fourth tap, kz = wave): the ring wrap of a tap's plane is then a per-lane base chosen by a
//    compile-time index (three bases per lane and K step), not a per-tap computation.
struct WgradRingArgs {
  const float* g;
  const float* x;
  int A, xld, xoff;
  int D, H, W;
  int NZ, S;                       // planes per unit, ring slots
  int U, upc, cps, total_chunks;   // units per sample, units per chunk, chunks per sample
  float* partial;                  // [gridDim.x][tilesA][tilesB][27][32][32]
  int tilesB;
  const unsigned* gmax_bits;
  const float* xcoef;              // as Wgrad16Args
  int abl;                         // experiment builds (-DCD_WGRAD_ABL): phases switched off, tools/wgrad_bench.py
};

template <bool ONE>
__global__ void __launch_bounds__(512, 1) wgrad_ring_f16x2_kernel(WgradRingArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int ta = blockIdx.y / a.tilesB, tb = blockIdx.y % a.tilesB;
  const int H = a.H, W = a.W, PV = H * W, vox = a.D * PV, NZ = a.NZ, S = a.S;
  // One-plane units (ONE): r rows WITHOUT the closing zero record -- a voxel's record is its index in the plane, so the four rows of
  // a transposing load are always 4 records apart (banks 0 / 16 / 32 / 48: conflict-free; SQ_LDS_BANK_CONFLICT 133 k -> 41 k cycles
  // per launch) -- and ONE zero record closing every plane slot: the r - 1 / r + 1 taps of the edge columns are redirected to it by
  // giving an edge lane a base whose tap offset lands on that record in whichever slot the tap's plane sits (one compare and one
  // select per lane and K step; selecting per tap cost more scalar and vector instructions than the conflicts it removed).
  const int pitch = ONE ? W : W + 1, prow = (H + 2) * pitch + (ONE ? 1 : 0);
  const int R = NZ * PV, RP = (R + 15) & ~15;
  char* const gL = wl;                                     // two dy images [2][RP][WG_VB]
  char* const xB = wl + (size_t)(2 * RP + 1) * WG_VB;      // record (slot 0, row 0, column 0); the record before it stays zero
  const int ringbytes = S * prow * WG_VB;
  float* const ctab = (float*)(xB + ringbytes);            // [32][4] GroupNorm coefficients of the chunk's sample (xcoef)

  float gscale, ginv;
  pow2_scale_for(*a.gmax_bits, &gscale, &ginv);
  for (int i = tid; i < ((2 * RP + 1) * WG_VB + ringbytes) / 16; i += 512) ((u32x4*)wl)[i] = u32x4{0u, 0u, 0u, 0u};

  const int ntap = wave < 3 ? 4 : 3;
  const int kz3 = wave < 3 ? wave : 0;
  int toff[4];
  {
    const int kh = wave / 3, kw = wave - 3 * kh;
#pragma unroll
    for (int t = 0; t < 3; ++t) toff[t] = (t * prow + kh * pitch + kw - 1) * WG_VB;
    toff[3] = (kz3 * prow + 2 * pitch + 1) * WG_VB;
  }
  f32x16 accA[4], accB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[t][r] = accB[t][r] = 0.f;

  // Transposing-load roles, laid out for the LDS banks: a ds_read_b64_tr_b16 is served in two groups of 32 lanes (16-lane groups
  // g4 = 0, 1 and 2, 3: both channel blocks of the same four voxel rows), each conflict-free when its 32 eight-byte pieces cover
  // the 64 banks once.  Records of 36 dwords: four CONSECUTIVE voxels start at banks 0, 36, 8, 44 and the two channel blocks of
  // wgrad_f16x2_kernel's record ([block][term]: 16 dwords apart) wrap onto each other -- every fragment read took twice its
  // LDS cycles, 560 of them per K step and CU beside 672 cycles of MFMAs per SIMD.  Here the record is [term][block] (a voxel's
  // two blocks are one 64-byte run) and the four rows of a load are voxels 4 q apart (K slot (half, j, q) <-> voxel
  // 4 q + 2 half + j of the step): 4 x 36 dwords = 16 mod 64, so the rows start at banks 0, 16, 32, 48 -- wherever a voxel's
  // record is its index: the dy image always, the x image in the one-plane form (see `pitch` above; with a closing record per r
  // row the next row is shifted by 36 dwords and most loads overlap banks again).  Measured: conflicts 133 k -> 41 k cycles per
  // launch, the launch time unchanged (profiles/r04_wgrad_pmc_summary.txt) -- the LDS array does not bound this K loop.
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int chan_off = (g4 & 1) * 32 + p4 * 8;
  const int vrow0 = 4 * q4 + 2 * (g4 >> 1);
  const int adv_h = 16 / W, adv_w = 16 - adv_h * W;
  const float inv_pv = 1.f / (float)PV, inv_w = 1.f / (float)W;
  auto div_pv = [&](int v) { return (int)(((float)v + 0.5f) * inv_pv); };
  auto div_w = [&](int v) { return (int)(((float)v + 0.5f) * inv_w); };

  // staging roles: a thread moves the channel quad qd of voxels vb, vb + 64, vb + 128 (R <= 192) of a unit
  const int qd = tid & 7, vb_ = tid >> 3;
  const int qoff = qd * 8;  // [term][block][16 fp16]: quad qd of term 0 at byte 8 qd, term 1 64 bytes on
  const float* const gcol = a.g + ta * 32 + qd * 4;
  const float* const xcol = a.x + a.xoff + tb * 32 + qd * 4;

  // The loads are UNCONDITIONAL (clamped addresses; validity is applied when the value is converted): a load under a divergent
  // branch makes the compiler wait for it at the join, which serialises the round trips the prefetch exists to overlap.
  // (the voxel index is laundered through an empty asm in each stage: hoisted out of the unit loop, the stages' index arithmetic
  // occupies ~30 registers across the K loop, and the reloads of what then spills sit between the prefetch and the MFMAs)
  // An explicit vmcnt(0) stands before every convert: a value first used under a divergent branch (`v < R`) is waited for inside
  // that branch only, the compiler's wait model then carries the load as still pending into the next K loop and drains the
  // PREFETCH in front of its first LDS read.
  constexpr int WAIT_VM0 = 0x0F70;  // vmcnt(0), expcnt / lgkmcnt untouched
  auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto ld_g = [&](int n, int z0, f32x4 (&r)[3]) {
    const int vb = fresh(vb_);
    const float* gs = gcol + ((size_t)n * vox + (size_t)z0 * PV) * a.A;
    const int last = min(R, vox - z0 * PV) - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = *(const f32x4*)(gs + (size_t)min(vb + 64 * k, last) * a.A);
  };
  auto st_g = [&](char* gbuf, int z0, const f32x4 (&r)[3]) {
    const int vb = fresh(vb_);
    const int nvalid = min(R, vox - z0 * PV);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int v = vb + 64 * k;
      if (v < R) {
        u32x2 t1, t2;
        split2(v < nvalid ? r[k] * gscale : f32x4{0.f, 0.f, 0.f, 0.f}, t1, t2);
        char* d = gbuf + v * WG_VB + qoff;
        *(u32x2*)d = t1;
        *(u32x2*)(d + 64) = t2;
      }
    }
  };
  // np <= NZ planes p0 .. p0 + np - 1 (zero outside the sample)
  auto ld_x = [&](int n, int p0, int np, f32x4 (&r)[3]) {
    const int vb = fresh(vb_);
    const float* xs = xcol + (size_t)n * vox * a.xld;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int v = min(vb + 64 * k, np * PV - 1);
      const int pl = div_pv(v), p = min(max(p0 + pl, 0), a.D - 1);
      r[k] = *(const f32x4*)(xs + ((size_t)p * PV + (v - pl * PV)) * a.xld);
    }
  };
  auto st_x = [&](int p0, int np, f32x4 (&r)[3]) {
    const int vb = fresh(vb_);
    const int slot0 = (p0 + 1) % S;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int v = vb + 64 * k;
      if (v < np * PV) {
        const int pl = div_pv(v), pv = v - pl * PV, h = div_w(pv), w = pv - h * W, p = p0 + pl;
        const bool inside = p >= 0 && p < a.D;
        f32x4 val = r[k];
        if (a.xcoef) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 c = *(const f32x4*)(ctab + ((qd * 4 + e) * 4));
            val[e] = cd_fast_silu(c[0] * val[e] + c[1]) + c[2];
          }
        }
        u32x2 t1, t2;
        split2(inside ? val : f32x4{0.f, 0.f, 0.f, 0.f}, t1, t2);
        int sl = slot0 + pl;
        if (sl >= S) sl -= S;
        char* d = xB + (sl * prow + (h + 1) * pitch + w) * WG_VB + qoff;
        *(u32x2*)d = t1;
        *(u32x2*)(d + 64) = t2;
        if (h == 0) {
          char* d2 = d + H * pitch * WG_VB;
          *(u32x2*)d2 = t1;
          *(u32x2*)(d2 + 64) = t2;
        }
        if (h == H - 1) {
          char* d2 = d - H * pitch * WG_VB;
          *(u32x2*)d2 = t1;
          *(u32x2*)(d2 + 64) = t2;
        }
      }
    }
  };

  for (int ch = blockIdx.x; ch < a.total_chunks; ch += gridDim.x) {
    const int n = ch / a.cps, cc = ch - n * a.cps;
    const int u0 = cc * a.upc, u1 = min(u0 + a.upc, a.U);
    const int zc = u0 * NZ;  // first plane of the chunk
    // (every wave left the previous chunk's last K loop at that unit's closing barrier: images and table are free)
    if (a.xcoef && tid < 32) *(f32x4*)(ctab + tid * 4) = *(const f32x4*)(a.xcoef + ((size_t)n * a.xld + a.xoff + tb * 32 + tid) * 4);
    __syncthreads();  // the coefficient table
    {
      // prologue: halo planes zc - 1, zc, then the first unit's own stage (planes zc + 1 .. zc + NZ, dy of unit u0) -- two rounds of
      // loads (all four sets in flight at once spill, and the reloads serialise the round trips)
      f32x4 ra[3], rb[3];
      const int npa = min(NZ, 2);
      ld_x(n, zc - 1, npa, ra);
      if (NZ == 1) ld_x(n, zc, 1, rb);
      __builtin_amdgcn_s_waitcnt(WAIT_VM0);
      st_x(zc - 1, npa, ra);
      if (NZ == 1) st_x(zc, 1, rb);
    }
    {
      f32x4 rc[3], rg[3];
      ld_x(n, zc + 1, NZ, rc);
      ld_g(n, zc, rg);
      __builtin_amdgcn_s_waitcnt(WAIT_VM0);
      st_x(zc + 1, NZ, rc);
      st_g(gL, zc, rg);
    }
    __syncthreads();

    for (int u = u0; u < u1; ++u) {
      const int zk = u * NZ;
#ifdef CD_WGRAD_ABL
      const bool more = u + 1 < u1 && !(a.abl & 4);
#else
      const bool more = u + 1 < u1;
#endif
      f32x4 rx[3], rg[3];
      if (more) {
        ld_x(n, zk + NZ + 1, NZ, rx);
        ld_g(n, zk + NZ, rg);
      }
      // ---- K loop of unit u: dy image (u - u0) & 1, x planes zk - 1 .. zk + NZ ----
      const char* gbuf = gL + ((u - u0) & 1) * RP * WG_VB;
      const int sb = zk % S;  // slot of plane zk - 1
#ifdef CD_WGRAD_ABL
      if (a.abl & 2) { __syncthreads(); continue; }
#endif
      // (Tried and measured, same box: ONE wave per SIMD -- four waves of seven tap slots, slot s + 1's fragments requested before slot
      // s's MFMAs, the next K step's under the last slot's, 452 registers, no spills: 55-62 us per level-0 launch against 46-49 here,
      // 35-37 against 29-30 at level 1: with one wave nothing covers the staging's and the prologue's memory round trips.)
      // (Tried: the second wave of each SIMD entering the K loop 3-10 x 64 cycles late, so that one wave's fragment reads run under
      // the other's MFMAs instead of both reading, then both multiplying: +0.6 ... +1.4 us per launch, the delay itself.  The sum
      // of the two phases is not a matter of phase: LDS reads issue slowly beside another wave's MFMA stream.)
      if constexpr (ONE) {
        // One plane per unit (level 0): every lane's voxel is in plane zk, so a tap's ring slot is a per-unit SCALAR and a lane only
        // advances a record offset.  ~20 vector instructions per K step instead of ~100: beside the other wave's MFMA stream a
        // wave issues one vector instruction per 10-17 cycles, and the general form's index arithmetic, not its 21 MFMAs
        // per SIMD (672 cycles), set the K step's ~1800 cycles.
        int tu[4];
        {
          int sk[3];
#pragma unroll
          for (int kz = 0; kz < 3; ++kz) {
            int sl = sb + kz;
            if (sl >= S) sl -= S;
            sk[kz] = sl * prow * WG_VB;
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) tu[t] = toff[t] - t * prow * WG_VB + sk[t];
          tu[3] = toff[3] - kz3 * prow * WG_VB + (kz3 == 0 ? sk[0] : (kz3 == 1 ? sk[1] : sk[2]));
        }
        int vloc[2], vw[2], roff[2], goff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int v = vrow0 + j;
          vloc[j] = v;
          const int vv = min(v, R - 1);
          const int h = div_w(vv);
          vw[j] = vv - h * W;
          roff[j] = v < R ? (h * pitch + vw[j]) * WG_VB + chan_off : chan_off;
          goff[j] = v * WG_VB + chan_off;
        }
        const int step_b = (adv_h * pitch + adv_w) * WG_VB;
        // edge lanes: base + (the tap's offset) = the zero record closing the tap's plane slot
        const int khA = wave / 3, kwA = wave - 3 * khA;     // (kh, kw) of this wave's taps 0..2 (tap 3, waves 0..2: (2, 2))
        const int ecA = kwA == 0 ? 0 : (kwA == 2 ? W - 1 : -1);
        const int zbA = ((H + 2) * pitch - (khA * pitch + kwA - 1)) * WG_VB + chan_off;
        const int zb3 = ((H + 2) * pitch - (2 * pitch + 1)) * WG_VB + chan_off;
        for (int c = 0; c < RP / 16; ++c) {
          fh4 g0[2], g1[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const char* gp = gbuf + goff[j];
            g0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp));
            g1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp + 64));
          }
          const fh8 G0 = cat8(g0[0], g0[1]), G1 = cat8(g1[0], g1[1]);
          int rA[2], r3[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            rA[j] = vw[j] == ecA ? zbA : roff[j];
            r3[j] = vw[j] == W - 1 ? zb3 : roff[j];
          }
          fh8 X0[4], X1[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t < 3 || ntap == 4) {
              fh4 x0[2], x1[2];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const char* xp = xB + (t < 3 ? rA[j] : r3[j]) + tu[t];
                x0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp));
                x1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp + 64));
              }
              X0[t] = cat8(x0[0], x0[1]);
              X1[t] = cat8(x1[0], x1[1]);
            }
          }
#if defined(CD_WGRAD_ABL) && CD_WGRAD_ABL == 1  // no MFMAs (the fragment reads stay)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < 3 || ntap == 4) {
              accA[t][0] += (float)X0[t][0] + (float)G0[0];
              accB[t][0] += (float)X1[t][0] + (float)G1[0];
            }
#else
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t < 3 || ntap == 4) {
              accA[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X0[t], accA[t], 0, 0, 0);
              accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X1[t], accB[t], 0, 0, 0);
              accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G1, X0[t], accB[t], 0, 0, 0);
            }
          }
#endif
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            vloc[j] += 16;
            goff[j] += 16 * WG_VB;
            vw[j] += adv_w;
            const bool carry = vw[j] >= W;
            vw[j] -= carry ? W : 0;
            roff[j] += step_b + (carry ? (pitch - W) * WG_VB : 0);
            if (vloc[j] >= R) roff[j] = chan_off;  // past the unit (dy is zero there): any valid record
          }
        }
      } else {
      int vloc[2], vz[2], vh[2], vw[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int v = vrow0 + j;
        vloc[j] = v;
        const int vv = min(v, R - 1);
        vz[j] = div_pv(vv);
        const int p = vv - vz[j] * PV;
        vh[j] = div_w(p);
        vw[j] = p - vh[j] * W;
      }
      for (int c = 0; c < RP / 16; ++c) {
        int lb[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool in = vloc[j] < R;  // past the unit: any valid record (dy is zero there)
          int mz = sb + (in ? vz[j] : 0);
          if (mz >= S) mz -= S;
          const int base = (mz * prow + (in ? vh[j] : 0) * pitch + (in ? vw[j] : 0)) * WG_VB + chan_off;
          lb[j][0] = base;
          lb[j][1] = base - (mz + 1 >= S ? ringbytes : 0);
          lb[j][2] = base - (mz + 2 >= S ? ringbytes : 0);
        }
        fh4 g0[2], g1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* gp = gbuf + (size_t)min(vloc[j], RP - 1) * WG_VB + chan_off;
          g0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp));
          g1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp + 64));
        }
        const fh8 G0 = cat8(g0[0], g0[1]), G1 = cat8(g1[0], g1[1]);
        // two taps' fragments at a time (all four at once, as wgrad_f16x2_kernel does, spill next to the prefetched unit)
#pragma unroll
        for (int tp = 0; tp < 4; tp += 2) {
          fh8 X0[2], X1[2];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int t = tp + tt;
            if (t < 3 || ntap == 4) {
              fh4 x0[2], x1[2];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int l = t < 3 ? lb[j][t] : (kz3 == 0 ? lb[j][0] : (kz3 == 1 ? lb[j][1] : lb[j][2]));
                const char* xp = xB + l + toff[t];
                x0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp));
                x1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp + 64));
              }
              X0[tt] = cat8(x0[0], x0[1]);
              X1[tt] = cat8(x1[0], x1[1]);
            }
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int t = tp + tt;
            if (t < 3 || ntap == 4) {
              accA[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X0[tt], accA[t], 0, 0, 0);
              accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X1[tt], accB[t], 0, 0, 0);
              accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G1, X0[tt], accB[t], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          vloc[j] += 16;
          vw[j] += adv_w;
          vh[j] += adv_h;
          if (vw[j] >= W) { vw[j] -= W; vh[j] += 1; }
          while (vh[j] >= H) { vh[j] -= H; vz[j] += 1; }
        }
      }
      }
      if (more) {
        __builtin_amdgcn_s_waitcnt(WAIT_VM0);
        st_x(zk + NZ + 1, NZ, rx);
        st_g(gL + ((u + 1 - u0) & 1) * RP * WG_VB, zk + NZ, rg);
      }
      __syncthreads();
    }
  }

  float* pbase = a.partial + (((size_t)blockIdx.x * (a.A / 32) + ta) * a.tilesB + tb) * (size_t)27 * 1024;
#ifdef CD_WGRAD_ABL
  if ((a.abl & 8) && accA[0][0] != 12345.f) return;
#endif
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntap) {
      const int tap = t < 3 ? 9 * t + wave : 9 * wave + 8;
      float* pp = pbase + (size_t)tap * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = (accA[t][r] + accB[t][r] * (1.f / 2048.f)) * ginv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same contraction for the STRIDED (KD, 4, 4) convs between the U-Net's levels (Downsample, models.py:360-365; and, with the two
// tensors' roles swapped, the transposed Upsample conv): dw[tap][a][b] = sum over coarse voxels o of g[o][a] * x[in(o, tap)][b],
// in(o, tap) = (oz SZ + kz - 1, 2 oh + kh - 1 (phi-periodic), 2 ow + kw - 1).  Round 4: these four weight gradients ran on the
// f32-input MFMA (wgrad_kernel<4>, K = 2 voxels per instruction, 85 us each = 4 % of the training step).  Here a unit is NZ coarse
// z-planes: their g rows and the fine planes they touch are staged as f16x2 record images -- the fine one with its phi halo rows
// (one above, two below) and two zero records closing every r row, so that every tap of every coarse voxel is "record + constant"
// -- and a K step of 16 coarse voxels takes its operands through the transposing LDS load exactly as wgrad_f16x2_kernel does,
// each lane addressing the fine record of ITS coarse voxel.  The KD x 16 taps are dealt to the 8 waves of TWO workgroups
// (blockIdx.z: 3 or 4 taps per wave, the accumulators of more would not fit a wave's registers).  Both operands are rescaled by a
// power of two from their maxima: whichever of them is the gradient (g for the down conv, x for the up conv) is far below the fp16
// normal range.
struct WgradS16Args {
  const float* g;   // coarse tensor (B, voxo, A), channel tile ta
  const float* x;   // fine tensor (B, voxi, xld) read at channel offset xoff, channel tile tb
  int A, xld, xoff;
  int Do, Ho, Wo, Di, Hi, Wi, KD, SZ, NZ;
  int units_per_sample, total_units;
  float* partial;   // [gridDim.x][tilesA][tilesB][KD * 16][32][32]
  int tilesB;
  const unsigned *gmax_bits, *xmax_bits;
};

template <int NTAP>
__global__ void __launch_bounds__(512, 1) wgrad_strided_f16x2_kernel(WgradS16Args a) {
  extern __shared__ __attribute__((aligned(16))) char wl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int ta = blockIdx.y / a.tilesB, tb = blockIdx.y % a.tilesB;
  const int T = a.KD * 16;
  const int Wo = a.Wo, PVo = a.Ho * a.Wo, voxo = a.Do * PVo;
  const int Hi = a.Hi, Wi = a.Wi, PVi = Hi * Wi, voxi = a.Di * PVi;
  const int pitch = Wi + 2, prow = (Hi + 3) * pitch;   // fine image: rows -1 .. Hi+1, two zero records after every row
  const int NZI = (a.NZ - 1) * a.SZ + a.KD;           // fine planes a unit touches
  const int R = a.NZ * PVo, RP = (R + 15) & ~15;
  char* const gL = wl;                                  // coarse image [RP][WG_VB]
  char* const xL = wl + (size_t)(RP + 1) * WG_VB;       // one zero lead record, then [NZI][Hi + 3][Wi + 2] records
  const int nxrec = NZI * prow;
  float gscale, ginv, xscale, xinv;
  pow2_scale_for(*a.gmax_bits, &gscale, &ginv);
  pow2_scale_for(*a.xmax_bits, &xscale, &xinv);
  for (int i = tid; i < ((RP + 1) * WG_VB + nxrec * WG_VB) / 16; i += 512) ((u32x4*)wl)[i] = u32x4{0u, 0u, 0u, 0u};

  // this wave's taps: tap = blockIdx.z * (T / 2) + wave + 8 t, t < NTAP
  int toff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tap = blockIdx.z * (T / 2) + wave + 8 * t;
    const int kz = tap >> 4, kh = (tap >> 2) & 3, kw = tap & 3;
    toff[t] = (kz * prow + kh * pitch + kw) * WG_VB;
  }
  f32x16 accA[NTAP], accB[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[t][r] = accB[t][r] = 0.f;
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int chan_off = (g4 & 1) * 64 + p4 * 8;
  const int vrow0 = 8 * (g4 >> 1) + q4;
  const int adv_h = 16 / Wo, adv_w = 16 - adv_h * Wo;

  for (int u = blockIdx.x; u < a.total_units; u += gridDim.x) {
    const int n = u / a.units_per_sample, uz = u - n * a.units_per_sample;
    const int oz0 = uz * a.NZ;
    const int zin0 = oz0 * a.SZ - 1;  // first fine plane of the image
    __syncthreads();
    {  // ---- coarse rows (rescaled) ----
      const float* gs = a.g + ((size_t)n * voxo + (size_t)oz0 * PVo) * a.A + ta * 32;
      const int nvalid = min(R, voxo - oz0 * PVo);
      for (int i = tid; i < RP * 8; i += 512) {
        const int v = i >> 3, qd = i & 7;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (v < nvalid) val = *(const f32x4*)(gs + (size_t)v * a.A + qd * 4);
        u32x2 t1, t2;
        split2(val * gscale, t1, t2);
        char* d = gL + v * WG_VB + (qd >> 2) * 64 + (qd & 3) * 8;
        *(u32x2*)d = t1;
        *(u32x2*)(d + 32) = t2;
      }
      // ---- fine planes zin0 .. zin0 + NZI - 1 (zero outside the sample): interior rows + the phi halo copies ----
      const float* xs = a.x + (size_t)n * voxi * a.xld + a.xoff + tb * 32;
      const int nst = NZI * PVi * 8;
      for (int i = tid; i < nst; i += 512) {
        const int v = i >> 3, qd = i & 7;
        const int zl = v / PVi, p = v - zl * PVi, h = p / Wi, w = p - h * Wi;
        const int z = zin0 + zl;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (z >= 0 && z < a.Di) val = *(const f32x4*)(xs + ((size_t)z * PVi + p) * a.xld + qd * 4);
        u32x2 t1, t2;
        split2(val * xscale, t1, t2);
        char* d = xL + (1 + zl * prow + (h + 1) * pitch + w) * WG_VB + (qd >> 2) * 64 + (qd & 3) * 8;
        *(u32x2*)d = t1;
        *(u32x2*)(d + 32) = t2;
        if (h <= 1) {  // rows 0 and 1 again below the last row (kh - 1 reaches 2)
          char* d2 = d + Hi * pitch * WG_VB;
          *(u32x2*)d2 = t1;
          *(u32x2*)(d2 + 32) = t2;
        }
        if (h == Hi - 1) {  // the last row again above the first
          char* d2 = d - Hi * pitch * WG_VB;
          *(u32x2*)d2 = t1;
          *(u32x2*)(d2 + 32) = t2;
        }
      }
    }
    __syncthreads();
    int vloc[2], vz[2], vh[2], vw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = vrow0 + 4 * j;
      vloc[j] = v;
      const int vv = min(v, R - 1);
      vz[j] = vv / PVo;
      const int p = vv - vz[j] * PVo;
      vh[j] = p / Wo;
      vw[j] = p - vh[j] * Wo;
    }
    for (int c = 0; c < RP / 16; ++c) {
      int xr[2];  // record of fine voxel (vz SZ + 0, 2 vh - 1, 2 vw - 1) = the (kz, kh, kw) = (0, 0, 0) tap; past the unit: any record
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool in = vloc[j] < R;
        xr[j] = in ? (vz[j] * a.SZ * prow + 2 * vh[j] * pitch + 2 * vw[j]) * WG_VB : 0;
      }
      fh4 g0[2], g1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* gp = gL + (size_t)min(vloc[j], RP - 1) * WG_VB + chan_off;
        g0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp));
        g1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(gp + 32));
      }
      const fh8 G0 = cat8(g0[0], g0[1]), G1 = cat8(g1[0], g1[1]);
      fh8 X0[NTAP], X1[NTAP];
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        fh4 x0[2], x1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* xp = xL + xr[j] + toff[t] + chan_off;
          x0[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp));
          x1[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(xp + 32));
        }
        X0[t] = cat8(x0[0], x0[1]);
        X1[t] = cat8(x1[0], x1[1]);
      }
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        accA[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X0[t], accA[t], 0, 0, 0);
        accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G0, X1[t], accB[t], 0, 0, 0);
        accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(G1, X0[t], accB[t], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        vloc[j] += 16;
        vw[j] += adv_w;
        vh[j] += adv_h;
        if (vw[j] >= Wo) { vw[j] -= Wo; vh[j] += 1; }
        while (vh[j] >= a.Ho) { vh[j] -= a.Ho; vz[j] += 1; }
      }
    }
  }
  float* pbase = a.partial + (((size_t)blockIdx.x * (a.A / 32) + ta) * a.tilesB + tb) * (size_t)T * 1024;
  const float inv = ginv * xinv;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tap = blockIdx.z * (T / 2) + wave + 8 * t;
    float* pp = pbase + (size_t)tap * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = (accA[t][r] + accB[t][r] * (1.f / 2048.f)) * inv;
  }
}

}  // namespace

// returns false when the geometry does not fit (caller falls back to the fp32 kernels).  gmax_word: a device word the
// caller owns (zeroed here); `partial` sized by wgrad_partial_floats.
namespace {
unsigned* g_absmax_word = nullptr;
const float* g_absmax_of = nullptr;  // tensor the word currently describes (stream order)
size_t g_absmax_n = 0;               // ... and its element count (0: a producer's note, size unknown)
}  // namespace
namespace {
bool g_absmax_fresh = false;  // the word was filled by the producer of g_absmax_of and has not been claimed yet
}
// A producer that can track max |x| while it writes x (gn_bwd_apply_kernel) takes the zeroed word here; the next
// launch_absmax_bits call claims it if (and only if) it asks for the same tensor -- a later tensor at a recycled workspace
// address can never match a stale note.
// The words come from a ring that is zeroed as a whole when it wraps (a hipMemsetAsync per request was ~80 four-byte fills per
// training step, 4 us each): a request takes the next word, still zero -- every launch that used the ring's previous round was
// enqueued on the stream before the wrap's memset.
namespace {
constexpr int ABSMAX_RING = 4096;
unsigned* g_absmax_ring = nullptr;
int g_absmax_next = 0;
unsigned* absmax_next_word(hipStream_t s) {
  if (!g_absmax_ring) {
    CD_HIP(hipMalloc((void**)&g_absmax_ring, sizeof(unsigned) * ABSMAX_RING));
    CD_HIP(hipMemsetAsync(g_absmax_ring, 0, sizeof(unsigned) * ABSMAX_RING, s));
    g_absmax_next = 0;
  }
  if (g_absmax_next == ABSMAX_RING) {
    CD_HIP(hipMemsetAsync(g_absmax_ring, 0, sizeof(unsigned) * ABSMAX_RING, s));
    g_absmax_next = 0;
  }
  return g_absmax_ring + g_absmax_next++;
}
}  // namespace
unsigned* absmax_word_fresh(const float* x, hipStream_t s) {
  g_absmax_word = absmax_next_word(s);
  g_absmax_of = x;
  g_absmax_n = 0;
  g_absmax_fresh = true;
  return g_absmax_word;
}
// The note is only good for the backward of ONE convolution: its owner drops it when that backward is enqueued, so a later tensor
// at a recycled workspace address (or the same tensor rewritten in place) can never pick up a stale maximum.
void absmax_note_drop() {
  g_absmax_of = nullptr;
  g_absmax_n = 0;
  g_absmax_fresh = false;
}
const unsigned* launch_absmax_bits(const float* x, size_t n, hipStream_t s) {
  if (g_absmax_fresh && g_absmax_of == x && g_absmax_word) {
    g_absmax_fresh = false;
    g_absmax_n = n;  // the note now has the consumer's size: the weight gradient of the same conv_backward re-uses it too
                     // (it was left at 0, so every weight gradient behind a producer-tracked dy ran its own pass: 28 per step)
    return g_absmax_word;
  }
  g_absmax_fresh = false;
  CD_REQUIRE(n % 4 == 0, "absmax: element count must be a multiple of 4");
  g_absmax_word = absmax_next_word(s);
  // one atomic per workgroup on ONE word: same-address atomics retire at ~12 ns each, so the 2048 workgroups this pass used to
  // launch spent 25 us in them whatever the tensor's size (rocprofv3, round 4: 35 calls x 25 us per training step).  256-512
  // workgroups with eight 16-byte loads in flight per thread stream a level-0 gradient (26 MB) in ~7 us.
  size_t blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n / 4, g_absmax_word);
  CD_HIP(hipGetLastError());
  g_absmax_of = x;
  g_absmax_n = n;
  return g_absmax_word;
}

static size_t wgrad16_lds(Dims3 d) {
  const int PV = d.h * d.w;
  const int RP = (WG_NZ * PV + 15) & ~15;
  return (size_t)(RP + 1) * WG_VB + (size_t)(WG_NZ + 2) * (d.h + 2) * (d.w + 1) * WG_VB;
}
// geometry of the z-sliding form: NZ planes per unit (units of <= 144 voxels, <= 192 for one large plane), a ring of 2 NZ + 2 plane
// slots, two dy images and the coefficient table within 160 KB
static bool wgrad_ring_geometry(Dims3 d, int* NZ_out, size_t* lds_out) {
  const int PV = d.h * d.w;
  if (PV > 192 || PV < 1) return false;
  int NZ = 144 / PV;
  if (NZ < 1) NZ = 1;
  if (NZ > d.d) NZ = d.d;
  for (; NZ >= 1; --NZ) {
    const int RP = (NZ * PV + 15) & ~15;
    const size_t lds = ((size_t)2 * RP + 1 + (size_t)(2 * NZ + 2) * (d.h + 2) * (d.w + 1)) * WG_VB + 512;
    if (lds <= 160 * 1024) {
      *NZ_out = NZ;
      *lds_out = lds;
      return true;
    }
  }
  return false;
}
bool wgrad_f16x2_eligible(Dims3 d) {
  if (getenv("CD_NO_WGRAD16")) return false;
  // (planes under 64 voxels were once left to the fp32 kernels: with the max-|dy| pass shared and at HBM speed the fp16 pipe
  // wins down to 8-voxel planes: 14.46 -> 13.3 ms per training step; CD_WGRAD16_MINPV restores a floor)
  static const int min_pv = getenv("CD_WGRAD16_MINPV") ? atoi(getenv("CD_WGRAD16_MINPV")) : 8;
  return d.d >= 1 && d.h * d.w >= min_pv && wgrad16_lds(d) <= 160 * 1024;
}
bool try_launch_wgrad_f16x2(const float* g, int A, const float* x, int Bc, int xld, int xoff, Dims3 d, int batch, float* partial,
                            unsigned* gmax_word, int* nblk_out, hipStream_t s, const float* xcoef) {
  if (!wgrad_f16x2_eligible(d)) return false;
  const size_t lds = wgrad16_lds(d);
  Wgrad16Args f;
  f.xcoef = xcoef;
  f.g = g; f.x = x; f.A = A; f.xld = xld; f.xoff = xoff; f.D = d.d; f.H = d.h; f.W = d.w;
  f.units_per_sample = (d.d + WG_NZ - 1) / WG_NZ;
  f.total_units = f.units_per_sample * batch;
  f.partial = partial; f.tilesB = Bc / 32;
  // max |dy|: reuse the word if the caller (conv_backward) already computed it for this tensor
  // (same pointer AND same size, noted since the last absmax_note_drop(): the dx convolution of this very backward)
  const size_t gn = (size_t)batch * d.vox() * A;
  f.gmax_bits = (g_absmax_of == g && g_absmax_n == gn && g_absmax_word) ? g_absmax_word : launch_absmax_bits(g, gn, s);
  (void)gmax_word;
  const int tiles = (A / 32) * (Bc / 32);
  int nblk = 256 / tiles;
  if (nblk < 32) nblk = 32;
  {
    // z-sliding form (wgrad_ring_f16x2_kernel): every workgroup one chunk of consecutive units of one sample where the batch allows
    int NZ = 0;
    size_t rlds = 0;
    static const bool no_ring = getenv("CD_NO_WGRAD_RING") != nullptr;
    if (!no_ring && wgrad_ring_geometry(d, &NZ, &rlds)) {
      WgradRingArgs r;
      r.g = g; r.x = x; r.A = A; r.xld = xld; r.xoff = xoff; r.D = d.d; r.H = d.h; r.W = d.w;
      r.NZ = NZ; r.S = 2 * NZ + 2;
      r.U = (d.d + NZ - 1) / NZ;
      int want = nblk;
      if (const char* e = getenv("CD_WGRAD_RING_NBLK")) want = atoi(e) > 0 ? atoi(e) : nblk;  // tests: few workgroups => long chunks, several per workgroup
      if (want > nblk) want = nblk;
      int cps = want / batch;
      if (cps < 1) cps = 1;
      if (cps > r.U) cps = r.U;
      r.upc = (r.U + cps - 1) / cps;
      r.cps = (r.U + r.upc - 1) / r.upc;
      r.total_chunks = r.cps * batch;
      r.partial = partial; r.tilesB = Bc / 32; r.gmax_bits = f.gmax_bits; r.xcoef = xcoef;
      r.abl = getenv("CD_WGRAD_ABL") ? atoi(getenv("CD_WGRAD_ABL")) : 0;
      int rblk = want < r.total_chunks ? want : r.total_chunks;
      static bool ring_attr = false;
      if (!ring_attr) {
        CD_HIP(hipFuncSetAttribute((const void*)wgrad_ring_f16x2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CD_HIP(hipFuncSetAttribute((const void*)wgrad_ring_f16x2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ring_attr = true;
      }
      static const bool no_one = getenv("CD_WGRAD_RING_GENERAL") != nullptr;  // A/B: the general K loop for one-plane units too
      if (NZ == 1 && !no_one) hipLaunchKernelGGL(wgrad_ring_f16x2_kernel<true>, dim3(rblk, tiles), dim3(512), rlds, s, r);
      else hipLaunchKernelGGL(wgrad_ring_f16x2_kernel<false>, dim3(rblk, tiles), dim3(512), rlds, s, r);
      CD_HIP(hipGetLastError());
      *nblk_out = rblk;
      return true;
    }
  }
  if (nblk > f.total_units) nblk = f.total_units;
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)wgrad_f16x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_f16x2_kernel, dim3(nblk, tiles), dim3(512), lds, s, f);
  CD_HIP(hipGetLastError());
  *nblk_out = nblk;
  return true;
}

// Strided (KD, 4, 4) / stride (SZ, 2, 2) weight gradient on the fp16 pipe (wgrad_strided_f16x2_kernel).  dg = the coarse grid (g), dx
// the fine grid (x).  max_slots: capacity of `partial` in [A x Bc x T]-float slots.  Returns false when it does not apply (the
// caller runs the f32-MFMA kernel).
bool try_launch_wgrad_strided_f16x2(const float* g, int A, Dims3 dg, const float* x, int Bc, int xld, int xoff, Dims3 dx, int kd, int sz,
                                    int batch, float* partial, int max_slots, int* nblk_out, hipStream_t s) {
  static const bool off = getenv("CD_NO_WGRAD16") != nullptr || getenv("CD_NO_WGRAD16_STRIDED") != nullptr;
  if (off || (kd != 3 && kd != 4) || dx.h < 2 || A % 32 || Bc % 32 || xld % 4 || xoff % 4) return false;
  // the coarse grid must be the strided conv's output of the fine one (padding 1 everywhere, circular in phi)
  if (dg.d != (dx.d + 2 - kd) / sz + 1 || dg.h != (dx.h - 2) / 2 + 1 || dg.w != (dx.w - 2) / 2 + 1) return false;
  const int PVo = dg.h * dg.w;
  const int pitch = dx.w + 2, prow = (dx.h + 3) * pitch;
  auto lds_for = [&](int nz) {
    const int RP = (nz * PVo + 15) & ~15;
    return (size_t)(RP + 1) * WG_VB + (size_t)((nz - 1) * sz + kd) * prow * WG_VB;
  };
  int NZ = 0;
  for (int nz : {4, 2, 1})
    if (nz <= dg.d && lds_for(nz) <= 160 * 1024) { NZ = nz; break; }
  if (!NZ) return false;
  WgradS16Args f;
  f.g = g; f.x = x; f.A = A; f.xld = xld; f.xoff = xoff;
  f.Do = dg.d; f.Ho = dg.h; f.Wo = dg.w; f.Di = dx.d; f.Hi = dx.h; f.Wi = dx.w; f.KD = kd; f.SZ = sz; f.NZ = NZ;
  f.units_per_sample = (dg.d + NZ - 1) / NZ;
  f.total_units = f.units_per_sample * batch;
  f.partial = partial; f.tilesB = Bc / 32;
  // both operands are rescaled from their maxima; the one the caller's input-gradient conv already measured is re-used (it must
  // be looked up BEFORE the other one's pass replaces the note)
  const size_t gn = (size_t)batch * dg.vox() * A, xn = (size_t)batch * dx.vox() * xld;
  auto noted = [&](const float* p, size_t n) { return (g_absmax_of == p && g_absmax_n == n && g_absmax_word) ? g_absmax_word : nullptr; };
  const unsigned* wg = noted(g, gn);
  const unsigned* wx = noted(x, xn);
  if (!wg) wg = launch_absmax_bits(g, gn, s);
  if (!wx) wx = launch_absmax_bits(x, xn, s);
  f.gmax_bits = wg; f.xmax_bits = wx;
  const int tiles = (A / 32) * (Bc / 32);
  int nblk = 128 / tiles;  // x two tap groups = one round of the 256 CUs
  if (nblk < 16) nblk = 16;
  if (nblk > f.total_units) nblk = f.total_units;
  if (nblk > max_slots) nblk = max_slots;
  if (nblk < 1) return false;
  const size_t lds = lds_for(NZ);
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)wgrad_strided_f16x2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CD_HIP(hipFuncSetAttribute((const void*)wgrad_strided_f16x2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (kd == 3) hipLaunchKernelGGL(wgrad_strided_f16x2_kernel<3>, dim3(nblk, tiles, 2), dim3(512), lds, s, f);
  else hipLaunchKernelGGL(wgrad_strided_f16x2_kernel<4>, dim3(nblk, tiles, 2), dim3(512), lds, s, f);
  CD_HIP(hipGetLastError());
  *nblk_out = nblk;
  return true;
}

}  // namespace cd
