// Backward kernels of the training step (Diffusion.compute_loss -> loss.backward()).
// The reference relies on torch autograd through stock modules; here every gradient is a hand-written kernel on the
// same channels-last layout as the forward path.  Input gradients of the convolutions reuse the forward conv kernels
// with re-packed (transposed / tap-flipped) weights; this file holds what has no forward counterpart:
//   * weight gradients: dW[a][b][tap] = sum over voxels of g[o][a] * x[in(o, tap)][b]  (contraction over voxels on the
//     fp32 MFMA), generic in taps / stride so that it serves 3x3x3, 1x1x1, the strided down conv and -- with the roles
//     of the two tensors swapped -- the transposed up conv;
//   * GroupNorm(+SiLU) backward (statistics pass + elementwise pass);
//   * small elementwise pieces (loss gradient, softmax backward of the linear attention, head).
#include "cd_common.h"
#include <cstdio>
#include <cstdlib>

namespace cd {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------------------
// Weight gradient.
//   g : (B, Og, A)  channels-last "output side" tensor (A = rows of dW), voxel grid (Dg, Hg, Wg)
//   x : (B, Ox, Bc) channels-last "input side" tensor (Bc = columns of dW), voxel grid (Dx, Hx, Wx)
//   dW[a][b][tap] = sum_{n, o} g[n][o][a] * x[n][in(o, tap)][b],  in(o,tap) = (oz*SZ + kz - 1, (oh*S + kh - 1) mod Hx, ow*S + kw - 1)
//   (zero outside z / r).  For a transposed conv the caller passes g = layer input, x = output gradient: same geometry.
// One workgroup = one 32x32 (a, b) tile x one voxel chunk; its 4 waves split the taps; every lane half takes one voxel of
// a pair (K = 2 per MFMA).  Partials [chunk][tap][32][32] are reduced in a fixed order by wgrad_reduce_kernel
// (deterministic, no float atomics), which also writes the torch layout.
// ------------------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* g;
  const float* x;
  int A, Bc;              // channel counts (ld of g / x)
  int xld, xoff;          // x may be a channel slice of a wider tensor (skip concat): row stride and offset
  int Dg, Hg, Wg, Dx, Hx, Wx;
  int KD, KH, KW, SZ, S;
  int batch;
  int per_sample;         // 1: no reduction over the batch (attention context gradient); partial index includes n
  int chunk_vox;          // output voxels per chunk (even)
  int nchunks;            // chunks per sample
  float* partial;         // [(n if per_sample)][chunk (x batch if !per_sample)][tileA][tileB][tap][32][32]
};

template <int TPW>  // taps per wave (upper bound); blockDim = 64 * ceil(T / TPW)
__global__ void __launch_bounds__(1024) wgrad_kernel(WgradArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int T = a.KD * a.KH * a.KW;
  const int tilesB = a.Bc / 32;
  const int ta = blockIdx.y / tilesB, tb = blockIdx.y % tilesB;
  const int chunk = blockIdx.x % a.nchunks;
  const int n = blockIdx.x / a.nchunks;
  const int Og = a.Dg * a.Hg * a.Wg, Ox = a.Dx * a.Hx * a.Wx;
  const int tap0 = wave * TPW;
  const int ntap = min(TPW, T - tap0);

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* gb = a.g + (size_t)n * Og * a.A + ta * 32 + col;
  const float* xb = a.x + (size_t)n * Ox * a.xld + a.xoff + tb * 32 + col;
  const int o0 = chunk * a.chunk_vox;
  const int o1 = min(o0 + a.chunk_vox, Og);
  if (T == 1 && a.S == 1 && a.SZ == 1 && TPW == 1) {
    // pointwise conv: in(o) = o.  Eight voxel pairs per trip: 16 loads in flight, no index arithmetic (the general loop below
    // spent ~100 VALU instructions of div/mod per 64-cycle MFMA and one memory round trip per voxel pair)
    constexpr int U = 8;
    int ob = o0;  // wave-uniform pair base: both lane halves run the same MFMAs (EXEC does not mask an MFMA)
    for (; ob + 2 * U <= o1; ob += 2 * U) {
      const int o = ob + half;
      float gv[U], xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        gv[u] = gb[(size_t)(o + 2 * u) * a.A];
        xv[u] = xb[(size_t)(o + 2 * u) * a.xld];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc[0] = MFMA32(gv[u], xv[u], acc[0]);
    }
    for (; ob < o0 + a.chunk_vox; ob += 2) {
      const int o = ob + half;
      const bool ov = o < o1;
      const float gv = ov ? gb[(size_t)o * a.A] : 0.f;
      const float xv = ov ? xb[(size_t)o * a.xld] : 0.f;
      acc[0] = MFMA32(gv, xv, acc[0]);
    }
  } else if (ntap > 0) {
    // tap geometry of this wave, hoisted out of the voxel loop (div / mod by run-time kernel extents)
    int dz[TPW], dh[TPW], dw[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tap = min(tap0 + t, T - 1);
      const int kw = tap % a.KW, kh = (tap / a.KW) % a.KH, kz = tap / (a.KW * a.KH);
      dz[t] = kz - (a.KD == 1 ? 0 : 1);
      dh[t] = kh - (a.KH == 1 ? 0 : 1);
      dw[t] = kw - (a.KW == 1 ? 0 : 1);
    }
    // U voxel pairs per trip: their U * (1 + TPW) loads are all in flight before the first MFMA
    constexpr int U = TPW >= 3 ? 3 : 4;
    for (int ob = o0; ob < o0 + a.chunk_vox; ob += 2 * U) {
      float gv[U], xv[U][TPW];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int o = ob + 2 * u + half;
        const bool ov = o < o1;
        const int oo = ov ? o : o0;
        const float g0 = gb[(size_t)oo * a.A];
        gv[u] = ov ? g0 : 0.f;
        const int ow = oo % a.Wg;
        const int t2 = oo / a.Wg;
        const int oh = t2 % a.Hg, oz = t2 / a.Hg;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int iz = oz * a.SZ + dz[t], iw = ow * a.S + dw[t];
          int ih = oh * a.S + dh[t];
          ih = ih < 0 ? ih + a.Hx : (ih >= a.Hx ? ih - a.Hx : ih);
          ih = ih >= a.Hx ? ih - a.Hx : ih;
          const bool ok = ov && iz >= 0 && iz < a.Dx && iw >= 0 && iw < a.Wx;
          const float v = xb[ok ? ((size_t)(iz * a.Hx + ih) * a.Wx + iw) * a.xld : 0];
          xv[u][t] = ok ? v : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
          if (t < ntap) acc[t] = MFMA32(gv[u], xv[u][t], acc[t]);
    }
  }
  // C layout: col = lane&31 (b), row = (r&3) + 8*(r>>2) + 4*half (a)
  const size_t slot = a.per_sample ? ((size_t)n * a.nchunks + chunk) : ((size_t)chunk * a.batch + n);
  float* pbase = a.partial + ((slot * (a.A / 32) + ta) * tilesB + tb) * (size_t)T * 1024;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t < ntap) {
      float* p = pbase + (size_t)(tap0 + t) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = acc[t][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1x1 conv (ResnetBlock shortcuts, the attention's to_qkv / to_out and -- per sample -- its context
// gradient): dW[a][b] = sum over rows of g[row][a] x[row][b], a tiny output over a very long K, i.e. a streaming pass over the two
// tensors.  Round 4: wgrad_kernel<1> ran it as ~12 k single-wave workgroups of dword loads, a pre-reduction of their 4096 partials
// and the final reduction -- three launches and ~60 us for the 107 MB of the level-0 to_qkv gradient, 27 times per training step.
// Here a workgroup of four waves loops over units of R rows: both tensors' rows are staged in LDS with 16-byte loads (fp32, row
// pitch = 32 mod 64 floats so that the two half-waves of an operand read hit disjoint banks), wave w runs rows 32 w .. 32 w + 31
// of the unit as 16 K-steps of v_mfma_f32_32x32x2_f32 for each of the workgroup's (a, b) tiles (lanes are channels, the lane
// half is the row of the pair: exact fp32, nothing to rescale), accumulators live across all units, and the four waves' sums are
// added in a fixed order through LDS into ONE partial per workgroup.  Segments: the whole batch (rows contiguous across samples)
// or, per_sample, one sample each.
// ------------------------------------------------------------------------------------------------------------
struct Wgrad1Args {
  const float* g;   // (rows, A)
  const float* x;   // (rows, xld) read at channel offset xoff, Bc channels
  int A, Bc, xld, xoff;
  int ldA, ldB;     // LDS row pitches (floats)
  int R;            // rows per unit (multiple of 128)
  long long rows_per_seg;
  int units_per_seg, wgs_per_seg;
  int tilesB, ntiles;
  float* partial;   // [gridDim.x][tilesA][tilesB][32][32]
  int dbg;          // ablation (CD_W1_DBG): 1 = no MFMAs, 2 = no global loads
};

// NQ: 16-byte quads of a unit per thread = ceil(R (A + Bc) / 4 / 256); W1_NT: (a, b) tiles per workgroup (blockIdx.y takes the next
// W1_NT; the launcher picks a divisor of the tile count, so every tile index below is valid)
template <int NQ, int W1_NT>
__global__ void __launch_bounds__(256) wgrad1x1_kernel(Wgrad1Args a) {
  extern __shared__ __attribute__((aligned(16))) float w1[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int seg = blockIdx.x / a.wgs_per_seg, wq = blockIdx.x % a.wgs_per_seg;
  const int tile0 = blockIdx.y * W1_NT;
  float* const sG = w1;
  float* const sX = w1 + (size_t)a.R * a.ldA;
  int ga[W1_NT], xb[W1_NT];  // channel offsets of this lane's operand element per tile
#pragma unroll
  for (int t = 0; t < W1_NT; ++t) {
    const int tile = tile0 + t;
    ga[t] = (tile / a.tilesB) * 32 + col;
    xb[t] = (tile % a.tilesB) * 32 + col;
  }
  f32x16 acc[W1_NT];
#pragma unroll
  for (int t = 0; t < W1_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* gseg = a.g + (size_t)seg * a.rows_per_seg * a.A;
  const float* xseg = a.x + (size_t)seg * a.rows_per_seg * a.xld + a.xoff;
  const int qa = a.A >> 2, qb = a.Bc >> 2, qrow = qa + qb;  // 16-byte quads per row: g's, then x's
  const int nq = a.R * qrow;
  // exact small-integer division by reciprocal: (i + 0.5) / qrow is never within float error of an integer for i < 2^20
  const float inv_qrow = 1.f / (float)qrow;
  auto row_of = [&](int i) { return (int)(((float)i + 0.5f) * inv_qrow); };
  // a unit's rows travel global -> registers -> LDS; the loads of unit u + 1 are issued before the MFMAs of unit u and land under
  // them (one memory round trip per unit was most of a workgroup's time: the first version took 45 us for the 107 MB of the level-0
  // to_qkv gradient)
  f32x4 v[NQ];
  auto issue = [&](int u) {
    const long long r0 = (long long)u * a.R;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int i = tid + k * 256;
      const int row = row_of(i), q = i - row * qrow;
      v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < nq && r0 + row < a.rows_per_seg && !(a.dbg & 2))
        v[k] = q < qa ? *(const f32x4*)(gseg + (size_t)(r0 + row) * a.A + q * 4)
                      : *(const f32x4*)(xseg + (size_t)(r0 + row) * a.xld + (q - qa) * 4);
    }
  };
  int u = wq;
  if (u < a.units_per_seg) issue(u);
  for (; u < a.units_per_seg; u += a.wgs_per_seg) {
    __syncthreads();  // the previous unit has been consumed
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int i = tid + k * 256;
      if (i < nq) {
        const int row = row_of(i), q = i - row * qrow;
        if (q < qa) *(f32x4*)(sG + (size_t)row * a.ldA + q * 4) = v[k];
        else *(f32x4*)(sX + (size_t)row * a.ldB + (q - qa) * 4) = v[k];
      }
    }
    __syncthreads();
    if (u + a.wgs_per_seg < a.units_per_seg) issue(u + a.wgs_per_seg);
    for (int rb = wave * 32; rb < ((a.dbg & 1) ? 0 : a.R); rb += 128) {
      // eight K-steps' operands are requested before their MFMAs (as a plain loop every MFMA waited for its own LDS round trip
      // behind a branch: 400 cycles per 64-cycle instruction)
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += 8) {
        float gv[8][W1_NT], xv[8][W1_NT];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const int row = rb + 2 * (s0 + s2) + half;
#pragma unroll
          for (int t = 0; t < W1_NT; ++t) {
            gv[s2][t] = sG[row * a.ldA + ga[t]];
            xv[s2][t] = sX[row * a.ldB + xb[t]];
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
          for (int t = 0; t < W1_NT; ++t) acc[t] = MFMA32(gv[s2][t], xv[s2][t], acc[t]);
      }
    }
  }
  // the four waves' sums, wave 0 first, through LDS: one partial per workgroup
  __syncthreads();
#pragma unroll
  for (int t = 0; t < W1_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) w1[((wave * W1_NT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = acc[t][r];
  __syncthreads();
  const int tilesA = a.A / 32;
  for (int i = tid; i < W1_NT * 1024; i += 256) {
    const int t = i >> 10, e = i & 1023;
    const float v = ((w1[(0 * W1_NT + t) * 1024 + e] + w1[(1 * W1_NT + t) * 1024 + e]) + w1[(2 * W1_NT + t) * 1024 + e]) +
                    w1[(3 * W1_NT + t) * 1024 + e];
    const int tile = tile0 + t;
    a.partial[(((size_t)blockIdx.x * tilesA + tile / a.tilesB) * a.tilesB + tile % a.tilesB) * 1024 + e] = v;
  }
}

// returns false when the shape does not fit (the caller runs wgrad_kernel<1>)
static bool try_launch_wgrad1x1(const float* g, int A, const float* x, int Bc, int xld, int xoff, int64_t vox, int batch, bool per_sample,
                                float* partial, size_t partial_slots, int* nslots_out, hipStream_t s) {
  static const bool off = getenv("CD_NO_WGRAD1X1") != nullptr;
  if (off || A % 32 || Bc % 32 || xld % 4 || xoff % 4) return false;
  Wgrad1Args a;
  a.g = g; a.x = x; a.A = A; a.Bc = Bc; a.xld = xld; a.xoff = xoff;
  a.ldA = A + (A % 64 == 32 ? 0 : 32);
  a.ldB = Bc + (Bc % 64 == 32 ? 0 : 32);
  a.tilesB = Bc / 32;
  a.ntiles = (A / 32) * (Bc / 32);
  a.R = 128;
  const int NT = a.ntiles % 4 == 0 ? 4 : (a.ntiles % 3 == 0 ? 3 : (a.ntiles % 2 == 0 ? 2 : (a.ntiles == 1 ? 1 : 0)));
  if (!NT) return false;
  const size_t stage = (size_t)a.R * (a.ldA + a.ldB) * 4, red = (size_t)4 * NT * 4096;
  const size_t lds = stage > red ? stage : red;
  if (lds > 160 * 1024) return false;
  a.rows_per_seg = per_sample ? vox : (int64_t)batch * vox;
  const int64_t units = (a.rows_per_seg + a.R - 1) / a.R;
  CD_REQUIRE(units < (1ll << 30), "wgrad 1x1: too many rows");
  a.units_per_seg = (int)units;
  const int groups = a.ntiles / NT;
  // workgroups: two per CU where the staging fits twice (one streams while the other multiplies), at least 2 units each
  const int nseg = per_sample ? batch : 1;
  int64_t want = (lds <= 80 * 1024 ? 512 : 256) / groups / nseg;
  if (want < 1) want = 1;
  if (want > units) want = units;
  if ((size_t)want * nseg > partial_slots) want = (int64_t)(partial_slots / nseg);
  if (want < 1) return false;
  a.wgs_per_seg = (int)want;
  a.partial = partial;
  static const int dbg = getenv("CD_W1_DBG") ? atoi(getenv("CD_W1_DBG")) : 0;
  a.dbg = dbg;
  const int nqt = (a.R * ((A + Bc) / 4) + 255) / 256;
  const dim3 grid((unsigned)(a.wgs_per_seg * nseg), (unsigned)groups);
#define W1_CASE(N, T)                                                                                                           \
  if (nqt <= N && NT == T) {                                                                                                    \
    static bool attr = false;                                                                                                   \
    if (!attr) {                                                                                                                \
      CD_HIP(hipFuncSetAttribute((const void*)wgrad1x1_kernel<N, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   \
      attr = true;                                                                                                              \
    }                                                                                                                           \
    hipLaunchKernelGGL((wgrad1x1_kernel<N, T>), grid, dim3(256), lds, s, a);                                                    \
    CD_HIP(hipGetLastError());                                                                                                  \
    *nslots_out = a.wgs_per_seg;                                                                                                \
    return true;                                                                                                                \
  }
#define W1_CASES(N) W1_CASE(N, 1) W1_CASE(N, 2) W1_CASE(N, 3) W1_CASE(N, 4)
  W1_CASES(8) W1_CASES(16) W1_CASES(24) W1_CASES(32)
#undef W1_CASES
#undef W1_CASE
  return false;  // (wider than 128 + 128 channels: the general kernel)
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient of the stride-1 3x3x3 conv, LDS-staged and persistent (the hot backward kernel).
// A workgroup loops over (sample, 256-voxel flat range) units; per unit it stages the output-gradient rows [R][32] and
// the input z-planes the range touches (+1 halo plane each side, zero outside) as [voxel][32] fp32, plus a per-voxel
// record (LDS index, phi/r edge flags).  Its waves split the 27 taps; a wave keeps one 32x32 accumulator per tap in
// registers across ALL its units and writes ONE partial per workgroup (grid = CU count => 256 partial slots instead of one
// per voxel chunk).  K = 2 voxels per v_mfma_f32_32x32x2_f32: lane half h takes voxel 2p+h, lanes are channels, so both
// LDS operand reads are conflict-free 128-B rows.
// ------------------------------------------------------------------------------------------------------------
struct WgradFlatArgs {
  const float* g;   // (B, vox, A)
  const float* x;   // (B, vox, xld) read at channel offset xoff
  int A, xld, xoff;
  int D, H, W;
  int R, P;         // voxels per unit, plane capacity
  int units_per_sample, total_units;
  float* partial;   // [gridDim.x][tilesA][tilesB][27][32][32]
  int tilesB;
};

template <int TPW>
__global__ void __launch_bounds__(64 * ((27 + TPW - 1) / TPW)) wgrad_flat_kernel(WgradFlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NW = (27 + TPW - 1) / TPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int ta = blockIdx.y / a.tilesB, tb = blockIdx.y % a.tilesB;
  const int HW = a.H * a.W, vox = a.D * HW;
  float* gL = lds;                                   // [R][32]
  float* xL = lds + a.R * 32;                        // [P*HW][32]
  int* tbl = (int*)(xL + (size_t)a.P * HW * 32);     // [R]

  // this wave's taps
  int toff[TPW], tdh[TPW], tdw[TPW];
  int ntap = 0;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave + t * NW;
    const int dz = tap / 9 - 1, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
    toff[t] = dz * HW + dh * a.W + dw;
    tdh[t] = dh;
    tdw[t] = dw;
    if (tap < 27) ntap = t + 1;
  }
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int u = blockIdx.x; u < a.total_units; u += gridDim.x) {
    const int n = u / a.units_per_sample, ux = u - n * a.units_per_sample;
    const int v0 = ux * a.R, vend = min(v0 + a.R, vox);
    const int zA = v0 / HW - 1, zB = (vend - 1) / HW + 1;
    const int nstage = (zB - zA + 1) * HW, gbase = zA * HW;
    __syncthreads();  // previous unit fully consumed
    {
      const float* gs = a.g + ((size_t)n * vox + v0) * a.A + ta * 32;
      for (int i0 = tid; i0 < a.R * 8; i0 += 4 * blockDim.x) {
        f32x4 val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * blockDim.x;
          val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (i < a.R * 8 && v0 + (i >> 3) < vend) val[k] = *(const f32x4*)(gs + (size_t)(i >> 3) * a.A + (i & 7) * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * blockDim.x;
          if (i < a.R * 8) *(f32x4*)(gL + (i >> 3) * 32 + (i & 7) * 4) = val[k];
        }
      }
      const float* xs = a.x + (size_t)n * vox * a.xld + a.xoff + tb * 32;
      for (int i0 = tid; i0 < nstage * 8; i0 += 4 * blockDim.x) {
        f32x4 val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * blockDim.x;
          const int gv = gbase + (i >> 3);
          val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (i < nstage * 8 && gv >= 0 && gv < vox) val[k] = *(const f32x4*)(xs + (size_t)gv * a.xld + (i & 7) * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * blockDim.x;
          if (i < nstage * 8) *(f32x4*)(xL + (i >> 3) * 32 + (i & 7) * 4) = val[k];
        }
      }
      for (int v = tid; v < a.R; v += blockDim.x) {
        const int gv = v0 + v;
        int rec = -1;
        if (gv < vend) {
          const int r = gv % HW;
          const int h = r / a.W, w = r - h * a.W;
          rec = (gv - gbase) | (w == 0 ? 1 << 20 : 0) | (w == a.W - 1 ? 1 << 21 : 0) | (h == 0 ? 1 << 22 : 0) |
                (h == a.H - 1 ? 1 << 23 : 0);
        }
        tbl[v] = rec;
      }
    }
    __syncthreads();
    // 4 voxel pairs per trip: all their LDS operands are requested before the first MFMA of the group issues
    constexpr int UP = 4;
    for (int p0 = 0; p0 < a.R; p0 += 2 * UP) {
      float gv[UP], xv[UP][TPW];
#pragma unroll
      for (int k = 0; k < UP; ++k) {
        const int p = p0 + 2 * k;
        const int rec = p < a.R ? tbl[p + half] : -1;
        gv[k] = p < a.R ? gL[(p + half) * 32 + col] : 0.f;
        const int nb = rec & 0xfffff;
        const bool vvalid = rec >= 0;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          int nidx = nb + toff[t];
          if (tdh[t] < 0 && (rec & (1 << 22))) nidx += HW;        // wrap phi: row -1 -> H-1
          if (tdh[t] > 0 && (rec & (1 << 23))) nidx -= HW;        // row H -> 0
          const bool ok = vvalid && t < ntap && !(tdw[t] < 0 && (rec & (1 << 20))) && !(tdw[t] > 0 && (rec & (1 << 21)));
          xv[k][t] = ok ? xL[nidx * 32 + col] : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < UP; ++k)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
          if (t < ntap) acc[t] = MFMA32(gv[k], xv[k][t], acc[t]);
    }
  }
  float* pbase = a.partial + (((size_t)blockIdx.x * (a.A / 32) + ta) * a.tilesB + tb) * (size_t)27 * 1024;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave + t * NW;
    if (tap < 27) {
      float* pp = pbase + (size_t)tap * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = acc[t][r];
    }
  }
}

// dW (torch layout) = sum over slots of the partial tiles.
//   transposed_out = 0: dW[a][b][tap] (Conv3d weight, a = out channel)   1: dW[b][a][tap]
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int A, int Bc, int T, int nslots,
                                    int accumulate, int transposed_out, size_t sample_stride_partial, size_t sample_stride_out,
                                    int b_total, int b_off, int slot_step = 1) {
  const size_t total = (size_t)A * Bc * T;
  const size_t sstride = total * (size_t)slot_step;  // the slots left by wgrad_prereduce_kernel are slot_step apart
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = blockIdx.y;
  // idx enumerates the partial-tile order: [ta][tb][tap][row a][col b]
  const int cb = idx & 31, ra = (idx >> 5) & 31;
  size_t rest = idx >> 10;
  const int tap = rest % T;
  rest /= T;
  const int tilesB = Bc / 32;
  const int tb = rest % tilesB, ta = rest / tilesB;
  const float* p = partial + (size_t)n * sample_stride_partial + idx;
  // 16 independent loads in flight per thread (the slot loop is latency-bound otherwise); fixed summation order
  float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 16 <= nslots; k += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(k + j) * sstride];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc8[j & 7] += v[j];
  }
  for (; k < nslots; ++k) acc8[k & 7] += p[(size_t)k * sstride];
  const float s = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
  const int ga = ta * 32 + ra, gb2 = tb * 32 + cb;
  // b_total / b_off: the b columns are a slice of a wider weight (second source of a channel concat)
  const size_t o = (transposed_out ? ((size_t)(gb2 + b_off) * A + ga) : ((size_t)ga * b_total + b_off + gb2)) * T + tap;
  float* d = dw + (size_t)n * sample_stride_out + o;
  *d = accumulate ? *d + s : s;
}

// The same reduction with hundreds of slots (one per workgroup of the persistent weight-gradient kernels), two levels in one
// launch: a block of 256 threads = 64 consecutive outputs x 4 slot lanes (one wave each: a wave's load is 256 contiguous bytes of
// one slot); wave w sums slots w, w + 4, ... with 16 loads in flight, the four waves are combined in a fixed order through LDS.
// Deterministic; 4 rounds of loads for 256 slots where the single-level loop needs 16.
__global__ void __launch_bounds__(256) wgrad_reduce1_kernel(const float* __restrict__ partial, float* __restrict__ dw, int A, int Bc,
                                                            int nslots, int accumulate, int transposed_out,
                                                            size_t sample_stride_partial, size_t sample_stride_out, int b_total, int b_off,
                                                            int T = 1) {
  const size_t total = (size_t)A * Bc * T;
  const int oi = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t idx = (size_t)blockIdx.x * 64 + oi;
  const int n = blockIdx.y;
  float acc = 0.f;
  if (idx < total) {
    const float* p = partial + (size_t)n * sample_stride_partial + idx;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = sl;
    for (; k + 4 * 15 < nslots; k += 4 * 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(k + 4 * j) * total];
#pragma unroll
      for (int j = 0; j < 16; ++j) a8[j & 7] += v[j];
    }
    for (int j = 0; k < nslots; k += 4, ++j) a8[j & 7] += p[(size_t)k * total];
    acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  __shared__ float sR[4][64];
  sR[sl][oi] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && idx < total) {
    const float sum = (sR[0][oi] + sR[1][oi]) + (sR[2][oi] + sR[3][oi]);
    const int cb = idx & 31, ra = (idx >> 5) & 31;
    size_t rest = idx >> 10;
    const int tap = rest % T;
    rest /= T;
    const int tilesB = Bc / 32;
    const int tb = rest % tilesB, ta = rest / tilesB;
    const int ga = ta * 32 + ra, gb2 = tb * 32 + cb;
    const size_t o = (transposed_out ? ((size_t)(gb2 + b_off) * A + ga) : ((size_t)ga * b_total + b_off + gb2)) * T + tap;
    float* d = dw + (size_t)n * sample_stride_out + o;
    *d = accumulate ? *d + sum : sum;
  }
}
// All queued slot reductions of a training step in one launch (WgradReduceQueue, cd_common.h): a block finds its job by binary
// search over the jobs' first blocks, then runs wgrad_reduce1_kernel's body (64 consecutive outputs x 4 slot lanes).
struct WgradReduceJobs {
  WgradReduceJob job[WgradReduceQueue::kMax];
  int n;
};
__global__ void __launch_bounds__(256) wgrad_reduce_jobs_kernel(WgradReduceJobs J) {
  int lo = 0, hi = J.n - 1;
  while (lo < hi) {  // last job whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (J.job[mid].first_block <= blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const WgradReduceJob& j = J.job[lo];
  const int A = j.A, Bc = j.Bc, T = j.T, nslots = j.nslots;
  const size_t total = (size_t)A * Bc * T;
  const int oi = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t idx = (size_t)(blockIdx.x - j.first_block) * 64 + oi;
  float acc = 0.f;
  if (idx < total) {
    const float* p = j.partial + idx;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = sl;
    for (; k + 4 * 15 < nslots; k += 4 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(k + 4 * u) * total];
#pragma unroll
      for (int u = 0; u < 16; ++u) a8[u & 7] += v[u];
    }
    for (int u = 0; k < nslots; k += 4, ++u) a8[u & 7] += p[(size_t)k * total];
    acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  __shared__ float sR[4][64];
  sR[sl][oi] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && idx < total) {
    const float sum = (sR[0][oi] + sR[1][oi]) + (sR[2][oi] + sR[3][oi]);
    const int cb = idx & 31, ra = (idx >> 5) & 31;
    size_t rest = idx >> 10;
    const int tap = rest % T;
    rest /= T;
    const int tilesB = Bc / 32;
    const int tb = rest % tilesB, ta = rest / tilesB;
    const int ga = ta * 32 + ra, gb2 = tb * 32 + cb;
    const size_t o = ((j.flags & 2) ? ((size_t)(gb2 + j.b_off) * A + ga) : ((size_t)ga * j.b_total + j.b_off + gb2)) * T + tap;
    float* d = j.dw + o;
    *d = (j.flags & 1) ? *d + sum : sum;
  }
}
namespace {
thread_local WgradReduceQueue* g_wq = nullptr;
}
void wgrad_queue_set(WgradReduceQueue* q) { g_wq = q; }
void wgrad_queue_flush(WgradReduceQueue* q, hipStream_t s) {
  if (!q || q->n <= 0) return;
  WgradReduceJobs J;
  for (int i = 0; i < q->n; ++i) J.job[i] = q->job[i];
  J.n = q->n;
  hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3(q->blocks), dim3(256), 0, s, J);
  CD_HIP(hipGetLastError());
  q->n = 0;
  q->blocks = 0;
}
// a partial buffer of the current queue (or nullptr: none current / no room); counts the request either way
static float* wgrad_queue_alloc(size_t nfloats) {
  WgradReduceQueue* q = g_wq;
  if (!q) return nullptr;
  nfloats = (nfloats + 63) & ~(size_t)63;
  q->need += nfloats;
  if (!q->base || q->used + nfloats > q->cap) return nullptr;
  float* r = q->base + q->used;
  q->used += nfloats;
  return r;
}
// the reduction of `partial` (a buffer of the current queue) as a job; false if it has to be launched here
static bool wgrad_queue_push(const float* partial, float* dw, int A, int Bc, int T, int nslots, bool accumulate, bool transposed_out,
                             int b_total, int b_off, hipStream_t s) {
  WgradReduceQueue* q = g_wq;
  if (!q || !q->base || partial < q->base || partial >= q->base + q->cap) return false;
  if (q->n == WgradReduceQueue::kMax) wgrad_queue_flush(q, s);  // (deeper networks than the shipped ones: flush and go on)
  WgradReduceJob& j = q->job[q->n++];
  j.partial = partial; j.dw = dw; j.A = A; j.Bc = Bc; j.T = T; j.nslots = nslots;
  j.flags = (accumulate ? 1 : 0) | (transposed_out ? 2 : 0);
  j.b_total = b_total; j.b_off = b_off; j.first_block = q->blocks;
  q->blocks += (unsigned)(((size_t)A * Bc * T + 63) / 64);
  return true;
}

// the slot reduction of the 27- and 48-tap weight gradients (up to 256 partials, one per workgroup): two-level from 64 slots on
static void launch_wgrad_reduce_slots(const float* partial, float* dw, int A, int Bc, int T, int nslots, bool accumulate,
                                      bool transposed_out, int b_total, int b_off, hipStream_t s) {
  const size_t total = (size_t)A * Bc * T;
  if (wgrad_queue_push(partial, dw, A, Bc, T, nslots, accumulate, transposed_out, b_total, b_off, s)) return;
  static const bool one_level = getenv("CD_WGRAD_REDUCE_1LEVEL") != nullptr;
  if (nslots >= 64 && !one_level)
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((unsigned)((total + 63) / 64), 1), dim3(256), 0, s, partial, dw, A, Bc, nslots,
                       accumulate ? 1 : 0, transposed_out ? 1 : 0, (size_t)nslots * total, total, b_total, b_off, T);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1), dim3(256), 0, s, partial, dw, A, Bc, T, nslots,
                       accumulate ? 1 : 0, transposed_out ? 1 : 0, (size_t)nslots * total, total, b_total, b_off);
  CD_HIP(hipGetLastError());
}

// First level of the slot reduction when there are many slots (the 1x1 convs write 4096 four-KiB partials: a single-level
// reduce is 1024 threads x 4096 serial loads = 75 us of a 100 us weight gradient).  Group g sums its `per` consecutive slots
// in a fixed order into the group's first slot, in place; wgrad_reduce_kernel then sums the group heads (slot_step = per).
__global__ void wgrad_prereduce_kernel(float* __restrict__ partial, size_t total, int nslots, int per) {
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k0 = blockIdx.y * per, k1 = min(k0 + per, nslots);
  float* p = partial + idx;
  float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = k0;
  for (; k + 16 <= k1; k += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(k + j) * total];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc8[j & 7] += v[j];
  }
  for (; k < k1; ++k) acc8[(k - k0) & 7] += p[(size_t)k * total];
  p[(size_t)k0 * total] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
}
// returns the slot step for wgrad_reduce_kernel and updates *nslots to the number of group heads
static int wgrad_prereduce(float* partial, size_t total, int* nslots, hipStream_t s) {
  if (*nslots < 256) return 1;
  const int per = 64, groups = (*nslots + per - 1) / per;
  hipLaunchKernelGGL(wgrad_prereduce_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)groups), dim3(256), 0, s, partial, total,
                     *nslots, per);
  CD_HIP(hipGetLastError());
  *nslots = groups;
  return per;
}

// chunks per sample: enough workgroups to fill the chip, chunks of >= 128 voxels, partial volume <= 32 MiB
int wgrad_chunks(int64_t out_vox, int batch, bool per_sample, int A, int Bc, int T) {
  int64_t want = per_sample ? 8 : (512 + batch - 1) / batch;
  if (T == 1) want = per_sample ? 32 : (4096 + batch - 1) / batch;  // single-wave workgroups, 4 KiB partials: use many
  const int64_t cap = (out_vox + (T == 1 ? 63 : 127)) / (T == 1 ? 64 : 128);
  if (want > cap) want = cap;
  const int64_t per_slot = (int64_t)A * Bc * T * 4;
  int64_t mem = (32ll << 20) / (per_slot * (per_sample ? 1 : batch));
  if (mem < 1) mem = 1;
  if (want > mem) want = mem;
  return (int)(want < 1 ? 1 : want);
}
size_t wgrad_partial_floats(int64_t out_vox, int batch, bool per_sample, int A, int Bc, int T) {
  size_t slots = (size_t)wgrad_chunks(out_vox, batch, per_sample, A, Bc, T) * batch;
  if (T == 27 && slots < 256) slots = 256;  // the persistent stride-1 kernel writes one partial per workgroup
  return slots * A * Bc * T;
}

bool wgrad_x_norm_supported(Dims3 dg, Dims3 dx, int kd, int kh, int kw, int sz, int sxy) {
  return kd * kh * kw == 27 && sz == 1 && sxy == 1 && dg.vox() == dx.vox() && wgrad_f16x2_eligible(dg);
}

void launch_wgrad(const float* g, int A, Dims3 dg, const float* x, int Bc, int xld, int xoff, Dims3 dx, int kd, int kh, int kw,
                  int sz, int sxy, int batch, bool per_sample, float* partial, float* dw, bool accumulate, bool transposed_out,
                  hipStream_t s, int b_total, int b_off, const float* xcoef) {
  CD_REQUIRE(A % 32 == 0 && Bc % 32 == 0, "wgrad: channel counts must be multiples of 32");
  CD_REQUIRE(!xcoef || (!per_sample && wgrad_x_norm_supported(dg, dx, kd, kh, kw, sz, sxy)),
             "wgrad: a normalised x operand is only read by the fp16-pipe 3x3x3 kernel");
  if (b_total <= 0) b_total = Bc;
  if (!per_sample) {  // training step: a buffer that outlives the caller's, its reduction deferred (WgradReduceQueue)
    if (float* qp = wgrad_queue_alloc(wgrad_partial_floats(dg.vox(), batch, false, A, Bc, kd * kh * kw))) partial = qp;
  }
  WgradArgs a;
  a.g = g; a.x = x; a.A = A; a.Bc = Bc; a.xld = xld; a.xoff = xoff;
  a.Dg = dg.d; a.Hg = dg.h; a.Wg = dg.w; a.Dx = dx.d; a.Hx = dx.h; a.Wx = dx.w;
  a.KD = kd; a.KH = kh; a.KW = kw; a.SZ = sz; a.S = sxy; a.batch = batch; a.per_sample = per_sample ? 1 : 0;
  const int Tt = kd * kh * kw;
  if (Tt == 27 && sz == 1 && sxy == 1 && !per_sample && dg.vox() == dx.vox()) {
    // fp16 matrix pipe (f16x2), transposing LDS loads: kernels_wgrad16.hip
    static unsigned* gmax_word = nullptr;
    if (!gmax_word) CD_HIP(hipMalloc((void**)&gmax_word, 64));
    int nblk16 = 0;
    if (wgrad_f16x2_eligible(dg)) {
      char cat16[96];
      std::snprintf(cat16, sizeof cat16, "wgrad T27 C%dx%d n%ld", A, Bc, (long)dg.vox());
      prof::Scope scope16(cat16, s, 2.0 * 27 * A * Bc * (double)dg.vox() * batch, 4.0 * batch * (double)dg.vox() * (A + Bc));
      CD_REQUIRE(try_launch_wgrad_f16x2(g, A, x, Bc, xld, xoff, dg, batch, partial, gmax_word, &nblk16, s, xcoef), "internal: wgrad f16x2");
      launch_wgrad_reduce_slots(partial, dw, A, Bc, 27, nblk16, accumulate, transposed_out, b_total, b_off, s);
      return;
    }
  }
  if (Tt == 27 && sz == 1 && sxy == 1 && !per_sample && dg.vox() == dx.vox() && !getenv("CD_NO_WGRAD_FLAT")) {
    // LDS-staged persistent kernel
    const int HW = dg.h * dg.w;
    int R = 256;
    while (R > 32 && (int64_t)(R / 2) >= dg.vox()) R /= 2;
    int P = (R - 1) / HW + 4;
    size_t lds = ((size_t)R * 32 + (size_t)P * HW * 32 + R) * 4;
    while (lds > 150 * 1024 && R > 32) {
      R /= 2;
      P = (R - 1) / HW + 4;
      lds = ((size_t)R * 32 + (size_t)P * HW * 32 + R) * 4;
    }
    if (lds <= 150 * 1024) {
      WgradFlatArgs f;
      f.g = g; f.x = x; f.A = A; f.xld = xld; f.xoff = xoff; f.D = dg.d; f.H = dg.h; f.W = dg.w; f.R = R; f.P = P;
      f.units_per_sample = (int)((dg.vox() + R - 1) / R);
      f.total_units = f.units_per_sample * batch;
      f.partial = partial; f.tilesB = Bc / 32;
      const int tiles = (A / 32) * (Bc / 32);
      int nblk = 256 / tiles;  // partial slots: one per workgroup
      if (nblk < 32) nblk = 32;
      if (nblk > f.total_units) nblk = f.total_units;
      char cat[96];
      std::snprintf(cat, sizeof cat, "wgrad T27 C%dx%d n%ld", A, Bc, (long)dg.vox());
      prof::Scope scope(cat, s, 2.0 * 27 * A * Bc * (double)dg.vox() * batch, 4.0 * batch * (double)dg.vox() * (A + Bc));
      static bool attr_set = false;
      if (!attr_set) {
        CD_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
      }
      hipLaunchKernelGGL(wgrad_flat_kernel<4>, dim3(nblk, tiles), dim3(64 * 7), lds, s, f);
      CD_HIP(hipGetLastError());
      launch_wgrad_reduce_slots(partial, dw, A, Bc, 27, nblk, accumulate, transposed_out, b_total, b_off, s);
      return;
    }
  }
  if (kh == 4 && kw == 4 && sxy == 2 && !per_sample) {
    // the strided convs between the levels: fp16 matrix pipe (round 4; kernels_wgrad16.hip)
    const int Ts = kd * 16;
    const int max_slots = wgrad_chunks(dg.vox(), batch, false, A, Bc, Ts) * batch;
    int nblk = 0;
    char cats[96];
    std::snprintf(cats, sizeof cats, "wgrad T%d C%dx%d n%ld", Ts, A, Bc, (long)dg.vox());
    prof::Scope scope_s(cats, s, 2.0 * Ts * A * Bc * (double)dg.vox() * batch, 4.0 * batch * ((double)dg.vox() * A + (double)dx.vox() * Bc));
    if (try_launch_wgrad_strided_f16x2(g, A, dg, x, Bc, xld, xoff, dx, kd, sz, batch, partial, max_slots, &nblk, s)) {
      launch_wgrad_reduce_slots(partial, dw, A, Bc, Ts, nblk, accumulate, transposed_out, b_total, b_off, s);
      return;
    }
  }
  if (Tt == 1 && sz == 1 && sxy == 1 && dg.vox() == dx.vox()) {
    const size_t cap_slots = (size_t)wgrad_chunks(dg.vox(), batch, per_sample, A, Bc, 1) * batch;  // what the caller's buffer holds
    int ns = 0;
    char cat1[96];
    std::snprintf(cat1, sizeof cat1, "wgrad T1 C%dx%d n%ld", A, Bc, (long)dg.vox());
    prof::Scope scope1(cat1, s, 2.0 * A * Bc * (double)dg.vox() * batch, 4.0 * batch * (double)dg.vox() * (A + Bc));
    if (try_launch_wgrad1x1(g, A, x, Bc, xld, xoff, dg.vox(), batch, per_sample, partial, cap_slots, &ns, s)) {
      const size_t total = (size_t)A * Bc;
      if (!per_sample && wgrad_queue_push(partial, dw, A, Bc, 1, ns, accumulate, transposed_out, b_total, b_off, s)) return;
      dim3 rg1((unsigned)((total + 63) / 64), per_sample ? batch : 1);
      hipLaunchKernelGGL(wgrad_reduce1_kernel, rg1, dim3(256), 0, s, partial, dw, A, Bc, ns, accumulate ? 1 : 0,
                         transposed_out ? 1 : 0, (size_t)ns * total, total, b_total, b_off);
      CD_HIP(hipGetLastError());
      return;
    }
  }
  a.nchunks = wgrad_chunks(dg.vox(), batch, per_sample, A, Bc, kd * kh * kw);
  int cv = (int)((dg.vox() + a.nchunks - 1) / a.nchunks);
  a.chunk_vox = (cv + 1) & ~1;
  a.partial = partial;
  const int T = kd * kh * kw;
  char cat[96];
  std::snprintf(cat, sizeof cat, "wgrad T%d C%dx%d n%ld", T, A, Bc, (long)dg.vox());
  prof::Scope scope(cat, s, 2.0 * T * A * Bc * (double)dg.vox() * batch, 4.0 * batch * ((double)dg.vox() * A + (double)dx.vox() * Bc));
  dim3 grid((unsigned)(a.nchunks * batch), (unsigned)((A / 32) * (Bc / 32)));
  // few taps per wave => small accumulator footprint => many resident waves to hide the operand-load latency
  if (T == 1) hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(64), 0, s, a);  // one wave per workgroup (nothing to split)
  else if (T <= 4) hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(64 * T), 0, s, a);
  else if (T <= 27) hipLaunchKernelGGL(wgrad_kernel<3>, grid, dim3(64 * ((T + 2) / 3)), 0, s, a);
  else hipLaunchKernelGGL(wgrad_kernel<4>, grid, dim3(64 * ((T + 3) / 4)), 0, s, a);
  CD_HIP(hipGetLastError());
  const size_t total = (size_t)A * Bc * T;
  int nslots = per_sample ? a.nchunks : a.nchunks * batch;
  const size_t sample_stride = (size_t)nslots * total;
  const int step = per_sample ? 1 : wgrad_prereduce(partial, total, &nslots, s);
  dim3 rg((unsigned)((total + 255) / 256), per_sample ? batch : 1);
  hipLaunchKernelGGL(wgrad_reduce_kernel, rg, dim3(256), 0, s, partial, dw, A, Bc, T, nslots, accumulate ? 1 : 0,
                     transposed_out ? 1 : 0, sample_stride, total, b_total, b_off, step);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Per-channel sums over (batch, voxels): bias gradients.  part: channel partials [B][units][C][2] (only the sums are used)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ part, int units, int batch, int channels,
                                                        float* __restrict__ db, int accumulate) {
  // block = 8 channels x 32 slices of the (batch x units) partial rows; fixed-order tree => deterministic
  __shared__ double sh[32][8];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const int rows = batch * units;
  double s = 0.0;
  if (c < channels)
    for (int r = sl; r < rows; r += 32) s += (double)part[((size_t)r * channels + c) * 2];
  sh[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < channels) {
    double t = 0.0;
    for (int k = 0; k < 32; ++k) t += sh[k][cl];
    db[c] = accumulate ? db[c] + (float)t : (float)t;
  }
}
void launch_bias_grad(const float* part, int units, int batch, int channels, float* db, bool accumulate, hipStream_t s) {
  hipLaunchKernelGGL(bias_grad_kernel, dim3((channels + 7) / 8), dim3(256), 0, s, part, units, batch, channels, db, accumulate ? 1 : 0);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU)(+add) backward.   forward: z = scale*h + shift (scale = rstd*gamma, shift = beta - mean*scale),
// y = act(z) + add [+ residual].  Given dy:
//   dz = dy * act'(z);  dgamma[c] = sum dz*hhat;  dbeta[c] = sum dz;  dadd[b][c] = sum_v dy
//   dh = rstd * (dhhat - mean_g(dhhat) - hhat * mean_g(dhhat*hhat)),  dhhat = dz*gamma,  hhat = (h - mean)*rstd
// Pass 1 (gn_bwd_stats_kernel): per (b, split, c): {sum dz, sum dz*hhat, sum dy, sum (h - mean)}.  Pass 2 (gn_bwd_apply_kernel).
// `stat` = saved {mean, rstd} per (b, g).
// Two more parameter gradients fall out of those sums without another pass over a tensor (round 4: ch_stats + bias_grad were two
// launches and a re-read of dh / dy per convolution bias):
//   the bias of the convolution that PRODUCED h:  sum_v dh = A sum_v dz - rstd (vox m1 + rstd m2 sum_v (h - mean))
//       (dh = A dz + Bh h + C0 with A = rstd gamma, Bh = -rstd^2 m2, C0 = rstd (-m1 + mean rstd m2): gn_bwd_finalize_kernel)
//   the bias of a convolution that adds into y (a ResnetBlock's 1x1 shortcut):  sum_v dy.
// ------------------------------------------------------------------------------------------------------------
// (on the transcendental unit, like the forward's cd_fast_silu: ~3 ulp; libm's expf and a division were ~40 vector instructions
// per element in kernels that stream two tensors)
__device__ __forceinline__ float silu_grad(float z) {
  const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
  return sg * (1.f + z * (1.f - sg));
}

__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                           const float* __restrict__ coef, const float* __restrict__ stat,
                                                           float* __restrict__ part, int channels, int64_t vox, int groups,
                                                           int silu, int nsplit) {
  __shared__ double sP[256][4];
  const int tid = threadIdx.x;
  const int split = blockIdx.x, b = blockIdx.y;
  const int cols = channels >> 2, rows = 256 / cols;
  const int64_t per = (vox + nsplit - 1) / nsplit;
  const int64_t v0 = split * per, v1 = (v0 + per < vox) ? v0 + per : vox;
  const int colid = tid % cols, row = tid / cols;
  const int c = colid * 4, cpg = channels / groups;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (row < rows) {
    f32x4 cf[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(coef + ((size_t)b * channels + c + e) * 4);
    const float mean = stat[((size_t)b * groups + c / cpg) * 2], rstd = stat[((size_t)b * groups + c / cpg) * 2 + 1];
    const size_t sb = (size_t)b * vox * channels + c;
    // four voxels per trip: eight loads in flight (one pair per trip ran at 3.2 TB/s at level 0); the order of the additions into
    // a thread's sums is that of the plain loop
    int64_t v = v0 + row;
    for (; v + 3 * rows < v1; v += 4 * rows) {
      f32x4 g4[4], h4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g4[k] = *(const f32x4*)(dy + sb + (size_t)(v + k * rows) * channels);
        h4[k] = *(const f32x4*)(h + sb + (size_t)(v + k * rows) * channels);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = cf[e][0] * h4[k][e] + cf[e][1];
          const float dz = silu ? g4[k][e] * silu_grad(z) : g4[k][e];
          s0[e] += dz;
          s1[e] += dz * (h4[k][e] - mean) * rstd;
          s2[e] += g4[k][e];
          s3[e] += h4[k][e] - mean;
        }
    }
    for (; v < v1; v += rows) {
      const f32x4 g = *(const f32x4*)(dy + sb + (size_t)v * channels);
      const f32x4 hv = *(const f32x4*)(h + sb + (size_t)v * channels);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = cf[e][0] * hv[e] + cf[e][1];
        const float dz = silu ? g[e] * silu_grad(z) : g[e];
        s0[e] += dz;
        s1[e] += dz * (hv[e] - mean) * rstd;
        s2[e] += g[e];
        s3[e] += hv[e] - mean;
      }
    }
  }
  float* dst = part + (((size_t)b * nsplit + split) * channels) * 4;
  if (cols == 8 || cols == 16) {  // (32 and 64 channels; the scratch below is 8 KB = 4 waves x 16 quads x 16 doubles)
    // the threads of one channel quad are `cols` lanes apart: fp64 xor tree inside the wave, then the four waves through LDS (fixed
    // order).  The LDS-only form below serialised 4 x (barrier, `rows` fp64 loads by `cols` threads, barrier): ~6 us of a 15 us launch
    // whose workgroups stream 250 voxels each.
    double d[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[e][0] = (double)s0[e]; d[e][1] = (double)s1[e]; d[e][2] = (double)s2[e]; d[e][3] = (double)s3[e];
    }
    for (int m = cols; m < 64; m <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) d[e][k] += __shfl_xor(d[e][k], m, 64);
    }
    const int lane = tid & 63, wave = tid >> 6;
    double* sW = &sP[0][0];  // [4 waves][cols][16]
    if (lane < cols) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) sW[(wave * cols + lane) * 16 + e * 4 + k] = d[e][k];
    }
    __syncthreads();
    for (int i = tid; i < cols * 4; i += 256) {  // (quad, element)
      const int q = i >> 2, e = i & 3;
      double a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = ((sW[(0 * cols + q) * 16 + e * 4 + k] + sW[(1 * cols + q) * 16 + e * 4 + k]) + sW[(2 * cols + q) * 16 + e * 4 + k]) + sW[(3 * cols + q) * 16 + e * 4 + k];
      *(f32x4*)(dst + (q * 4 + e) * 4) = f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
    }
    return;
  }
  for (int e = 0; e < 4; ++e) {
    sP[tid][0] = (double)s0[e]; sP[tid][1] = (double)s1[e]; sP[tid][2] = (double)s2[e]; sP[tid][3] = (double)s3[e];
    __syncthreads();
    if (tid < cols) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int r = 0; r < rows; ++r) {
        a0 += sP[r * cols + tid][0]; a1 += sP[r * cols + tid][1]; a2 += sP[r * cols + tid][2]; a3 += sP[r * cols + tid][3];
      }
      *(f32x4*)(dst + (tid * 4 + e) * 4) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
    }
    __syncthreads();
  }
}

// one block per sample: reduces the partials; writes gcoef[b][c] = {gamma*rstd, m1, m2*rstd... } for the apply pass,
// accumulates dadd[b][c]; dgamma/dbeta are reduced over the batch by a second tiny kernel.
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ gamma,
                                                              const float* __restrict__ stat, float* __restrict__ gcoef,
                                                              float* __restrict__ sums_bc, float* __restrict__ dadd, int dadd_ld,
                                                              int channels, int groups, int64_t vox) {
  __shared__ double s0[256], s1[256];
  __shared__ float m1[64], m2[64];
  const int b = blockIdx.x, c = threadIdx.x;
  const int cpg = channels / groups;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c < channels) {
    const float* p = part + ((size_t)b * nsplit * channels + c) * 4;
    for (int u = 0; u < nsplit; ++u) {
      const f32x4 v = *(const f32x4*)(p + (size_t)u * channels * 4);
      a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
    s0[c] = a0 * (double)gamma[c];   // sum dhhat
    s1[c] = a1 * (double)gamma[c];   // sum dhhat*hhat
    sums_bc[((size_t)b * channels + c) * 4] = (float)a0;      // dbeta contribution of this sample
    sums_bc[((size_t)b * channels + c) * 4 + 1] = (float)a1;  // dgamma contribution
    sums_bc[((size_t)b * channels + c) * 4 + 3] = (float)a2;  // sum_v dy: bias of a conv that adds into y (the shortcut)
    if (dadd) dadd[(size_t)b * dadd_ld + c] = (float)a2;
  }
  __syncthreads();
  if (c < groups) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < cpg; ++k) { t0 += s0[c * cpg + k]; t1 += s1[c * cpg + k]; }
    const double cnt = (double)vox * cpg;
    m1[c] = (float)(t0 / cnt);
    m2[c] = (float)(t1 / cnt);
  }
  __syncthreads();
  if (c < channels) {
    const int g = c / cpg;
    const float mean = stat[((size_t)b * groups + g) * 2], rstd = stat[((size_t)b * groups + g) * 2 + 1];
    // dh = rstd*(gamma*dz - m1 - hhat*m2) = A*dz + Bh*h + C0 with hhat = (h-mean)*rstd
    f32x4 o;
    o[0] = rstd * gamma[c];                 // * dz
    o[1] = -rstd * rstd * m2[g];            // * h
    o[2] = rstd * (-m1[g] + mean * rstd * m2[g]);
    o[3] = 0.f;
    *(f32x4*)(gcoef + ((size_t)b * channels + c) * 4) = o;
    // sum_v dh of this sample and channel (the bias gradient of the conv that produced h), from the sums in hand
    sums_bc[((size_t)b * channels + c) * 4 + 2] =
        (float)((double)o[0] * a0 - (double)rstd * ((double)m1[g] * (double)vox + (double)rstd * (double)m2[g] * a3));
  }
}

__global__ void param_grad_from_samples_kernel(const float* __restrict__ sums_bc, int batch, int channels, float* __restrict__ dgamma,
                                               float* __restrict__ dbeta, int accumulate, float* __restrict__ dbias,
                                               float* __restrict__ dsumdy) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  double g = 0.0, bt = 0.0, bs = 0.0, sy = 0.0;
  int n = 0;
  for (; n + 8 <= batch; n += 8) {  // eight samples' rows in flight (fixed summation order)
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(sums_bc + ((size_t)(n + u) * channels + c) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) { bt += (double)v[u][0]; g += (double)v[u][1]; bs += (double)v[u][2]; sy += (double)v[u][3]; }
  }
  for (; n < batch; ++n) {
    const f32x4 v = *(const f32x4*)(sums_bc + ((size_t)n * channels + c) * 4);
    bt += (double)v[0]; g += (double)v[1]; bs += (double)v[2]; sy += (double)v[3];
  }
  dgamma[c] = accumulate ? dgamma[c] + (float)g : (float)g;
  dbeta[c] = accumulate ? dbeta[c] + (float)bt : (float)bt;
  if (dbias) dbias[c] = (float)bs;
  if (dsumdy) dsumdy[c] = (float)sy;
}

// dh = gc0*dz + gc1*h + gc2 (+ dh_accum), dz = dy*act'(scale*h+shift)
// `fold` (round 4): the arithmetic of gn_bwd_finalize_kernel in the prologue of every workgroup of the sample, from the statistics
// partials (a few KB per sample) -- same operations in the same order, so the result equals the three-launch form bit for bit;
// the sample's first workgroup writes the per-sample sums for the parameter gradients.  One launch less per GroupNorm layer.
struct GnBwdFold {
  const float* part = nullptr;   // [B][nsplit][C][4] of gn_bwd_stats_kernel; null: gcoef is read from memory
  const float* gamma = nullptr;
  const float* stat = nullptr;
  float* sums_bc = nullptr;
  float* dadd = nullptr;
  int nsplit = 0, dadd_ld = 0, groups = 0;
};
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                           const float* __restrict__ coef, const float* __restrict__ gcoef,
                                                           float* __restrict__ dh, int channels, int64_t vox, int silu,
                                                           int blocks_per_sample, unsigned* __restrict__ amax_out, GnBwdFold fold) {
  __shared__ float sAmax[4];
  __shared__ double fs0[256], fs1[256];
  __shared__ float fm1[64], fm2[64];
  __shared__ __attribute__((aligned(16))) float sGc[256][4];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  if (fold.part) {
    const int cc = tid, cpg = channels / fold.groups;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (cc < channels) {
      const float* p = fold.part + ((size_t)b * fold.nsplit * channels + cc) * 4;
      int u = 0;
      for (; u + 8 <= fold.nsplit; u += 8) {  // eight partials in flight (same order of additions as the plain loop)
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *(const f32x4*)(p + (size_t)(u + k) * channels * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0 += (double)v[k][0]; a1 += (double)v[k][1]; a2 += (double)v[k][2]; a3 += (double)v[k][3]; }
      }
      for (; u < fold.nsplit; ++u) {
        const f32x4 v = *(const f32x4*)(p + (size_t)u * channels * 4);
        a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
      }
      fs0[cc] = a0 * (double)fold.gamma[cc];
      fs1[cc] = a1 * (double)fold.gamma[cc];
    }
    __syncthreads();
    if (cc < fold.groups) {
      double t0 = 0.0, t1 = 0.0;
      for (int k = 0; k < cpg; ++k) { t0 += fs0[cc * cpg + k]; t1 += fs1[cc * cpg + k]; }
      const double cnt = (double)vox * cpg;
      fm1[cc] = (float)(t0 / cnt);
      fm2[cc] = (float)(t1 / cnt);
    }
    __syncthreads();
    if (cc < channels) {
      const int g = cc / cpg;
      const float mean = fold.stat[((size_t)b * fold.groups + g) * 2], rstd = fold.stat[((size_t)b * fold.groups + g) * 2 + 1];
      f32x4 o;
      o[0] = rstd * fold.gamma[cc];
      o[1] = -rstd * rstd * fm2[g];
      o[2] = rstd * (-fm1[g] + mean * rstd * fm2[g]);
      o[3] = 0.f;
      *(f32x4*)sGc[cc] = o;
      if (blk == 0) {
        const float sdh = (float)((double)o[0] * a0 - (double)rstd * ((double)fm1[g] * (double)vox + (double)rstd * (double)fm2[g] * a3));
        *(f32x4*)(fold.sums_bc + ((size_t)b * channels + cc) * 4) = f32x4{(float)a0, (float)a1, sdh, (float)a2};
        if (fold.dadd) fold.dadd[(size_t)b * fold.dadd_ld + cc] = (float)a2;
      }
    }
    __syncthreads();
  }
  const int cols = channels >> 2, rows = 256 / cols;
  const int64_t vper = (vox + blocks_per_sample - 1) / blocks_per_sample;
  const int64_t v0 = blk * vper, v1 = (v0 + vper < vox) ? v0 + vper : vox;
  const int colid = tid % cols, row = tid / cols;
  const int c = colid * 4;
  float am = 0.f;
  if (row < rows) {
  f32x4 cf[4], gc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cf[e] = *(const f32x4*)(coef + ((size_t)b * channels + c + e) * 4);
    gc[e] = fold.part ? *(const f32x4*)sGc[c + e] : *(const f32x4*)(gcoef + ((size_t)b * channels + c + e) * 4);
  }
  const size_t sb = (size_t)b * vox * channels + c;
  int64_t v = v0 + row;
  for (; v + 3 * rows < v1; v += 4 * rows) {  // four voxels per trip: eight loads in flight
    f32x4 g4[4], h4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g4[k] = *(const f32x4*)(dy + sb + (size_t)(v + k * rows) * channels);
      h4[k] = *(const f32x4*)(h + sb + (size_t)(v + k * rows) * channels);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = cf[e][0] * h4[k][e] + cf[e][1];
        const float dz = silu ? g4[k][e] * silu_grad(z) : g4[k][e];
        o[e] = gc[e][0] * dz + gc[e][1] * h4[k][e] + gc[e][2];
      }
      *(f32x4*)(dh + sb + (size_t)(v + k * rows) * channels) = o;
      am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
  }
  for (; v < v1; v += rows) {
    const f32x4 g = *(const f32x4*)(dy + sb + (size_t)v * channels);
    const f32x4 hv = *(const f32x4*)(h + sb + (size_t)v * channels);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = cf[e][0] * hv[e] + cf[e][1];
      const float dz = silu ? g[e] * silu_grad(z) : g[e];
      o[e] = gc[e][0] * dz + gc[e][1] * hv[e] + gc[e][2];
    }
    *(f32x4*)(dh + sb + (size_t)v * channels) = o;
    am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
  }
  }
  if (amax_out) {  // max |dh| for the power-of-two rescaling of the conv gradients that consume dh (saves their own pass over it)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((tid & 63) == 0) sAmax[tid >> 6] = am;
    __syncthreads();
    if (tid == 0) atomicMax(amax_out, __float_as_uint(fmaxf(fmaxf(sAmax[0], sAmax[1]), fmaxf(sAmax[2], sAmax[3]))));
  }
}

// The whole GroupNorm backward of ONE sample in one workgroup, for the grids where the three launches above are three prologues
// around microseconds of streaming (the deepest levels of the U-Net: <= 100 KB per sample and tensor; 24 of the 44 GroupNorm layers
// of a Dataset-2 training step): statistics -> the finalize arithmetic in LDS -> apply, the second pass over dy / h served by L2.
// Same summation structure as the split form with nsplit = 1 (per-thread float partials, fp64 column sums, fp64 group means).
__global__ void __launch_bounds__(512) gn_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                           const float* __restrict__ coef, const float* __restrict__ stat,
                                                           const float* __restrict__ gamma, float* __restrict__ dh,
                                                           float* __restrict__ sums_bc, float* __restrict__ dadd, int dadd_ld,
                                                           int channels, int64_t vox, int groups, int silu,
                                                           unsigned* __restrict__ amax_out) {
  // The launch is a chain of dependent latencies around microseconds of streaming (14-20 us for 24-100 KB per tensor); round 4's
  // second session removed four of them: gamma / mean / rstd are fetched into LDS beside the first pass's loads (they were read
  // from global in the middle of the fold, twice), the column sums of all four channels of a quad are formed at once -- rows of a
  // wave by shuffles, waves through ONE LDS hop (eight barrier-separated serial sums before) -- and a sample of at most four row
  // trips per thread (the deepest level) keeps dy / h in registers for the second pass.
  __shared__ double sW[8][32][4][4];   // [wave][column][e][sum]: per-wave column sums (channels <= 128: launcher)
  __shared__ double sCol[256][4];
  __shared__ float m1[64], m2[64];
  __shared__ __attribute__((aligned(16))) float sG[256][4];
  __shared__ float sGam[256], sMean[64], sRstd[64];
  __shared__ float sAmax[8];
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cols = channels >> 2, rows = 512 / cols;   // cols in {8, 16, 32} (launcher)
  const int colid = tid % cols, row = tid / cols;
  const int c = colid * 4, cpg = channels / groups;
  const size_t sb = (size_t)b * vox * channels + c;
  if (tid < channels) sGam[tid] = gamma[tid];
  if (tid < groups) {
    sMean[tid] = stat[((size_t)b * groups + tid) * 2];
    sRstd[tid] = stat[((size_t)b * groups + tid) * 2 + 1];
  }
  f32x4 cf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(coef + ((size_t)b * channels + c + e) * 4);
  const float mean = stat[((size_t)b * groups + c / cpg) * 2], rstd = stat[((size_t)b * groups + c / cpg) * 2 + 1];
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  auto acc1 = [&](const f32x4 g, const f32x4 hv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = cf[e][0] * hv[e] + cf[e][1];
      const float dz = silu ? g[e] * silu_grad(z) : g[e];
      s0[e] += dz;
      s1[e] += dz * (hv[e] - mean) * rstd;
      s2[e] += g[e];
      s3[e] += hv[e] - mean;
    }
  };
  const bool keep = vox <= (int64_t)4 * rows;  // block-uniform: the whole sample is one trip of four rows per thread
  f32x4 kg[4], kh[4];
  if (keep) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = row + (int64_t)u * rows;
      const int64_t vc = v < vox ? v : vox - 1;
      kg[u] = *(const f32x4*)(dy + sb + (size_t)vc * channels);
      kh[u] = *(const f32x4*)(h + sb + (size_t)vc * channels);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (row + (int64_t)u * rows < vox) acc1(kg[u], kh[u]);
  } else {
    // four voxel rows per trip: eight 16-byte loads in flight per thread
    int64_t v = row;
    for (; v + 3 * rows < vox; v += 4 * rows) {
      f32x4 g[4], hv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = *(const f32x4*)(dy + sb + (size_t)(v + u * rows) * channels);
        hv[u] = *(const f32x4*)(h + sb + (size_t)(v + u * rows) * channels);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc1(g[u], hv[u]);
    }
    for (; v < vox; v += rows) acc1(*(const f32x4*)(dy + sb + (size_t)v * channels), *(const f32x4*)(h + sb + (size_t)v * channels));
  }
  // column sums in fp64: the rows a wave holds of one column sit cols lanes apart (cols < 64) -> xor shuffles; then the eight waves
  // through LDS, summed in wave order by one thread per (column, e)
  {
    double p[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[e][0] = (double)s0[e]; p[e][1] = (double)s1[e]; p[e][2] = (double)s2[e]; p[e][3] = (double)s3[e];
    }
    for (int o = 32; o >= cols; o >>= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[e][q] += __shfl_xor(p[e][q], o, 64);
    }
    if (lane < cols) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) sW[wave][lane][e][q] = p[e][q];
    }
  }
  __syncthreads();
  if (tid < channels) {
    // thread = channel cc = 4 column + e (every wave holds rows of every column: cols <= 32)
    const int col = tid >> 2, e = tid & 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int w = 0; w < 8; ++w) {
      a0 += sW[w][col][e][0]; a1 += sW[w][col][e][1]; a2 += sW[w][col][e][2]; a3 += sW[w][col][e][3];
    }
    // (rounded to float like the split form's partials)
    sCol[tid][0] = (double)(float)a0; sCol[tid][1] = (double)(float)a1;
    sCol[tid][2] = (double)(float)a2; sCol[tid][3] = (double)(float)a3;
  }
  __syncthreads();
  if (tid < groups) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < cpg; ++k) {
      t0 += sCol[tid * cpg + k][0] * (double)sGam[tid * cpg + k];
      t1 += sCol[tid * cpg + k][1] * (double)sGam[tid * cpg + k];
    }
    const double cnt = (double)vox * cpg;
    m1[tid] = (float)(t0 / cnt);
    m2[tid] = (float)(t1 / cnt);
  }
  __syncthreads();
  if (tid < channels) {
    const int cc = tid, g = cc / cpg;
    const float mn = sMean[g], rs = sRstd[g];
    const float A = rs * sGam[cc];
    sG[cc][0] = A;
    sG[cc][1] = -rs * rs * m2[g];
    sG[cc][2] = rs * (-m1[g] + mn * rs * m2[g]);
    sG[cc][3] = 0.f;
    float* o = sums_bc + ((size_t)b * channels + cc) * 4;
    o[0] = (float)sCol[cc][0];
    o[1] = (float)sCol[cc][1];
    o[2] = (float)((double)A * sCol[cc][0] - (double)rs * ((double)m1[g] * (double)vox + (double)rs * (double)m2[g] * sCol[cc][3]));
    o[3] = (float)sCol[cc][2];
    if (dadd) dadd[(size_t)b * dadd_ld + cc] = (float)sCol[cc][2];
  }
  __syncthreads();
  float am = 0.f;
  {
    f32x4 gc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gc[e] = *(const f32x4*)sG[c + e];
    auto one = [&](const f32x4 g, const f32x4 hv, int64_t v) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = cf[e][0] * hv[e] + cf[e][1];
        const float dz = silu ? g[e] * silu_grad(z) : g[e];
        o[e] = gc[e][0] * dz + gc[e][1] * hv[e] + gc[e][2];
      }
      *(f32x4*)(dh + sb + (size_t)v * channels) = o;
      am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    };
    if (keep) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (row + (int64_t)u * rows < vox) one(kg[u], kh[u], row + (int64_t)u * rows);
    } else {
      int64_t v = row;
      for (; v + 3 * rows < vox; v += 4 * rows) {
        f32x4 g[4], hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          g[u] = *(const f32x4*)(dy + sb + (size_t)(v + u * rows) * channels);
          hv[u] = *(const f32x4*)(h + sb + (size_t)(v + u * rows) * channels);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(g[u], hv[u], v + u * rows);
      }
      for (; v < vox; v += rows) one(*(const f32x4*)(dy + sb + (size_t)v * channels), *(const f32x4*)(h + sb + (size_t)v * channels), v);
    }
  }
  if (amax_out) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((tid & 63) == 0) sAmax[tid >> 6] = am;
    __syncthreads();
    if (tid == 0) {
      float m = sAmax[0];
      for (int w = 1; w < 8; ++w) m = fmaxf(m, sAmax[w]);
      atomicMax(amax_out, __float_as_uint(m));
    }
  }
}

// dgamma / dbeta (/ the conv-bias gradients that fall out of the same sums) of MANY GroupNorm layers in one launch: the training
// step queues one job per layer while it walks the network backwards and flushes the queue at the end (44 launches -> 1).
__global__ void __launch_bounds__(64) param_grad_multi_kernel(GnParamJobs jobs) {
  const GnParamJob j = jobs.job[blockIdx.y];
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= j.channels) return;
  double g = 0.0, bt = 0.0, bs = 0.0, sy = 0.0;
  for (int n = 0; n < j.batch; ++n) {
    const f32x4 v = *(const f32x4*)(j.sums_bc + ((size_t)n * j.channels + c) * 4);
    bt += (double)v[0]; g += (double)v[1]; bs += (double)v[2]; sy += (double)v[3];
  }
  j.dgamma[c] = (float)g;
  j.dbeta[c] = (float)bt;
  if (j.dbias) j.dbias[c] = (float)bs;
  if (j.dsumdy) j.dsumdy[c] = (float)sy;
}
void launch_gn_param_jobs(const GnParamJobs& jobs, hipStream_t s) {
  if (jobs.n <= 0) return;
  int cmax = 0;
  for (int i = 0; i < jobs.n; ++i) cmax = jobs.job[i].channels > cmax ? jobs.job[i].channels : cmax;
  hipLaunchKernelGGL(param_grad_multi_kernel, dim3((unsigned)((cmax + 63) / 64), (unsigned)jobs.n), dim3(64), 0, s, jobs);
  CD_HIP(hipGetLastError());
}

void launch_gn_backward(const float* dy, const float* h, const float* coef, const float* stat, const float* gamma, float* dh,
                        float* dgamma, float* dbeta, float* dadd, int dadd_ld, int batch, int channels, int64_t vox, int groups,
                        int silu, float* scratch, bool accumulate_params, hipStream_t s, float* dbias, float* dsumdy,
                        GnParamQueue* queue) {
  CD_REQUIRE(channels % 4 == 0 && channels <= 256 && groups <= 64, "group norm backward: <= 256 channels, <= 64 groups");
  const int ns = gn_nsplit_for(vox, batch);
  float* part = scratch;                                         // [B][ns][C][4]
  float* gcoef = part + (size_t)batch * ns * channels * 4;        // [B][C][4]
  float* sums_bc = gcoef + (size_t)batch * channels * 4;          // [B][C][4]
  if (queue) {  // the per-sample sums go to the caller's persistent slot and the batch reduction joins the queue
    CD_REQUIRE(!accumulate_params, "gn backward: queued parameter gradients do not accumulate");
    if (queue->jobs.n == GnParamJobs::kMax) {  // (deeper networks than the shipped ones: flush and go on)
      launch_gn_param_jobs(queue->jobs, s);
      queue->jobs.n = 0;
    }
    sums_bc = queue->next_sums;
    queue->next_sums += (size_t)batch * channels * 4;
    GnParamJob& j = queue->jobs.job[queue->jobs.n++];
    j.sums_bc = sums_bc; j.batch = batch; j.channels = channels; j.dgamma = dgamma; j.dbeta = dbeta; j.dbias = dbias; j.dsumdy = dsumdy;
  }
  prof::Scope scope("gn_backward", s, 0, 4.0 * batch * (double)vox * channels * 5);
  static const bool no_small = getenv("CD_NO_GN_BWD_SMALL") != nullptr;
  // (one workgroup streams its sample twice: 24 KB at the deepest level in ~6 us against four launches' ~20; at 188 KB -- level 1
  // with 64 channels -- it took 35 us against the split form's 27, so the bound sits between the two)
  static const size_t small_max = getenv("CD_GN_BWD_SMALL_KB") ? (size_t)atoi(getenv("CD_GN_BWD_SMALL_KB")) * 1024 : 100 * 1024;
  if (!no_small && (size_t)vox * channels * 4 <= small_max && channels <= 128 && 512 % (channels >> 2) == 0) {
    unsigned* amax_word = absmax_word_fresh(dh, s);
    hipLaunchKernelGGL(gn_bwd_small_kernel, dim3((unsigned)batch), dim3(512), 0, s, dy, h, coef, stat, gamma, dh, sums_bc, dadd, dadd_ld,
                       channels, vox, groups, silu, amax_word);
    if (!queue)
      hipLaunchKernelGGL(param_grad_from_samples_kernel, dim3((channels + 63) / 64), dim3(64), 0, s, sums_bc, batch, channels, dgamma,
                         dbeta, accumulate_params ? 1 : 0, dbias, dsumdy);
    CD_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(ns, batch), dim3(256), 0, s, dy, h, coef, stat, part, channels, vox, groups, silu, ns);
  // the finalize arithmetic runs in the apply kernel's prologue when the parameter-gradient reduction is queued behind it
  // (stand-alone calls reduce over the batch right here, between the two, and keep the three-launch form)
  static const bool no_fold = getenv("CD_NO_GN_BWD_FOLD") != nullptr;
  GnBwdFold fold;
  if (queue && !no_fold) {
    fold.part = part; fold.gamma = gamma; fold.stat = stat; fold.sums_bc = sums_bc; fold.dadd = dadd; fold.nsplit = ns;
    fold.dadd_ld = dadd_ld; fold.groups = groups;
  } else {
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(batch), dim3(256), 0, s, part, ns, gamma, stat, gcoef, sums_bc, dadd, dadd_ld,
                       channels, groups, vox);
  }
  if (!queue)
    hipLaunchKernelGGL(param_grad_from_samples_kernel, dim3((channels + 63) / 64), dim3(64), 0, s, sums_bc, batch, channels, dgamma,
                       dbeta, accumulate_params ? 1 : 0, dbias, dsumdy);
  // one round of workgroups: each repeats the finalize arithmetic in its prologue, so fewer and longer-lived ones win (same-box A/B at
  // batch 32: 7.31 -> 7.22 ms per training step from ~1024 to 256 workgroups; the forward gn_apply is neutral to the same change)
  static const int bwd_wgs = getenv("CD_GN_BWD_APPLY_WGS") ? atoi(getenv("CD_GN_BWD_APPLY_WGS")) : 256;
  int bps = gn_apply_blocks_per_sample(batch, channels, vox);
  const int bps_cap = (bwd_wgs + batch - 1) / batch;
  if (fold.part && bps > bps_cap) bps = bps_cap < 1 ? 1 : bps_cap;
  unsigned* amax_word = absmax_word_fresh(dh, s);  // zeroed; the consumer's launch_absmax_bits(dh) finds it instead of re-reading dh
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)(batch * bps)), dim3(256), 0, s, dy, h, coef, gcoef, dh, channels, vox, silu, bps,
                     amax_word, fold);
  CD_HIP(hipGetLastError());
}
// out[v][c] = a[v][aoff + c] + (b ? b[v][boff + c] : 0)   (row strides lda / ldb / C): gradient fan-in, channel slices of
// the gradient of a concatenated tensor
__global__ void add_slices_kernel(const float* __restrict__ a, int lda, int aoff, const float* __restrict__ b, int ldb, int boff,
                                  float* __restrict__ out, int channels, int64_t rows) {
  const int cols = channels >> 2;
  if (256 % cols == 0) {
    // a thread keeps its channel quad and walks rows (the general form below divides a 64-bit index twice per element: ~200 vector
    // instructions around three 16-byte memory operations); four rows per trip, their loads issued together
    const int c = (threadIdx.x % cols) * 4, rpb = 256 / cols;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    int64_t r = (int64_t)blockIdx.x * rpb + threadIdx.x / cols;
    for (; r + 3 * stride < rows; r += 4 * stride) {
      f32x4 v[4], w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = *(const f32x4*)(a + (size_t)(r + k * stride) * lda + aoff + c);
        if (b) w[k] = *(const f32x4*)(b + (size_t)(r + k * stride) * ldb + boff + c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *(f32x4*)(out + (size_t)(r + k * stride) * channels + c) = b ? v[k] + w[k] : v[k];
    }
    for (; r < rows; r += stride) {
      f32x4 v = *(const f32x4*)(a + (size_t)r * lda + aoff + c);
      if (b) v += *(const f32x4*)(b + (size_t)r * ldb + boff + c);
      *(f32x4*)(out + (size_t)r * channels + c) = v;
    }
    return;
  }
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i % cols) * 4;
    f32x4 v = *(const f32x4*)(a + (size_t)r * lda + aoff + c);
    if (b) v += *(const f32x4*)(b + (size_t)r * ldb + boff + c);
    *(f32x4*)(out + (size_t)r * channels + c) = v;
  }
}
void launch_add_slices(const float* a, int lda, int aoff, const float* b, int ldb, int boff, float* out, int channels,
                       int64_t rows, hipStream_t s) {
  int64_t blocks = (rows * (channels / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_slices_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, lda, aoff, b, ldb, boff, out, channels, rows);
  CD_HIP(hipGetLastError());
}

// Input gradient of the strided (KD,4,4) down conv for ODD phi extents.  With an even phi ring the adjoint coincides with the
// up-conv gather kernel (conv_transpose_kernel); with an odd ring the circular halo rows break its parity classes, so this
// (rare: Dataset-1 grid) case takes a plain gather:  dx[i][ci] = sum_{o,k : in(o,k) = i} sum_co dy[o][co] * w[co][ci][k].
// One thread per (input voxel, ci).
__global__ void strided_dgrad_naive_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                           int cin, int cout, int D, int H, int W, int Do, int Ho, int Wo, int KD, int SZ) {
  const int64_t vox = (int64_t)D * H * W;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= vox * cin) return;
  const int ci = (int)(idx % cin);
  const int v = (int)(idx / cin);
  const int iw = v % W, ih = (v / W) % H, iz = v / (W * H);
  float acc = 0.f;
  for (int kz = 0; kz < KD; ++kz) {
    const int tz = iz + 1 - kz;
    if (tz < 0 || tz % SZ) continue;
    const int oz = tz / SZ;
    if (oz >= Do) continue;
    for (int kw = 0; kw < 4; ++kw) {
      const int tw = iw + 1 - kw;
      if (tw < 0 || (tw & 1)) continue;
      const int ow = tw >> 1;
      if (ow >= Wo) continue;
      for (int kh = 0; kh < 4; ++kh) {
        // padded row r = 2*oh + kh covers input row (r - 1) mod H for r in [0, H+2)
        for (int rr = 0; rr < 3; ++rr) {
          const int r = ih + 1 + (rr - 1) * H;
          if (r < 0 || r > H + 1) continue;
          const int th = r - kh;
          if (th < 0 || (th & 1)) continue;
          const int oh = th >> 1;
          if (oh >= Ho) continue;
          const float* g = dy + (((size_t)b * Do + oz) * Ho + oh) * (size_t)Wo * cout + (size_t)ow * cout;
          const float* wr = w + (size_t)ci * KD * 16 + (kz * 4 + kh) * 4 + kw;
          for (int co = 0; co < cout; ++co) acc = fmaf(g[co], wr[(size_t)co * cin * KD * 16], acc);
        }
      }
    }
  }
  dx[((size_t)b * vox + v) * cin + ci] = acc;
}
void launch_strided_dgrad_naive(const float* dy, const float* w, float* dx, int batch, int cin, int cout, Dims3 din, Dims3 dout,
                                int kd, int sz, hipStream_t s) {
  const int64_t total = din.vox() * cin;
  hipLaunchKernelGGL(strided_dgrad_naive_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, s, dy, w, dx, cin, cout,
                     din.d, din.h, din.w, dout.d, dout.h, dout.w, kd, sz);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Linear attention backward helpers (forward: kernels_norm_attn.hip; reference LinearAttention.forward models.py:301-318)
// ------------------------------------------------------------------------------------------------------------
// qs[n][d] = softmax over the 32 channels of q (q = channels [0,32) of the (B, n, 96) qkv tensor); one thread per voxel
// Eight lanes per row (a quad of the 32 channels each: a wave's load is 8 rows x 128 contiguous bytes), the row's max / sum by
// three xor shuffles, four rows per lane in flight.  (Round 4: one THREAD per row read its 128 bytes as eight 16-byte loads
// 384 bytes apart from its neighbours' -- every instruction touched 64 cache lines: 2.7 TB/s at level 0.)
__global__ void __launch_bounds__(256) softmax32_kernel(const float* __restrict__ qkv, float* __restrict__ qs, int64_t rows) {
  const int q = threadIdx.x & 7;
  const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;       // 32 rows per block and trip
  const int64_t stride = (int64_t)gridDim.x * 32;
  for (int64_t r = r0; r < rows; r += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ru = r + u * stride;
      v[u] = *(const f32x4*)(qkv + (size_t)(ru < rows ? ru : rows - 1) * 96 + q * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float m = fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3]));
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float ssum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[u][e] = expf(v[u][e] - m);
        ssum += v[u][e];
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) ssum += __shfl_xor(ssum, o, 64);
      const int64_t ru = r + u * stride;
      if (ru < rows) *(f32x4*)(qs + (size_t)ru * 32 + q * 4) = v[u] * (1.f / ssum);
    }
  }
}
void launch_softmax32(const float* qkv, float* qs, int64_t rows, hipStream_t s) {
  int64_t blocks = (rows + 127) / 128;  // four rows per lane
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(softmax32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, qkv, qs, rows);
  CD_HIP(hipGetLastError());
}

// dq[n][d] = qs*(dqs - sum_d' qs*dqs)  written into channels [0,32) of dqkv (row stride 96); same lane layout
__global__ void __launch_bounds__(256) softmax32_bwd_kernel(const float* __restrict__ qs, const float* __restrict__ dqs,
                                                            float* __restrict__ dqkv, int64_t rows) {
  const int q = threadIdx.x & 7;
  const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int64_t stride = (int64_t)gridDim.x * 32;
  for (int64_t r = r0; r < rows; r += 4 * stride) {
    f32x4 av[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ru = r + u * stride;
      const size_t o = (size_t)(ru < rows ? ru : rows - 1) * 32 + q * 4;
      av[u] = *(const f32x4*)(qs + o);
      gv[u] = *(const f32x4*)(dqs + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dot = (av[u][0] * gv[u][0] + av[u][1] * gv[u][1]) + (av[u][2] * gv[u][2] + av[u][3] * gv[u][3]);
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) dot += __shfl_xor(dot, o, 64);
      const int64_t ru = r + u * stride;
      if (ru < rows) *(f32x4*)(dqkv + (size_t)ru * 96 + q * 4) = av[u] * (gv[u] - dot);
    }
  }
}
void launch_softmax32_bwd(const float* qs, const float* dqs, float* dqkv, int64_t rows, hipStream_t s) {
  int64_t blocks = (rows + 127) / 128;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(softmax32_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, qs, dqs, dqkv, rows);
  CD_HIP(hipGetLastError());
}

// dk[n][d] = ks*(dks - r[d]),  ks = exp(k - M[d])/S[d],  r[d] = dscale * sum_e dctx[d][e]*ctx[d][e];  written to dqkv channels [32,64)
// kstat[b][d] = {M, 1/S};  ctx, dctx: [b][32][32] (row d, col e), both WITHOUT the q scale.
__global__ void __launch_bounds__(256) ksoftmax_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dks,
                                                           const float* __restrict__ kstat, const float* __restrict__ ctx,
                                                           const float* __restrict__ dctx, float dscale, float* __restrict__ dqkv,
                                                           int64_t vox) {
  __shared__ float sR[32], sM[32], sI[32];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 32) {
    float r = 0.f;
    for (int e = 0; e < 32; ++e) r += dctx[((size_t)b * 32 + tid) * 32 + e] * ctx[((size_t)b * 32 + tid) * 32 + e];
    sR[tid] = r * dscale;
    sM[tid] = kstat[((size_t)b * 32 + tid) * 2];
    sI[tid] = kstat[((size_t)b * 32 + tid) * 2 + 1];
  }
  __syncthreads();
  const int64_t total = vox * 8;  // float4 items
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i >> 3;
    const int c = (int)(i & 7) * 4;
    const f32x4 kv = *(const f32x4*)(qkv + ((size_t)b * vox + n) * 96 + 32 + c);
    const f32x4 g = *(const f32x4*)(dks + ((size_t)b * vox + n) * 32 + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = expf(kv[e] - sM[c + e]) * sI[c + e] * (g[e] - sR[c + e]);
    *(f32x4*)(dqkv + ((size_t)b * vox + n) * 96 + 32 + c) = o;
  }
}
void launch_ksoftmax_bwd(const float* qkv, const float* dks, const float* kstat, const float* ctx, const float* dctx, float dscale,
                         float* dqkv, int batch, int64_t vox, hipStream_t s) {
  int64_t bx = (vox * 8 + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(ksoftmax_bwd_kernel, dim3((unsigned)bx, batch), dim3(256), 0, s, qkv, dks, kstat, ctx, dctx, dscale, dqkv, vox);
  CD_HIP(hipGetLastError());
}

// per-sample 32x32 matrix -> packed 1x1 MFMA weights:  W[co][ci] = scale * (transpose ? m[co][ci] : m[ci][co])
__global__ void pack_sample32_kernel(const float* __restrict__ m, float* __restrict__ wpk, int transpose, float scale) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
    const int e4 = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3;
    const int co = lane & 31, ci = (lane >> 5) * 16 + q * 4 + e4;
    wpk[(size_t)b * 1024 + i] = scale * (transpose ? m[((size_t)b * 32 + co) * 32 + ci] : m[((size_t)b * 32 + ci) * 32 + co]);
  }
}
void launch_pack_sample32(const float* m, float* wpk, int batch, bool transpose, float scale, hipStream_t s) {
  hipLaunchKernelGGL(pack_sample32_kernel, dim3(batch), dim3(256), 0, s, m, wpk, transpose ? 1 : 0, scale);
  CD_HIP(hipGetLastError());
}
// both images of the same matrices in one launch: wpk_plain (transpose = false) and wpk_tr (transpose = true)
__global__ void pack_sample32_pair_kernel(const float* __restrict__ m, float* __restrict__ wpk_plain, float* __restrict__ wpk_tr,
                                          float scale) {
  const int b = blockIdx.x, tr = blockIdx.y;
  float* wpk = tr ? wpk_tr : wpk_plain;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
    const int e4 = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3;
    const int co = lane & 31, ci = (lane >> 5) * 16 + q * 4 + e4;
    wpk[(size_t)b * 1024 + i] = scale * (tr ? m[((size_t)b * 32 + co) * 32 + ci] : m[((size_t)b * 32 + ci) * 32 + co]);
  }
}
void launch_pack_sample32_pair(const float* m, float* wpk_plain, float* wpk_tr, int batch, float scale, hipStream_t s) {
  hipLaunchKernelGGL(pack_sample32_pair_kernel, dim3(batch, 2), dim3(256), 0, s, m, wpk_plain, wpk_tr, scale);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Head + loss backward (forward: head_kernel; loss.py:103-104,176):
//   L = sum_b w_b sum_v (x0 - data)^2 / (mean(w) * B * per),  x0 = c_skip*x + c_out*F,  F = sum_c Wh[c]*h[v][c] + bh
//   dF = 2 w_b (x0 - data) c_out[b] / (mean(w) B per);  dh[v][c] = dF*Wh[c];  dWh[c] = sum dF*h[v][c];  dbh = sum dF
// part: [blocks][33] partial sums (32 weights + bias)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_loss_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ data,
                                                            const float* __restrict__ noise,
                                                            const float* __restrict__ scal, const float* __restrict__ h,
                                                            const float* __restrict__ wh, float* __restrict__ dh,
                                                            float* __restrict__ part, int batch, int64_t vox, int loss_type,
                                                            int objective) {
  __shared__ float sW[32];
  __shared__ float sAcc[8][33];
  __shared__ float sNorm;
  const int tid = threadIdx.x, sub = tid & 7, grp = tid >> 3;
  if (tid < 32) sW[tid] = wh[tid];
  if (tid == 0) {
    double wsum = 0.0;
    for (int b = 0; b < batch; ++b) {
      wsum += (double)objective_weight(objective, 0, scal[b * 4 + 3]);
    }
    // d loss / d x0 = sNorm * w_b * f'(d):  l2: 2 w d / (mean(w) N);  mse: 2 d / N;  l1: sign(d) / N;  huber: clamp(d, -1, 1) / N
    sNorm = loss_type == 0 ? (float)(2.0 / ((wsum / batch) * (double)batch * (double)vox))
                           : (float)((loss_type == 2 ? 2.0 : 1.0) / ((double)batch * (double)vox));
  }
  __syncthreads();
  const int64_t total = (int64_t)batch * vox;
  f32x4 aw = {0.f, 0.f, 0.f, 0.f};
  float ab = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 32 + grp; i < total; i += (int64_t)gridDim.x * 32) {
    const int b = (int)(i / vox);
    const float sg = scal[b * 4 + 3];
    const float dd = objective_residual(objective, x0[i], data[i], objective == 1 ? noise[i] : 0.f, sg);
    const float fp = loss_type == 1 ? (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f))
                                    : (loss_type == 3 ? fminf(fmaxf(dd, -1.f), 1.f) : dd);
    // d pred / d F: c_out (hybrid), -sigma (noise_pred: out = x - sigma F and pred ~ out), 1 (mean_pred)
    const float chain = objective == 0 ? scal[b * 4 + 2] : (objective == 1 ? -sg : 1.0f);
    const float dF = sNorm * objective_weight(objective, loss_type, sg) * fp * chain;
    const f32x4 hv = *(const f32x4*)(h + (size_t)i * 32 + sub * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = dF * sW[sub * 4 + e];
      aw[e] += dF * hv[e];
    }
    *(f32x4*)(dh + (size_t)i * 32 + sub * 4) = o;
    if (sub == 0) ab += dF;
  }
  // reduce over the 32 voxel groups of the block: lanes with equal `sub` hold the same channels
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) aw[e] += __shfl_xor(aw[e], o, 64);
    ab += __shfl_xor(ab, o, 64);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sAcc[wave * 2][lane * 4 + e] = aw[e];
    if (lane == 0) sAcc[wave * 2][32] = ab;
  }
  __syncthreads();
  if (tid < 33) part[(size_t)blockIdx.x * 33 + tid] = sAcc[0][tid] + sAcc[2][tid] + sAcc[4][tid] + sAcc[6][tid];
}
// 7 slices of the partial rows per column, eight loads in flight, fixed-order tree (33 threads walking 1024 rows each was
// a 237 us serial chain)
__global__ void __launch_bounds__(256) head_grad_reduce_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dwh,
                                                               float* __restrict__ dbh) {
  __shared__ double sh[7][33];
  const int c = threadIdx.x % 33, sl = threadIdx.x / 33;
  if (sl < 7) {
    double s = 0.0;
    int k = sl;
    for (; k + 7 * 7 < nblocks; k += 7 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + 7 * u) * 33 + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < nblocks; k += 7) s += (double)part[(size_t)k * 33 + c];
    sh[sl][c] = s;
  }
  __syncthreads();
  if (threadIdx.x < 33) {
    double s = 0.0;
    for (int i = 0; i < 7; ++i) s += sh[i][threadIdx.x];
    if (threadIdx.x < 32) dwh[threadIdx.x] = (float)s;
    else dbh[0] = (float)s;
  }
}
int head_bwd_blocks(int batch, int64_t vox) {
  int64_t n = ((int64_t)batch * vox + 255) / 256;
  return (int)(n > 1024 ? 1024 : (n < 1 ? 1 : n));
}
void launch_head_loss_bwd(const float* x0, const float* data, const float* noise, const float* scal, const float* h, const float* wh,
                          float* dh, float* part, float* dwh, float* dbh, int batch, int64_t vox, hipStream_t s, int loss_type,
                          int objective) {
  const int nb = head_bwd_blocks(batch, vox);
  hipLaunchKernelGGL(head_loss_bwd_kernel, dim3(nb), dim3(256), 0, s, x0, data, noise, scal, h, wh, dh, part, batch, vox, loss_type,
                     objective);
  hipLaunchKernelGGL(head_grad_reduce_kernel, dim3(1), dim3(256), 0, s, part, nb, dwh, dbh);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// init conv weight gradient (few input channels, coordinate channels synthesised as in init_conv_kernel):
//   dW[co][ci][tap] = sum_{b,v} g[b][v][co] * xin[b][in(v,tap)][ci]
// lane = output channel; a wave walks a voxel range with 27*CIN accumulators per lane; partials [b][chunk][tap*CIN+ci][32]
// ------------------------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(256) init_wgrad_kernel(InitConvArgs a, const float* __restrict__ g, float* __restrict__ part,
                                                         int chunk_vox, int nchunks) {
  __shared__ float red[4][27 * CIN][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co = lane & 31, half = lane >> 5;
  const int chunk = blockIdx.x, b = blockIdx.y, ct = blockIdx.z;
  const int D = a.dims.d, H = a.dims.h, W = a.dims.w;
  const int64_t vox = a.dims.vox();
  const float sc = a.scale_b ? a.scale_b[(size_t)b * a.scale_stride] : 1.f;
  float acc[27 * CIN];
#pragma unroll
  for (int i = 0; i < 27 * CIN; ++i) acc[i] = 0.f;
  const int v0 = chunk * chunk_vox, v1 = min((int64_t)(v0 + chunk_vox), vox);
  for (int v = v0 + wave * 2 + half; v < v1; v += 8) {
    const float gv = g[((size_t)b * vox + v) * a.cout + ct * 32 + co];
    const int w = v % W, h = (v / W) % H, z = v / (W * H);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int zz = z + kd - 1;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        int hh = h + kh - 1;
        hh = hh < 0 ? hh + H : (hh >= H ? hh - H : hh);
        hh = hh % H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = w + kw - 1;
          const bool inb = zz >= 0 && zz < D && ww >= 0 && ww < W;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            float xv = 0.f;
            if (inb) {
              if (ci < a.cx) {
                xv = a.x[(((size_t)b * a.cx + ci) * D + zz) * H * W + (size_t)hh * W + ww];
                if (ci == 0) xv *= sc;
              } else {
                const int k = ci - a.cx;
                if (a.use_rz) xv = (k == 0) ? a.r_w[ww] : (k == 1 ? a.z_d[zz] : a.phi_h[hh]);
                else xv = a.phi_h[hh];
              }
            }
            acc[((kd * 3 + kh) * 3 + kw) * CIN + ci] = fmaf(gv, xv, acc[((kd * 3 + kh) * 3 + kw) * CIN + ci]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 27 * CIN; ++i) {
    const float t = acc[i] + __shfl_xor(acc[i], 32, 64);
    if (half == 0) red[wave][i][co] = t;
  }
  __syncthreads();
  float* dst = part + ((((size_t)b * nchunks + chunk) * gridDim.z + ct) * 27 * CIN) * 32;
  for (int i = tid; i < 27 * CIN * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    dst[i] = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
  }
}
__global__ void init_wgrad_reduce_kernel(const float* __restrict__ part, int nslots, int ctiles, int cin, int cout, float* __restrict__ dw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over cout*cin*27
  if (idx >= cout * cin * 27) return;
  const int tap = idx % 27, ci = (idx / 27) % cin, co = idx / (27 * cin);
  const int ct = co / 32, c = co % 32;
  double s = 0.0;
  for (int k = 0; k < nslots; ++k) s += (double)part[((((size_t)k * ctiles + ct) * 27 * cin) + tap * cin + ci) * 32 + c];
  dw[idx] = (float)s;
}
// The init conv's weight gradient through the general 3x3x3 weight-gradient kernels: its logical input (c_in * x and the
// synthesised coordinate channels) is written once as a 32-channel channels-last tensor (zero beyond cin), the 32 x 32 x 27
// gradient is computed like any other level-0 conv's (fp16 matrix pipe) and the first cin input columns are kept.  The scalar
// kernel above needs 81 broadcast loads per voxel pair: 0.98 ms per step against ~0.15 ms this way.
__global__ void __launch_bounds__(256) init_pad_input_kernel(InitConvArgs a, float* __restrict__ out) {
  const int64_t vox = a.dims.vox();
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= vox) return;
  const int H = a.dims.h, W = a.dims.w;
  const int w = (int)(n % W), h = (int)((n / W) % H), z = (int)(n / ((int64_t)W * H));
  float sc = a.scale_b ? a.scale_b[(size_t)b * a.scale_stride] : 1.f;
  if (a.sigma_b) {
    const float tv = a.sigma_b[b], sd = a.sigma_data;
    sc = 1.f / sqrtf(tv * tv + sd * sd);
  }
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ci = 0; ci < a.cin; ++ci) {
    if (ci < a.cx) {
      v[ci] = a.x[((size_t)b * a.cx + ci) * vox + n];
      if (ci == 0) v[ci] *= sc;
    } else {
      const int k = ci - a.cx;
      v[ci] = a.use_rz ? (k == 0 ? a.r_w[w] : (k == 1 ? a.z_d[z] : a.phi_h[h])) : a.phi_h[h];
    }
  }
  f32x4* o = (f32x4*)(out + ((size_t)b * vox + n) * 32);
  o[0] = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
  for (int q = 1; q < 8; ++q) o[q] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ void init_extract_dw_kernel(const float* __restrict__ dw32, float* __restrict__ dw, int cout, int cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over cout * cin * 27
  if (idx >= cout * cin * 27) return;
  const int tap = idx % 27, ci = (idx / 27) % cin, co = idx / (27 * cin);
  dw[idx] = dw32[((size_t)co * 32 + ci) * 27 + tap];
}
size_t init_wgrad_mfma_floats(int batch, int64_t vox, int cout) {  // padded input + 32-wide gradient + slot partials
  return (size_t)batch * vox * 32 + (size_t)cout * 32 * 27 + 64 + wgrad_partial_floats(vox, batch, false, cout, 32, 27) + 128;
}
void launch_init_wgrad_mfma(const InitConvArgs& a, const float* g, float* scratch, float* dw, hipStream_t s) {
  CD_REQUIRE(a.cin <= 4 && a.cout % 32 == 0, "init conv wgrad: 1..4 input channels, 32 k output channels");
  const int64_t vox = a.dims.vox();
  float* xin = scratch;
  float* dw32 = xin + (((size_t)a.batch * vox * 32 + 63) & ~(size_t)63);
  float* part = dw32 + (((size_t)a.cout * 32 * 27 + 63) & ~(size_t)63);
  hipLaunchKernelGGL(init_pad_input_kernel, dim3((unsigned)((vox + 255) / 256), (unsigned)a.batch), dim3(256), 0, s, a, xin);
  CD_HIP(hipGetLastError());
  {
    // dw32 is read right below: this reduction cannot wait in a queue
    WgradReduceQueue* const q = g_wq;
    g_wq = nullptr;
    try {
      launch_wgrad(g, a.cout, a.dims, xin, 32, 32, 0, a.dims, 3, 3, 3, 1, 1, a.batch, false, part, dw32, false, false, s);
    } catch (...) {
      g_wq = q;
      throw;
    }
    g_wq = q;
  }
  const int total = a.cout * a.cin * 27;
  hipLaunchKernelGGL(init_extract_dw_kernel, dim3((total + 255) / 256), dim3(256), 0, s, dw32, dw, a.cout, a.cin);
  CD_HIP(hipGetLastError());
}

size_t init_wgrad_partial_floats(int batch, int64_t vox, int cin, int cout) {
  const int nchunks = (int)((vox + 1023) / 1024);
  return (size_t)batch * nchunks * (cout / 32) * 27 * cin * 32;
}
void launch_init_wgrad(const InitConvArgs& a, const float* g, float* part, float* dw, hipStream_t s) {
  const int64_t vox = a.dims.vox();
  const int nchunks = (int)((vox + 1023) / 1024);
  dim3 grid(nchunks, a.batch, a.cout / 32);
  switch (a.cin) {
    case 1: hipLaunchKernelGGL(init_wgrad_kernel<1>, grid, dim3(256), 0, s, a, g, part, 1024, nchunks); break;
    case 2: hipLaunchKernelGGL(init_wgrad_kernel<2>, grid, dim3(256), 0, s, a, g, part, 1024, nchunks); break;
    case 3: hipLaunchKernelGGL(init_wgrad_kernel<3>, grid, dim3(256), 0, s, a, g, part, 1024, nchunks); break;
    case 4: hipLaunchKernelGGL(init_wgrad_kernel<4>, grid, dim3(256), 0, s, a, g, part, 1024, nchunks); break;
    default: CD_REQUIRE(false, "init conv wgrad: 1..4 input channels");
  }
  const int total = a.cout * a.cin * 27;
  hipLaunchKernelGGL(init_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, a.batch * nchunks, a.cout / 32, a.cin,
                     a.cout, dw);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// Conditioning MLPs backward (forward: embed_kernel).  One block per sample recomputes the tiny forward, back-propagates
// demb (gradient of every ResnetBlock projection output) down to the first layers and leaves, per sample, each Linear's
// input activation and output delta in `tape`; linear_wgrad_kernel then forms dW = sum_b delta x input, db = sum_b delta.
// tape row layout per sample (floats): see EmbedTapeLayout.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf_b(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
// y = W x + b (pre-activation) for all outputs; block-cooperative (thread per output row, serial dot: tiny sizes)
__device__ void dense_pre(const float* __restrict__ w, const float* __restrict__ bias, const float* in, float* pre, int nin, int nout) {
  for (int j = threadIdx.x; j < nout; j += blockDim.x) {
    float acc = bias[j];
    for (int k = 0; k < nin; ++k) acc = fmaf(w[(size_t)j * nin + k], in[k], acc);
    pre[j] = acc;
  }
  __syncthreads();
}
// din[k] = sum_j W[j][k] * dout[j]
__device__ void dense_bwd_in(const float* __restrict__ w, const float* dout, float* din, int nin, int nout) {
  for (int k = threadIdx.x; k < nin; k += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < nout; ++j) acc = fmaf(w[(size_t)j * nin + k], dout[j], acc);
    din[k] = acc;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) embed_bwd_kernel(EmbedArgs a, const float* __restrict__ demb, float* __restrict__ tape) {
  __shared__ float p1t[128], p2t[128], p1c[256], p2c[128], cat[256], sc[256], dcat[256], tmpA[256], tmpB[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int half = a.half, q = half / 2, hid = a.cond_hidden;
  const EmbedTapeLayout L = embed_tape_layout(a.cond_size, hid, half);
  float* T = tape + (size_t)b * L.total;
  const float tv = a.time_or_sigma[b];
  float t_in = tv;
  if (a.time_kind == 0) t_in = 0.5f * logf(tv);
  else if (a.time_kind == 1) t_in = tv / sqrtf(1.f + tv * tv);
  // ---- forward recompute (pre-activations kept) ----
  if (tid == 0) { tmpA[0] = t_in; T[L.t_in] = t_in; }
  __syncthreads();
  dense_pre(a.tw1, a.tb1, tmpA, p1t, 1, q);
  for (int i = tid; i < q; i += blockDim.x) { tmpB[i] = gelu_erf_b(p1t[i]); T[L.a1t + i] = tmpB[i]; }
  __syncthreads();
  dense_pre(a.tw2, a.tb2, tmpB, p2t, q, half);
  for (int i = tid; i < half; i += blockDim.x) { tmpA[i] = gelu_erf_b(p2t[i]); T[L.a2t + i] = tmpA[i]; }
  __syncthreads();
  dense_pre(a.tw3, a.tb3, tmpA, cat, half, half);
  for (int i = tid; i < a.cond_size; i += blockDim.x) { tmpA[i] = a.cond[(size_t)b * a.cond_size + i]; T[L.cond_in + i] = tmpA[i]; }
  __syncthreads();
  dense_pre(a.cw1, a.cb1, tmpA, p1c, a.cond_size, hid);
  for (int i = tid; i < hid; i += blockDim.x) { tmpB[i] = gelu_erf_b(p1c[i]); T[L.a1c + i] = tmpB[i]; }
  __syncthreads();
  dense_pre(a.cw2, a.cb2, tmpB, p2c, hid, half);
  for (int i = tid; i < half; i += blockDim.x) { tmpA[i] = gelu_erf_b(p2c[i]); T[L.a2c + i] = tmpA[i]; }
  __syncthreads();
  dense_pre(a.cw3, a.cb3, tmpA, cat + half, half, half);
  for (int i = tid; i < 2 * half; i += blockDim.x) {
    const float v = cat[i];
    sc[i] = v / (1.f + expf(-v));
    T[L.sc + i] = sc[i];
  }
  __syncthreads();
  // ---- backward: dsc = sum_l W_l^T demb_l ; dcat = dsc * silu'(cat) ----
  // (two thread groups take the even / odd projection layers, eight weight loads in flight each: one thread per k walking all
  // ~800 rows alone was a 200 us chain of L2 round trips)
  {
    const int nk = 2 * half, grp = tid / nk, k = tid - grp * nk, ngrp = blockDim.x / nk;  // nk <= 128 => ngrp >= 2
    float acc = 0.f;
    if (grp < 2) {
      for (int l = grp; l < a.n_layers; l += 2) {
        const EmbedLayer Ly = a.layers[l];
        const float* d = demb + (size_t)b * a.emb_ld + Ly.offset;
        const float* wk = Ly.w + k;
        int j = 0;
        for (; j + 8 <= Ly.cout; j += 8) {
          float wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) wv[u] = wk[(size_t)(j + u) * nk];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = fmaf(wv[u], d[j + u], acc);
        }
        for (; j < Ly.cout; ++j) acc = fmaf(wk[(size_t)j * nk], d[j], acc);
      }
      (grp == 0 ? tmpA : tmpB)[k] = acc;
    }
    (void)ngrp;
    __syncthreads();
    if (tid < nk) {
      const float v = cat[tid];
      const float sg = 1.f / (1.f + expf(-v));
      dcat[tid] = (tmpA[tid] + tmpB[tid]) * sg * (1.f + v * (1.f - sg));
    }
  }
  __syncthreads();
  // time branch: cat[0:half] = W3 a2t + b3
  for (int i = tid; i < half; i += blockDim.x) T[L.d3t + i] = dcat[i];
  dense_bwd_in(a.tw3, dcat, tmpA, half, half);                       // d a2t
  for (int i = tid; i < half; i += blockDim.x) { tmpA[i] *= gelu_grad(p2t[i]); T[L.d2t + i] = tmpA[i]; }
  __syncthreads();
  dense_bwd_in(a.tw2, tmpA, tmpB, q, half);                          // d a1t
  for (int i = tid; i < q; i += blockDim.x) { tmpB[i] *= gelu_grad(p1t[i]); T[L.d1t + i] = tmpB[i]; }
  __syncthreads();
  // cond branch: cat[half:] = W3c a2c + b3c
  for (int i = tid; i < half; i += blockDim.x) T[L.d3c + i] = dcat[half + i];
  dense_bwd_in(a.cw3, dcat + half, tmpA, half, half);
  for (int i = tid; i < half; i += blockDim.x) { tmpA[i] *= gelu_grad(p2c[i]); T[L.d2c + i] = tmpA[i]; }
  __syncthreads();
  dense_bwd_in(a.cw2, tmpA, tmpB, hid, half);
  for (int i = tid; i < hid; i += blockDim.x) { tmpB[i] *= gelu_grad(p1c[i]); T[L.d1c + i] = tmpB[i]; }
}
size_t embed_tape_floats(int cond_size, int hidden, int half) { return (size_t)embed_tape_layout(cond_size, hidden, half).total; }
void launch_embed_bwd(const EmbedArgs& a, const float* demb, float* tape, hipStream_t s) {
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(a.batch), dim3(256), 0, s, a, demb, tape);
  CD_HIP(hipGetLastError());
}

// dW[j][k] = sum_b delta[b][j] * in[b][k],  db[j] = sum_b delta[b][j];  blockIdx.y = job
__global__ void linear_wgrad_kernel(const LinearWgradJob* __restrict__ jobs, int batch) {
  const LinearWgradJob J = jobs[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < J.nout * J.nin) {
    const int j = idx / J.nin, k = idx % J.nin;
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s = fmaf(J.delta[(size_t)b * J.delta_ld + j], J.in[(size_t)b * J.in_ld + k], s);
    J.dw[idx] = s;
  }
  if (idx < J.nout) {
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s += J.delta[(size_t)b * J.delta_ld + idx];
    J.db[idx] = s;
  }
}
void launch_linear_wgrad(const LinearWgradJob* jobs_dev, int njobs, int max_elems, int batch, hipStream_t s) {
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((max_elems + 255) / 256, njobs), dim3(256), 0, s, jobs_dev, batch);
  CD_HIP(hipGetLastError());
}

// Up-sampling to an ODD phi extent (output_padding 1 along phi): the forward's last phi row duplicates row 0 (both read
// the same wrapped inputs), so its adjoint first folds the gradient of row H-1 into row 0 and then proceeds on the even
// ring of H-1 rows.  dst: (B, D, H-1, W, C) <- src: (B, D, H, W, C)
__global__ void fold_phi_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W, int C, int batch) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)batch * D * (H - 1) * W * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % c4);
    int64_t r = i / c4;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % (H - 1));
    const int64_t bz = r / (H - 1);
    const float* s0 = src + (((size_t)bz * H + h) * W + w) * C + q * 4;
    f32x4 v = *(const f32x4*)s0;
    if (h == 0) v += *(const f32x4*)(src + (((size_t)bz * H + (H - 1)) * W + w) * C + q * 4);
    *(f32x4*)(dst + (size_t)i * 4) = v;
  }
}
void launch_fold_phi(const float* src, float* dst, int batch, Dims3 d, int C, hipStream_t s) {
  int64_t blocks = ((int64_t)batch * d.d * (d.h - 1) * d.w * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fold_phi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, d.d, d.h, d.w, C, batch);
  CD_HIP(hipGetLastError());
}

size_t gn_backward_scratch_floats(int batch, int channels, int64_t vox) {
  const int ns = gn_nsplit_for(vox, batch);
  return (size_t)batch * ns * channels * 4 + (size_t)batch * channels * 8 + 64;
}

}  // namespace cd
