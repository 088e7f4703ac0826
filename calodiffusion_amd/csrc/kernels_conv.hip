// Convolution kernels for gfx950 (MI355X): phi-periodic 3D convolutions as implicit GEMMs on the fp32 matrix cores.
//
// All dense contractions use v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 FLOP/clk/SIMD = the chip's fp32 peak):
//   M = 32 output voxels (A operand, one channels-last voxel per lane, staged through LDS with its halo),
//   N = 32 output channels (B operand = pre-packed weights, 1-KiB coalesced wave loads, L2-resident),
//   K = 2 input channels per instruction: lane half h = lane>>5 owns channels [16h, 16h+16) of a 32-channel chunk,
//       so MFMA m of a tap contracts channels (m, 16+m).
// Reference semantics: CylindricalConv / CylindricalConvTrans / Downsample / Upsample,
// calodiffusion/models/models.py:25-96, 335-369: circular padding along phi (H), zero padding along z (D) and r (W).
#include "cd_common.h"
#include "split16.h"
#include "gn_defer.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace cd {

// process-wide arithmetic of the convolutions (cd_common.h); the plan flips it to bf16x3 for the re-run of a trajectory
// whose f16x2 pass left the fp16 range
static int g_conv_precision = -1;
static thread_local int tl_conv_precision_override = -1;
void set_conv_precision_override(int p) { tl_conv_precision_override = p; }
int conv_precision() {
  if (tl_conv_precision_override >= 0) return tl_conv_precision_override;
  if (g_conv_precision < 0) {
    const char* e = getenv("CD_CONV_PRECISION");
    g_conv_precision = !e ? PREC_F16X2 : (!strcmp(e, "f32") ? PREC_F32 : (!strcmp(e, "bf16x3") ? PREC_BF16X3 : PREC_F16X2));
  }
  return g_conv_precision;
}
void set_conv_precision(int p) { g_conv_precision = p; }

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

static constexpr int LDS_VOX_PAD = 4;  // floats of padding per LDS voxel: stride 36/68/100 words => conflict-free ds_read_b128

// ------------------------------------------------------------------------------------------------------------
// weight packing (see cd_common.h for the layout)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_weights_elem(size_t idx, const float* __restrict__ w, float* __restrict__ wpk, int cout, int cin,
                                                  int taps, int transposed, int flip) {
  const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3;
  size_t rest = idx >> 10;
  const int CT = (cout + 31) / 32;
  const int ct = rest % CT;
  rest /= CT;
  const int tap = rest % taps;
  const int chunk = rest / taps;
  const int h = lane >> 5, j = lane & 31, m = q * 4 + e;
  const int ci = chunk * 32 + h * 16 + m, co = ct * 32 + j;
  float v = 0.f;
  const int st = flip ? taps - 1 - tap : tap;  // source tap (flipped for the input gradient of a stride-1 conv)
  if (co < cout && ci < cin)
    v = transposed ? w[((size_t)ci * cout + co) * taps + st] : w[((size_t)co * cin + ci) * taps + st];
  wpk[idx] = v;
}
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int cout, int cin, int taps,
                                    int transposed, size_t total, int flip) {
  size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= total) return;
  pack_weights_elem(idx, w, wpk, cout, cin, taps, transposed, flip);
}

void launch_pack_weights(const float* w_torch, float* wpk, int cout, int cin, int taps, bool transposed, hipStream_t s, bool flip) {
  CD_REQUIRE(cin % 32 == 0, "MFMA convolutions need input channels in multiples of 32");
  size_t total = packed_weight_floats(cin, cout, taps);
  hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_torch, wpk, cout, cin, taps,
                     transposed ? 1 : 0, total, flip ? 1 : 0);
  CD_HIP(hipGetLastError());
}

// init conv weights: [tap][ci][cout] so that the 32 output-channel weights of one (tap, ci) are contiguous (scalar loads)
__device__ __forceinline__ void pack_init_weights_elem(int idx, const float* __restrict__ w, float* __restrict__ wpk, int cout, int cin) {
  const int co = idx % cout;
  const int ci = (idx / cout) % cin;
  const int tap = idx / (cout * cin);
  wpk[idx] = w[((size_t)co * cin + ci) * 27 + tap];
}
__global__ void pack_init_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int cout, int cin) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cout * cin * 27) return;
  pack_init_weights_elem(idx, w, wpk, cout, cin);
}
void launch_pack_init_weights(const float* w_torch, float* wpk, int cout, int cin, hipStream_t s) {
  int total = cout * cin * 27;
  hipLaunchKernelGGL(pack_init_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_torch, wpk, cout, cin);
  CD_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// forward conv (3x3x3 stride 1, and the strided (3,4,4) down-sampling conv)
// ------------------------------------------------------------------------------------------------------------
struct ConvKArgs {
  const float* in0;
  const float* in1;
  int c0, c1;
  const float* wpk;
  const float* bias;
  float* out;
  int Din, Hin, Win, Do, Ho, Wo;
  int KD, KH, KW, SZ, SH, SW;
  int TZ, TH, nTZ, nTH;  // output tile (z, phi) extents and tile counts; tiles span the full r extent
  int IZ, IH;            // staged input tile extents (with halo)
  int cout, CTtot;
  const float* coef;     // fused GroupNorm(+SiLU) of the input, see ConvFlatArgs
  int act;
};

template <int VT, int CT>
__global__ void __launch_bounds__(512) conv_mfma_kernel(ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  int bid = blockIdx.x;
  const int thi = bid % a.nTH;
  bid /= a.nTH;
  const int tzi = bid % a.nTZ;
  const int b = bid / a.nTZ;
  const int ct0 = blockIdx.y * CT;
  const int oz0 = tzi * a.TZ, oh0 = thi * a.TH;
  const int tileVox = a.IZ * a.IH * a.Win;
  const int ZERO = tileVox * 36;
  const int half = lane >> 5, col = lane & 31;
  if (tid < 36) lds[ZERO + tid] = 0.f;

  int abase[VT], ooff[VT];
  unsigned wmask[VT];
  bool any_valid = false;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int v = (wave * VT + vt) * 32 + col;
    const int ow = v % a.Wo;
    const int t = v / a.Wo;
    const int oh = t % a.TH, oz = t / a.TH;
    const bool valid = (oz < a.TZ) && (oz0 + oz < a.Do) && (oh0 + oh < a.Ho);
    abase[vt] = ((oz * a.SZ * a.IH + oh * a.SH) * a.Win + ow * a.SW - 1) * 36 + half * 16;
    unsigned m = 0;
    for (int kw = 0; kw < a.KW; ++kw) {
      const int iw = ow * a.SW + kw - 1;
      if (valid && iw >= 0 && iw < a.Win) m |= 1u << kw;
    }
    wmask[vt] = m;
    ooff[vt] = valid ? (((oz0 + oz) * a.Ho + oh0 + oh) * a.Wo + ow) * a.cout : -1;
    any_valid |= valid;
  }
  const bool wave_active = __any(any_valid);

  f32x16 acc[VT][CT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[vt][ct][r] = 0.f;

  const int nchunk = (a.c0 + a.c1) >> 5;
  const int T = a.KD * a.KH * a.KW;
  const int gz0 = oz0 * a.SZ - 1, gh0 = oh0 * a.SH - 1;
  const int items = tileVox * 8;
  const size_t in_vox = (size_t)a.Din * a.Hin * a.Win;

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const float* src;
    int ldc, coff;
    if (chunk * 32 < a.c0) {
      src = a.in0; ldc = a.c0; coff = chunk * 32;
    } else {
      src = a.in1; ldc = a.c1; coff = chunk * 32 - a.c0;
    }
    src += (size_t)b * in_vox * ldc + coff;
    __syncthreads();  // all reads of the previous chunk's tile are done
    for (int idx = tid; idx < items; idx += nthreads) {
      const int q = idx & 7, vox = idx >> 3;
      const int iw = vox % a.Win;
      const int r = vox / a.Win;
      const int ih = r % a.IH, iz = r / a.IH;
      const int gz = gz0 + iz;
      int gh = (gh0 + ih) % a.Hin;
      if (gh < 0) gh += a.Hin;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (gz >= 0 && gz < a.Din) {
        val = *(const f32x4*)(src + ((size_t)(gz * a.Hin + gh) * a.Win + iw) * ldc + q * 4);
        if (a.coef) {
          const float* cfp = a.coef + ((size_t)b * (a.c0 + a.c1) + chunk * 32 + q * 4) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 cf = *(const f32x4*)(cfp + e * 4);
            float t = cf[0] * val[e] + cf[1];
            if (a.act) t = cd_fast_silu(t);
            val[e] = t + cf[2];
          }
        }
      }
      *(f32x4*)(lds + vox * 36 + q * 4) = val;
    }
    __syncthreads();
    if (!wave_active) continue;

    const f32x4* wq = (const f32x4*)a.wpk + (size_t)chunk * T * a.CTtot * 256 + lane;
    for (int kd = 0; kd < a.KD; ++kd) {
      for (int kh = 0; kh < a.KH; ++kh) {
        const int rowoff = (kd * a.IH + kh) * a.Win * 36;
        for (int kw = 0; kw < a.KW; ++kw) {
          const int tap = (kd * a.KH + kh) * a.KW + kw;
          f32x4 bw[CT][4];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) bw[ct][q] = wq[((size_t)tap * a.CTtot + ct0 + ct) * 256 + q * 64];
#pragma unroll
          for (int vt = 0; vt < VT; ++vt) {
            const int off = ((wmask[vt] >> kw) & 1u) ? abase[vt] + rowoff + kw * 36 : ZERO + half * 16;
            f32x4 av[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = *(const f32x4*)(lds + off + q * 4);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[vt][ct] = MFMA32(av[q][e], bw[ct][q][e], acc[vt][ct]);
          }
        }
      }
    }
  }

  // epilogue: C/D layout of the 32x32 tile: column (output channel) = lane&31, row (voxel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* outb = a.out + (size_t)b * a.Do * a.Ho * a.Wo * a.cout;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int off = __shfl(ooff[vt], row, 64);
      if (off >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int co = (ct0 + ct) * 32 + col;
          const float bv = a.bias ? a.bias[co] : 0.f;
          outb[off + co] = acc[vt][ct][r] + bv;
        }
      }
    }
  }
}

namespace {
// One-time on-device selection among candidate tilings of one conv geometry (all candidates give bit-identical
// results: the per-output summation order does not depend on the tiling).  Never runs during stream capture.
std::map<std::string, int>& tune_cache() {
  static std::map<std::string, int> c;
  return c;
}
template <typename F>
int autotune(const std::string& key, int ncand, F&& run, hipStream_t s) {
  auto it = tune_cache().find(key);
  if (it != tune_cache().end()) return it->second;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (getenv("CD_NO_AUTOTUNE") || hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone || prof::enabled())
    return -1;  // caller falls back to its heuristic (not cached)
  hipEvent_t e0, e1;
  CD_HIP(hipEventCreate(&e0));
  CD_HIP(hipEventCreate(&e1));
  int best = 0;
  float best_ms = 1e30f;
  for (int i = 0; i < ncand; ++i) {
    run(i);  // warm-up (also sets function attributes)
    CD_HIP(hipEventRecord(e0, s));
    run(i);
    run(i);
    CD_HIP(hipEventRecord(e1, s));
    CD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    CD_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (getenv("CD_TUNE_VERBOSE") && atoi(getenv("CD_TUNE_VERBOSE")) > 1) std::fprintf(stderr, "[calodiff autotune]   %s cand %d: %.1f us\n", key.c_str(), i, ms * 500.f);
    if (ms < best_ms) { best_ms = ms; best = i; }
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  tune_cache()[key] = best;
  if (getenv("CD_TUNE_VERBOSE")) std::fprintf(stderr, "[calodiff autotune] %s -> candidate %d (%.1f us)\n", key.c_str(), best, best_ms * 500.f);
  return best;
}

struct ConvTile {
  int TZ, TH, NW, VT;
  size_t lds;
};

// Candidate output tiles / wave layouts, ranked by an estimate of whole-chip MFMA time (tile quantisation in 32-voxel
// MFMA rows, SIMD balance, tail rounds over 256 CUs) subject to the 160 KiB LDS per CU; the best few are timed on the
// device once per geometry (autotune above), the top-ranked one is the fallback when timing is not possible.
std::vector<ConvTile> conv_tile_candidates(const ConvGeom& g, int batch, int CT, int keep, int vox_bytes = 144) {
  std::vector<std::pair<double, ConvTile>> all;
  const int max_vt = 8 / CT;
  for (int TZ = 1; TZ <= g.out.d && TZ <= 12; ++TZ) {
    for (int nth = 1; nth <= g.out.h; ++nth) {
      const int TH = (g.out.h + nth - 1) / nth;
      if (nth > 1 && (g.out.h + nth - 2) / (nth - 1) == TH) continue;  // same TH as previous nth
      const int IZ = (TZ - 1) * g.sz + g.kd, IH = (TH - 1) * g.sh + g.kh;
      const size_t lds = ((size_t)IZ * IH * g.in.w + 1) * vox_bytes;
      if (lds > 150 * 1024) continue;
      const int tiles = (TZ * TH * g.out.w + 31) / 32;
      for (int NW = 1; NW <= 8; ++NW) {
        const int VT = (tiles + NW - 1) / NW;
        if (VT > max_vt || VT < 1) continue;
        if (NW > 1 && (tiles + NW - 2) / (NW - 1) == VT) continue;  // a smaller NW already covers it with the same VT
        const int nTZ = (g.out.d + TZ - 1) / TZ;
        const long nblocks = (long)batch * nTZ * nth;
        int bpc = (int)(160 * 1024 / lds);
        bpc = bpc < 1 ? 1 : bpc;
        while (bpc > 1 && bpc * NW > 16) --bpc;
        const long per_cu = (nblocks + 255) / 256;
        const long rounds = (per_cu + bpc - 1) / bpc;
        const int resident = (int)(per_cu < bpc ? per_cu : bpc);
        double per_round = (double)((resident * NW + 3) / 4) * VT;
        const double useful = (double)(g.out.d * g.out.h * g.out.w) / ((double)nTZ * nth * NW * VT * 32);
        const double cost = rounds * per_round / (useful > 0 ? 1.0 : 1.0) + 0.02 * (double)IZ * IH * g.in.w / 32.0 * rounds;
        all.push_back({cost, ConvTile{TZ, TH, NW, VT, lds}});
      }
    }
  }
  CD_REQUIRE(!all.empty(), "no convolution tiling fits in LDS (grid too wide in r?)");
  std::stable_sort(all.begin(), all.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  std::vector<ConvTile> out;
  for (auto& c : all) {
    out.push_back(c.second);
    if ((int)out.size() >= keep) break;
  }
  return out;
}

template <int VT, int CT>
void launch_conv_inst(const ConvKArgs& a, dim3 grid, int threads, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_mfma_kernel<VT, CT>), grid, dim3(threads), lds, s, a);
  CD_HIP(hipGetLastError());
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// 3x3x3 stride-1 conv, "flat range" variant (the hot kernel: 92 % of the model's FLOPs).
//
// A workgroup owns R = 32*NT consecutive voxels of ONE sample in flattened (z, phi, r) order, so every MFMA row tile is
// full whatever the grid extents are (45x16x9 has 144-voxel planes = 4.5 tiles).  It stages the z-planes that range
// touches (+1 halo plane each side, zero-filled outside the tensor) as WHOLE planes: phi wraps by index arithmetic,
// r edges are predicated to a zero slot, so no halo rows/columns are stored.  Input channels stream through LDS in
// 16-channel sub-chunks (64 B per voxel, XOR-swizzled 16-B slots => conflict-free ds_read_b128 without padding):
// ~46 KiB for R = 256 on Dataset-2, i.e. three workgroups per CU whose staging and MFMA phases overlap.
// The 27 taps are fully unrolled: weight fragments (1-KiB wave loads, L1/L2 resident) and LDS fragments of tap t+1
// are in flight while the 8*VT*CT MFMAs of tap t issue.
// Sub-chunk k' of a 32-channel chunk = channels [8k', 8k'+8) (lane half 0) and [16+8k', 16+8k'+8) (lane half 1), so the
// packed weight layout of cd_common.h is used unchanged (fragments q = 2k', 2k'+1).
// ------------------------------------------------------------------------------------------------------------
struct ConvFlatArgs {
  const float* in0;
  const float* in1;
  int c0, c1;
  const float* wpk;
  const float* bias;
  float* out;
  int D, H, W;     // input extents
  int Do, Ho, Wo;  // output extents (== input for stride 1)
  int R;       // output voxels per workgroup (multiple of 32)
  int P;       // plane capacity of the LDS tile
  int cout, CTtot;
  int dbg;     // timing experiments only (CD_FLAT_DBG): 1 = skip staging, 2 = skip the MFMA taps
  // fused GroupNorm: `coef` = per-(sample, input channel) {scale, shift, add, -} applied (with SiLU if `act`) while the
  // input is staged; `ch_part` = per-(sample, workgroup, output channel) {sum, sum of squares} of this conv's output.
  const float* coef;
  int act;
  float* ch_part;
  int* status = nullptr;  // f16x2 only: bit 0 <- a staged value exceeded the fp16 range
  GnDefer defer;          // split-16 kernels: fold the input normalisation in the prologue (table at lds + coef_lds_off)
  int coef_lds_off = 0;
  const unsigned* in_absmax = nullptr;  // f16x2: power-of-two input rescaling (ConvFusion::in_absmax)
  const float* add_src = nullptr;       // split-16 kernels: out = conv + add_src (ConvFusion::add_src)
};

template <int VT, int CT>
__global__ void __launch_bounds__(512, (VT * CT <= 2 ? 3 : 2)) conv3_flat_kernel(ConvFlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int b = blockIdx.y;
  const int ct0 = blockIdx.z * CT;
  const int HW = a.H * a.W;
  const int vox = a.D * HW;
  const int v0 = blockIdx.x * a.R;
  const int vend = min(v0 + a.R, vox);
  const int zA = v0 / HW - 1;
  const int zB = (vend - 1) / HW + 1;
  const int nstage = (zB - zA + 1) * HW;   // voxels staged per sub-chunk
  const int NZ = a.P * HW;                 // index of the all-zero voxel
  const int half = lane >> 5, col = lane & 31;
  if (tid < 16) lds[NZ * 16 + tid] = 0.f;

  // per-lane geometry of its voxel in each of the wave's VT row tiles
  int nb[VT], rowm[VT], rowp[VT], ooff[VT];
  unsigned wmask[VT];
  bool any_valid = false;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int v = v0 + (wave * VT + vt) * 32 + col;
    const bool valid = v < vend;
    const int vv = valid ? v : v0;
    const int r = vv % HW;
    const int h = r / a.W, w = r - h * a.W;
    nb[vt] = vv - zA * HW;
    rowm[vt] = (h == 0 ? a.H - 1 : -1) * a.W;
    rowp[vt] = (h == a.H - 1 ? -(a.H - 1) : 1) * a.W;
    unsigned m = 0;
    if (valid) m = (w > 0 ? 1u : 0u) | 2u | (w + 1 < a.W ? 4u : 0u);
    wmask[vt] = m;
    ooff[vt] = valid ? v * a.cout : -1;
    any_valid |= valid;
  }
  const bool wave_active = __any(any_valid);

  f32x16 acc[VT][CT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[vt][ct][r] = 0.f;

  const int nsub = (a.c0 + a.c1) >> 4;
  const int gbase = zA * HW;  // global voxel index of LDS voxel 0
  const int nslots = nstage * 4;

  // LDS image: voxel n = 64 B = two 32-B pairs; pair (lane half) hp sits at ((hp ^ (n>>2)) & 1): 2-way conflicts at most
  // on the ds_read_b128 fragment reads, no padding.  A lane's two fragments are adjacent (immediate offset +16 B).
  auto frag_addr = [&](int n) -> const float* { return lds + n * 16 + ((half ^ (n >> 2)) & 1) * 8; };
  auto tap_voxel = [&](int vt, int tap) -> int {
    const int dz = tap / 9 - 1, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
    const int n = nb[vt] + dz * HW + (dh < 0 ? rowm[vt] : (dh > 0 ? rowp[vt] : 0)) + dw;
    return ((wmask[vt] >> (dw + 1)) & 1u) ? n : NZ;
  };

  for (int sc = 0; sc < nsub; ++sc) {
    const int chunk = sc >> 1, kq = sc & 1;
    const float* src;
    int ldc, coff;
    if (chunk * 32 < a.c0) {
      src = a.in0; ldc = a.c0; coff = chunk * 32;
    } else {
      src = a.in1; ldc = a.c1; coff = chunk * 32 - a.c0;
    }
    src += (size_t)b * vox * ldc + coff + kq * 8;
    // this thread always stages the same 4 channels of a sub-chunk (slot index mod 4 is tid mod 4)
    f32x4 cf[4];
    if (a.coef) {
      const int c = chunk * 32 + kq * 8 + ((tid & 3) >> 1) * 16 + (tid & 1) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(a.coef + ((size_t)b * (a.c0 + a.c1) + c + e) * 4);
    }
    __syncthreads();
    // stage: 4 independent 16-B loads in flight per thread before the first LDS write
    for (int s0 = tid; s0 < ((a.dbg & 1) ? 0 : nslots); s0 += 4 * nthreads) {
      f32x4 val[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sidx = s0 + k * nthreads;
        const int n = sidx >> 2, p = sidx & 3;
        const int g = gbase + n;
        val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sidx < nslots && g >= 0 && g < vox) {
          val[k] = *(const f32x4*)(src + (size_t)g * ldc + (p >> 1) * 16 + (p & 1) * 4);
          if (a.coef) {  // zero padding applies to the NORMALISED activation, so only in-range voxels are transformed
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = cf[e][0] * val[k][e] + cf[e][1];
              if (a.act) t = cd_fast_silu(t);
              val[k][e] = t + cf[e][2];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sidx = s0 + k * nthreads;
        const int n = sidx >> 2, p = sidx & 3;
        if (sidx < nslots) *(f32x4*)(lds + n * 16 + ((((p >> 1) ^ (n >> 2)) & 1) * 2 + (p & 1)) * 4) = val[k];
      }
    }
    __syncthreads();
    if (!wave_active || (a.dbg & 2)) continue;

    // keep the per-tap address arithmetic inside this loop (hoisting 27*VT addresses costs ~100 VGPRs)
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) asm volatile("" : "+v"(nb[vt]));

    const f32x4* wq = (const f32x4*)a.wpk + ((size_t)chunk * 27 * a.CTtot + ct0) * 256 + kq * 128 + lane;
    // software pipeline over the 27 taps: fragments of tap t+1 are requested before the MFMAs of tap t issue
    f32x4 bw[2][CT][2], av[2][VT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      bw[0][ct][0] = wq[(size_t)ct * 256];
      bw[0][ct][1] = wq[(size_t)ct * 256 + 64];
    }
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      const float* p = frag_addr(tap_voxel(vt, 0));
      av[0][vt][0] = *(const f32x4*)p;
      av[0][vt][1] = *(const f32x4*)(p + 4);
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int cur = tap & 1, nxt = cur ^ 1;
      if (tap + 1 < 27) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          bw[nxt][ct][0] = wq[((size_t)(tap + 1) * a.CTtot + ct) * 256];
          bw[nxt][ct][1] = wq[((size_t)(tap + 1) * a.CTtot + ct) * 256 + 64];
        }
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
          const float* p = frag_addr(tap_voxel(vt, tap + 1));
          av[nxt][vt][0] = *(const f32x4*)p;
          av[nxt][vt][1] = *(const f32x4*)(p + 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of this tap's MFMAs (hipcc otherwise sinks them)
#pragma unroll
      for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[vt][ct] = MFMA32(av[cur][vt][0][e], bw[cur][ct][0][e], acc[vt][ct]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[vt][ct] = MFMA32(av[cur][vt][1][e], bw[cur][ct][1][e], acc[vt][ct]);
        }
    }
  }

  float* outb = a.out + (size_t)b * vox * a.cout;
  float bv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bv[ct] = a.bias ? a.bias[(ct0 + ct) * 32 + col] : 0.f;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int off = __shfl(ooff[vt], row, 64);
      if (off >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) outb[off + (ct0 + ct) * 32 + col] = acc[vt][ct][r] + bv[ct];
      }
    }
  }
  if (a.ch_part) {
    // per-channel {sum, sum of squares} of this workgroup's outputs, reduced in a fixed order (deterministic)
    float s1[CT], s2[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) s1[ct] = s2[ct] = 0.f;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = __shfl(ooff[vt], row, 64) >= 0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float v = ok ? acc[vt][ct][r] + bv[ct] : 0.f;
          s1[ct] += v;
          s2[ct] += v * v;
        }
      }
    __syncthreads();  // every wave is done with the LDS tile
    const int nw = nthreads >> 6;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float t1 = s1[ct] + __shfl_xor(s1[ct], 32, 64), t2 = s2[ct] + __shfl_xor(s2[ct], 32, 64);
      if (half == 0) {
        lds[((wave * CT + ct) * 32 + col) * 2] = t1;
        lds[((wave * CT + ct) * 32 + col) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    for (int i = tid; i < CT * 32; i += nthreads) {
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < nw; ++w) {
        t1 += lds[((w * CT * 32) + i) * 2];
        t2 += lds[((w * CT * 32) + i) * 2 + 1];
      }
      float* dst = a.ch_part + (((size_t)b * gridDim.x + blockIdx.x) * a.cout + ct0 * 32 + i) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

namespace {
template <int VT, int CT>
void launch_flat_inst(const ConvFlatArgs& a, dim3 grid, int threads, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv3_flat_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3_flat_kernel<VT, CT>), grid, dim3(threads), lds, s, a);
  CD_HIP(hipGetLastError());
}

struct FlatTile {
  int NT = 0, VT = 0;
  size_t lds = 0;
};

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// 3x3x3 stride-1 conv on the bf16 matrix pipe with fp32-grade accuracy ("bf16x3").
//
// gfx950's f32-input MFMA runs at 1/16 of the bf16 rate.  Every fp32 operand is therefore split exactly into three bf16
// terms, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 significant bits), and the
// product is formed from the six term pairs whose magnitude is >= 2^-16 of the leading one:
//     x*w ~= x1*w1 + (x1*w2 + x2*w1) + (x1*w3 + x2*w2 + x3*w1)          (dropped terms <= 2^-24 relative)
// Each bf16 x bf16 product is exact in fp32 and accumulation is fp32 inside v_mfma_f32_32x32x16_bf16, so the result has
// fp32 rounding-level error (measured: ~2x the error of an fp32 FMA chain, 1e-6 relative on K = 864), at 6/16 of the
// matrix-pipe time of the f32 MFMA.  Weights are split once at pack time; activations are split while they are staged
// into LDS (after the optional fused GroupNorm+SiLU), 96 B per voxel per 16-channel sub-chunk.
// Same flat-range tiling, LDS plane image, software pipeline, and fused statistics epilogue as conv3_flat_kernel.
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
// exact three-way split of 4 floats -> three 8-byte groups of 4 bf16
__device__ __forceinline__ void split3(const f32x4 x, u32x2& t1, u32x2& t2, u32x2& t3) {
  f32x4 r = x;
  t1 = u32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
  r[0] -= __uint_as_float(t1[0] << 16); r[1] -= __uint_as_float(t1[0] & 0xffff0000u);
  r[2] -= __uint_as_float(t1[1] << 16); r[3] -= __uint_as_float(t1[1] & 0xffff0000u);
  t2 = u32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
  r[0] -= __uint_as_float(t2[0] << 16); r[1] -= __uint_as_float(t2[0] & 0xffff0000u);
  r[2] -= __uint_as_float(t2[1] << 16); r[3] -= __uint_as_float(t2[1] & 0xffff0000u);
  t3 = u32x2{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
}

// packed bf16x3 weights: [sub-chunk = ci/16][tap][ct][term][lane = h*32+j][8 bf16] = W_term[co = ct*32+j][ci = sc*16+8h+0..7]
__device__ __forceinline__ void pack_weights_bf16x3_elem(size_t idx, const float* __restrict__ w, u32x4* __restrict__ wpk, int cout,
                                                         int cin, int taps, int transposed, int flip) {
  const int lane = idx & 63;
  size_t rest = idx >> 6;
  const int CT = (cout + 31) / 32;
  const int ct = rest % CT;
  rest /= CT;
  const int tap = rest % taps;
  const int sc = rest / taps;
  const int h = lane >> 5, j = lane & 31;
  const int co = ct * 32 + j;
  f32x4 v[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = sc * 16 + h * 8 + e;
    const int st = flip ? taps - 1 - tap : tap;
    const size_t src = transposed ? ((size_t)ci * cout + co) * taps + st : ((size_t)co * cin + ci) * taps + st;
    v[e >> 2][e & 3] = (co < cout && ci < cin) ? w[src] : 0.f;
  }
  u32x2 a1, a2, a3, b1, b2, b3;
  split3(v[0], a1, a2, a3);
  split3(v[1], b1, b2, b3);
  u32x4* dst = wpk + (((size_t)(sc * taps + tap) * CT + ct) * 3) * 64 + lane;
  dst[0] = u32x4{a1[0], a1[1], b1[0], b1[1]};
  dst[64] = u32x4{a2[0], a2[1], b2[0], b2[1]};
  dst[128] = u32x4{a3[0], a3[1], b3[0], b3[1]};
}
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wpk, int cout, int cin, int taps,
                                           size_t total, int transposed, int flip) {
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;  // one thread per (sc, tap, ct, lane)
  if (idx >= total) return;
  pack_weights_bf16x3_elem(idx, w, wpk, cout, cin, taps, transposed, flip);
}

// every tensor of a plan in one launch: blockIdx.y = job; raw copy, then the f32 (or init) image, then the bf16x3 image, each read
// from the caller's tensor
__global__ void __launch_bounds__(256) pack_jobs_kernel(const PackJob* __restrict__ jobs) {
  const PackJob j = jobs[blockIdx.y];
  const size_t t0 = blockIdx.x * (size_t)256 + threadIdx.x, stride = gridDim.x * (size_t)256;
  for (size_t i = t0; i < j.numel; i += stride) j.raw[i] = j.src[i];
  if (j.pk) {
    if (j.kind == 3) {
      for (size_t i = t0; i < j.n_pk; i += stride) pack_init_weights_elem((int)i, j.src, j.pk, j.cout, j.cin);
    } else {
      for (size_t i = t0; i < j.n_pk; i += stride) pack_weights_elem(i, j.src, j.pk, j.cout, j.cin, j.taps, j.kind == 2 || j.tr, j.flip);
    }
  }
  if (j.bf3)
    for (size_t i = t0; i < j.n_bf3; i += stride)
      pack_weights_bf16x3_elem(i, j.src, (u32x4*)j.bf3, j.cout, j.cin, j.taps, j.kind == 2 || j.tr, j.flip);
}
void launch_pack_jobs(const PackJob* d_jobs, int njobs, hipStream_t s) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(pack_jobs_kernel, dim3(48, (unsigned)njobs), dim3(256), 0, s, d_jobs);
  CD_HIP(hipGetLastError());
}

void launch_pack_weights_bf16x3(const float* w_torch, void* wpk, int cout, int cin, int taps, hipStream_t s, bool transposed,
                                bool flip) {
  CD_REQUIRE(cin % 16 == 0, "bf16x3 convolution needs input channels in multiples of 16");
  const size_t total = (size_t)(cin / 16) * taps * ((cout + 31) / 32) * 64;
  hipLaunchKernelGGL(pack_weights_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_torch, (u32x4*)wpk, cout,
                     cin, taps, total, transposed ? 1 : 0, flip ? 1 : 0);
  CD_HIP(hipGetLastError());
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// Geometry is a template parameter: (KD,KH,KW) taps, z stride SZ, phi/r stride SXY; padding is always (1, circular 1, 1).
// Instantiated for the 3x3x3 stride-1 conv and the (3,4,4) down-sampling conv with z stride 2 or 1.
// NTERM = 3: bf16x3 (96 B per voxel per sub-chunk); NTERM = 2: f16x2 (split16.h; 64 B + 16 B pad = 80 B, an odd number of
// 16-B slots => conflict-free ds_read_b128), two accumulators per tile folded after the K loop.
template <int VT, int CT, int KD, int KH, int KW, int SZ, int SXY, int NTERM>
__global__ void __launch_bounds__(512, (VT * CT * (NTERM == 2 ? 2 : 1) <= 2 ? 3 : 2)) conv3_flat_bf16x3_kernel(ConvFlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = (char*)lds;
  constexpr int T = KD * KH * KW;
  constexpr int VB = NTERM == 3 ? 96 : 80;   // bytes per staged voxel
  constexpr int WS = 64 * NTERM;             // u32x4 per (tap, ct) in the packed weights
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int b = blockIdx.y;
  const int ct0 = blockIdx.z * CT;
  const int HW = a.H * a.W;          // input plane
  const int vox = a.D * HW;          // input voxels per sample
  const int HWo = a.Ho * a.Wo;
  const int voxo = a.Do * HWo;       // output voxels per sample
  const int v0 = blockIdx.x * a.R;
  const int vend = min(v0 + a.R, voxo);
  // (index arithmetic by reciprocal -- (v + 0.5) / d is never within float error of an integer for v < 2^20: a run-time integer
  // division is ~40 vector instructions, and this prologue had four to ten of them in workgroups that live ~15 us)
  const float inv_hwo = 1.f / (float)HWo, inv_wo = 1.f / (float)a.Wo;
  auto fdiv = [](int x, float inv) { return (int)(((float)x + 0.5f) * inv); };
  const int zA = fdiv(v0, inv_hwo) * SZ - 1;                   // first staged input plane (may be -1: zero plane)
  const int zB = fdiv(vend - 1, inv_hwo) * SZ + KD - 2;        // last staged input plane
  const int nstage = (zB - zA + 1) * HW;
  const int NZ = a.P * HW;  // all-zero voxel
  const int half = lane >> 5, col = lane & 31;
  if (a.defer.part) gn_defer_to_lds(a.defer, b, (float*)(ldsb + a.coef_lds_off), ldsb + a.coef_lds_off + a.defer.C * 16);
  const bool normed = a.coef || a.defer.part;
  if (tid < VB / 4) ((float*)(ldsb + (size_t)NZ * VB))[tid] = 0.f;

  // per-lane geometry of its output voxel in each of the wave's VT row tiles: LDS index of the (kz=0, kh=1, kw=1) tap,
  // phi-row offsets with wrap-around for each kh, r-validity bit for each kw
  int nb[VT], rowoff[VT][KH], ooff[VT];
  unsigned wmask[VT];
  bool any_valid = false;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int v = v0 + (wave * VT + vt) * 32 + col;
    const bool valid = v < vend;
    const int vv = valid ? v : v0;
    const int oz = fdiv(vv, inv_hwo);
    const int r = vv - oz * HWo;
    const int oh = fdiv(r, inv_wo), ow = r - oh * a.Wo;
    const int ih0 = oh * SXY, iw0 = ow * SXY;
    nb[vt] = (oz * SZ - 1 - zA) * HW + ih0 * a.W + iw0;
    unsigned m = 0;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      int ih = ih0 + kh - 1;
      ih = ih < 0 ? ih + a.H : (ih >= a.H ? ih - a.H : ih);
      ih = ih >= a.H ? ih - a.H : ih;  // H == 2 with a 4-wide kernel wraps twice
      rowoff[vt][kh] = (ih - ih0) * a.W;
    }
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int iw = iw0 + kw - 1;
      if (valid && iw >= 0 && iw < a.W) m |= 1u << kw;
    }
    wmask[vt] = m;
    ooff[vt] = valid ? v * a.cout : -1;
    any_valid |= valid;
  }
  const bool wave_active = __any(any_valid);

  f32x16 acc[VT][CT], accB[NTERM == 2 ? VT : 1][NTERM == 2 ? CT : 1];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[vt][ct][r] = 0.f;
        if (NTERM == 2) accB[vt][ct][r] = 0.f;
      }

  float amax = 0.f;
  float gscale = 1.f, ginv = 1.f;
  if (NTERM == 2 && a.in_absmax) pow2_scale_for(*a.in_absmax, &gscale, &ginv);
  const int nsub = (a.c0 + a.c1) >> 4;
  const int gbase = zA * HW;
  const int nslots = nstage * 4;  // one slot = 4 channels of one voxel

  auto tap_voxel = [&](int vt, int tap) -> int {
#ifdef CD_FLAT_ABL_TAPS  // ablation (experiment builds only): what the per-tap address arithmetic costs -- WRONG results
    return nb[vt] + tap;
#endif
    const int kz = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
    const int n = nb[vt] + kz * HW + rowoff[vt][kh] + kw - 1;
    return ((wmask[vt] >> kw) & 1u) ? n : NZ;
  };

  for (int sc = 0; sc < nsub; ++sc) {
    const float* src;
    int ldc, coff;
    if (sc * 16 < a.c0) {
      src = a.in0; ldc = a.c0; coff = sc * 16;
    } else {
      src = a.in1; ldc = a.c1; coff = sc * 16 - a.c0;
    }
    const int pq = tid & 3;  // this thread always stages channel quad pq of a voxel
    src += (size_t)b * vox * ldc + coff + pq * 4;
    f32x4 cf[4];
    if (a.defer.part) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(ldsb + a.coef_lds_off + (sc * 16 + pq * 4 + e) * 16);
    } else if (a.coef) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(a.coef + ((size_t)b * (a.c0 + a.c1) + sc * 16 + pq * 4 + e) * 4);
    }
    __syncthreads();
    for (int s0 = tid; s0 < ((a.dbg & 1) ? 0 : nslots); s0 += 4 * nthreads) {
      f32x4 val[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sidx = s0 + k * nthreads;
        const int g = gbase + (sidx >> 2);
        val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sidx < nslots && g >= 0 && g < vox) {
          val[k] = *(const f32x4*)(src + (size_t)g * ldc);
          if (normed) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = cf[e][0] * val[k][e] + cf[e][1];
              if (a.act) t = cd_fast_silu(t);
              val[k][e] = t + cf[e][2];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sidx = s0 + k * nthreads;
        if (sidx < nslots) {
          char* d = ldsb + (size_t)(sidx >> 2) * VB + pq * 8;
          if (NTERM == 3) {
            u32x2 t1, t2, t3;
            split3(val[k], t1, t2, t3);
            *(u32x2*)d = t1;
            *(u32x2*)(d + 32) = t2;
            *(u32x2*)(d + 64) = t3;
          } else {
            const f32x4 vs = val[k] * gscale;
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(vs[0]), fabsf(vs[1])), fmaxf(fabsf(vs[2]), fabsf(vs[3]))));
            u32x2 t1, t2;
            split2(vs, t1, t2);
            *(u32x2*)d = t1;
            *(u32x2*)(d + 32) = t2;
          }
        }
      }
    }
    __syncthreads();
    if (!wave_active || (a.dbg & 2)) continue;

#pragma unroll
    for (int vt = 0; vt < VT; ++vt) asm volatile("" : "+v"(nb[vt]));

    const u32x4* wq = (const u32x4*)a.wpk + ((size_t)sc * T * a.CTtot + ct0) * WS + lane;
    // Register rings: weight fragments (L1/L2) are requested WD taps ahead, LDS fragments AD taps ahead.
#ifndef CD_FLAT_WD
#define CD_FLAT_WD 3
#endif
#ifndef CD_FLAT_AD
#define CD_FLAT_AD 2
#endif
    // Measured again in round 3 (same box, alternating runs): WD 1 / AD 1 -> 3 / 2 takes the strided 32->32 conv from 48.8 to 41.8 us
    // and the 128->32 conv at 23x8x4 from 46.9 to 41.3 us (one tap of cover = 3 VT CT MFMAs is less than an L2 round trip for the
    // narrow tilings), -2 % on the Dataset-2 step, -4 % on HGCal; 4 / 2 the same, 5 / 3 slower (registers).
    // (the f16x2 arm only: the three-term bf16 arm spills hundreds of registers with the deeper rings and keeps one tap of cover;
    // restricting them to the narrow f16x2 tilings as well was measured 0.8 % slower on the Dataset-2 step)
    constexpr bool DEEP = NTERM == 2;
    constexpr int WD = DEEP ? CD_FLAT_WD : 1, AD = DEEP ? CD_FLAT_AD : 1;
    u32x4 bw[WD + 1][CT][NTERM], av[AD + 1][VT][NTERM];
    auto load_w = [&](int tap) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < NTERM; ++t) bw[tap % (WD + 1)][ct][t] = wq[((size_t)tap * a.CTtot + ct) * WS + t * 64];
    };
    auto load_a = [&](int tap) {
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        const char* p = ldsb + (size_t)tap_voxel(vt, tap) * VB + half * 16;
#pragma unroll
        for (int t = 0; t < NTERM; ++t) av[tap % (AD + 1)][vt][t] = *(const u32x4*)(p + t * 32);
      }
    };
#pragma unroll
    for (int t0 = 0; t0 < WD; ++t0) load_w(t0);
#pragma unroll
    for (int t0 = 0; t0 < AD; ++t0) load_a(t0);
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      if (tap + WD < T) load_w(tap + WD);
      if (tap + AD < T) load_a(tap + AD);
      __builtin_amdgcn_sched_barrier(0);
      const int wc = tap % (WD + 1), ac = tap % (AD + 1);
#pragma unroll
      for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (NTERM == 3) {
            f32x16 c = acc[vt][ct];
            c = MFMA_BF16(av[ac][vt][NTERM - 1], bw[wc][ct][0], c);  // x3*w1
            c = MFMA_BF16(av[ac][vt][1], bw[wc][ct][1], c);          // x2*w2
            c = MFMA_BF16(av[ac][vt][0], bw[wc][ct][NTERM - 1], c);  // x1*w3
            c = MFMA_BF16(av[ac][vt][1], bw[wc][ct][0], c);          // x2*w1
            c = MFMA_BF16(av[ac][vt][0], bw[wc][ct][1], c);          // x1*w2
            c = MFMA_BF16(av[ac][vt][0], bw[wc][ct][0], c);          // x1*w1
            acc[vt][ct] = c;
          } else {
            acc[vt][ct] = MFMA_F16(av[ac][vt][0], bw[wc][ct][0], acc[vt][ct]);    // x1*w1
            accB[vt][ct] = MFMA_F16(av[ac][vt][0], bw[wc][ct][1], accB[vt][ct]);  // x1*w2'
            accB[vt][ct] = MFMA_F16(av[ac][vt][1], bw[wc][ct][0], accB[vt][ct]);  // x2'*w1
          }
        }
    }
  }

  if (NTERM == 2) {
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[vt][ct][r] = (acc[vt][ct][r] + accB[vt][ct][r] * (1.f / 2048.f)) * ginv;
  }
  if (NTERM == 2 && a.status && amax > 65504.f) atomicOr(a.status, 1);
  float* outb = a.out + (size_t)b * voxo * a.cout;
  float bv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bv[ct] = a.bias ? a.bias[(ct0 + ct) * 32 + col] : 0.f;
  if (a.add_src) {  // (the loads of all rows first: one round trip, not one per row)
    const float* addb = a.add_src + (size_t)b * voxo * a.cout;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      float ad[16][CT];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = __shfl(ooff[vt], (r & 3) + 8 * (r >> 2) + 4 * half, 64);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) ad[r][ct] = off >= 0 ? addb[off + (ct0 + ct) * 32 + col] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[vt][ct][r] += ad[r][ct];
    }
  }
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int off = __shfl(ooff[vt], row, 64);
      if (off >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) outb[off + (ct0 + ct) * 32 + col] = acc[vt][ct][r] + bv[ct];
      }
    }
  }
  if (a.ch_part) {
    float s1[CT], s2[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) s1[ct] = s2[ct] = 0.f;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = __shfl(ooff[vt], row, 64) >= 0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float v = ok ? acc[vt][ct][r] + bv[ct] : 0.f;
          s1[ct] += v;
          s2[ct] += v * v;
        }
      }
    __syncthreads();
    const int nw = nthreads >> 6;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float t1 = s1[ct] + __shfl_xor(s1[ct], 32, 64), t2 = s2[ct] + __shfl_xor(s2[ct], 32, 64);
      if (half == 0) {
        lds[((wave * CT + ct) * 32 + col) * 2] = t1;
        lds[((wave * CT + ct) * 32 + col) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    for (int i = tid; i < CT * 32; i += nthreads) {
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < nw; ++w) {
        t1 += lds[((w * CT * 32) + i) * 2];
        t2 += lds[((w * CT * 32) + i) * 2 + 1];
      }
      float* dst = a.ch_part + (((size_t)b * gridDim.x + blockIdx.x) * a.cout + ct0 * 32 + i) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Warp-specialised, persistent variant of the bf16x3 flat conv.
//
// One workgroup per CU loops over its share of (sample, voxel-range) units.  NLW loader waves stage the NEXT
// 16-channel sub-chunk (global -> fused GroupNorm/SiLU -> exact bf16 split -> LDS buffer B) on the vector ALU while NMW
// matrix waves run the taps of the CURRENT sub-chunk out of LDS buffer A on the matrix pipe; one barrier per phase swaps
// the buffers.  Staging (VALU / LDS-write bound) and MFMAs (matrix-pipe bound) therefore overlap inside a CU instead of
// alternating, and the pipeline runs seamlessly across units (the first sub-chunk of unit u+1 is staged during the last
// phase of unit u).  Each matrix wave owns one 32-voxel row tile of the unit (R = 32*NMW voxels); the per-unit channel
// statistics are handed to the loader waves through a small LDS scratch and written by them one phase later.
// ------------------------------------------------------------------------------------------------------------
template <int CT, int KD, int KH, int KW, int SZ, int SXY>
__global__ void __launch_bounds__(768) conv_flat_ws_kernel(ConvFlatArgs a, int nlw, int units_per_sample, int total_units) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int T = KD * KH * KW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nmw = (blockDim.x >> 6) - nlw;
  const bool loader = wave < nlw;
  const int ct0 = blockIdx.z * CT;
  const int HW = a.H * a.W, vox = a.D * HW;
  const int HWo = a.Ho * a.Wo, voxo = a.Do * HWo;
  const int NZ = a.P * HW;
  const size_t buf_bytes = ((size_t)NZ + 1) * 96;
  char* bufp[2] = {(char*)lds, (char*)lds + buf_bytes};
  float* scratch = (float*)((char*)lds + 2 * buf_bytes);  // [nmw][CT*32][2]
  const int half = lane >> 5, col = lane & 31;
  const int nsub = (a.c0 + a.c1) >> 4;
  if (tid < 48) ((float*)(bufp[tid / 24] + (size_t)NZ * 96))[tid % 24] = 0.f;

  // units of this workgroup: u = blockIdx.x + k*gridDim.x
  const int my_units = (total_units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nph = my_units * nsub;

  // ---- loader side ------------------------------------------------------------------------------------------
  auto stage = [&](int ph, char* dst) {
    const int u = blockIdx.x + (ph / nsub) * gridDim.x, sc = ph % nsub;
    const int b = u / units_per_sample, ux = u - b * units_per_sample;
    const int v0 = ux * a.R, vend = min(v0 + a.R, voxo);
    const int zA = (v0 / HWo) * SZ - 1, zB = ((vend - 1) / HWo) * SZ + KD - 2;
    const int nslots = (zB - zA + 1) * HW * 4, gbase = zA * HW;
    const float* src;
    int ldc, coff;
    if (sc * 16 < a.c0) {
      src = a.in0; ldc = a.c0; coff = sc * 16;
    } else {
      src = a.in1; ldc = a.c1; coff = sc * 16 - a.c0;
    }
    const int pq = tid & 3, nth = nlw * 64;
    src += (size_t)b * vox * ldc + coff + pq * 4;
    f32x4 cf[4];
    if (a.coef) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(a.coef + ((size_t)b * (a.c0 + a.c1) + sc * 16 + pq * 4 + e) * 4);
    }
    // all loads of a batch are issued before the first conversion: with only nlw waves loading, memory-level
    // parallelism (bytes in flight per CU), not issue rate, sets the staging time
    constexpr int LB = 12;
    for (int s0 = tid; s0 < nslots; s0 += LB * nth) {
      f32x4 val[LB];
#pragma unroll
      for (int k = 0; k < LB; ++k) {
        const int sidx = s0 + k * nth;
        const int g = gbase + (sidx >> 2);
        val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sidx < nslots && g >= 0 && g < vox) {
          val[k] = *(const f32x4*)(src + (size_t)g * ldc);
          if (a.coef) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = cf[e][0] * val[k][e] + cf[e][1];
              if (a.act) t = cd_fast_silu(t);
              val[k][e] = t + cf[e][2];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < LB; ++k) {
        const int sidx = s0 + k * nth;
        if (sidx < nslots) {
          u32x2 t1, t2, t3;
          split3(val[k], t1, t2, t3);
          char* d = dst + (size_t)(sidx >> 2) * 96 + pq * 8;
          *(u32x2*)d = t1;
          *(u32x2*)(d + 32) = t2;
          *(u32x2*)(d + 64) = t3;
        }
      }
    }
  };

  if (loader) stage(0, bufp[0]);
  __syncthreads();

  // ---- matrix-wave state --------------------------------------------------------------------------------------
  f32x16 acc[CT];
  int nb = 0, ooff = -1, rowoff[KH];
  unsigned wmask = 0;
  int cur_b = 0;

  for (int ph = 0; ph < nph; ++ph) {
    const int cur = ph & 1;
    const int sc = ph % nsub;
    if (loader) {
      if (ph + 1 < nph) stage(ph + 1, bufp[cur ^ 1]);
      // statistics of the unit that finished in the previous phase
      if (a.ch_part && sc == 0 && ph > 0) {
        const int u = blockIdx.x + (ph / nsub - 1) * gridDim.x;
        for (int i = tid; i < CT * 32; i += nlw * 64) {
          float t1 = 0.f, t2 = 0.f;
          for (int w = 0; w < nmw; ++w) {
            t1 += scratch[((w * CT * 32) + i) * 2];
            t2 += scratch[((w * CT * 32) + i) * 2 + 1];
          }
          float* dst = a.ch_part + ((size_t)u * a.cout + ct0 * 32 + i) * 2;  // u = b*units_per_sample + ux
          dst[0] = t1;
          dst[1] = t2;
        }
      }
    } else {
      const int mw = wave - nlw;
      if (sc == 0) {
        const int u = blockIdx.x + (ph / nsub) * gridDim.x;
        cur_b = u / units_per_sample;
        const int ux = u - cur_b * units_per_sample;
        const int v0 = ux * a.R, vend = min(v0 + a.R, voxo);
        const int zA = (v0 / HWo) * SZ - 1;
        const int v = v0 + mw * 32 + col;
        const bool valid = v < vend;
        const int vv = valid ? v : v0;
        const int oz = vv / HWo;
        const int r = vv - oz * HWo;
        const int oh = r / a.Wo, ow = r - oh * a.Wo;
        const int ih0 = oh * SXY, iw0 = ow * SXY;
        nb = (oz * SZ - 1 - zA) * HW + ih0 * a.W + iw0;
        unsigned m = 0;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          int ih = ih0 + kh - 1;
          ih = ih < 0 ? ih + a.H : (ih >= a.H ? ih - a.H : ih);
          ih = ih >= a.H ? ih - a.H : ih;
          rowoff[kh] = (ih - ih0) * a.W;
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int iw = iw0 + kw - 1;
          if (valid && iw >= 0 && iw < a.W) m |= 1u << kw;
        }
        wmask = m;
        ooff = valid ? v * a.cout : -1;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r2 = 0; r2 < 16; ++r2) acc[ct][r2] = 0.f;
      }
      if (__any(ooff >= 0)) {
        const char* ldsb = bufp[cur];
        auto tap_ptr = [&](int tap) -> const char* {
          const int kz = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
          const int n = nb + kz * HW + rowoff[kh] + kw - 1;
          return ldsb + (size_t)(((wmask >> kw) & 1u) ? n : NZ) * 96 + half * 16;
        };
        asm volatile("" : "+v"(nb));
        const u32x4* wq = (const u32x4*)a.wpk + ((size_t)sc * T * a.CTtot + ct0) * 192 + lane;
        u32x4 bw[2][CT][3], av[2][3];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int t = 0; t < 3; ++t) bw[0][ct][t] = wq[(size_t)ct * 192 + t * 64];
        {
          const char* p = tap_ptr(0);
#pragma unroll
          for (int t = 0; t < 3; ++t) av[0][t] = *(const u32x4*)(p + t * 32);
        }
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
          const int c = tap & 1, nx = c ^ 1;
          if (tap + 1 < T) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int t = 0; t < 3; ++t) bw[nx][ct][t] = wq[((size_t)(tap + 1) * a.CTtot + ct) * 192 + t * 64];
            const char* p = tap_ptr(tap + 1);
#pragma unroll
            for (int t = 0; t < 3; ++t) av[nx][t] = *(const u32x4*)(p + t * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            f32x16 cc = acc[ct];
            cc = MFMA_BF16(av[c][2], bw[c][ct][0], cc);
            cc = MFMA_BF16(av[c][1], bw[c][ct][1], cc);
            cc = MFMA_BF16(av[c][0], bw[c][ct][2], cc);
            cc = MFMA_BF16(av[c][1], bw[c][ct][0], cc);
            cc = MFMA_BF16(av[c][0], bw[c][ct][1], cc);
            cc = MFMA_BF16(av[c][0], bw[c][ct][0], cc);
            acc[ct] = cc;
          }
        }
      }
      if (sc == nsub - 1) {
        float* outb = a.out + (size_t)cur_b * voxo * a.cout;
        float bv[CT], s1[CT], s2[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          bv[ct] = a.bias ? a.bias[(ct0 + ct) * 32 + col] : 0.f;
          s1[ct] = s2[ct] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int off = __shfl(ooff, row, 64);
          if (off >= 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const float v = acc[ct][r] + bv[ct];
              outb[off + (ct0 + ct) * 32 + col] = v;
              s1[ct] += v;
              s2[ct] += v * v;
            }
          }
        }
        if (a.ch_part) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float t1 = s1[ct] + __shfl_xor(s1[ct], 32, 64), t2 = s2[ct] + __shfl_xor(s2[ct], 32, 64);
            if (half == 0) {
              scratch[((mw * CT + ct) * 32 + col) * 2] = t1;
              scratch[((mw * CT + ct) * 32 + col) * 2 + 1] = t2;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // statistics of the last unit
  if (loader && a.ch_part && nph > 0) {
    const int u = blockIdx.x + (my_units - 1) * gridDim.x;
    for (int i = tid; i < CT * 32; i += nlw * 64) {
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < nmw; ++w) {
        t1 += scratch[((w * CT * 32) + i) * 2];
        t2 += scratch[((w * CT * 32) + i) * 2 + 1];
      }
      float* dst = a.ch_part + ((size_t)u * a.cout + ct0 * 32 + i) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

namespace {
template <int CT, int KD, int KH, int KW, int SZ, int SXY>
void launch_ws_geo(const ConvFlatArgs& a, dim3 grid, int threads, size_t lds, int nlw, int ups, int total, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv_flat_ws_kernel<CT, KD, KH, KW, SZ, SXY>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_flat_ws_kernel<CT, KD, KH, KW, SZ, SXY>), grid, dim3(threads), lds, s, a, nlw, ups, total);
  CD_HIP(hipGetLastError());
}
template <int CT>
void launch_ws_inst(const ConvFlatArgs& a, dim3 grid, int threads, size_t lds, int nlw, int ups, int total, hipStream_t s, int geo) {
  if (geo == 0) launch_ws_geo<CT, 3, 3, 3, 1, 1>(a, grid, threads, lds, nlw, ups, total, s);
  else if (geo == 1) launch_ws_geo<CT, 3, 4, 4, 2, 2>(a, grid, threads, lds, nlw, ups, total, s);
  else if (geo == 2) launch_ws_geo<CT, 3, 4, 4, 1, 2>(a, grid, threads, lds, nlw, ups, total, s);
  else launch_ws_geo<CT, 4, 4, 4, 2, 2>(a, grid, threads, lds, nlw, ups, total, s);
}
}  // namespace

namespace {
template <int VT, int CT, int KD, int KH, int KW, int SZ, int SXY, int NTERM>
void launch_flat3_geo(const ConvFlatArgs& a, dim3 grid, int threads, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv3_flat_bf16x3_kernel<VT, CT, KD, KH, KW, SZ, SXY, NTERM>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3_flat_bf16x3_kernel<VT, CT, KD, KH, KW, SZ, SXY, NTERM>), grid, dim3(threads), lds, s, a);
  CD_HIP(hipGetLastError());
}
// geo: 0 = 3x3x3 stride 1, 1 = (3,4,4) stride (2,2,2), 2 = (3,4,4) stride (1,2,2), 3 = (4,4,4) stride (2,2,2)
template <int VT, int CT, int NTERM>
void launch_flat3_inst(const ConvFlatArgs& a, dim3 grid, int threads, size_t lds, hipStream_t s, int geo = 0) {
  if (geo == 0) launch_flat3_geo<VT, CT, 3, 3, 3, 1, 1, NTERM>(a, grid, threads, lds, s);
  else if (geo == 1) launch_flat3_geo<VT, CT, 3, 4, 4, 2, 2, NTERM>(a, grid, threads, lds, s);
  else if (geo == 2) launch_flat3_geo<VT, CT, 3, 4, 4, 1, 2, NTERM>(a, grid, threads, lds, s);
  else launch_flat3_geo<VT, CT, 4, 4, 4, 2, 2, NTERM>(a, grid, threads, lds, s);
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// (TZ, TH) halo-tiled bf16x3 conv: same split-bf16 arithmetic as conv3_flat_bf16x3_kernel for grids whose z-planes are
// too wide for the whole-plane LDS image (Dataset-3 level 0: 50x18 = 900-voxel planes).  Tile geometry as in
// conv_mfma_kernel: output tile (TZ, TH, full r), staged input tile with phi halo rows (wrapped) and z halo planes
// (zero-filled); 96 B per voxel per 16-channel sub-chunk; runtime tap loop.
// ------------------------------------------------------------------------------------------------------------
struct ConvTiled3Args {
  ConvKArgs k;       // geometry / tiling / pointers (wpk = packed bf16x3 weights)
  float* ch_part;    // optional channel statistics of the output: [B][nTZ*nTH][cout][2]
  int* status;       // f16x2: bit 0 <- a staged value exceeded the fp16 range
};

// NTERM = 3: bf16x3; NTERM = 2: f16x2 (split16.h: 64 B + 16 B pad per voxel and sub-chunk, two accumulators per tile folded after
// the K loop, fp16 range flag) -- the arithmetic of the other f16x2 kernels for the convs only this tiling fits (Dataset-3's
// down-sampling conv out of 50x18 planes: 370 us per launch as bf16x3).
template <int VT, int CT, int NTERM>
__global__ void __launch_bounds__(512, (VT * CT * (NTERM == 2 ? 2 : 1) <= 2 ? 3 : 2)) conv_tiled_bf16x3_kernel(ConvTiled3Args args) {
  constexpr int VB = NTERM == 3 ? 96 : 80;  // bytes per staged voxel
  constexpr int WS = 64 * NTERM;            // u32x4 per (tap, ct) in the packed weights
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = (char*)lds;
  const ConvKArgs& a = args.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  int bid = blockIdx.x;
  const int thi = bid % a.nTH;
  bid /= a.nTH;
  const int tzi = bid % a.nTZ;
  const int b = bid / a.nTZ;
  const int ct0 = blockIdx.y * CT;
  const int oz0 = tzi * a.TZ, oh0 = thi * a.TH;
  const int tileVox = a.IZ * a.IH * a.Win;
  const int ZERO = tileVox * VB;  // byte offset of the all-zero voxel
  const int half = lane >> 5, col = lane & 31;
  if (tid < 24) ((float*)(ldsb + ZERO))[tid] = 0.f;

  int abase[VT], ooff[VT];
  unsigned wmask[VT];
  bool any_valid = false;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int v = (wave * VT + vt) * 32 + col;
    const int ow = v % a.Wo;
    const int t = v / a.Wo;
    const int oh = t % a.TH, oz = t / a.TH;
    const bool valid = (oz < a.TZ) && (oz0 + oz < a.Do) && (oh0 + oh < a.Ho);
    abase[vt] = ((oz * a.SZ * a.IH + oh * a.SH) * a.Win + ow * a.SW - 1) * VB + half * 16;
    unsigned m = 0;
    for (int kw = 0; kw < a.KW; ++kw) {
      const int iw = ow * a.SW + kw - 1;
      if (valid && iw >= 0 && iw < a.Win) m |= 1u << kw;
    }
    wmask[vt] = m;
    ooff[vt] = valid ? (((oz0 + oz) * a.Ho + oh0 + oh) * a.Wo + ow) * a.cout : -1;
    any_valid |= valid;
  }
  const bool wave_active = __any(any_valid);

  f32x16 acc[VT][CT], accB[NTERM == 2 ? VT : 1][NTERM == 2 ? CT : 1];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[vt][ct][r] = 0.f;
        if (NTERM == 2) accB[vt][ct][r] = 0.f;
      }
  float amax = 0.f;

  const int nsub = (a.c0 + a.c1) >> 4;
  const int T = a.KD * a.KH * a.KW;
  const int gz0 = oz0 * a.SZ - 1, gh0 = oh0 * a.SH - 1;
  const int items = tileVox * 4;
  const size_t in_vox = (size_t)a.Din * a.Hin * a.Win;

  for (int sc = 0; sc < nsub; ++sc) {
    const float* src;
    int ldc, coff;
    if (sc * 16 < a.c0) {
      src = a.in0; ldc = a.c0; coff = sc * 16;
    } else {
      src = a.in1; ldc = a.c1; coff = sc * 16 - a.c0;
    }
    const int pq = tid & 3;
    src += (size_t)b * in_vox * ldc + coff + pq * 4;
    f32x4 cf[4];
    if (a.coef) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cf[e] = *(const f32x4*)(a.coef + ((size_t)b * (a.c0 + a.c1) + sc * 16 + pq * 4 + e) * 4);
    }
    __syncthreads();
    // a thread keeps its channel quad (pq) and walks the tile's voxels in steps of nthreads / 4, four voxels per trip with their loads
    // in flight together; each voxel's (iw, ih, iz) advances by one trip's stride with carries -- five integer divisions by
    // run-time values per item (~40 instructions each) were most of this loop
    {
      const int vq = nthreads >> 2;          // voxels between a thread's slots
      int viw[4], vih[4], viz[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int vox = (tid >> 2) + k * vq;
        viw[k] = vox % a.Win;
        const int r = vox / a.Win;
        vih[k] = r % a.IH;
        viz[k] = r / a.IH;
      }
      const int DW = nthreads % a.Win, dr = nthreads / a.Win, DH = dr % a.IH, DZ = dr / a.IH;  // one trip = nthreads voxels on
      for (int i0 = tid; i0 < items; i0 += 4 * nthreads) {
        f32x4 val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = i0 + k * nthreads;
          val[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (idx < items) {
            const int gz = gz0 + viz[k];
            int gh = gh0 + vih[k];  // (circular in phi: gh0 >= -1, the tile's rows reach at most Hin - 1 + its halo)
            gh = gh < 0 ? gh + a.Hin : gh;
            gh = gh >= a.Hin ? gh - a.Hin : gh;
            gh = gh >= a.Hin ? gh - a.Hin : gh;
            if (gz >= 0 && gz < a.Din) {
              val[k] = *(const f32x4*)(src + ((size_t)(gz * a.Hin + gh) * a.Win + viw[k]) * ldc);
              if (a.coef) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float t = cf[e][0] * val[k][e] + cf[e][1];
                  if (a.act) t = cd_fast_silu(t);
                  val[k][e] = t + cf[e][2];
                }
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = i0 + k * nthreads;
          if (idx < items) {
            char* d = ldsb + (size_t)(idx >> 2) * VB + pq * 8;
            if (NTERM == 3) {
              u32x2 t1, t2, t3;
              split3(val[k], t1, t2, t3);
              *(u32x2*)d = t1;
              *(u32x2*)(d + 32) = t2;
              *(u32x2*)(d + 64) = t3;
            } else {
              amax = fmaxf(amax, fmaxf(fmaxf(fabsf(val[k][0]), fabsf(val[k][1])), fmaxf(fabsf(val[k][2]), fabsf(val[k][3]))));
              u32x2 t1, t2;
              split2(val[k], t1, t2);
              *(u32x2*)d = t1;
              *(u32x2*)(d + 32) = t2;
            }
          }
          viw[k] += DW; vih[k] += DH; viz[k] += DZ;
          if (viw[k] >= a.Win) { viw[k] -= a.Win; vih[k] += 1; }
          if (vih[k] >= a.IH) { vih[k] -= a.IH; viz[k] += 1; }
        }
      }
    }
    __syncthreads();
    if (!wave_active) continue;

    const u32x4* wq = (const u32x4*)a.wpk + ((size_t)sc * T * a.CTtot + ct0) * WS + lane;
    // The taps as one flat sequence, software-pipelined over a ring of three weight sets: the (L2) weight loads of tap t + 2 are
    // requested before the MFMAs of tap t.  (Loaded inside the tap they were an L2 round trip per VT x CT MFMA blocks.)  Requests
    // past the end repeat the last tap instead of being conditional.
    u32x4 bw0[CT][NTERM], bw1[CT][NTERM], bw2[CT][NTERM];
    auto loadw = [&](u32x4 (&bw)[CT][NTERM], int tap) {
      const int tc = min(tap, T - 1);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < NTERM; ++t) bw[ct][t] = wq[((size_t)tc * a.CTtot + ct) * WS + t * 64];
    };
    int kd = 0, kh = 0, kw = 0;  // of the tap whose MFMAs run next
    auto run_tap = [&](const u32x4 (&bw)[CT][NTERM]) {
      const int rowoff = (kd * a.IH + kh) * a.Win * VB;
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        const int off = ((wmask[vt] >> kw) & 1u) ? abase[vt] + rowoff + kw * VB : ZERO + half * 16;
        u32x4 av[NTERM];
#pragma unroll
        for (int t = 0; t < NTERM; ++t) av[t] = *(const u32x4*)(ldsb + off + t * 32);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (NTERM == 3) {
            f32x16 c = acc[vt][ct];
            c = MFMA_BF16(av[2], bw[ct][0], c);
            c = MFMA_BF16(av[NTERM - 2], bw[ct][NTERM - 2], c);
            c = MFMA_BF16(av[0], bw[ct][NTERM - 1], c);
            c = MFMA_BF16(av[NTERM - 2], bw[ct][0], c);
            c = MFMA_BF16(av[0], bw[ct][NTERM - 2], c);
            c = MFMA_BF16(av[0], bw[ct][0], c);
            acc[vt][ct] = c;
          } else {
            acc[vt][ct] = MFMA_F16(av[0], bw[ct][0], acc[vt][ct]);
            accB[vt][ct] = MFMA_F16(av[0], bw[ct][NTERM - 1], accB[vt][ct]);
            accB[vt][ct] = MFMA_F16(av[NTERM - 1], bw[ct][0], accB[vt][ct]);
          }
        }
      }
      if (++kw == a.KW) { kw = 0; if (++kh == a.KH) { kh = 0; ++kd; } }
    };
    loadw(bw0, 0);
    loadw(bw1, 1);
    for (int tap = 0; tap < T; tap += 3) {
      loadw(bw2, tap + 2);
      __builtin_amdgcn_sched_barrier(0);
      run_tap(bw0);
      __builtin_amdgcn_sched_barrier(0);
      loadw(bw0, tap + 3);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < T) run_tap(bw1);
      __builtin_amdgcn_sched_barrier(0);
      loadw(bw1, tap + 4);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 2 < T) run_tap(bw2);
    }
  }

  if (NTERM == 2) {
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[vt][ct][r] += accB[vt][ct][r] * (1.f / 2048.f);
    if (args.status && amax > 65504.f) atomicOr(args.status, 1);
  }
  float* outb = a.out + (size_t)b * a.Do * a.Ho * a.Wo * a.cout;
  float bv[CT], s1[CT], s2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    bv[ct] = a.bias ? a.bias[(ct0 + ct) * 32 + col] : 0.f;
    s1[ct] = s2[ct] = 0.f;
  }
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int off = __shfl(ooff[vt], row, 64);
      if (off >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float v = acc[vt][ct][r] + bv[ct];
          outb[off + (ct0 + ct) * 32 + col] = v;
          s1[ct] += v;
          s2[ct] += v * v;
        }
      }
    }
  }
  if (args.ch_part) {
    __syncthreads();
    const int nw = nthreads >> 6;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float t1 = s1[ct] + __shfl_xor(s1[ct], 32, 64), t2 = s2[ct] + __shfl_xor(s2[ct], 32, 64);
      if (half == 0) {
        lds[((wave * CT + ct) * 32 + col) * 2] = t1;
        lds[((wave * CT + ct) * 32 + col) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    const int unit = tzi * a.nTH + thi, units = a.nTZ * a.nTH;
    for (int i = tid; i < CT * 32; i += nthreads) {
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < nw; ++w) {
        t1 += lds[((w * CT * 32) + i) * 2];
        t2 += lds[((w * CT * 32) + i) * 2 + 1];
      }
      float* dst = args.ch_part + (((size_t)b * units + unit) * a.cout + ct0 * 32 + i) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

namespace {
template <int VT, int CT, int NTERM = 3>
void launch_tiled3_inst(const ConvTiled3Args& a, dim3 grid, int threads, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv_tiled_bf16x3_kernel<VT, CT, NTERM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_tiled_bf16x3_kernel<VT, CT, NTERM>), grid, dim3(threads), lds, s, a);
  CD_HIP(hipGetLastError());
}
}  // namespace

// returns false when the whole-plane LDS tile does not fit (wide grids such as Dataset-3's 50x18 planes).
// bf16x3 = true runs the split-bf16 kernel on `wpk` = packed bf16x3 weights; otherwise the f32 MFMA kernel
// (stride-1 3x3x3 only).
static bool try_launch_conv3_flat(const float* in0, int c0, const float* in1, int c1, const void* wpk, const float* bias,
                                  float* out, int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu,
                                  int prec /* 0 = f32 MFMA, 3 = bf16x3, 2 = f16x2 */) {
  const bool bf16x3 = prec != 0;  // any 16-bit split kernel
  if (getenv("CD_NO_FLAT")) return false;
  int geo = -1;
  if (g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sz == 1 && g.sh == 1 && g.sw == 1) geo = 0;
  else if (g.kd == 3 && g.kh == 4 && g.kw == 4 && g.sh == 2 && g.sw == 2 && (g.sz == 1 || g.sz == 2)) geo = g.sz == 2 ? 1 : 2;
  else if (g.kd == 4 && g.kh == 4 && g.kw == 4 && g.sh == 2 && g.sw == 2 && g.sz == 2) geo = 3;  // input gradient of an up conv
  if (geo < 0 || (geo != 0 && !bf16x3)) return false;
  const Dims3 d = g.in;
  const int CTtot = cout / 32;
  const int CTmax = CTtot <= 3 ? CTtot : 2;
  if (CTtot % CTmax) return false;
  const int HW = d.h * d.w, HWo = g.out.h * g.out.w;
  const size_t vox_bytes = prec == 3 ? 96 : (prec == 2 ? 80 : 64);
  auto planes = [&](int NT) { return ((32 * NT - 1) / HWo + 1) * g.sz + g.kd; };
  // CT = output-channel tiles per workgroup: CTmax shares one staged input tile between them; 1 spreads them over
  // gridDim.z (shorter MFMA chains: wins on the deep, latency-bound levels)
  auto launch = [&](int NT, int VT, int CT) -> bool {
    if (VT < 0) {  // warp-specialised persistent kernel
      const int NLW = -VT;
      ConvFlatArgs a;
      a.in0 = in0; a.in1 = in1; a.c0 = c0; a.c1 = c1; a.wpk = (const float*)wpk; a.bias = bias; a.out = out;
      a.D = d.d; a.H = d.h; a.W = d.w; a.Do = g.out.d; a.Ho = g.out.h; a.Wo = g.out.w;
      a.R = 32 * NT; a.P = planes(NT); a.cout = cout; a.CTtot = CTtot; a.dbg = 0;
      a.coef = fu.coef; a.act = fu.act; a.ch_part = fu.ch_part;
      const size_t lds = 2 * ((size_t)a.P * HW + 1) * 96 + (size_t)NT * CT * 64 * 4;
      const int ups = (int)((g.out.vox() + a.R - 1) / a.R);
      const int total = ups * batch;
      const int nblk = total < 256 ? total : 256;
      dim3 grid((unsigned)nblk, 1, (unsigned)(CTtot / CT));
      if (fu.units) *fu.units = ups;
      const int threads = (NT + NLW) * 64;
      switch (CT) {
        case 1: launch_ws_inst<1>(a, grid, threads, lds, NLW, ups, total, s, geo); break;
        case 2: launch_ws_inst<2>(a, grid, threads, lds, NLW, ups, total, s, geo); break;
        case 3: launch_ws_inst<3>(a, grid, threads, lds, NLW, ups, total, s, geo); break;
        default: return false;
      }
      return true;
    }
    ConvFlatArgs a;
    a.in0 = in0; a.in1 = in1; a.c0 = c0; a.c1 = c1; a.wpk = (const float*)wpk; a.bias = bias; a.out = out;
    a.D = d.d; a.H = d.h; a.W = d.w; a.Do = g.out.d; a.Ho = g.out.h; a.Wo = g.out.w;
    a.R = 32 * NT; a.P = planes(NT); a.cout = cout; a.CTtot = CTtot;
    a.dbg = getenv("CD_FLAT_DBG") ? atoi(getenv("CD_FLAT_DBG")) : 0;
    a.coef = fu.coef; a.act = fu.act; a.ch_part = fu.ch_part; a.status = fu.status; a.in_absmax = fu.in_absmax;
    a.add_src = prec >= 2 && !bias ? fu.add_src : nullptr;  // (the split-16 kernels only: the caller adds the tensor itself otherwise)
    size_t lds = ((size_t)a.P * HW + 1) * vox_bytes;
    const size_t red = (size_t)(NT / VT) * CT * 32 * 2 * 4;  // cross-wave reduction scratch of the stats epilogue
    if (lds < red) lds = red;
    if (fu.defer.part) {  // (only reached with prec != 0: launch_conv_mfma materialises the table for the f32 kernels)
      lds = (lds + 15) & ~(size_t)15;
      a.defer = fu.defer;
      a.coef_lds_off = (int)lds;
      lds += (size_t)fu.defer.C * 16 + gn_defer_scratch_bytes(fu.defer.C);
      if (lds > 160 * 1024) return false;
    }
    dim3 grid((unsigned)((g.out.vox() + a.R - 1) / a.R), (unsigned)batch, (unsigned)(CTtot / CT));
    if (fu.units) *fu.units = (int)grid.x;
    const int threads = (NT / VT) * 64;
#define CD_FLAT_CASE(V, C)                                                   \
  if (VT == V && CT == C) {                                                  \
    if (prec == 3) launch_flat3_inst<V, C, 3>(a, grid, threads, lds, s, geo); \
    else if (prec == 2) {                                                    \
      if constexpr (V * C <= 4) launch_flat3_inst<V, C, 2>(a, grid, threads, lds, s, geo); \
      else return false;                                                     \
    } else launch_flat_inst<V, C>(a, grid, threads, lds, s);                 \
    if (a.add_src && fu.add_done) *fu.add_done = 1;                          \
    return true;                                                             \
  }
    CD_FLAT_CASE(1, 1) CD_FLAT_CASE(2, 1) CD_FLAT_CASE(3, 1) CD_FLAT_CASE(4, 1)
    CD_FLAT_CASE(1, 2) CD_FLAT_CASE(2, 2) CD_FLAT_CASE(3, 2) CD_FLAT_CASE(4, 2)
    CD_FLAT_CASE(1, 3) CD_FLAT_CASE(2, 3)
#undef CD_FLAT_CASE
    return false;
  };
  if (const char* ov = getenv("CD_FLAT_TILE")) {
    int nt, vt;
    if (sscanf(ov, "%d,%d", &nt, &vt) == 2 && nt % vt == 0 && nt / vt <= 8 && vt * CTmax <= 8 &&
        ((size_t)planes(nt) * HW + 1) * vox_bytes <= 160 * 1024 && (prec != 2 || vt * CTmax <= 4))
      return launch(nt, vt, CTmax);
  }
  // candidate tilings: (tiles per workgroup, tiles per wave)
  static const int kCand[][2] = {{8, 2}, {4, 1}, {8, 1}, {12, 3}, {16, 2}, {16, 4}, {4, 2}, {6, 2}, {6, 3}, {2, 1},
                                 {3, 1}, {1, 1}, {2, 2}, {12, 2}, {8, 4}, {6, 1}, {3, 3}};
  struct Cand { int nt, vt, ct; };
  std::vector<Cand> cand;
  const bool small = g.out.vox() * batch <= 64 * 1024;  // deep levels: also try one output tile per workgroup
  for (int pass = 0; pass < (small && CTmax > 1 ? 2 : 1); ++pass) {
    const int CT = pass == 0 ? CTmax : 1;
    for (auto& c : kCand) {
      const int NT = c[0], VT = c[1];
      if (VT * CT > 8 || (CT == 3 && VT > 2)) continue;
      if (prec == 2 && VT * CT > 4) continue;
      // (more row tiles than the sample has: skipped -- except one tile per wave on grids of <= 128 output voxels, where the
      // surplus waves have no rows but share the staging, whose few threads are the latency of such launches)
      static const bool no_extra_waves = getenv("CD_FLAT_NO_EXTRA_WAVES") != nullptr;
      const bool extra_ok = !no_extra_waves && g.out.vox() <= 128 && VT == 1 && NT <= 8;
      if ((int64_t)32 * (NT - 1) >= g.out.vox() && !extra_ok) continue;
      if (((size_t)planes(NT) * HW + 1) * vox_bytes > 150 * 1024) continue;
      if (pass == 1 && NT > 4) continue;
      cand.push_back({NT, VT, CT});
    }
  }
  if (prec == 3 && !getenv("CD_NO_WS")) {
    // warp-specialised persistent variants: (matrix waves, -loader waves); two LDS buffers + statistics scratch
    static const int kWs[][2] = {{8, 4}, {8, 2}, {4, 2}, {4, 4}, {8, 3}, {6, 2}, {2, 2}, {3, 1}, {1, 1}, {2, 1}};
    for (auto& c : kWs) {
      const int NMW = c[0], NLW = c[1];
      if ((int64_t)32 * (NMW - 1) >= g.out.vox()) continue;
      const size_t lds = 2 * ((size_t)planes(NMW) * HW + 1) * 96 + (size_t)NMW * CTmax * 64 * 4;
      if (lds > 160 * 1024 - 256) continue;
      cand.push_back({NMW, -NLW, CTmax});
    }
  }
  if (cand.empty()) return false;
  char key[192];
  std::snprintf(key, sizeof key, "flat%s g%d %dx%dx%d c%d+%d->%d b%d", prec == 3 ? "_bf16x3" : (prec == 2 ? "_f16x2" : "_f32"), geo,
                d.d, d.h, d.w, c0, c1, cout, batch);
  const int pick = autotune(key, (int)cand.size(), [&](int i) { launch(cand[i].nt, cand[i].vt, cand[i].ct); }, s);
  const Cand& c = cand[pick < 0 ? 0 : pick];
  return launch(c.nt, c.vt, c.ct);
}

void launch_conv_mfma(const float* in0, int c0, const float* in1, int c1, const float* wpk, const float* bias, float* out,
                      int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu_in) {
  CD_REQUIRE((!fu_in.coef && !fu_in.defer.part) || c1 == 0, "conv: a fused input normalisation needs a single (non-concatenated) source");
  if (fu_in.units) *fu_in.units = 0;  // set by kernels that produce the output statistics themselves
  CD_REQUIRE(c0 % 32 == 0 && c1 % 32 == 0 && c0 > 0, "conv: channel counts must be multiples of 32");
  CD_REQUIRE(cout % 32 == 0, "conv: output channels must be a multiple of 32");
  CD_REQUIRE(g.kw <= 4, "conv: r kernel extent > 4 unsupported");
  const int CTtot = cout / 32;
  int CT = CTtot <= 3 ? CTtot : 2;
  CD_REQUIRE(CTtot % CT == 0, "conv: unsupported output channel count");
  char cat[128];
  std::snprintf(cat, sizeof cat, "conv%dx%dx%d_s%d C%d->%d @%dx%dx%d", g.kd, g.kh, g.kw, g.sh, c0 + c1, cout, g.in.d, g.in.h, g.in.w);
  const double taps = (double)g.kd * g.kh * g.kw;
  prof::Scope scope(cat, s, 2.0 * taps * (c0 + c1) * cout * (double)g.out.vox() * batch,
                    4.0 * batch * ((double)g.in.vox() * (c0 + c1) + (double)g.out.vox() * cout));
  // deferred input normalisation: the split-16 kernels fold it in their prologue; for every other kernel the coefficient
  // table is materialised first by a gn_finalize launch
  ConvFusion fu = fu_in;
  auto materialise = [&]() {
    if (!fu.defer.part) return;
    if (fu.defer.coef_out) fu.coef_buf = fu.defer.coef_out;  // (training tape: the table is wanted in memory anyway)
    CD_REQUIRE(fu.coef_buf, "conv: deferred normalisation needs a coefficient buffer for kernels without the prologue");
    launch_gn_finalize(fu.defer.part, fu.defer.units, fu.defer.gamma, fu.defer.beta, fu.defer.add, fu.defer.add_ld, fu.coef_buf, batch,
                       fu.defer.C, fu.defer.groups, fu.defer.vox, s, fu.defer.stat_out);
    fu.coef = fu.coef_buf;
    fu.defer = GnDefer();
  };
  {
    const bool want_f32 = conv_precision() == PREC_F32, want_bf16x3 = conv_precision() == PREC_BF16X3;
    if (fu.wpk_bf16x3 && !want_f32 && !want_bf16x3 &&
        try_launch_conv_zslide(in0, c0, in1, c1, (const char*)fu.wpk_bf16x3 + packed_bf16x3_bytes(c0 + c1, cout, g.kd * g.kh * g.kw),
                               bias, out, batch, cout, g, s, fu))
      return;
    if (fu.wpk_bf16x3 && !want_f32 && !want_bf16x3 &&
        try_launch_conv_small(in0, c0, in1, c1, (const char*)fu.wpk_bf16x3 + packed_bf16x3_bytes(c0 + c1, cout, g.kd * g.kh * g.kw),
                              bias, out, batch, cout, g, s, fu))
      return;
    if (fu.wpk_bf16x3 && !want_f32 && !want_bf16x3 &&
        try_launch_conv3_flat(in0, c0, in1, c1, (const char*)fu.wpk_bf16x3 + packed_bf16x3_bytes(c0 + c1, cout, g.kd * g.kh * g.kw), bias,
                              out, batch, cout, g, s, fu, 2))
      return;
    materialise();
    if (fu.wpk_bf16x3 && !want_f32 && try_launch_conv3_flat(in0, c0, in1, c1, fu.wpk_bf16x3, bias, out, batch, cout, g, s, fu, 3))
      return;
    if (try_launch_conv3_flat(in0, c0, in1, c1, wpk, bias, out, batch, cout, g, s, fu, 0)) return;
  }
  {
    const bool want_f32 = conv_precision() == PREC_F32;
    if (fu.wpk_bf16x3 && !want_f32) {
      // f16x2 arm of the tiled kernel first (two accumulators per tile: at most two tiles per wave) unless the exact split is asked
      // for; bf16x3 if no such tiling fits
      const bool f16_ok = conv_precision() == PREC_F16X2 && !fu.in_absmax;
      for (int pass = f16_ok ? 0 : 1; pass < 2; ++pass) {
      const bool f16 = pass == 0;
      const std::vector<ConvTile> cand3 = conv_tile_candidates(g, batch, CT, 14, f16 ? 80 : 96);
      auto launch3 = [&](const ConvTile& t) {
        ConvTiled3Args a3;
        ConvKArgs& a = a3.k;
        a3.status = fu.status;
        a.in0 = in0; a.in1 = in1; a.c0 = c0; a.c1 = c1; a.bias = bias; a.out = out;
        a.wpk = f16 ? (const float*)((const char*)fu.wpk_bf16x3 + packed_bf16x3_bytes(c0 + c1, cout, g.kd * g.kh * g.kw))
                    : (const float*)fu.wpk_bf16x3;
        a.Din = g.in.d; a.Hin = g.in.h; a.Win = g.in.w; a.Do = g.out.d; a.Ho = g.out.h; a.Wo = g.out.w;
        a.KD = g.kd; a.KH = g.kh; a.KW = g.kw; a.SZ = g.sz; a.SH = g.sh; a.SW = g.sw;
        a.TZ = t.TZ; a.TH = t.TH; a.nTZ = (g.out.d + t.TZ - 1) / t.TZ; a.nTH = (g.out.h + t.TH - 1) / t.TH;
        a.IZ = (t.TZ - 1) * g.sz + g.kd; a.IH = (t.TH - 1) * g.sh + g.kh;
        a.cout = cout; a.CTtot = CTtot; a.coef = fu.coef; a.act = fu.act;
        const int units = a.nTZ * a.nTH;
        const int64_t cap = (g.out.vox() + 31) / 32;  // capacity of the caller's partial buffer (units per sample)
        a3.ch_part = (fu.ch_part && units <= cap) ? fu.ch_part : nullptr;
        if (fu.units) *fu.units = a3.ch_part ? units : 0;
        dim3 grid((unsigned)(batch * a.nTZ * a.nTH), (unsigned)(CTtot / CT));
        size_t lds = t.lds;
        const size_t red = (size_t)t.NW * CT * 32 * 2 * 4;
        if (lds < red) lds = red;
        if (f16) {
          if (t.VT == 1 && CT == 1) { launch_tiled3_inst<1, 1, 2>(a3, grid, t.NW * 64, lds, s); return; }
          if (t.VT == 2 && CT == 1) { launch_tiled3_inst<2, 1, 2>(a3, grid, t.NW * 64, lds, s); return; }
          if (t.VT == 1 && CT == 2) { launch_tiled3_inst<1, 2, 2>(a3, grid, t.NW * 64, lds, s); return; }
          CD_REQUIRE(false, "conv: no f16x2 tiled kernel instance for the chosen tiling");
        }
#define CD_T3_CASE(V, C)                                              \
  if (t.VT == V && CT == C) {                                         \
    launch_tiled3_inst<V, C>(a3, grid, t.NW * 64, lds, s);            \
    return;                                                           \
  }
        CD_T3_CASE(1, 1) CD_T3_CASE(2, 1) CD_T3_CASE(3, 1) CD_T3_CASE(4, 1)
        CD_T3_CASE(1, 2) CD_T3_CASE(2, 2) CD_T3_CASE(3, 2) CD_T3_CASE(4, 2)
        CD_T3_CASE(1, 3) CD_T3_CASE(2, 3)
#undef CD_T3_CASE
        CD_REQUIRE(false, "conv: no bf16x3 tiled kernel instance for the chosen tiling");
      };
      std::vector<ConvTile> ok;
      for (auto& t : cand3)
        if (f16 ? (t.VT * CT <= 2) : (t.VT <= 4 && !(CT == 3 && t.VT > 2))) ok.push_back(t);
      if (!ok.empty()) {
        char key3[192];
        std::snprintf(key3, sizeof key3, "tiled_%s %s b%d", f16 ? "f16x2" : "bf16x3", cat, batch);
        const int pick3 = autotune(key3, (int)ok.size(), [&](int i) { launch3(ok[i]); }, s);
        launch3(ok[pick3 < 0 ? 0 : pick3]);
        return;
      }
      }
    }
  }
  const std::vector<ConvTile> cand = conv_tile_candidates(g, batch, CT, 14);
  auto launch = [&](const ConvTile& t) {
    ConvKArgs a;
    a.in0 = in0; a.in1 = in1; a.c0 = c0; a.c1 = c1; a.wpk = wpk; a.bias = bias; a.out = out;
    a.Din = g.in.d; a.Hin = g.in.h; a.Win = g.in.w; a.Do = g.out.d; a.Ho = g.out.h; a.Wo = g.out.w;
    a.KD = g.kd; a.KH = g.kh; a.KW = g.kw; a.SZ = g.sz; a.SH = g.sh; a.SW = g.sw;
    a.TZ = t.TZ; a.TH = t.TH; a.nTZ = (g.out.d + t.TZ - 1) / t.TZ; a.nTH = (g.out.h + t.TH - 1) / t.TH;
    a.IZ = (t.TZ - 1) * g.sz + g.kd; a.IH = (t.TH - 1) * g.sh + g.kh;
    a.cout = cout; a.CTtot = CTtot; a.coef = fu.coef; a.act = fu.act;
    dim3 grid((unsigned)(batch * a.nTZ * a.nTH), (unsigned)(CTtot / CT));
    const int threads = t.NW * 64;
#define CD_CONV_CASE(V, C)                                        \
  if (t.VT == V && CT == C) {                                     \
    launch_conv_inst<V, C>(a, grid, threads, t.lds, s);           \
    return;                                                       \
  }
    CD_CONV_CASE(1, 1) CD_CONV_CASE(2, 1) CD_CONV_CASE(3, 1) CD_CONV_CASE(4, 1)
    CD_CONV_CASE(5, 1) CD_CONV_CASE(6, 1) CD_CONV_CASE(7, 1) CD_CONV_CASE(8, 1)
    CD_CONV_CASE(1, 2) CD_CONV_CASE(2, 2) CD_CONV_CASE(3, 2) CD_CONV_CASE(4, 2)
    CD_CONV_CASE(1, 3) CD_CONV_CASE(2, 3)
#undef CD_CONV_CASE
    CD_REQUIRE(false, "conv: no kernel instance for the chosen tiling");
  };
  char key[192];
  std::snprintf(key, sizeof key, "tiled %s b%d", cat, batch);
  int pick = autotune(key, (int)cand.size(), [&](int i) { launch(cand[i]); }, s);
  launch(cand[pick < 0 ? 0 : pick]);
  return;
#define CD_CONV_CASE(V, C)
#undef CD_CONV_CASE
  CD_REQUIRE(false, "conv: no kernel instance for the chosen tiling");
}

// ------------------------------------------------------------------------------------------------------------
// transposed conv (Upsample): gather form, output voxels grouped by stride-parity class so that all 32 voxels of an
// MFMA tile share one set of valid taps.
//   out[o] = sum_k in[(o + pad - k)/s] * w[ci][co][k]   over k with (o + pad - k) % s == 0
//   pad = (1, kH-1 after a circular halo of 1, 1)  (models.py:45,59-61)
// ------------------------------------------------------------------------------------------------------------
struct ConvTArgs {
  const float* in;
  int cin;
  const float* wpk;
  const float* bias;
  float* out;
  int Din, Hin, Win, Do, Ho, Wo;
  int KZ, SZ;
  int cout, CTtot;
  int TZ, TH, nTZ, nTH;  // tile extents in class-index space: oz = SZ*a + pz, oh = 2*b + ph
  int Cw;                // ceil(Wo/2)
  int CS;                // LDS voxel stride in floats
  const u32x4* wpk16;    // f16x2 image [k-step][tap][ct][term][lane] (conv_transpose_f16x2_kernel)
  int* status;           // bit 0: a staged value exceeded the fp16 range
  const unsigned* in_absmax;  // power-of-two input rescaling (gradients: ConvFusion::in_absmax) or null
  int tr_off = 0;             // (f16x2 kernel) float offset of the per-wave output transpose tiles behind the input tile
};

template <int CT>
__global__ void __launch_bounds__(256) conv_transpose_kernel(ConvTArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  int bid = blockIdx.x;
  const int thi = bid % a.nTH;
  bid /= a.nTH;
  const int tzi = bid % a.nTZ;
  const int b = bid / a.nTZ;
  const int a0 = tzi * a.TZ, b0 = thi * a.TH;
  const int PZ = a.TZ + 2, PH = a.TH + 2;
  const int tileVox = PZ * PH * a.Win;
  const int ZERO = tileVox * a.CS;
  const int half = lane >> 5, col = lane & 31;
  for (int i = tid; i < a.CS; i += blockDim.x) lds[ZERO + i] = 0.f;

  // stage all input channels of the haloed tile
  {
    const int c4 = a.cin >> 2;
    const int items = tileVox * c4;
    const float* src = a.in + (size_t)b * a.Din * a.Hin * a.Win * a.cin;
    for (int idx = tid; idx < items; idx += blockDim.x) {
      const int q = idx % c4, vox = idx / c4;
      const int iw = vox % a.Win;
      const int r = vox / a.Win;
      const int lh = r % PH, lz = r / PH;
      const int gz = a0 - 1 + lz;
      int gh = (b0 - 1 + lh) % a.Hin;
      if (gh < 0) gh += a.Hin;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (gz >= 0 && gz < a.Din) val = *(const f32x4*)(src + ((size_t)(gz * a.Hin + gh) * a.Win + iw) * a.cin + q * 4);
      *(f32x4*)(lds + vox * a.CS + q * 4) = val;
    }
  }
  __syncthreads();

  const int ncls = a.SZ * 4;
  const int njt = (a.TZ * a.TH * a.Cw + 31) / 32;
  const int nchunk = a.cin >> 5;
  const int T = a.KZ * 16;
  float* outb = a.out + (size_t)b * a.Do * a.Ho * a.Wo * a.cout;

  for (int job = wave; job < ncls * njt; job += nw) {
    const int cls = job / njt, jt = job % njt;
    const int pz = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int v = jt * 32 + col;
    const int c = v % a.Cw;
    const int t = v / a.Cw;
    const int bb = t % a.TH, aa = t / a.TH;
    const int oz = a.SZ * (a0 + aa) + pz, oh = 2 * (b0 + bb) + ph, ow = 2 * c + pw;
    const bool valid = (aa < a.TZ) && (oz < a.Do) && (oh < a.Ho) && (ow < a.Wo);
    if (!__any(valid)) continue;
    const int ooff = valid ? ((oz * a.Ho + oh) * a.Wo + ow) * a.cout : -1;

    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    for (int kz = (pz + 1) % a.SZ; kz < a.KZ; kz += a.SZ) {
      const int lz = aa + (pz + 1 - kz) / a.SZ + 1;
      for (int kh = (ph + 3) & 1; kh < 4; kh += 2) {
        const int lh = bb + (ph + 3 - kh) / 2;  // (.. )/2 - 1 (circular halo) + 1 (tile halo)
        for (int kw = (pw + 1) & 1; kw < 4; kw += 2) {
          const int iw = c + (pw + 1 - kw) / 2;
          const bool ok = valid && iw >= 0 && iw < a.Win;
          const int off = ok ? ((lz * PH + lh) * a.Win + iw) * a.CS + half * 16 : ZERO + half * 16;
          const int tap = (kz * 4 + kh) * 4 + kw;
          for (int chunk = 0; chunk < nchunk; ++chunk) {
            f32x4 av[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = *(const f32x4*)(lds + off + chunk * 32 + q * 4);
            const f32x4* wq = (const f32x4*)a.wpk + ((size_t)(chunk * T + tap) * a.CTtot) * 256 + lane;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              f32x4 bw[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) bw[q] = wq[ct * 256 + q * 64];
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ct] = MFMA32(av[q][e], bw[q][e], acc[ct]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int off = __shfl(ooff, row, 64);
      if (off >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int co = ct * 32 + col;
          outb[off + co] = acc[ct][r] + (a.bias ? a.bias[co] : 0.f);
        }
      }
    }
  }
}

// The same gather on the fp16 matrix pipe (f16x2, see kernels_conv_zs.hip): the haloed input tile is split into two fp16
// terms while it is staged (record = [k-step][term][16 fp16] + 16 B pad, the byte size of the fp32 record), every valid
// (tap, 16-channel k-step) costs two ds_read_b128 and three MFMAs per 32 output channels instead of eight f32 MFMAs of twice
// the duration.
template <int CT>
__global__ void __launch_bounds__(256) conv_transpose_f16x2_kernel(ConvTArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  int bid = blockIdx.x;
  const int thi = bid % a.nTH;
  bid /= a.nTH;
  const int tzi = bid % a.nTZ;
  const int b = bid / a.nTZ;
  const int a0 = tzi * a.TZ, b0 = thi * a.TH;
  const int PZ = a.TZ + 2, PH = a.TH + 2;
  const int tileVox = PZ * PH * a.Win;
  const int ZERO = tileVox * a.CS;
  const int half = lane >> 5, col = lane & 31;
  for (int i = tid; i < a.CS; i += blockDim.x) lds[ZERO + i] = 0.f;

  float gscale = 1.f, ginv = 1.f;
  if (a.in_absmax) pow2_scale_for(*a.in_absmax, &gscale, &ginv);
  {  // stage + split all input channels of the haloed tile
    const int c4 = a.cin >> 2;
    const float* src = a.in + (size_t)b * a.Din * a.Hin * a.Win * a.cin;
    float amax = 0.f;
    auto stage = [&](int vox, int q, int iw, int lh, int lz) {
      const int gz = a0 - 1 + lz;
      int gh = b0 - 1 + lh;  // (circular halo: -1 .. b0 + TH < 2 Hin)
      gh = gh < 0 ? gh + a.Hin : gh;
      gh = gh >= a.Hin ? gh - a.Hin : gh;
      gh = gh >= a.Hin ? gh - a.Hin : gh;  // (a tile of a ring shorter than its halo: TH + 2 <= 3 Hin always)
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (gz >= 0 && gz < a.Din) val = *(const f32x4*)(src + ((size_t)(gz * a.Hin + gh) * a.Win + iw) * a.cin + q * 4);
      val *= gscale;
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3]))));
      u32x2 t1, t2;
      split2(val, t1, t2);
      char* dst = (char*)(lds + vox * a.CS) + (q >> 2) * 64 + (q & 3) * 8;
      *(u32x2*)dst = t1;
      *(u32x2*)(dst + 32) = t2;
    };
    const int nthr = blockDim.x;
    if (nthr % c4 == 0) {
      // a thread keeps its channel quad and walks the tile's voxels in steps of nthr / c4, its (iw, lh, lz) advanced with carries:
      // six integer divisions by run-time values per item (~40 instructions each) were most of this loop
      const int q = tid % c4, vstep = nthr / c4;
      int vox = tid / c4;
      int iw = vox % a.Win, r = vox / a.Win;
      int lh = r % PH, lz = r / PH;
      const int dw = vstep % a.Win, dr = vstep / a.Win, dh = dr % PH, dz = dr / PH;
      for (; vox < tileVox; vox += vstep) {
        stage(vox, q, iw, lh, lz);
        iw += dw; lh += dh; lz += dz;
        if (iw >= a.Win) { iw -= a.Win; lh += 1; }
        if (lh >= PH) { lh -= PH; lz += 1; }
      }
    } else {
      const int items = tileVox * c4;
      for (int idx = tid; idx < items; idx += nthr) {
        const int q = idx % c4, vox = idx / c4;
        const int iw = vox % a.Win;
        const int r = vox / a.Win;
        stage(vox, q, iw, r % PH, r / PH);
      }
    }
    if (a.status && amax > 65504.f) atomicOr(a.status, 1);
  }
  __syncthreads();

  // (the z stride is 1 or 2 -- the launcher checks it: shifts and masks below where run-time integer divisions by a.SZ cost ~30 vector
  // instructions each, per tap and parity class, in a kernel that PMC shows bound by vector issue: 21 VALU instructions per MFMA)
  const int szs = a.SZ - 1;  // log2(SZ)
  const int ncls = a.SZ * 4;
  const int njt = (a.TZ * a.TH * a.Cw + 31) / 32;
  const int nks = a.cin >> 4;
  const int T = a.KZ * 16;
  float* outb = a.out + (size_t)b * a.Do * a.Ho * a.Wo * a.cout;

  // Work = (row tile jt, parity class cls).  With at least one tile per wave a wave takes tiles jt = wave, wave + nw, .. and runs all
  // SZ x 4 classes on each: the tile's 32 class-space positions are decomposed once (four integer divisions by run-time values per
  // lane) instead of once per class.  Tiles of fewer row tiles than waves (the deepest levels) spread (class, tile) pairs over the
  // waves instead.
  const bool tile_major = njt >= nw;
  for (int item = wave; item < (tile_major ? njt : ncls * njt); item += nw) {
    const int jt = tile_major ? item : item % njt;
    const int v = jt * 32 + col;
    const int c = v % a.Cw;
    const int t = v / a.Cw;
    const int bb = t % a.TH, aa = t / a.TH;
  for (int ci = 0; ci < (tile_major ? ncls : 1); ++ci) {
    const int cls = tile_major ? ci : item / njt;
    const int pz = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int oz = ((a0 + aa) << szs) + pz, oh = 2 * (b0 + bb) + ph, ow = 2 * c + pw;
    const bool valid = (aa < a.TZ) && (oz < a.Do) && (oh < a.Ho) && (ow < a.Wo);
    if (!__any(valid)) continue;
    const int ooff = valid ? ((oz * a.Ho + oh) * a.Wo + ow) * a.cout : -1;

    f32x16 accA[CT], accB[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[ct][r] = 0.f; accB[ct][r] = 0.f; }

    // The class's taps -- kz in {kz0, kz0 + SZ, ..}, two kh, two kw -- times the k-steps as ONE flat sequence of stages, software-
    // pipelined over a ring of four: the fragment (LDS) and weight (L2) loads of stage j + 3 are requested before the MFMAs of
    // stage j -- three stages = 9 MFMAs = ~300 cycles of cover for an L2 round trip.  (As nested loops every k-step waited for its
    // own weight loads: a round trip per 3 MFMAs.)  Requests past the end repeat the last stage instead of being conditional: a
    // conditional load in a pipelined loop costs a full vmcnt(0) per trip.
    const int kz0 = (pz + 1) & szs, kh0 = (ph + 3) & 1, kw0 = (pw + 1) & 1;
    const int nkz = (a.KZ - kz0 + a.SZ - 1) >> szs;
    const int nstage = nkz * 4 * nks;
    struct Stage {
      u32x4 x1, x2, w[CT][2];
    };
    int ti_n = 0, ks_n = 0, tap_n = 0;  // the next stage to request: tap number (kz-major), k-step; its weight tap index
    const char* rec_n = nullptr;        // ... and this lane's record of that tap
    auto setup = [&](int ti) {
      const int kz = kz0 + ((ti >> 2) << szs), kh = kh0 + ((ti >> 1) & 1) * 2, kw = kw0 + (ti & 1) * 2;
      const int lz = aa + ((pz + 1 - kz) >> szs) + 1;  // (pz + 1 - kz is a multiple of SZ, possibly negative: the arithmetic shift is exact)
      const int lh = bb + ((ph + 3 - kh) >> 1);  // (.. )/2 - 1 (circular halo) + 1 (tile halo); even by the choice of kh0
      const int iw = c + ((pw + 1 - kw) >> 1);   // even by the choice of kw0
      const bool ok = valid && iw >= 0 && iw < a.Win;
      rec_n = (const char*)(lds + (ok ? ((lz * PH + lh) * a.Win + iw) * a.CS : ZERO)) + half * 16;
      tap_n = (kz * 4 + kh) * 4 + kw;
    };
    setup(0);
    auto fetch = [&](Stage& st) {
      st.x1 = *(const u32x4*)(rec_n + ks_n * 64);
      st.x2 = *(const u32x4*)(rec_n + ks_n * 64 + 32);
      const u32x4* wq = a.wpk16 + ((size_t)(ks_n * T + tap_n) * a.CTtot) * 128 + lane;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        st.w[ct][0] = wq[ct * 128];
        st.w[ct][1] = wq[ct * 128 + 64];
      }
      if (ks_n + 1 < nks) ++ks_n;
      else if (ti_n + 1 < nkz * 4) { ks_n = 0; setup(++ti_n); }
    };
    auto mfmas = [&](const Stage& st) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        accA[ct] = MFMA_F16(st.x1, st.w[ct][0], accA[ct]);
        accB[ct] = MFMA_F16(st.x1, st.w[ct][1], accB[ct]);
        accB[ct] = MFMA_F16(st.x2, st.w[ct][0], accB[ct]);
      }
    };
    Stage s0, s1, s2, s3;
    fetch(s0);
    fetch(s1);
    fetch(s2);
    for (int j = 0; j < nstage; j += 4) {
      fetch(s3);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(s0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < nstage) mfmas(s1);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s1);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 2 < nstage) mfmas(s2);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s2);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 3 < nstage) mfmas(s3);
    }
    // the tile's rows (output voxels of one parity class: 128 contiguous bytes each per channel tile) leave as 16-byte quads after
    // a transpose through the wave's LDS tile (behind the input tile): row 8 k + (lane >> 3), channels 4 (lane & 7) .. + 3
    float* tr = lds + a.tr_off + wave * (32 * 36);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float bv = a.bias ? a.bias[ct * 32 + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tr[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + col] = (accA[ct][r] + accB[ct][r] * (1.f / 2048.f)) * ginv + bv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes are visible to its reads in order)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 8 * k + (lane >> 3);
        const int off = __shfl(ooff, row, 64);
        const f32x4 q = *(const f32x4*)(tr + row * 36 + (lane & 7) * 4);
        if (off >= 0) *(f32x4*)(outb + off + ct * 32 + (lane & 7) * 4) = q;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the tile buffer is reused)
    }
  }
  }
}

template <int CT>
static void launch_convT_inst(const ConvTArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    CD_HIP(hipFuncSetAttribute((const void*)conv_transpose_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CD_HIP(hipFuncSetAttribute((const void*)conv_transpose_f16x2_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (a.wpk16) hipLaunchKernelGGL((conv_transpose_f16x2_kernel<CT>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv_transpose_kernel<CT>), grid, dim3(256), lds, s, a);
  CD_HIP(hipGetLastError());
}

void launch_conv_transpose_mfma(const float* in, int cin, const float* wpk, const float* bias, float* out, int batch,
                                int cout, Dims3 din, Dims3 dout, int kz, int sz, hipStream_t s, const void* wpk_f16x2,
                                int* status, const unsigned* in_absmax) {
  CD_REQUIRE(cin % 32 == 0 && cout % 32 == 0, "conv_transpose: channels must be multiples of 32");
  CD_REQUIRE(sz == 1 || sz == 2, "conv_transpose: z stride must be 1 or 2");
  const bool full_range = conv_precision() != PREC_F16X2;
  ConvTArgs a;
  a.wpk16 = full_range ? nullptr : (const u32x4*)wpk_f16x2;
  a.status = status;
  a.in_absmax = in_absmax;
  a.in = in; a.cin = cin; a.wpk = wpk; a.bias = bias; a.out = out;
  a.Din = din.d; a.Hin = din.h; a.Win = din.w; a.Do = dout.d; a.Ho = dout.h; a.Wo = dout.w;
  a.KZ = kz; a.SZ = sz; a.cout = cout; a.CTtot = cout / 32;
  a.Cw = (dout.w + 1) / 2;
  a.CS = cin + LDS_VOX_PAD;
  const int Az = (dout.d + sz - 1) / sz, Bh = (dout.h + 1) / 2;
  // candidate (TZ, TH) tiles in class-index space; ranked by useful/haloed volume, a spread of them is timed once
  struct TT { int tz, th; double score; };
  std::vector<TT> all;
  for (int TZ = 1; TZ <= Az; ++TZ)
    for (int TH = 1; TH <= Bh; ++TH) {
      const size_t lds = ((size_t)(TZ + 2) * (TH + 2) * din.w + 1) * a.CS * 4;
      if (lds > 140 * 1024) break;  // (+ 18 KB of output transpose tiles in the f16x2 kernel)
      if (TH != Bh && (Bh + TH - 1) / TH == (Bh + TH) / (TH + 1)) continue;  // a larger TH gives the same tile count
      if (TZ != Az && (Az + TZ - 1) / TZ == (Az + TZ) / (TZ + 1)) continue;
      const long nblocks = (long)batch * ((Az + TZ - 1) / TZ) * ((Bh + TH - 1) / TH);
      const double ratio = (double)TZ * TH / ((double)(TZ + 2) * (TH + 2));
      const double fill = nblocks >= 512 ? 1.0 : (double)nblocks / 512.0;
      all.push_back({TZ, TH, ratio * fill});
    }
  CD_REQUIRE(!all.empty(), "conv_transpose: no tile fits in LDS");
  std::stable_sort(all.begin(), all.end(), [](const TT& x, const TT& y) { return x.score > y.score; });
  std::vector<TT> cand;
  for (size_t i = 0; i < all.size() && cand.size() < 8; ++i) cand.push_back(all[i]);
  for (size_t i = 8; i < all.size() && cand.size() < 14; i += (all.size() - 8) / 6 + 1) cand.push_back(all[i]);
  char cat[128];
  std::snprintf(cat, sizeof cat, "convT%dx4x4 C%d->%d @%dx%dx%d", kz, cin, cout, din.d, din.h, din.w);
  prof::Scope scope(cat, s, 2.0 * kz * 16 * cin * cout * (double)din.vox() * batch,
                    4.0 * batch * ((double)din.vox() * cin + (double)dout.vox() * cout));
  auto launch = [&](const TT& t) {
    ConvTArgs b = a;
    b.TZ = t.tz; b.TH = t.th;
    b.nTZ = (Az + b.TZ - 1) / b.TZ; b.nTH = (Bh + b.TH - 1) / b.TH;
    size_t lds = ((size_t)(b.TZ + 2) * (b.TH + 2) * din.w + 1) * b.CS * 4;
    if (b.wpk16) {  // four per-wave 32 x 36 float tiles behind the input tile (16-byte aligned)
      lds = (lds + 15) & ~(size_t)15;
      b.tr_off = (int)(lds / 4);
      lds += 4 * 32 * 36 * 4;
    }
    dim3 grid((unsigned)(batch * b.nTZ * b.nTH));
    switch (b.CTtot) {
      case 1: launch_convT_inst<1>(b, grid, lds, s); break;
      case 2: launch_convT_inst<2>(b, grid, lds, s); break;
      case 3: launch_convT_inst<3>(b, grid, lds, s); break;
      case 4: launch_convT_inst<4>(b, grid, lds, s); break;
      default: CD_REQUIRE(false, "conv_transpose: more than 128 output channels unsupported");
    }
  };
  char key[192];
  std::snprintf(key, sizeof key, "%s k%d s%d b%d%s", cat, kz, sz, batch, a.wpk16 ? " f16x2" : "");
  const int pick = autotune(key, (int)cand.size(), [&](int i) { launch(cand[i]); }, s);
  launch(cand[pick < 0 ? 0 : pick]);
}

// ------------------------------------------------------------------------------------------------------------
// pointwise (1x1x1) conv = per-voxel channel GEMM, A operand straight from global memory (HBM-bound).
// Optional A prologues: GroupNorm(1) affine (PreNorm -> to_qkv) or a 32-way channel softmax (q of linear attention).
// Optional per-sample weights (the folded  W_out * context^T  of linear attention) and residual add.
// ------------------------------------------------------------------------------------------------------------
template <int CT, int PRO>
__global__ void __launch_bounds__(256) pointwise_kernel(PointwiseArgs a, int CTtot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int ct0 = blockIdx.z * CT;
  const int64_t n0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const int64_t n = n0 + col;
  const bool valid = n < a.vox;

  // fused block close (PointwiseArgs::gn_res): the normalisation's coefficients folded here, and this lane's 16 x CT values of the
  // normalised tensor requested before the matrix loop
  __shared__ __attribute__((aligned(16))) float sGn[128 * 4];
  __shared__ __attribute__((aligned(16))) char sGnScratch[128 * 16 + 64 * 8];
  float hv[CT][16];
  // `full`: the wave's 32 voxels and the workgroup's channel tiles all exist (every tile but a sample's last): the epilogue then
  // addresses its 16 rows as 32-bit offsets from one wave-uniform pointer, without a predicate per element (the general form costs
  // ~25 vector instructions per element in 64-bit index arithmetic and exec masking)
  const bool full = n0 + 32 <= a.vox && (ct0 + CT) * 32 <= a.cout;
  const int rl = 4 * half;  // accumulator register r of a tile = row (r & 3) + 8 (r >> 2) + rl, column col
  if (a.gn_res) {
    gn_defer_to_lds(a.gn_defer, b, sGn, sGnScratch);
    if (full) {
      const float* hp = a.gn_res + ((size_t)b * a.vox + n0) * a.cout + ct0 * 32 + col;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[ct][r] = hp[((r & 3) + 8 * (r >> 2) + rl) * a.cout + ct * 32];
    } else {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int co = min((ct0 + ct) * 32 + col, a.cout - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t nr = min(n0 + (r & 3) + 8 * (r >> 2) + 4 * half, a.vox - 1);
          hv[ct][r] = a.gn_res[((size_t)b * a.vox + nr) * a.cout + co];
        }
      }
    }
  }

  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const int nchunk = (a.c0 + a.c1) >> 5;
  const float* wb = a.wpk + (size_t)b * a.w_batch_stride;
  if (PRO == A_NONE && a.wpk16) {
    // fp16 pipe.  The lane keeps the f32 path's loads -- its voxel's channels 16 half .. 16 half + 15 of the chunk, 64 contiguous
    // bytes -- and runs them as two k-steps of 8: k-slot (half, j) of k-step s' is channel 16 half + 8 s' + j, which in the packed
    // image (k-step s: slot (h, j) = channel 16 s + 8 h + j) is what lane (h = s', col) of k-step s = half holds -- the same image,
    // another lane's entry.
    f32x16 accB[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[ct][r] = 0.f;
    float amax = 0.f;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const float* src;
      if (chunk * 32 < a.c0) src = a.in0 + ((size_t)b * a.vox + (valid ? n : 0)) * a.ld0 + a.off0 + chunk * 32 + half * 16;
      else src = a.in1 + ((size_t)b * a.vox + (valid ? n : 0)) * a.ld1 + (chunk * 32 - a.c0) + half * 16;
      f32x4 av[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        av[q] = *(const f32x4*)(src + q * 4);
        if (!valid) av[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(av[q][0])), fabsf(av[q][1])), fmaxf(fabsf(av[q][2]), fabsf(av[q][3])));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x2 h0, l0, h1, l1;
        split2(av[2 * ks], h0, l0);
        split2(av[2 * ks + 1], h1, l1);
        const u32x4 a1 = {h0[0], h0[1], h1[0], h1[1]}, a2 = {l0[0], l0[1], l1[0], l1[1]};
        const u32x4* wp = (const u32x4*)a.wpk16 + ((size_t)(chunk * 2 + half) * CTtot + ct0) * 128 + ks * 32 + col;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const u32x4 w1 = wp[ct * 128], w2 = wp[ct * 128 + 64];
          acc[ct] = MFMA_F16(a1, w1, acc[ct]);
          accB[ct] = MFMA_F16(a1, w2, accB[ct]);
          accB[ct] = MFMA_F16(a2, w1, accB[ct]);
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] += accB[ct][r] * (1.f / 2048.f);
    if (a.status && amax > 65504.f) atomicOr(a.status, 1);
  } else
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const float* src;
    if (chunk * 32 < a.c0) src = a.in0 + ((size_t)b * a.vox + (valid ? n : 0)) * a.ld0 + a.off0 + chunk * 32 + half * 16;
    else src = a.in1 + ((size_t)b * a.vox + (valid ? n : 0)) * a.ld1 + (chunk * 32 - a.c0) + half * 16;
    f32x4 av[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      av[q] = *(const f32x4*)(src + q * 4);
      if (!valid) av[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (PRO == A_AFFINE) {
      const float* cfp = a.coef + ((size_t)b * (a.c0 + a.c1) + chunk * 32 + half * 16) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 cf = *(const f32x4*)(cfp + (q * 4 + e) * 4);
          av[q][e] = cf[0] * av[q][e] + cf[1];
        }
    } else if (PRO == A_EXPNORM) {
      const float* cfp = a.coef + ((size_t)b * (a.c0 + a.c1) + chunk * 32 + half * 16) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) av[q][e] = valid ? expf(av[q][e] - cfp[(q * 4 + e) * 2]) * cfp[(q * 4 + e) * 2 + 1] : 0.f;
    } else if (PRO == A_SOFTMAX32) {
      float m = av[0][0];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, av[q][e]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float ssum = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          av[q][e] = expf(av[q][e] - m);
          ssum += av[q][e];
        }
      ssum += __shfl_xor(ssum, 32, 64);
      const float inv = 1.f / ssum;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) av[q][e] *= inv;
    }
    const f32x4* wq = (const f32x4*)wb + ((size_t)chunk * CTtot + ct0) * 256 + lane;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 bw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bw[q] = wq[ct * 256 + q * 64];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[ct] = MFMA32(av[q][e], bw[q][e], acc[ct]);
    }
  }

  // final values in place (bias, residual, fused block close), then the stores and the channel statistics read them
  if (full) {
    const int ld = a.out_ld ? a.out_ld : a.cout;
    // the output tile leaves as 16-byte quads (row 8 k + (lane >> 3), channels 4 (lane & 7) .. + 3) after a transpose through LDS:
    // 16 scalar row stores per lane in accumulator layout ran at a fraction of the HBM rate (see init_conv_f16x2_kernel)
    __shared__ __attribute__((aligned(16))) float sTr[4][32 * 36];
    float* tr = sTr[wave];
    float* op = a.out + ((size_t)b * a.vox + n0) * ld + a.out_off + ct0 * 32 + (lane & 7) * 4;
    const float* rp = a.residual ? a.residual + ((size_t)b * a.vox + n0) * a.cout + ct0 * 32 + col : nullptr;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float bv = a.bias ? a.bias[(ct0 + ct) * 32 + col] : 0.f;
      f32x4 cf = {0.f, 0.f, 0.f, 0.f};
      if (a.gn_res) cf = *(const f32x4*)(sGn + ((ct0 + ct) * 32 + col) * 4);
      float rv[16];
      if (rp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = rp[((r & 3) + 8 * (r >> 2) + rl) * a.cout + ct * 32];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[ct][r] + bv;
        if (rp) v += rv[r];
        if (a.gn_res) {
          const float u = cf[0] * hv[ct][r] + cf[1];
          v += u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f)) + cf[2];
        }
        acc[ct][r] = v;
        tr[((r & 3) + 8 * (r >> 2) + rl) * 36 + col] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes are visible to its reads in order)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 8 * k + (lane >> 3);
        *(f32x4*)(op + (size_t)row * ld + ct * 32) = *(const f32x4*)(tr + row * 36 + (lane & 7) * 4);
      }
      if (ct + 1 < CT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the tile buffer is reused by the next channel tile)
    }
  } else {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int co = (ct0 + ct) * 32 + col;
    const bool cok = co < a.cout;
    const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
    f32x4 cf = {0.f, 0.f, 0.f, 0.f};
    if (a.gn_res && cok) cf = *(const f32x4*)(sGn + co * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t nr = n0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[ct][r] + bv;
      if (a.residual && cok && nr < a.vox) v += a.residual[((size_t)b * a.vox + nr) * a.cout + co];
      if (a.gn_res) {  // + silu(scale h + shift) + add, SiLU on the transcendental unit as in gn_apply_kernel
        const float u = cf[0] * hv[ct][r] + cf[1];
        v += u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f)) + cf[2];
      }
      acc[ct][r] = v;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    const int64_t nr = n0 + row;
    if (nr < a.vox) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int co = (ct0 + ct) * 32 + col;
        if (co < a.cout) a.out[((size_t)b * a.vox + nr) * (a.out_ld ? a.out_ld : a.cout) + a.out_off + co] = acc[ct][r];
      }
    }
  }
  }
  if (a.ch_part) {  // per-channel {sum, sum of squares} of this workgroup's 128 output voxels
    __shared__ float red[4][CT * 32][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (full || n0 + row < a.vox) {
          const float v = acc[ct][r];
          s1 += v;
          s2 += v * v;
        }
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (half == 0) {
        red[wave][ct * 32 + col][0] = s1;
        red[wave][ct * 32 + col][1] = s2;
      }
    }
    __syncthreads();
    if (tid < CT * 32 && ct0 * 32 + tid < a.cout) {
      const float t1 = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
      const float t2 = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
      float* dst = a.ch_part + (((size_t)b * gridDim.x + blockIdx.x) * a.cout + ct0 * 32 + tid) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

void launch_pointwise(const PointwiseArgs& a, hipStream_t s) {
  CD_REQUIRE(a.c0 % 32 == 0 && a.c1 % 32 == 0 && a.c0 > 0, "pointwise conv: channels must be multiples of 32");
  CD_REQUIRE(a.prologue != A_SOFTMAX32 || (a.c0 == 32 && a.c1 == 0), "softmax prologue needs exactly 32 channels");
  CD_REQUIRE(!a.gn_res || (a.gn_defer.part && a.gn_defer.C == a.cout && a.cout <= 128 && !a.out_ld),
             "pointwise conv: the fused block close normalises a packed tensor of the output's width (<= 128 channels)");
  CD_REQUIRE(a.out_off % 4 == 0 && a.out_ld % 4 == 0 && (a.cout % 4 == 0 || a.out_ld), "pointwise conv: output rows must be 16-byte aligned");
  CD_REQUIRE(!a.wpk16 || (a.prologue == A_NONE && !a.w_batch_stride && a.cout % 32 == 0),
             "pointwise conv: the fp16-pipe form takes shared weights, whole 32-channel tiles and no input prologue");
  const int CTtot = (a.cout + 31) / 32;
  const int CT = CTtot <= 3 ? CTtot : (CTtot % 2 == 0 ? 2 : 1);
  dim3 grid((unsigned)((a.vox + 127) / 128), (unsigned)a.batch, (unsigned)(CTtot / CT));
  char cat[128];
  std::snprintf(cat, sizeof cat, "pointwise_p%d C%d->%d n%ld", a.prologue, a.c0 + a.c1, a.cout, (long)a.vox);
  prof::Scope scope(cat, s, 2.0 * (a.c0 + a.c1) * a.cout * (double)a.vox * a.batch,
                    4.0 * a.batch * (double)a.vox * (a.c0 + a.c1 + a.cout + (a.residual ? a.cout : 0)));
#define CD_PW_CASE(C, P)                                                                   \
  if (CT == C && a.prologue == P) {                                                        \
    hipLaunchKernelGGL((pointwise_kernel<C, P>), grid, dim3(256), 0, s, a, CTtot);         \
    CD_HIP(hipGetLastError());                                                             \
    return;                                                                                \
  }
  CD_PW_CASE(1, A_NONE) CD_PW_CASE(2, A_NONE) CD_PW_CASE(3, A_NONE)
  CD_PW_CASE(1, A_AFFINE) CD_PW_CASE(2, A_AFFINE) CD_PW_CASE(3, A_AFFINE)
  CD_PW_CASE(1, A_SOFTMAX32) CD_PW_CASE(2, A_SOFTMAX32) CD_PW_CASE(3, A_SOFTMAX32)
  CD_PW_CASE(1, A_EXPNORM) CD_PW_CASE(2, A_EXPNORM) CD_PW_CASE(3, A_EXPNORM)
#undef CD_PW_CASE
  CD_REQUIRE(false, "pointwise conv: no kernel instance");
}

// ------------------------------------------------------------------------------------------------------------
// init conv: 3x3x3 cylindrical conv from a few planar channels (x, and the constant R / Z / phi coordinate images,
// synthesised from their 1-D profiles instead of being materialised: calodiffusion.py:121-142) to 32*k channels-last.
// One thread per output voxel; weights are wave-uniform => scalar loads.
// ------------------------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(256) init_conv_kernel(InitConvArgs a) {
  const int64_t vox = a.dims.vox();
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int ct = blockIdx.z;
  const bool valid = n < vox;
  const int D = a.dims.d, H = a.dims.h, W = a.dims.w;
  const int nn = valid ? (int)n : 0;
  const int w = nn % W, h = (nn / W) % H, z = nn / (W * H);
  float sc = (a.x && a.scale_b) ? a.scale_b[(size_t)b * a.scale_stride] : 1.f;
  if (a.x && a.sigma_b) {  // same expression as embed_kernel's c_in
    const float tv = a.sigma_b[b], sd = a.sigma_data;
    sc = 1.f / sqrtf(tv * tv + sd * sd);
  }

  float acc[32];
  const float* __restrict__ bias = a.bias;
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = bias ? bias[ct * 32 + j] : 0.f;

  // this channel tile's weights in LDS (27 CIN rows of 32): read from global as wave-uniform scalar loads they were 81 dependent
  // round trips per thread -- 52 us for the batch-1 coordinate table the training step refreshes every step (26 workgroups)
  __shared__ __attribute__((aligned(16))) float wsm[27 * CIN * 32];
  for (int i = threadIdx.x; i < 27 * CIN * 32; i += 256) wsm[i] = a.wpk[(size_t)(i >> 5) * a.cout + ct * 32 + (i & 31)];
  __syncthreads();
  for (int kd = 0; kd < 3; ++kd) {
    const int zz = z + kd - 1;
    for (int kh = 0; kh < 3; ++kh) {
      int hh = h + kh - 1;
      hh = hh < 0 ? hh + H : (hh >= H ? hh - H : hh);
      hh = hh % H;  // H == 1 or 2
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = w + kw - 1;
        const bool inb = valid && zz >= 0 && zz < D && ww >= 0 && ww < W;
        const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          float v = 0.f;
          if (inb) {
            if (ci < a.cx) {
              if (a.x) {  // null: only the synthesised channels contribute (coordinate table of the matrix-core path)
                v = a.x[(((size_t)b * a.cx + ci) * D + zz) * H * W + (size_t)hh * W + ww];
                if (ci == 0) v *= sc;
              }
            } else {
              const int k = ci - a.cx;
              if (a.use_rz) v = (k == 0) ? a.r_w[ww] : (k == 1 ? a.z_d[zz] : a.phi_h[hh]);
              else v = a.phi_h[hh];
            }
          }
          const float* wr = wsm + (tap * CIN + ci) * 32;
#pragma unroll
          for (int j = 0; j < 32; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
        }
      }
    }
  }
  if (valid) {
    f32x4* o = (f32x4*)(a.out + ((size_t)b * vox + n) * a.cout + ct * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  }
}

// The same conv on the matrix cores when x is the only data channel (the denoiser's case).  The conv is linear in its input
// channels: the coordinate channels' contribution (+ bias) is the same for every sample and step -- `table` (vox, cout), filled
// by one batch-1 launch of the kernel above with x = null -- and what remains is a 27-tap, one-channel conv of c_in * x:
// K = 27 padded to 32 = two fp16 k-steps (f16x2: three MFMAs each) per 32 voxels instead of 27 * cin * 32 scalar FMAs per
// voxel.  A workgroup owns TZ z-planes of one sample: c_in * x of those planes (+ halo: zero planes / columns outside the
// grid, phi rows wrapped) sits in LDS as fp32, so every tap of every voxel is "base + constant".
__global__ void __launch_bounds__(256) init_conv_f16x2_kernel(InitConvArgs a, const float* __restrict__ table, int TZ) {
  extern __shared__ __attribute__((aligned(16))) float img[];
  __shared__ __attribute__((aligned(16))) float trn[4 * 32 * 36];  // per-wave output tile on its way to row-major quads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int D = a.dims.d, H = a.dims.h, W = a.dims.w, PV = H * W;
  const int z0 = blockIdx.x * TZ, nz = min(TZ, D - z0);
  const int HP = H + 2, WP = W + 2;
  float sc = a.scale_b ? a.scale_b[(size_t)b * a.scale_stride] : 1.f;
  if (a.sigma_b) {  // same expression as embed_kernel's c_in
    const float tv = a.sigma_b[b], sd = a.sigma_data;
    sc = 1.f / sqrtf(tv * tv + sd * sd);
  }
  {
    const float* xb = a.x + (size_t)b * D * PV;
    float amax = 0.f;
    const float inv_wp = 1.f / (float)WP, inv_hp = 1.f / (float)HP;
    for (int i = tid; i < (nz + 2) * HP * WP; i += 256) {
      const int r = (int)(((float)i + 0.5f) * inv_wp), lw = i - r * WP, lz = (int)(((float)r + 0.5f) * inv_hp), lh = r - lz * HP;
      const int gz = z0 - 1 + lz, gw = lw - 1;
      int gh = lh - 1;
      gh = gh < 0 ? gh + H : (gh >= H ? gh - H : gh);
      gh = gh % H;  // H == 1 or 2
      float v = 0.f;
      if (gz >= 0 && gz < D && gw >= 0 && gw < W) v = xb[((size_t)gz * H + gh) * W + gw] * sc;
      amax = fmaxf(amax, fabsf(v));
      img[i] = v;
    }
    if (a.status && amax > 65504.f) atomicOr(a.status, 1);
  }
  // this lane's 16 im2col columns: k = ks*16 + half*8 + e = tap index (k >= 27: zero weight, any readable cell)
  int toff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ks * 16 + half * 8 + e;
      const int kz = k / 9, kh = (k / 3) % 3, kw = k % 3;
      toff[ks][e] = k < 27 ? (kz * HP + kh) * WP + kw : 0;
    }
  __syncthreads();
  const int nvox = nz * PV, ntiles = (nvox + 31) / 32;
  const int64_t vox = a.dims.vox();
  const float inv_pv = 1.f / (float)PV, inv_w = 1.f / (float)W;
  for (int ct = 0; ct < a.cout / 32; ++ct) {
    u32x4 w1[2], w2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 wv[2];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 16 + half * 8 + e;
        wv[e >> 2][e & 3] = k < 27 ? a.wpk[((size_t)k * a.cin) * a.cout + ct * 32 + col] : 0.f;
      }
      u32x2 a1, a2, b1, b2;
      split2(wv[0], a1, a2);
      split2(wv[1], b1, b2);
      w1[ks] = u32x4{a1[0], a1[1], b1[0], b1[1]};
      w2[ks] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    }
    // the table rows of a tile as whole 16-byte quads, row 8 k + (lane >> 3), channels 4 (lane & 7) .. + 3 -- the layout the tile is
    // stored in after a transpose through LDS (16 scalar row stores per lane in accumulator layout ran at a third of the HBM rate) --
    // requested ONE TILE AHEAD (round 4): with two waves per SIMD a tile's gather and six MFMAs are ~150 ns, an L2 round trip several
    // times that, and the tile ended up waiting for its own table rows
    auto load_table = [&](int tile, f32x4 (&t4)[4]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int vr = min(tile * 32 + 8 * k + (lane >> 3), nvox - 1);
        t4[k] = *(const f32x4*)(table + ((size_t)z0 * PV + vr) * a.cout + ct * 32 + (lane & 7) * 4);
      }
    };
    f32x4 tb[4], tbn[4];
    load_table(min(wave, ntiles - 1), tb);
    for (int tile = wave; tile < ntiles; tile += 4) {
      const int v = min(tile * 32 + col, nvox - 1);
      // (exact small-integer division by reciprocal: (v + 0.5) / d is never within float error of an integer for v < 2^20; two run-time
      // integer divisions were ~80 of a tile's ~300 instructions)
      const int lz = (int)(((float)v + 0.5f) * inv_pv), p = v - lz * PV, h = (int)(((float)p + 0.5f) * inv_w), w = p - h * W;
      const float* base = img + (lz * HP + h) * WP + w;
      load_table(min(tile + 4, ntiles - 1), tbn);
      f32x16 accA, accB;
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 xv[2];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e >> 2][e & 3] = base[toff[ks][e]];
        u32x2 a1, a2, b1, b2;
        split2(xv[0], a1, a2);
        split2(xv[1], b1, b2);
        const u32x4 x1 = {a1[0], a1[1], b1[0], b1[1]}, x2 = {a2[0], a2[1], b2[0], b2[1]};
        accA = MFMA_F16(x1, w1[ks], accA);
        accB = MFMA_F16(x1, w2[ks], accB);
        accB = MFMA_F16(x2, w1[ks], accB);
      }
      float* tr = trn + wave * (32 * 36);  // this wave's 32 x 32 tile, rows padded to 36 floats
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + col] = accA[r] + accB[r] * (1.f / 2048.f);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes are visible to its reads in order)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 8 * k + (lane >> 3);
        const int vr = tile * 32 + row;
        const f32x4 o = *(const f32x4*)(tr + row * 36 + (lane & 7) * 4) + tb[k];
        if (vr < nvox) *(f32x4*)(a.out + ((size_t)b * vox + (size_t)z0 * PV + vr) * a.cout + ct * 32 + (lane & 7) * 4) = o;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) tb[k] = tbn[k];
    }
  }
}

void launch_init_coord_table(const InitConvArgs& a, hipStream_t s) {
  CD_REQUIRE(a.coord_table && a.cout % 32 == 0 && a.cin >= 1 && a.cin <= 4, "init conv table: bad arguments");
  // the scalar kernel at batch 1 without x
  InitConvArgs t = a;
  t.x = nullptr; t.cx = 1; t.batch = 1; t.out = a.coord_table; t.scale_b = nullptr; t.sigma_b = nullptr;
  dim3 tgrid((unsigned)((a.dims.vox() + 255) / 256), 1u, (unsigned)(a.cout / 32));
  switch (a.cin) {
    case 1: hipLaunchKernelGGL(init_conv_kernel<1>, tgrid, dim3(256), 0, s, t); break;
    case 2: hipLaunchKernelGGL(init_conv_kernel<2>, tgrid, dim3(256), 0, s, t); break;
    case 3: hipLaunchKernelGGL(init_conv_kernel<3>, tgrid, dim3(256), 0, s, t); break;
    case 4: hipLaunchKernelGGL(init_conv_kernel<4>, tgrid, dim3(256), 0, s, t); break;
  }
  CD_HIP(hipGetLastError());
}

void launch_init_conv(const InitConvArgs& a, hipStream_t s) {
  CD_REQUIRE(a.cout % 32 == 0, "init conv: output channels must be a multiple of 32");
  CD_REQUIRE(a.cin >= 1 && a.cin <= 4 && a.cx <= a.cin, "init conv: 1..4 input channels supported");
  const bool full_range = conv_precision() != PREC_F16X2;
  static const bool no_mfma = getenv("CD_NO_INIT_MFMA") != nullptr;
  if (a.coord_table && a.cx == 1 && a.x && !full_range && !no_mfma) {
    prof::Scope scope("init_conv", s, 2.0 * 27 * a.cin * a.cout * (double)a.dims.vox() * a.batch,
                      4.0 * a.batch * (double)a.dims.vox() * (a.cx + a.cout));
    if (!a.table_ready) launch_init_coord_table(a, s);  // (~25 us of scalar-kernel latency: callers that can, keep the table)
    // 2. the x part on the matrix cores: TZ planes per workgroup, about two rounds of workgroups
    const int D = a.dims.d;
    static const int init_wgs = getenv("CD_INIT_WGS") ? atoi(getenv("CD_INIT_WGS")) : 512;
    int slabs = (init_wgs + a.batch - 1) / a.batch;
    slabs = slabs < 1 ? 1 : (slabs > D ? D : slabs);
    int TZ = (D + slabs - 1) / slabs;
    while (TZ > 1 && (size_t)(TZ + 2) * (a.dims.h + 2) * (a.dims.w + 2) * 4 > 60 * 1024) --TZ;
    const size_t lds = (size_t)(TZ + 2) * (a.dims.h + 2) * (a.dims.w + 2) * 4;
    CD_REQUIRE(lds <= 64 * 1024, "init conv: plane too large for the LDS image");
    dim3 grid((unsigned)((D + TZ - 1) / TZ), (unsigned)a.batch);
    hipLaunchKernelGGL(init_conv_f16x2_kernel, grid, dim3(256), lds, s, a, (const float*)a.coord_table, TZ);
    CD_HIP(hipGetLastError());
    return;
  }
  dim3 grid((unsigned)((a.dims.vox() + 255) / 256), (unsigned)a.batch, (unsigned)(a.cout / 32));
  prof::Scope scope("init_conv", s, 2.0 * 27 * a.cin * a.cout * (double)a.dims.vox() * a.batch,
                    4.0 * a.batch * (double)a.dims.vox() * (a.cx + a.cout));
  switch (a.cin) {
    case 1: hipLaunchKernelGGL(init_conv_kernel<1>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(init_conv_kernel<2>, grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(init_conv_kernel<3>, grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(init_conv_kernel<4>, grid, dim3(256), 0, s, a); break;
  }
  CD_HIP(hipGetLastError());
}

}  // namespace cd
