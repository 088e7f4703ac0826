// Shared declarations for the gfx950 CaloDiffusion hot-path library (internal; the public ABI is include/calodiff.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace cd {

// ---- error plumbing ---------------------------------------------------------------------------------
void set_error(const std::string& msg);
struct Fail {
  int code;
  std::string msg;
};
#define CD_HIP(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      throw cd::Fail{-2, std::string(#expr) + ": " + hipGetErrorString(_e)};                          \
  } while (0)
#define CD_REQUIRE(cond, msg)                                                                         \
  do {                                                                                                \
    if (!(cond)) throw cd::Fail{-1, std::string(msg)};                                                \
  } while (0)

// ---- optional per-launch profiling with HIP events (eager mode only; bench.py's roofline leg) -----------------
namespace prof {
bool enabled();
void begin();
// writes a JSON object {"<category>": {"launches": n, "ms": total, "flops": per_launch, "bytes": per_launch}, ...}
int end(char* buf, int cap);
struct Scope {
  Scope(const char* category, hipStream_t s, double flops, double bytes);
  ~Scope();
  int idx;
  hipStream_t stream;
};
}  // namespace prof

// ---- arithmetic of the 3x3x3 / strided / transposed convolutions (process-wide; CD_CONV_PRECISION sets the initial value) ---
//   0 = f16x2 (default: two-term fp16 split, fp16 range), 1 = bf16x3 (exact three-term bf16 split, fp32 range), 2 = f32 MFMA
enum ConvPrecision { PREC_F16X2 = 0, PREC_BF16X3 = 1, PREC_F32 = 2 };
int conv_precision();
void set_conv_precision(int p);
// this thread's override of the process-wide setting (-1 = none): the bf16x3 re-run of a range fallback must not switch the
// kernels of other plans / threads / ranks of the process in mid-call
void set_conv_precision_override(int p);

// ---- channels-last tensor view ------------------------------------------------------------------------
// Internal activation layout: (B, D, H, W, C) fp32, C a multiple of 32 => every voxel is a whole number of 128-B lines.
struct Dims3 {
  int d, h, w;
  __host__ __device__ int64_t vox() const { return (int64_t)d * h * w; }
};

// Packed weight layout shared by every MFMA kernel (32x32x2 f32):
//   wpk[chunk][tap][ct][q][lane][e]   chunk = ci/32, ct = co/32, lane = h*32 + j, m = 4q+e
//   holds W[co = ct*32 + j][ci = chunk*32 + h*16 + m][tap]
// so that one wave-wide 16-B load (fixed q) is 1 KiB contiguous and lane (j,h) receives the B-operand
// values of MFMAs m = 4q..4q+3 for its output column j and k-half h.
inline size_t packed_weight_floats(int cin, int cout, int taps) {
  return (size_t)(cin / 32) * taps * ((cout + 31) / 32) * 2048;
}

// ---- kernel launchers (defined in the .hip files) ---------------------------------------------------------
struct ConvGeom {
  Dims3 in, out;
  int kd, kh, kw;   // kernel extents
  int sz, sh, sw;   // strides
};

// SiLU on the transcendental unit: t * rcp(1 + exp2(-t log2 e)), ~3 ulp (libm's expf + a division is ~30 VALU instructions
// per element in kernels whose staging is VALU-bound)
__device__ __forceinline__ float cd_fast_silu(float t) {
  return t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
}

// GroupNorm fusion around a conv (all optional):
//   coef     [B][Cin][4] = {scale, shift, add, -}: y = act(scale*x + shift) + add applied to the input as it is staged
//            (GroupNorm affine folded per sample and channel by launch_gn_finalize; `add` = the time/cond embedding);
//   ch_part  [B][units][Cout][2] = per-workgroup {sum, sum of squares} of the conv output, units <= ceil(vox/32);
//            *units is set to the number of workgroups per sample that wrote partials, or 0 if the chosen kernel
//            cannot (the caller then runs launch_ch_stats on the output).
// Deferred GroupNorm coefficients (gn_defer.h): instead of a coefficient table the consumer gets the producer's channel
// partials and the affine parameters and folds them in its prologue.  part == nullptr: not deferred.
struct GnDefer {
  const float* part = nullptr;  // [B][units][C][2]
  int units = 0;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* add = nullptr;   // [B][add_ld] added after the activation, or null
  int add_ld = 0;
  int C = 0, groups = 0;
  int64_t vox = 0;
  // training tape: the folded table ([B][C][4]) and {mean, rstd} per (sample, group) ([B][G][2]) are also written to memory, by
  // every workgroup that folds them (the fold is deterministic, so they all store the same values): the backward pass reads what
  // the forward used, and no gn_finalize launch is needed for it
  float* coef_out = nullptr;
  float* stat_out = nullptr;
};

struct ConvFusion {
  const float* coef = nullptr;
  int act = 0;
  float* ch_part = nullptr;
  int* units = nullptr;
  // 16-bit split images of the weights (launch_pack_weights_split16): the bf16x3 image, followed at
  // packed_bf16x3_bytes() by the f16x2 image
  const void* wpk_bf16x3 = nullptr;
  int* status = nullptr;  // device word; bit 0 is set when a value staged for an f16x2 conv exceeds the fp16 range
  // alternative to `coef`: fold the input normalisation from `defer` in the kernel's prologue.  Kernels without that
  // prologue get the table materialised into `coef_buf` ([B][Cin][4]) by a gn_finalize launch first.
  GnDefer defer;
  float* coef_buf = nullptr;
  // max |input| as a float bit pattern in a device word (launch_absmax_bits), or null.  Given, the f16x2 kernels multiply the
  // staged input by 2^s (bringing that maximum to ~2^10) and their output by 2^-s: the input gradients of a conv are
  // O(1e-6), deep in the fp16 subnormals, and the conv is linear.  Needs bias == null and no input normalisation.
  const unsigned* in_absmax = nullptr;
  // out = conv + add_src (a tensor shaped like `out`, bias == null): the identity shortcut's share of a ResnetBlock's input gradient
  // joins the input-gradient conv's epilogue instead of a separate elementwise pass.  The f16x2 kernels (z-slide, flat, small-grid)
  // honour it and set *add_done = 1 on the host; any other kernel ignores it and the caller adds the tensor itself.
  const float* add_src = nullptr;
  int* add_done = nullptr;
  // Output side of a ResnetBlock's second conv on a grid small enough for one workgroup to see a whole (sample, 32-channel tile)
  // (kernels_conv_small.hip): the kernel applies the block's closing GroupNorm + SiLU and adds the shortcut itself,
  //   out = silu(gn(conv + bias)) + (res0 | res1),
  // and sets *gn_out.done = 1 on the host; every other kernel ignores the request and the caller runs gn_apply as before.
  struct GnOut {
    const float* gamma = nullptr;  // (cout); null = no request
    const float* beta = nullptr;
    int groups = 0;
    const float* res0 = nullptr;   // shortcut: (B, vox, res_c0) [and res1: (B, vox, cout - res_c0), an un-materialised concat]
    const float* res1 = nullptr;
    int res_c0 = 0;
    float* part_out = nullptr;     // [B][1][cout][2] channel partials of `out` (for a following PreNorm), or null
    int* done = nullptr;
  } gn_out;
};
// device word holding max |x| (bit pattern) of the tensor last passed to launch_absmax_bits; valid in stream order
const unsigned* launch_absmax_bits(const float* x, size_t n, hipStream_t s);
// the zeroed word for a producer that tracks max |x| itself while writing x; claimed by the next launch_absmax_bits(x)
unsigned* absmax_word_fresh(const float* x, hipStream_t s);
// forget which tensor the word describes (end of the convolution backward that computed it)
void absmax_note_drop();
// 2^s and 2^-s for a tensor whose max |x| has bit pattern mb: s = 10 - floor(log2 max)
__host__ __device__ inline void pow2_scale_for(unsigned mb, float* scale, float* inv) {
  const int e = (int)((mb >> 23) & 0xff) - 127;
  int sexp = mb == 0u ? 0 : 10 - e;
  sexp = sexp < -100 ? -100 : (sexp > 100 ? 100 : sexp);
  union { unsigned u; float f; } a, b;
  a.u = (unsigned)(127 + sexp) << 23;
  b.u = (unsigned)(127 - sexp) << 23;
  *scale = a.f;
  *inv = b.f;
}
inline size_t packed_bf16x3_bytes(int cin, int cout, int taps) {
  return (size_t)(cin / 16) * taps * ((cout + 31) / 32) * 3 * 64 * 16;
}
inline size_t packed_f16x2_bytes(int cin, int cout, int taps) {
  return (size_t)(cin / 16) * taps * ((cout + 31) / 32) * 2 * 64 * 16;
}
inline size_t packed_split16_bytes(int cin, int cout, int taps) {
  return packed_bf16x3_bytes(cin, cout, taps) + packed_f16x2_bytes(cin, cout, taps);
}
void launch_pack_weights_bf16x3(const float* w_torch, void* wpk, int cout, int cin, int taps, hipStream_t s,
                                bool transposed = false, bool flip = false);
void launch_pack_weights_f16x2(const float* w_torch, void* wpk, int cout, int cin, int taps, hipStream_t s,
                               bool transposed = false, bool flip = false);
inline void launch_pack_weights_split16(const float* w_torch, void* wpk, int cout, int cin, int taps, hipStream_t s,
                                        bool transposed = false, bool flip = false) {
  launch_pack_weights_bf16x3(w_torch, wpk, cout, cin, taps, s, transposed, flip);
  launch_pack_weights_f16x2(w_torch, (char*)wpk + packed_bf16x3_bytes(cin, cout, taps), cout, cin, taps, s, transposed, flip);
}

// packed image Wp[co'][ci'][tap'] (see above) of torch weights: transposed = stored [ci'][co'][tap] (ConvTranspose3d, or the
// input gradient of a Conv3d, whose roles of in/out channels swap); flip = tap' <- taps-1-tap' (input gradient, stride 1)
void launch_pack_weights(const float* w_torch, float* wpk, int cout, int cin, int taps, bool transposed, hipStream_t s,
                         bool flip = false);
void launch_pack_init_weights(const float* w_torch, float* wpk, int cout, int cin, hipStream_t s);
// All weights of a plan in two launches (cd_plan_set_weights: the re-pack after every optimizer step was ~400 launches and 3 ms
// of host time per training step).  One job per tensor; null images are skipped.
struct PackJob {
  const float* src;  // the caller's tensor (device)
  float* raw;        // copy in the plan arena
  float* pk;         // f32 MFMA image (kind 1, 2) or the init conv's [tap][ci][cout] image (kind 3), or null
  void* bf3;         // bf16x3 image or null
  void* f16;         // f16x2 image or null
  int cout, cin, taps;
  int kind;          // 0 = raw only, 1 = conv, 2 = transposed conv (images in transposed channel order), 3 = init conv
  int tr, flip;      // (the input-gradient images of the training step: plan.hip dgrad_images) stored [ci][co][tap] / taps reversed;
                     // raw may then be null (numel 0) and src is the plan's own copy of the tensor
  unsigned long long numel, n_pk, n_bf3, n_f16;  // work items of the four phases
};
void launch_pack_jobs(const PackJob* d_jobs, int njobs, hipStream_t s);        // raw copy, f32 / init and bf16x3 images (kernels_conv.hip)
void launch_pack_jobs_f16x2(const PackJob* d_jobs, int njobs, hipStream_t s);  // f16x2 images (kernels_conv_zs.hip)

void launch_conv_mfma(const float* in0, int c0, const float* in1, int c1, const float* wpk, const float* bias, float* out,
                      int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu = ConvFusion());
// z-slide f16x2 kernel for the full-resolution 3x3x3 convs (kernels_conv_zs.hip); false = geometry not eligible
bool try_launch_conv_zslide(const float* in0, int c0, const float* in1, int c1, const void* wpk_f16x2, const float* bias, float* out,
                            int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu);
// whole-sample-in-LDS f16x2 kernel for the deepest levels (kernels_conv_small.hip); false = geometry not eligible
bool try_launch_conv_small(const float* in0, int c0, const float* in1, int c1, const void* wpk_f16x2, const float* bias, float* out,
                           int batch, int cout, const ConvGeom& g, hipStream_t s, const ConvFusion& fu);
void launch_conv_transpose_mfma(const float* in, int cin, const float* wpk, const float* bias, float* out, int batch,
                                int cout, Dims3 din, Dims3 dout, int kz, int sz, hipStream_t s,
                                const void* wpk_f16x2 = nullptr, int* status = nullptr, const unsigned* in_absmax = nullptr);

enum A_Prologue { A_NONE = 0, A_AFFINE = 1, A_SOFTMAX32 = 2, A_EXPNORM = 3 };
struct PointwiseArgs {
  const float* in0 = nullptr;  // (B, vox, ld0) read at channel offset off0, c0 channels
  int ld0 = 0, off0 = 0, c0 = 0;
  const float* in1 = nullptr;  // optional concat source
  int ld1 = 0, c1 = 0;
  const float* wpk = nullptr;
  int64_t w_batch_stride = 0;  // floats between per-sample weight sets (0 = shared)
  const float* bias = nullptr;
  const float* residual = nullptr;  // (B, vox, cout) added in the epilogue
  float* out = nullptr;
  int batch = 0, cout = 0;
  int64_t vox = 0;
  int prologue = A_NONE;
  const float* coef = nullptr;  // A_AFFINE: [B][Cin][4] {scale, shift, -, -} (a folded GroupNorm, see launch_gn_finalize);
                                // A_EXPNORM: [B][Cin][2] {max, 1/sum}: a <- exp(a - max)/sum (the voxel softmax of k)
  int out_ld = 0, out_off = 0;  // output row stride / channel offset (0 = cout, packed)
  float* ch_part = nullptr;     // optional channel partials of the output: [B][ceil(vox/128)][cout][2]
  // Block close of a ResnetBlock whose shortcut is this 1x1 conv (models.py:200): the epilogue adds  silu(gn(gn_res)) + add  --
  // gn_res the block's second conv output (B, vox, cout), normalisation folded from gn_defer's channel partials -- so the
  // shortcut tensor is never written and the block's own elementwise pass (gn_apply) never runs.  `out` may be gn_res itself.
  const float* gn_res = nullptr;
  GnDefer gn_defer;
  // f16x2 image of the weights ([k-step][cout tile][w1 | w2'] as the conv kernels hold them; shared weights only): with prologue
  // A_NONE the products run on the fp16 pipe -- two-term splits, 3 x 32-cycle MFMAs per 16 channels instead of 8 x 64-cycle fp32
  // ones -- in fp16 RANGE: an input beyond it sets bit 0 of *status (the caller's range fallback re-runs at full range, wpk16 = null)
  const void* wpk16 = nullptr;
  int* status = nullptr;
};
inline int pointwise_units(int64_t vox) { return (int)((vox + 127) / 128); }
void launch_pointwise(const PointwiseArgs& a, hipStream_t s);

struct InitConvArgs {
  const float* x = nullptr;       // (B, cx, D, H, W) planar
  int cx = 0;                     // planar channels present in x
  int cin = 0;                    // logical input channels of the conv (cx + synthesised coordinate channels)
  const float* scale_b = nullptr; // per-sample multiplier of channel 0 (c_in) or null: scale_b[b * scale_stride]
  int scale_stride = 1;
  const float* sigma_b = nullptr; // alternative to scale_b: channel 0 is scaled by c_in = 1/sqrt(sigma_b[b]^2 + sigma_data^2)
  float sigma_data = 1.f;         // (get_scaling, loss.py:29-41) so that the conv does not wait for the embedding kernel
  const float* r_w = nullptr;     // coordinate profiles, used when cin > cx
  const float* z_d = nullptr;
  const float* phi_h = nullptr;
  int use_rz = 0, use_phi = 0;
  const float* wpk = nullptr;     // [tap][ci][cout]
  const float* bias = nullptr;
  float* out = nullptr;           // channels-last
  int batch = 0, cout = 0;
  Dims3 dims{};
  float* coord_table = nullptr;   // optional (vox, cout) buffer: enables the matrix-core path (cx == 1), which splits the conv
                                  // into the sample-independent coordinate-channel part (+ bias) and the x part
  bool table_ready = false;       // coord_table already holds that part (launch_init_coord_table); else it is filled first
  int* status = nullptr;          // bit 0: a staged value exceeded the fp16 range (matrix-core path)
};
void launch_init_conv(const InitConvArgs& a, hipStream_t s);
// coordinate channels + bias of the init conv into a.coord_table (vox, cout): changes only with the weights / profiles
void launch_init_coord_table(const InitConvArgs& a, hipStream_t s);

int gn_nsplit_for(int64_t vox, int batch);
// channel partials [B][nsplit][C][2] of a tensor whose producer has no stats epilogue
void launch_ch_stats(const float* x, float* part, int batch, int channels, int64_t vox, int nsplit, hipStream_t s);
// coef[B][C][4] = {rstd*gamma, beta - mean*rstd*gamma, add_bc[b][c] or 0, 0} from `units` channel partials per sample
void launch_gn_finalize(const float* part, int units, const float* gamma, const float* beta, const float* add_bc, int add_ld,
                        float* coef, int batch, int channels, int groups, int64_t vox, hipStream_t s,
                        float* stat_out = nullptr /* [B][G][2] = {mean, rstd}, kept for the backward pass */);

// ---- backward (kernels_bwd.hip) ---------------------------------------------------------------------------------
size_t wgrad_partial_floats(int64_t out_vox, int batch, bool per_sample, int A, int Bc, int T);
// Deferred slot reductions of the weight gradients (training step).  Every launch_wgrad ends in a reduction of its per-workgroup
// partials: ~28 MB read for a few hundred KB of dw whatever the level, 65 launches of 4-7 us per step.  While a queue is current
// (wgrad_queue_set), launch_wgrad takes its partial buffer from the queue's own region instead of the caller's (which the caller
// reuses for its next weight gradient) and records the reduction as a job; wgrad_queue_flush runs all of them in ONE launch (same
// two-level fixed-order sum per output: deterministic).  A queue without room (region too small, job table full) simply leaves
// the reduction where it was.
struct WgradReduceJob {
  const float* partial;
  float* dw;
  int A, Bc, T, nslots, flags /* 1 accumulate, 2 transposed_out */, b_total, b_off;
  unsigned first_block;
};
struct WgradReduceQueue {
  static constexpr int kMax = 80;   // (the job table travels as a kernel argument: 48 B x 80 < 4 KB)
  WgradReduceJob job[kMax];
  int n = 0;
  unsigned blocks = 0;
  float* base = nullptr;            // region the owner keeps until the flush
  size_t cap = 0, used = 0;         // floats
  size_t need = 0;                  // floats all requests of this pass asked for (the owner sizes the next pass's region by it)
};
void wgrad_queue_set(WgradReduceQueue* q);  // nullptr: reductions are launched where they arise
void wgrad_queue_flush(WgradReduceQueue* q, hipStream_t s);
// dW[a][b][tap] (or [b][a][tap] if transposed_out) = sum_{n,o} g[n][o][a] * x[n][in(o,tap)][xoff + b]; see kernels_bwd.hip
void launch_wgrad(const float* g, int A, Dims3 dg, const float* x, int Bc, int xld, int xoff, Dims3 dx, int kd, int kh, int kw,
                  int sz, int sxy, int batch, bool per_sample, float* partial, float* dw, bool accumulate, bool transposed_out,
                  hipStream_t s, int b_total = 0, int b_off = 0,
                  // x is read through silu(coef[0] x + coef[1]) + coef[2] per (sample, channel) ([B][xld][4]): only where
                  // wgrad_x_norm_supported() says so (the fp16-pipe 3x3x3 kernel)
                  const float* xcoef = nullptr);
bool wgrad_x_norm_supported(Dims3 dg, Dims3 dx, int kd, int kh, int kw, int sz, int sxy);
// stride-1 3x3x3 weight gradient on the fp16 matrix pipe (kernels_wgrad16.hip); false = geometry not eligible
bool wgrad_f16x2_eligible(Dims3 d);
bool try_launch_wgrad_f16x2(const float* g, int A, const float* x, int Bc, int xld, int xoff, Dims3 d, int batch, float* partial,
                            unsigned* gmax_word, int* nblk_out, hipStream_t s, const float* xcoef = nullptr);
// the strided (KD, 4, 4) convs between the levels (and the transposed conv with the tensors' roles swapped): kernels_wgrad16.hip
bool try_launch_wgrad_strided_f16x2(const float* g, int A, Dims3 dg, const float* x, int Bc, int xld, int xoff, Dims3 dx, int kd, int sz,
                                    int batch, float* partial, int max_slots, int* nblk_out, hipStream_t s);
void launch_strided_dgrad_naive(const float* dy, const float* w, float* dx, int batch, int cin, int cout, Dims3 din, Dims3 dout,
                                int kd, int sz, hipStream_t s);
void launch_softmax32(const float* qkv, float* qs, int64_t rows, hipStream_t s);
void launch_softmax32_bwd(const float* qs, const float* dqs, float* dqkv, int64_t rows, hipStream_t s);
void launch_ksoftmax_bwd(const float* qkv, const float* dks, const float* kstat, const float* ctx, const float* dctx, float dscale,
                         float* dqkv, int batch, int64_t vox, hipStream_t s);
void launch_pack_sample32(const float* m, float* wpk, int batch, bool transpose, float scale, hipStream_t s);
void launch_pack_sample32_pair(const float* m, float* wpk_plain, float* wpk_tr, int batch, float scale, hipStream_t s);
int head_bwd_blocks(int batch, int64_t vox);
// loss_type (here and below): CD_LOSS_* of calodiff.h -- 0 l2 (hybrid weight), 1 l1, 2 mse, 3 huber (models/loss.py:97-116)
// objective: CD_OBJ_* (objective_residual / objective_weight below); noise is only read for noise_pred
void launch_head_loss_bwd(const float* x0, const float* data, const float* noise, const float* scal, const float* h, const float* wh,
                          float* dh, float* part, float* dwh, float* dbh, int batch, int64_t vox, hipStream_t s, int loss_type = 0,
                          int objective = 0);
struct LinearWgradJob {
  const float* delta;  // (B, delta_ld) rows, nout used
  const float* in;     // (B, in_ld) rows, nin used
  float* dw;           // (nout, nin)
  float* db;           // (nout)
  int nout, nin, delta_ld, in_ld;
};
void launch_linear_wgrad(const LinearWgradJob* jobs_dev, int njobs, int max_elems, int batch, hipStream_t s);
void launch_fold_phi(const float* src, float* dst, int batch, Dims3 d, int C, hipStream_t s);
void launch_add_slices(const float* a, int lda, int aoff, const float* b, int ldb, int boff, float* out, int channels,
                       int64_t rows, hipStream_t s);
void launch_bias_grad(const float* part, int units, int batch, int channels, float* db, bool accumulate, hipStream_t s);
size_t gn_backward_scratch_floats(int batch, int channels, int64_t vox);
// backward of y = act(scale*h + shift) + add: dh, dgamma, dbeta (accumulated if asked), dadd[b][c] (optional)
// Deferred batch reduction of the GroupNorm parameter gradients: one job per layer, passed to the kernel by value
struct GnParamJob {
  const float* sums_bc;  // [batch][channels][4] = {dbeta, dgamma, conv-bias, sum dy} contributions per sample
  float *dgamma, *dbeta, *dbias, *dsumdy;
  int batch, channels;
};
struct GnParamJobs {
  static constexpr int kMax = 64;
  GnParamJob job[kMax];
  int n = 0;
};
struct GnParamQueue {
  GnParamJobs jobs;
  float* next_sums = nullptr;  // bump pointer into a region of gn_param_queue_floats(...) the caller keeps until the flush
};
void launch_gn_param_jobs(const GnParamJobs& jobs, hipStream_t s);
void launch_gn_backward(const float* dy, const float* h, const float* coef, const float* stat, const float* gamma, float* dh,
                        float* dgamma, float* dbeta, float* dadd, int dadd_ld, int batch, int channels, int64_t vox, int groups,
                        int silu, float* scratch, bool accumulate_params, hipStream_t s,
                        // (optional, round 4) the bias gradient of the conv that produced h, and sum_{b,v} dy per channel (the bias
                        // gradient of a conv that adds into y): both fall out of the statistics pass, see kernels_bwd.hip
                        float* dbias = nullptr, float* dsumdy = nullptr,
                        // (optional) queue the batch reduction of the parameter gradients instead of launching it (training step)
                        GnParamQueue* queue = nullptr);
int gn_apply_blocks_per_sample(int batch, int channels, int64_t vox);
// y = act(scale*x + shift) + add (+ residual; residual1/res_c0: shortcut read from a two-source channel concat);
// part_out (optional): channel partials of y, [B][gn_apply_blocks_per_sample][C][2]
// defer (optional): fold the coefficients in the kernel's prologue instead of reading `coef`
void launch_gn_apply(const float* x, float* y, const float* coef, int batch, int channels, int64_t vox, int silu,
                     const float* residual, const float* residual1, int res_c0, float* part_out, hipStream_t s,
                     const GnDefer* defer = nullptr);

int attn_nsplit_for(int64_t vox, int batch);
size_t attn_partial_floats(int batch, int nsplit);
void launch_attn_context(const float* qkv, float* partials, int batch, int64_t vox, int nsplit, hipStream_t s);
void launch_attn_combine(const float* partials, int nsplit, const float* w_out /*torch (C,32)*/, int cout, float* wpk_b,
                         int batch, float scale, hipStream_t s, float* ctx_out = nullptr /* [B][32][32] unscaled context */,
                         float* kstat_out = nullptr /* [B][32][2] = {max, 1/sum} of the k softmax */,
                         bool layout_T = false /* wpk_b in the K order of launch_attn_out instead of the pointwise one */);

// fused linear attention of the sampling path (kernels_attn.hip): qkv is never materialised
int attn_fused_nsplit_for(int64_t vox, int batch);
// whole ResnetBlock in one launch where a sample is <= 128 voxels and the block is 32 channels wide (kernels_conv_small.hip)
bool try_launch_res_block_small(const float* x0, int c0, const float* x1, int c1, const void* w1_f16x2, const float* b1,
                                const float* gn1_gamma, const float* gn1_beta, const float* emb, int emb_ld, const void* w2_f16x2,
                                const float* b2, const float* gn2_gamma, const float* gn2_beta, int groups, const float* res0,
                                const float* res1, int res_c0, float* h1, float* out, float* part_out, int batch, int cout,
                                Dims3 dims, int* status, hipStream_t s);
// The whole deepest level -- downs[-1], mid blocks, ups[0]: six ResnetBlocks and up to three attention blocks -- in ONE launch,
// one workgroup per sample with the activations resident in LDS (kernels_deep.hip).  Block order: downs r1, downs r2, mid1, mid2,
// ups r1 (input = cat(x, skip)), ups r2; attention after downs r2, mid1 and ups r2.
struct DeepLevelDesc {
  Dims3 dims{};
  int Ca = 0, Cb = 0, groups = 0;  // channels entering / leaving the level (layer_sizes[-2]) and inside it (layer_sizes[-1])
  struct Res {
    int c0 = 0, c1 = 0, cout = 0;
    const void *w1 = nullptr, *w2 = nullptr;  // f16x2 images of the two 3x3x3 convs
    const float *b1 = nullptr, *b2 = nullptr, *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
    const float* emb = nullptr;               // (B, emb_ld) slice or null
    int emb_ld = 0;
    const void* wres = nullptr;               // f16x2 image of the 1x1 shortcut conv or null (identity)
    const float* bres = nullptr;
  } res[6];
  struct Attn {
    int C = 0;
    const float *ng = nullptr, *nb = nullptr, *wout = nullptr, *bout = nullptr, *gg = nullptr, *gb = nullptr;
    const void* wqkv = nullptr;               // f16x2 image of to_qkv
  } attn[3];
  int has_attn[3] = {0, 0, 0};
};
bool deep_level_eligible(const DeepLevelDesc& d);
void launch_deep_level(const DeepLevelDesc& d, const float* x_in, float* x_out, int batch, int* status, hipStream_t s);
// whole attention block in one launch for small grids (kernels_attn.hip: attn_small_kernel)
bool attn_small_eligible(int64_t vox);
void launch_attn_small(const float* x, int C, const float* coef, const void* wqkv_f16x2, float* partials, const float* w_out,
                       float scale, const float* bias, const float* out_gamma, const float* out_beta, float* y, float* ch_part,
                       int batch, int64_t vox, hipStream_t s, const GnDefer* defer, int* status = nullptr,
                       // capacity of partials ([B][max_parts][1088]) and ch_part ([B][max_parts][C][2]) in parts per sample: with
                       // CD_ATTN_COOP the sample's voxels are dealt to up to that many co-operating workgroups
                       int max_parts = 1);
// moments ([B][nsplit][1056], attn_moment_floats): pass 1 also accumulates the first and second moments of softmax(q), from which
// pass 2 (given the same buffer and the closing GroupNorm's parameters) writes the BLOCK's output gn(y) + x directly -- no y
// tensor, no channel sums, no gn_apply pass
bool attn_moments_eligible(int C);
size_t attn_moment_floats(int batch, int nsplit);
void launch_attn_kv_context(const float* x, int C, const float* coef, const void* wqkv_f16x2, float* partials, int batch,
                            int64_t vox, int nsplit, hipStream_t s, const GnDefer* defer = nullptr, int* status = nullptr,
                            float* moments = nullptr);
void launch_attn_out(const float* x, int C, const float* coef, const void* wqkv_f16x2, const float* wT_b, const float* bias,
                     float* y, float* ch_part /* [B][nsplit][C][2] */, int batch, int64_t vox, int nsplit, hipStream_t s,
                     const GnDefer* defer = nullptr,
                     // wT_b == null: every workgroup merges the pass-1 partials and folds W_out itself (no combine launch)
                     const float* partials = nullptr, const float* w_out = nullptr, float scale = 0.f, int* status = nullptr,
                     const float* moments = nullptr, const float* out_gamma = nullptr, const float* out_beta = nullptr);

struct EmbedLayer {
  const float* w;  // (cout, 128) torch layout
  const float* b;
  int cout;
  int offset;  // column offset in the per-sample embedding row
};
struct EmbedArgs {
  const float* cond = nullptr;  // (B, cond_size)
  const float* time_or_sigma = nullptr;  // (B,)
  int time_kind = 0;
  float sigma_data = 1.f;
  int cond_size = 0, cond_hidden = 0, half = 0;
  const float *tw1, *tb1, *tw2, *tb2, *tw3, *tb3;
  const float *cw1, *cb1, *cw2, *cb2, *cw3, *cb3;
  const EmbedLayer* layers = nullptr;  // device array
  int n_layers = 0;
  float* emb = nullptr;  // (B, emb_ld)
  int emb_ld = 0;
  float* scal = nullptr;  // (B, 4): c_in, c_skip, c_out, sigma
  int batch = 0;
  // SinusoidalPositionEmbeddings(half/2) (models.py:132-144) in place of the first Linear+GELU of the time / cond branch
  // (CondUnet(time_embed=True / cond_embed=True), models.py:578-601): tw1/tb1 resp. cw1/cb1 are unused then, and cond is (B,)
  int time_sin = 0, cond_sin = 0;
  // Several steps of one trajectory in one launch (the sampler loop's embeddings, a chunk of steps ahead): workgroup b embeds
  // sample b % cond_rows at the time value time_or_sigma[(b / cond_rows) * time_stride].  cond_rows = 0: one step, as before.
  int cond_rows = 0, time_stride = 0;
  // The ResnetBlock projections are linear in SiLU(cat(t, c)) (models.py:176-180, 707): W SiLU(cat(t, c)) + b = [W_t SiLU(t) + b] +
  // [W_c SiLU(c)].  part = 1: row b is a TIME row (time value time_or_sigma[b * time_stride], no condition): the first bracket and
  // the scalings; part = 2: row b is a CONDITION row: the second bracket, no bias, no scalings.  0: the whole expression.
  int part = 0;
};
void launch_embed(const EmbedArgs& a, hipStream_t s);
void launch_silu_linear(const float* cond, const float* w, const float* bias, float* out, int batch, int nin, int nout,
                        hipStream_t s);

struct HeadArgs {
  const float* h = nullptr;   // (B, vox, 32) channels-last
  const float* w = nullptr;   // (32,) torch layout of the 1x1x1 head
  const float* bias = nullptr;
  const float* x = nullptr;   // (B, vox) network input before c_in scaling (null for raw unet_forward)
  const float* scal = nullptr;  // (B,4) or null
  int objective = 0;
  float* out = nullptr;       // (B, vox): F (raw) or x0
  int batch = 0;
  int64_t vox = 0;
  // The closing GroupNorm + SiLU + identity shortcut of the final ResnetBlock applied on the fly (defer.part != null): h is then
  // that block's second conv output and `res` its input; saves the block's own elementwise pass over the level-0 tensor.
  GnDefer defer;
  const float* res = nullptr;
  // DDim.__call__'s update of the running sample (models/sample.py:88-107) in the same pass (needs x and scal): with
  // stepvals = {sigma, sigma_prev*[t>0], ddim_sigma, denom}:  x_next = x0 + sigma_prev (x - x0) / sigma + ddim_sigma noise / denom
  // -- the arithmetic of launch_ddim_update, whose launch and pass over x / x0 it saves.  upd_x_next may alias x.
  const float* upd_stepvals = nullptr;  // null: no update
  const float* upd_noise = nullptr;
  float* upd_x_next = nullptr;
  float* upd_xs = nullptr;
  float* upd_x0s = nullptr;
};
void launch_head(const HeadArgs& a, hipStream_t s);

// x_next = x0 + sigma_prev*((x - x0)/sigma) + ddim_sigma*noise/denom, scalars read from stepvals[0..3]
void launch_ddim_update(const float* x, const float* x0, const float* noise, const float* stepvals, float* x_next,
                        float* xs_slot, float* x0s_slot, int64_t n, hipStream_t s);
// stepvals <- table[*counter]; sigma_b[0..B) <- stepvals.sigma; (*counter)++.  chunk (optional): the step's slice of the
// embeddings / scalings computed a chunk of steps ahead -- slot (*counter) % chunk_steps of emb_src / scal_src -> emb_dst / scal_dst
struct StepChunk {
  const float* emb_src = nullptr;  // [chunk_steps][emb_floats]
  float* emb_dst = nullptr;
  int emb_floats = 0;
  const float* scal_src = nullptr;  // [chunk_steps][scal_floats]
  float* scal_dst = nullptr;
  int scal_floats = 0;
  int chunk_steps = 0;
  // separable form: emb_src / scal_src hold ONE row per step (EmbedArgs::part = 1), emb_cond one row per sample (part = 2):
  // emb_dst[b] = emb_src[slot] + emb_cond[b], scal_dst[b] = scal_src[slot]; emb_floats / scal_floats are then the ROW lengths
  const float* emb_cond = nullptr;
};
void launch_load_step(const float* table, int* counter, float* stepvals, float* sigma_b, int batch, hipStream_t s,
                      const StepChunk* chunk = nullptr);
void launch_scale(const float* x, float* y, const float* stepvals_sigma, int64_t n, hipStream_t s);
void launch_scale_imm(const float* x, float* y, float scale, int64_t n, hipStream_t s);
void launch_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s);
// noise tensor number (*step_counter - 1) * per_step + index of a sampler run, from {seed, base offset, stride} in device
// memory: stream position = base + tensor number * stride (graph-replayable; stride = the GLOBAL tensor size, so that batch
// shards on different GPUs draw their slices of the one stream a single-GPU run of the whole batch would use)
void launch_randn_step(float* out, int64_t n, const uint64_t* seed_offset_stride_dev, const int* step_counter, hipStream_t s,
                       int per_step = 1, int index = 0);
// generic sampler programs (cd_sampler_run): per-step scalars are columns of row (*step_counter - 1) of a device table
void launch_step_advance(int* counter, hipStream_t s);  // (*counter)++
void launch_or_word(int* word, int bits, hipStream_t s);  // *word |= bits
void launch_fill_from_table(float* dst, int count, const float* table, int ncol, int col, const int* step_counter, hipStream_t s);
// out[i] = sum_k table[row][col + k] * src[k][i]   (nsrc <= 6; out may alias a source)
void launch_lincomb(float* out, const float* const* src, int nsrc, const float* table, int ncol, int col, const int* step_counter,
                    int64_t n, hipStream_t s);
void launch_lincomb_div(float* out, const float* const* src, int nsrc, const float* table, int ncol, int col, const int* step_counter,
                    int64_t n, hipStream_t s);
// traj[(*step_counter - 1) * n + i] = src[i]
void launch_record_step(float* traj, const float* src, const int* step_counter, int64_t n, hipStream_t s);
void launch_axpy_sigma(const float* data, const float* noise, const float* sigma_b, float* out, int batch, int64_t per,
                       hipStream_t s);
// The training objectives of models/loss.py on top of the denoiser's output `out` (calodiffusion.py:161-169), objective =
// CD_OBJ_*:  what the loss compares, its per-sample weight (used by 'l2' only) and d pred / d F for the backward pass
//   0 hybrid_weight (:163-179)  pred = out = c_skip x + c_out F        target = data    w = 1 + sigma^-2   dpred/dF = c_out
//   1 noise_pred    (:181-196)  pred = (data - (data - sigma out)) / sigma, out = x - sigma F   target = noise   w = 1   dpred/dF = -sigma
//   2 mean_pred     (:198-210)  pred = out = F                         target = data    w = sigma^-2       dpred/dF = 1
// (noise_pred's round trip through x0_pred is evaluated as written, product / difference / quotient each rounded on its own)
__device__ __forceinline__ float objective_residual(int objective, float out, float data, float noise, float sigma) {
#pragma clang fp contract(off)
  if (objective == 1) {
    const float so = sigma * out;
    const float x0p = data - so;
    const float num = data - x0p;
    const float pred = num / sigma;
    return pred - noise;
  }
  return out - data;
}
__device__ __forceinline__ float objective_weight(int objective, int loss_type, float sg) {
  if (loss_type != 0 || objective == 1) return 1.0f;
  return objective == 0 ? 1.0f + 1.0f / (sg * sg) : 1.0f / (sg * sg);
}

void launch_loss_partial(const float* x0, const float* data, const float* noise, const float* sigma_b, double* partial, int batch,
                         int64_t per, hipStream_t s, int loss_type = 0, int objective = 0);
void launch_loss_final(const double* partial, const float* sigma_b, double* loss, int batch, int64_t per, hipStream_t s,
                       int loss_type = 0, int objective = 0);
void launch_transpose_to_cl(const float* ncdhw, float* ndhwc, int batch, int channels, int64_t vox, hipStream_t s);
void launch_transpose_to_planar(const float* ndhwc, float* ncdhw, int batch, int channels, int64_t vox, hipStream_t s);

struct ReverseNormArgs {
  const float* voxels;  // (B, 1, D, H, W) normalised-space showers
  const float* energy;  // (B) incident energies (physical units)
  const float* layerE;  // (B, 1 + D) normalised {total, layers} or null
  float* out;           // (B, D*H*W)
  int batch, D, H, W, layer_mode;
  float logit_mean, logit_std, totalE_mean, totalE_std, layers_mean, layers_std, max_deposit, ecut;
  // 0: everything (CaloChallenge regular grids); 1: un-normalise + inverse logit only; 2: layer renormalisation + scaling of
  // already-decoded showers (the two halves of ReverseNormHGCal around its geometry decode, utils/HGCal_utils.py:167-292)
  int stage = 0;
  float alpha = 1e-6f;      // reverse_logit's alpha (utils.py:233: 1e-6; HGCal_utils.py:13: 1e-8)
  float layer_eps = 1e-6f;  // "essentially zero" layer (utils.py:539-547: 1e-6; HGCal_utils.py:262-268: 1e-8)
};
void launch_reverse_norm(const ReverseNormArgs& a, hipStream_t s);

// LayerDiffusion's layer-energy MLP: raw forward (mode 0), EDM denoise (mode 1) or a whole sampler trajectory (mode 2)
struct LayerMlpArgs {
  const float* w[64];    // (weight, bias) per Linear in the reference ResNet's state_dict order
  int dim_in, hidden, cond_emb, cond_size, n_res, time_kind, objective, mode, batch, n_steps;
  float sigma_data;
  const float* x;        // (B, dim_in): input / start noise
  const float* cond;     // (B, cond_size)
  const float* tsig;     // (B): time (mode 0) or sigma (mode 1)
  const float* table;    // device (n_steps, 4) step table (mode 2)
  const float* noise;    // (n_steps, B, dim_in) or null
  float* out;            // (B, dim_in)
  float* xs;             // (n_steps, B, dim_in) or null
  float* x0s;
};
void launch_layer_mlp(const LayerMlpArgs& a, hipStream_t s);

// training step of the layer-energy MLP (kernels_mlp_train.hip)
struct LayerTapeLayout {
  int xin, t_in, a1t, a2t, cin, a1c, a2c, g, hprev[8], h1[8], hfin;       // inputs of the Linear layers
  int dpred, dv[8], du[8], de[8], dh0, d3t, d2t, d1t, d3c, d2c, d1c;     // their output deltas
  int total;
};
LayerTapeLayout layer_tape_layout(int dim, int hidden, int cond_emb, int cond_size, int n_res);
struct LayerMlpTrainArgs {
  const float* w[64];
  int dim_in, hidden, cond_emb, cond_size, n_res, time_kind, batch;
  int loss_type = 0;  // CD_LOSS_* (Loss._loss, models/loss.py:97-116): 0 l2 (weighted), 1 l1, 2 mse, 3 huber
  float sigma_data;
  const float *data, *noise, *sigma, *cond;  // (B, dim), (B, dim), (B), (B, cond_size)
  LayerTapeLayout layout;
  float* tape;        // [B][layout.total]
  double* loss_part;  // [B]
};
size_t layer_train_workspace_bytes(const LayerMlpTrainArgs& a);
void launch_layer_mlp_train(LayerMlpTrainArgs a, float* grads, double* loss_out, void* workspace, hipStream_t s);

// fused Adam over up to 48 tensors per launch (kernel-argument table)
struct AdamChunk {
  float* p[48];
  const float* g[48];
  float* m[48];
  float* v[48];
  int64_t n[48];
};
void launch_adam(const AdamChunk& c, int ntensors, int64_t max_numel, double lr, double beta1, double beta2, float eps,
                 float weight_decay, int step, hipStream_t s);

size_t init_wgrad_partial_floats(int batch, int64_t vox, int cin, int cout);
void launch_init_wgrad(const InitConvArgs& a, const float* g, float* part, float* dw, hipStream_t s);
// the same through the general weight-gradient kernels (padded 32-channel input); scratch: init_wgrad_mfma_floats floats
size_t init_wgrad_mfma_floats(int batch, int64_t vox, int cout);
void launch_init_wgrad_mfma(const InitConvArgs& a, const float* g, float* scratch, float* dw, hipStream_t s);
// per-sample record the embedding backward leaves for the Linear weight gradients
struct EmbedTapeLayout {
  int t_in, a1t, a2t, a1c, a2c, sc;      // inputs of the Linears (time: 1, q, half; cond: cond(copied), hidden, half; proj: 2*half)
  int cond_in;
  int d1t, d2t, d3t, d1c, d2c, d3c;      // deltas at the Linear outputs
  int total;
};
__host__ __device__ inline EmbedTapeLayout embed_tape_layout(int cond_size, int hidden, int half) {
  EmbedTapeLayout L;
  int o = 0;
  const int q = half / 2;
  L.t_in = o; o += 1;
  L.a1t = o; o += q;
  L.a2t = o; o += half;
  L.cond_in = o; o += cond_size;
  L.a1c = o; o += hidden;
  L.a2c = o; o += half;
  L.sc = o; o += 2 * half;
  L.d1t = o; o += q;
  L.d2t = o; o += half;
  L.d3t = o; o += half;
  L.d1c = o; o += hidden;
  L.d2c = o; o += half;
  L.d3c = o; o += half;
  L.total = (o + 3) & ~3;
  return L;
}

size_t embed_tape_floats(int cond_size, int hidden, int half);
void launch_embed_bwd(const EmbedArgs& a, const float* demb, float* tape, hipStream_t s);

}  // namespace cd
