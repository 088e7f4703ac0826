/*
 * calodiff.h -- C ABI of the MI355X (gfx950) CaloDiffusion denoising hot path.
 *
 * The reference (OzAmram/CaloDiffusion) has no FFI of its own: its seam for this path is a Python
 * class protocol.  Each entry point below states the reference interface it
 * stands behind (file:line into the reference tree).  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative CD_E* code; cd_last_error() gives the message
 *     (thread-local, valid until the next failing call on that thread).  Nothing throws across the ABI.
 *   - all tensor pointers are DEVICE pointers to contiguous fp32 unless marked "host".
 *     User-facing activations are NCDHW (D = layer/z, H = phi, periodic, W = r), as in the reference
 *     (calodiffusion/models/models.py:26,66).  Channels-last (NDHWC) is the library's internal layout
 *     and appears only in the cd_op_* primitive entry points, which say so.
 *   - memory is owned by the caller (PyTorch's caching allocator in the shipped host code).  The
 *     library allocates only plan-private metadata and the packed-weight arena at plan creation /
 *     cd_plan_set_weight time, never inside a compute call (compute calls are hipGraph-capturable).
 *   - `stream` is a hipStream_t passed as void*; all work of a call is enqueued on it and the call
 *     returns without synchronising.
 *   - a plan is not thread-safe; use one plan per device per process.
 */
#ifndef CALODIFF_H
#define CALODIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CD_OK 0
#define CD_EINVAL -1       /* bad argument / unsupported configuration */
#define CD_EHIP -2         /* HIP runtime error */
#define CD_ENOGPU -3       /* no gfx950 device visible */
#define CD_EWEIGHTS -4     /* a weight tensor is missing or has the wrong size */
#define CD_EWORKSPACE -5   /* workspace too small */

#define CD_MAX_SIZES 8

/* Bumped whenever a struct layout or an argument list of this header changes.  cd_abi_version() returns the value the
 * library was built with: a binder compares it with the CD_ABI_VERSION it was written against before any other call, and
 * every descriptor struct starts with its own sizeof (struct_size), which the library checks -- a binder written against an
 * older header gets CD_EINVAL from cd_plan_create instead of the library reading past its struct. */
#define CD_ABI_VERSION 3

#define CD_TIME_LOG 0      /* t_emb = 0.5*ln(sigma)         (calodiffusion.py:150) */
#define CD_TIME_SIGMA 1    /* t_emb = sigma/sqrt(1+sigma^2) (calodiffusion.py:149) */
#define CD_TIME_RAW 2      /* t_emb = the value passed (CondUnet.forward's `time` argument) */

#define CD_OBJ_HYBRID 0     /* c_skip*x + c_out*F  (calodiffusion.py:166-167) */
#define CD_OBJ_NOISE_PRED 1 /* x - sigma*F         (calodiffusion.py:161-162) */
#define CD_OBJ_MEAN_PRED 2  /* F                   (calodiffusion.py:164-165) */

typedef struct CdPlan CdPlan;

/* Mirrors the arguments of CondUnet.__init__ (models/models.py:525-543) as CaloDiffusion.init_model
 * derives them from the config (models/calodiffusion.py:39-81). */
typedef struct CdUnetDesc {
  uint32_t struct_size;            /* = sizeof(CdUnetDesc) of the header the caller was built against */
  int32_t grid[3];                 /* D, H, W of SHAPE_FINAL */
  int32_t in_channels;             /* `channels`: 1 (+2 if R_Z_INPUT) (+1 if PHI_INPUT) */
  int32_t n_sizes;                 /* len(LAYER_SIZE_UNET) */
  int32_t layer_sizes[CD_MAX_SIZES];
  int32_t groups;                  /* BLOCK_GROUPS (8) */
  int32_t block_attn, mid_attn, compress_z;
  int32_t cond_size;               /* width of cat(E, layers) */
  int32_t cond_dim;                /* COND_SIZE_UNET */
  int32_t rz_input, phi_input;     /* which coordinate channels cd_denoise synthesises */
  int32_t time_embed_kind;         /* CD_TIME_* used by cd_denoise */
  int32_t objective;               /* CD_OBJ_*  used by cd_denoise */
  float sigma_data;                /* Loss.sigma_data (models/loss.py:18-25) */
  /* CondUnet(time_embed=True / cond_embed=True): SinusoidalPositionEmbeddings (models.py:132-144, 578-601) instead of the
   * first Linear of the time / cond MLP.  Reachable through cd_unet_forward only: the reference's own denoise path raises
   * KeyError for TIME_EMBED 'sin' (calodiffusion.py:148-152), and so do cd_denoise / the samplers / cd_train_step. */
  int32_t time_sin, cond_sin;
} CdUnetDesc;

/* One row per sampler loop iteration; computed on the host exactly as DDim.__call__ does in fp32
 * (models/sample.py:45-101): sigma = sqrt(1-abar_t)/sqrt(abar_t); sigma_prev = sqrt(1-abar_prev-ddim_sigma^2)/denom
 * (already multiplied by the t>0 mask); ddim_sigma = eta*sqrt(...); denom = sqrt(abar_{max(t-1,0)}). */
typedef struct CdStep {
  float sigma;
  float sigma_prev_masked;
  float ddim_sigma;
  float denom;
} CdStep;

const char* cd_last_error(void);
int cd_abi_version(void);  /* CD_ABI_VERSION of the build */
/* 0 if a gfx950 device is usable by this process, CD_ENOGPU otherwise. Fills name (may be NULL). */
int cd_device_check(char* name, int cap);

/* ---- plan ------------------------------------------------------------------------------------------- */
/* Replaces CondUnet.__init__ + CaloDiffusion.init_model (models.py:525-699, calodiffusion.py:39-81). */
int cd_plan_create(const CdUnetDesc* desc, CdPlan** plan);
int cd_plan_destroy(CdPlan* plan);
/* Names follow CondUnet.state_dict() (e.g. "downs.0.0.block1.proj.conv.weight"); iteration helpers so the host
 * can check it feeds every tensor.  *numel is the element count the plan expects. */
int cd_plan_num_weights(const CdPlan* plan, int* n);
int cd_plan_weight_name(const CdPlan* plan, int idx, char* name, int cap, int64_t* numel);
/* Copies/re-packs one state_dict tensor (device pointer, torch layout) into the plan's arena on `stream`.
 * Replaces nn.Module.load_state_dict for this path (calodiffusion.py:31-37). Call again after an optimizer step. */
int cd_plan_set_weight(CdPlan* plan, const char* name, const float* dev_ptr, int64_t numel, void* stream);
/* The same for every tensor at once: dev_ptrs[i] is the tensor cd_plan_weight_name(plan, i, ...) names (n = cd_plan_num_weights),
 * all copied and re-packed by two launches.  This is the call for the training loop -- after optimizer.step() every parameter
 * has changed (train/train.py:144-170); one tensor at a time that was ~400 launches and 3 ms of host time per step. */
int cd_plan_set_weights(CdPlan* plan, int n, const float* const* dev_ptrs, void* stream);
/* Host arrays: the 1-D profiles of the constant R (len W), Z (len D) and phi (len H) input images
 * (utils/utils.py:33-150, calodiffusion.py:17-20). */
int cd_plan_set_coords(CdPlan* plan, const float* r_w, const float* z_d, const float* phi_h, void* stream);
int cd_plan_workspace_bytes(CdPlan* plan, int batch, size_t* bytes);

/* ---- hot path ------------------------------------------------------------------------------------------ */
/* CondUnet.forward(x, cond, time) (models.py:701-748).  x: (B, in_channels, D, H, W); cond: (B, cond_size);
 * time: (B,); out: (B, 1, D, H, W). */
int cd_unet_forward(CdPlan* plan, int batch, const float* x, const float* cond, const float* time, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);
/* CaloDiffusion.denoise / __call__ (calodiffusion.py:154-173) incl. Loss.get_scaling (loss.py:29-41),
 * do_time_embed (calodiffusion.py:144-152), forward + add_RZPhi (calodiffusion.py:86-98,121-142).
 * x: (B,1,D,H,W); sigma: (B,); cond: (B, cond_size) = cat(E, layers); out: (B,1,D,H,W). */
int cd_denoise(CdPlan* plan, int batch, const float* x, const float* sigma, const float* cond, float* out,
               void* workspace, size_t workspace_bytes, void* stream);
/* cd_denoise with the range fallback of the sampler entry points (below): if an operand of the fp16-pipe kernels left the fp16
 * range during the call, the call is run again with the full-range kernels (bf16x3 convolutions, f32-MFMA attention) before it
 * returns and *fell_back is set to 1 (fell_back == NULL: cd_plan_status reports bit 1 instead).  Unlike cd_denoise it SYNCHRONISES `stream`
 * (to read the flag) and is therefore not graph-capturable: it is the entry point for samplers that call the model back from
 * host code (models/sample.py: `model(x, sigma=, E=, layers=)`, e.g. DPMAdaptive :188-309), which must not die mid-trajectory. */
int cd_denoise_safe(CdPlan* plan, int batch, const float* x, const float* sigma, const float* cond, float* out,
                    void* workspace, size_t workspace_bytes, int* fell_back, void* stream);
/* DDim.__call__ / DDPM (models/sample.py:41-121) with Diffusion.sample's start tensor (diffusion.py:77-104).
 * start: (B,1,D,H,W) unit normal; steps: host array of n_steps rows; x_out: (B,1,D,H,W).
 * step_noise: NULL (DDIM, eta = 0: the reference draws and discards it) or device (n_steps, B,1,D,H,W);
 * if NULL and any ddim_sigma != 0 the noise comes from the device Philox stream (seed, offset).
 * xs / x0s: NULL or device (n_steps, B,1,D,H,W) trajectories (`debug`). use_graph: capture one step as a
 * hipGraph and replay it. */
int cd_ddim_sample(CdPlan* plan, int batch, const float* start, const float* cond, const CdStep* steps, int n_steps,
                   const float* step_noise, uint64_t seed, uint64_t offset, uint64_t noise_stride, float* x_out, float* xs,
                   float* x0s, int use_graph, void* workspace, size_t workspace_bytes, void* stream);
/* noise_stride (both sampler entry points): distance in the Philox stream between the noise tensors of consecutive draws;
 * 0 = this call's own tensor size.  A rank holding rows [lo, hi) of a global batch passes offset + lo * voxels and
 * noise_stride = global_batch * voxels: the union of the shards then IS the single-GPU result of the same seed.
 *
 * Range fallback (both sampler entry points, cd_denoise_safe): the default arithmetic (f16x2 convolutions, the fused
 * attention's fp16-pipe products) covers the fp16 range only.  If an operand leaves it during the call (a flag private to the
 * call: a bit 0 left in the sticky word by an earlier cd_denoise is neither consumed nor lost) the call re-runs the whole
 * trajectory with the full-range kernels -- exact bf16x3 convolutions, attention on the f32-input MFMA -- before it returns,
 * synchronising `stream` for the check; cd_plan_status then reports bit 1 (fallback taken).  The switch of arithmetic is local
 * to the calling thread: other plans / threads of the process keep their kernels. */

/* ---- every other sampler of models/sample.py on the same device loop ------------------------------------------------
 * A sampler is a "step program": per step a short list of ops over a few (B,1,D,H,W) buffers, whose scalars are columns of
 * that step's row of a host coefficient table.  Buffer 0 is the running sample x (= x_out); buffers 1 .. n_bufs-1 live in
 * the workspace and start as zeros.  Steps that share one op list (op_begin == NULL) are captured once as a hipGraph and
 * replayed (the device step counter selects the table row, the Philox position and the trajectory slot); otherwise
 * step i runs ops[op_begin[i] .. op_begin[i+1]) eagerly (Restart, DPM-Solver-fast).
 * The host side (calodiffusion_amd/sample.py) builds the programs of EDM Euler(+churn) / Heun / DPM2 (sample.py:577-727,
 * 771-851), LMS (:729-769), Restart (:853-954), DPM / DPM++2S / DPM++2M (:124-186, 311-344, 415-449) and Consistency
 * (:957-1011). */
#define CD_SOP_LINCOMB 0 /* buf[dst] = sum_k coef[col + k] * buf[src[k]], k < nsrc <= 6 (dst may be a source) */
#define CD_SOP_DENOISE 1 /* buf[dst] = denoise(buf[src[0]], sigma = coef[col])  (CaloDiffusion.denoise, as cd_denoise) */
#define CD_SOP_RANDN 2   /* buf[dst] = unit normals: the next tensor of step_noise, or of the Philox stream */
#define CD_SOP_RECORD 3  /* trajectory slot of this step <- buf[src[0]]; dst: 0 = xs, 1 = x0s (skipped if that pointer is NULL) */
#define CD_SOP_LINDIV 4  /* buf[dst] = (((coef[col] * buf[src[0]]) + coef[col+1] * buf[src[1]]) + ...) / coef[col + nsrc]: like \
                            LINCOMB, but in the operation order of a chain of torch elementwise ops -- every product, sum and the \
                            final division rounded to fp32 on its own, no fused multiply-add (DPM-Solver's eps = (x - D) / sigma \
                            and its cancelling updates, utils/sampling.py:402-456) */
typedef struct CdSamplerOp {
  int32_t kind, dst, nsrc;
  int32_t src[6];
  int32_t col;
} CdSamplerOp;
int cd_plan_sampler_workspace_bytes(CdPlan* plan, int batch, int n_bufs, int n_steps, int n_coef, size_t* bytes);
/* start: (B,1,D,H,W) unit normal, x = start * start_scale first.  coefs: HOST (n_steps, n_coef) fp32.  ops: n_ops entries
 * (op_begin == NULL: the one list of every step; else op_begin has n_steps + 1 entries).  step_noise: NULL or DEVICE
 * (number of RANDN ops executed, B,1,D,H,W), consumed in execution order.  xs / x0s: NULL or (n_steps, B,1,D,H,W). */
int cd_sampler_run(CdPlan* plan, int batch, const float* start, float start_scale, const float* cond, int n_bufs, int n_steps,
                   const CdSamplerOp* ops, int n_ops, const int32_t* op_begin, const float* coefs, int n_coef,
                   const float* step_noise, uint64_t seed, uint64_t offset, uint64_t noise_stride, float* x_out, float* xs,
                   float* x0s, int use_graph, void* workspace, size_t workspace_bytes, void* stream);
/* Diffusion.noise_generation (diffusion.py:58-61): n unit normals from Philox4x32-10 + Box-Muller;
 * element i depends only on (seed, offset + i), so shards of one global stream can be drawn per rank. */
int cd_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* Loss.__call__ + hybrid_weight.loss_function + l2_loss forward value (models/loss.py:103-104,118-142,163-179):
 * x_noisy = data + sigma*noise; x0 = denoise(x_noisy); loss = sum(w (x0-data)^2) / (mean(w) numel), w = 1 + sigma^-2.
 * sigma: (B,) device. loss_out: 1 double on device. */
int cd_loss_hybrid_l2(CdPlan* plan, int batch, const float* data, const float* noise, const float* sigma,
                      const float* cond, double* loss_out, void* workspace, size_t workspace_bytes, void* stream);
/* The same for every LOSS_TYPE of Loss._loss (models/loss.py:97-116) under hybrid_weight.loss_function (:163-179), which passes
 * (pred = x0, target = data, weight = 1 + sigma^-2):
 *   CD_LOSS_L2    sum(w (x0-data)^2) / (mean(w) numel)          (the only one that uses the weight; the shipped configs)
 *   CD_LOSS_L1    torch.nn.functional.l1_loss:        mean |x0 - data|
 *   CD_LOSS_MSE   torch.nn.functional.mse_loss:       mean (x0 - data)^2
 *   CD_LOSS_HUBER torch.nn.functional.smooth_l1_loss: mean of d^2/2 where |d| < 1, |d| - 1/2 elsewhere (the reference's CI
 *                 fixture trains with it, tests/test_execution.py:94) */
#define CD_LOSS_L2 0
#define CD_LOSS_L1 1
#define CD_LOSS_MSE 2
#define CD_LOSS_HUBER 3
/* The plan's objective (CdUnetDesc.objective) selects which loss class of models/loss.py this is -- the entry point keeps its
 * name from the shipped configs' hybrid_weight:
 *   CD_OBJ_HYBRID     hybrid_weight (:163-179)  pred = denoise(x_noisy), target = data, weight 1 + sigma^-2
 *   CD_OBJ_NOISE_PRED noise_pred    (:181-196)  pred = (data - (data - sigma denoise(x_noisy))) / sigma, target = noise, weight 1
 *   CD_OBJ_MEAN_PRED  mean_pred     (:198-210)  pred = denoise(x_noisy) = F, target = data, weight sigma^-2
 * (the weight enters CD_LOSS_L2 only).  minsnr (:144-161) cannot be constructed in the reference (its __init__ takes no
 * loss_type, models/diffusion.py:30 passes one) and has no counterpart here. */
int cd_loss_hybrid(CdPlan* plan, int batch, const float* data, const float* noise, const float* sigma, const float* cond,
                   int loss_type, double* loss_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- training step ----------------------------------------------------------------------------------------------- */
/* Body of TrainDiffusion.training_loop (train/train_diffusion.py:52-63) up to loss.backward(): the loss of the plan's objective
 * (as cd_loss_hybrid: hybrid_weight / noise_pred / mean_pred, any CD_LOSS_* type) AND the gradient of that loss with respect to every parameter, written to `grads`, a flat fp32
 * buffer laid out as cd_plan_grad_layout reports (tensor idx of cd_plan_weight_name starts at *offset, torch layout;
 * *total_floats = size of the buffer).  Workspace: cd_plan_train_workspace_bytes (the forward's activations are kept
 * until the backward has consumed them). */
int cd_plan_grad_layout(const CdPlan* plan, int idx, int64_t* offset, int64_t* total_floats);
int cd_plan_train_workspace_bytes(CdPlan* plan, int batch, size_t* bytes);

/* Sticky range flags of the compute calls issued on this plan since the last query (synchronises `stream`, then clears):
 *   bit 0: an operand of an fp16-pipe kernel (an activation staged for an f16x2 convolution; the normalised input, v or the
 *          folded output weights of the fused attention) exceeded the fp16 range (|x| > 65504): the outputs of that call
 *          contain inf/NaN.  The reference computes in fp32 throughout; rerun with CD_CONV_PRECISION=bf16x3 (full fp32 range).
 *          (cd_denoise / cd_unet_forward / cd_train_step; cd_denoise_safe and the sampler entry points recover by themselves)
 *   bit 1: a sampler / cd_denoise_safe call took the full-range fallback (its result is valid). */
int cd_plan_status(CdPlan* plan, int* flags, void* stream);
int cd_train_step(CdPlan* plan, int batch, const float* data, const float* noise, const float* sigma, const float* cond,
                  int loss_type /* CD_LOSS_* */, double* loss_out, float* grads, void* workspace, size_t workspace_bytes,
                  void* stream);

/* torch.optim.Adam step (train/train.py:144: Adam(model.parameters(), lr); no amsgrad) over n tensors in ceil(n / 48)
 * launches: params / grads / exp_avg / exp_avg_sq are HOST arrays of n DEVICE pointers, numel their lengths.  step is the
 * 1-based step count after this update (torch's state['step']).  Same element-wise formulas as torch:
 *   m += (1-beta1)(g - m);  v = beta2 v + (1-beta2) g^2;  p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps);
 * lr and the betas are doubles (python floats): 1-beta and the bias corrections are formed in double, as torch does. */
int cd_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int64_t* numel, double lr, double beta1, double beta2, float eps, float weight_decay, int step, void* stream);

/* Inverse pre-processing of generated showers on the device: utils.ReverseNormCaloChall (calodiffusion/utils/utils.py:446-573)
 * for the regular grids (dataset_num 2 / 3; showerMap 'layer-logit-norm' when layerE != NULL, 'logit-norm' otherwise).
 * voxels (B,1,D,H,W) normalised; energy (B) incident energies already in physical units (emin*(emax/emin)^e on the host);
 * layerE (B, 1+D) normalised {total, layer} energies or NULL; out (B, D*H*W).  consts = {logit_mean, logit_std, totalE_mean,
 * totalE_std, layers_mean, layers_std} (utils/consts.py:82-116). */
int cd_reverse_norm(const float* voxels, const float* energy, const float* layerE, float* out, int batch, const int32_t dims[3],
                    const float consts[6], float max_deposit, float ecut, void* stream);

/* The HGCal variant, utils.ReverseNormHGCal (calodiffusion/utils/HGCal_utils.py:167-292), has a geometry decode in the middle
 * (NN_embed.dec_batches: needs a geometry file that does not ship with the reference).  Its arithmetic either side of the decode:
 *   stage 1  out = reverse_logit(voxels * logit_std + logit_mean, alpha)                     (any shape: dims only give the count)
 *   stage 2  voxels = DECODED showers (B, L, cells) given as dims = {L, 1, cells}: negatives clamped, every layer rescaled to
 *            the layer energy of layerE unless layer or sum < layer_eps, then x max_deposit x energy
 *   stage 0  = cd_reverse_norm with explicit alpha / layer_eps.
 * HGCal: alpha 1e-8, layer_eps 1e-8, energy = emin + (emax - emin) e[:, 0] (host), ecut 0 (the reference's cut is disabled). */
int cd_reverse_norm_staged(const float* voxels, const float* energy, const float* layerE, float* out, int batch,
                           const int32_t dims[3], const float consts[6], float max_deposit, float ecut, float alpha, float layer_eps,
                           int stage, void* stream);

/* ---- LayerDiffusion's layer-energy model --------------------------------------------------------------------------
 * The conditional residual MLP `ResNet` (calodiffusion/models/models.py:391-457) that LayerDiffusion
 * (calodiffusion/models/layerdiffusion.py:35-38, 114-132) samples the (B, D+1) {total, per-layer} energies with.
 * Stateless: `weights` is a HOST array of n_weights = 2*(8 + 3*n_res) DEVICE pointers, (weight, bias) per nn.Linear in the
 * module's state_dict order: time_mlp.{1,3,5}, cond_mlp.{0,2,4}, in_lay, hidden_layers.i.{embeder.1, dense1.0, dense2.0},
 * out_lay; torch (out, in) row-major fp32. */
typedef struct CdLayerMlpDesc {
  uint32_t struct_size;     /* = sizeof(CdLayerMlpDesc) */
  int32_t dim_in;           /* SHAPE_FINAL[2] + 1 */
  int32_t hidden;           /* 256 */
  int32_t cond_emb;         /* 128: cat(cond_mlp, time_mlp) */
  int32_t cond_size;        /* 1 (3 for HGCal) */
  int32_t n_res;            /* num_layers - 1 ResDense blocks */
  int32_t time_embed_kind;  /* CD_TIME_* (calodiffusion.py:144-152) */
  int32_t objective;        /* CD_OBJ_* */
  float sigma_data;
} CdLayerMlpDesc;

/* ResNet.forward(x, cond, time) (models.py:444-457): x (B, dim_in), cond (B, cond_size), time (B) -> out (B, dim_in). */
int cd_layer_forward(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* x,
                     const float* cond, const float* time, float* out, void* stream);
/* CaloDiffusion.denoise on the layer model (calodiffusion.py:154-169 with layerdiffusion.py:109-112): sigma (B). */
int cd_layer_denoise(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* x,
                     const float* sigma, const float* cond, float* out, void* stream);
/* LayerDiffusion.sample_layers' sampler loop (layerdiffusion.py:114-132 -> models/sample.py:40-110) in ONE launch.
 * steps_dev: DEVICE (n_steps, 4) table of CdStep rows; step_noise (n_steps, B, dim_in) or NULL (deterministic);
 * xs / x0s (n_steps, B, dim_in) or NULL. */
int cd_layer_sample(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* start,
                    const float* cond, const CdStep* steps_dev, int n_steps, const float* step_noise, float* x_out, float* xs,
                    float* x0s, void* stream);

/* Training step of the layer model (LayerDiffusion.compute_loss in the layer state, models/layerdiffusion.py:52-57, with
 * the hybrid_weight / l2 loss of models/loss.py:103-104,118-142,163-179): data = layer energies (B, dim_in), noise (B, dim_in),
 * sigma (B), cond (B, cond_size).  loss_out: one double; grads: ONE flat device buffer holding the gradient of every parameter
 * in the order of `weights` (weight, bias, weight, bias, ...), each with the parameter's element count.  The objective must be
 * CD_OBJ_HYBRID.  workspace: cd_layer_train_workspace_bytes. */
int cd_layer_train_workspace_bytes(const CdLayerMlpDesc* desc, int batch, size_t* bytes);
int cd_layer_train_step(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* data,
                        const float* noise, const float* sigma, const float* cond, double* loss_out, float* grads,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same for every LOSS_TYPE of Loss._loss (models/loss.py:97-116; cd_layer_train_step is loss_type CD_LOSS_L2): the reference's
 * CI fixture trains the layer model with 'huber' (tests/test_execution.py:94). */
int cd_layer_train_step_loss(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* data,
                             const float* noise, const float* sigma, const float* cond, int loss_type /* CD_LOSS_* */,
                             double* loss_out, float* grads, void* workspace, size_t workspace_bytes, void* stream);

/* Arithmetic of the matrix-core kernels, process-wide: "f16x2" (default; fp32 operands as two-term fp16 splits, 3 MFMAs per
 * block, fp16 RANGE -- the 3x3x3 / strided / transposed convolutions and the fused attention's projections and products),
 * "bf16x3" (convolutions on an exact three-term bf16 split, 6 MFMAs; attention unfused on the f32-input MFMA: full fp32 range)
 * or "f32" (everything on the f32-input MFMA).  Initial value: environment variable CD_CONV_PRECISION.  Cached step graphs are
 * dropped by the next sampler call. */
int cd_set_conv_precision(const char* mode);
const char* cd_get_conv_precision(void);

/* ---- measurement ---------------------------------------------------------------------------------------------- */
/* Per-launch timing with HIP events on the launch stream (eager mode; graphs are bypassed while active).
 * cd_profile_end synchronises the device and writes a JSON object
 *   {"<kernel category>": {"launches": n, "ms": total_ms, "flops": algorithmic_per_launch, "bytes": algorithmic_per_launch,
 *                          "flops_total": sum over the launches, "bytes_total": sum over the launches}}
 * (per-launch figures are those of the category's last launch; the totals serve categories that mix shapes). */
int cd_profile_begin(void);
int cd_profile_end(char* json, int cap);

/* ---- primitives (parity tests of the individual kernels; activations CHANNELS-LAST (B, D, H, W, C)) ---- */
int cd_op_to_channels_last(const float* ncdhw, float* ndhwc, int batch, int channels, int64_t voxels, void* stream);
int cd_op_to_ncdhw(const float* ndhwc, float* ncdhw, int batch, int channels, int64_t voxels, void* stream);
/* phi-periodic Conv3d (CylindricalConv, models.py:65-96; Downsample, :360-365). w: torch layout (Cout,Cin,kD,kH,kW).
 * kernel (kD,kH,kW) in {(3,3,3),(3,4,4),(1,1,1)}; padding 1 (z,r zero; phi circular) unless 1x1x1.
 * x0/x1: two channel-concatenated sources (c1 may be 0).  scratch: >= cd_op_scratch_bytes(). */
int cd_op_cyl_conv(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias, float* y,
                   int batch, int cout, const int32_t dims_in[3], const int32_t kernel[3], const int32_t stride[3],
                   void* scratch, void* stream);
/* CylindricalConvTrans as built by Upsample (models.py:25-62, 335-348). w: (Cin,Cout,kD,4,4); padding (1, circ, 1). */
int cd_op_cyl_conv_transpose(const float* x, const float* w, const float* bias, float* y, int batch, int channels,
                             const int32_t dims_in[3], int kernel_z, int stride_z, const int32_t out_pad[3],
                             void* scratch, void* stream);
/* small-Cin planar 3x3x3 conv (init_conv, models.py:562-564): x NCDHW (B,cin,D,H,W) -> y channels-last (B,D,H,W,cout). */
int cd_op_init_conv(const float* x_ncdhw, const float* w, const float* bias, float* y, int batch, int cin, int cout,
                    const int32_t dims[3], void* scratch, void* stream);
/* GroupNorm (+SiLU) (+ per-(b,c) additive embedding) (+ residual), Block.forward / PreNorm (models.py:160-169,321-329). */
int cd_op_group_norm(const float* x, float* y, const float* gamma, const float* beta, int batch, int channels,
                     int64_t voxels, int groups, int silu, const float* add_bc, const float* residual,
                     void* scratch, void* stream);
/* ResnetBlock.forward (models.py:172-200) on channels-last input(s) x0 (+ x1 concatenated).  w: 12 device pointers in
 * torch layout: block1.proj.conv.{weight,bias}, block1.norm.{weight,bias}, block2.proj.conv.{weight,bias},
 * block2.norm.{weight,bias}, mlp.1.{weight,bias} (NULL without conditioning), res_conv.conv.{weight,bias} (NULL if
 * cin == cout).  cond: (B, 128) or NULL. */
int cd_op_resnet_block(const float* x0, int c0, const float* x1, int c1, const float* const* w, const float* cond, float* y,
                       int batch, int cout, const int32_t dims[3], int groups, void* workspace, size_t workspace_bytes,
                       void* stream);
/* Residual(PreNorm(LinearAttention)) (models.py:111-117, 281-329).  w: 7 device pointers: fn.norm.{weight,bias},
 * fn.fn.to_qkv.conv.weight, fn.fn.to_out.0.conv.{weight,bias}, fn.fn.to_out.1.{weight,bias}. */
int cd_op_linear_attention(const float* x, const float* const* w, float* y, int batch, int channels, const int32_t dims[3],
                           void* workspace, size_t workspace_bytes, void* stream);
/* ---- backward primitives (training path; parity-tested against torch autograd on the oracle) ------------------------- */
/* Gradients of y = cyl_conv(cat(x0, x1), w) + b given dy: dx (B, vox_in, c0+c1) or NULL, dw (torch layout), db or NULL.
 * Same geometry rules as cd_op_cyl_conv (models.py:65-96, 360-365). */
int cd_op_conv_backward(const float* x0, int c0, const float* x1, int c1, const float* w, const float* dy, float* dx, float* dw,
                        float* db, int batch, int cout, const int32_t dims_in[3], const int32_t kernel[3],
                        const int32_t stride[3], void* workspace, size_t workspace_bytes, void* stream);
/* Gradients of the Upsample transposed conv (models.py:25-62, 335-348): dx, dw (cin, cout, kz, 4, 4), db. */
int cd_op_conv_transpose_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int batch,
                                  int channels, const int32_t dims_in[3], int kernel_z, int stride_z, const int32_t out_pad[3],
                                  void* workspace, size_t workspace_bytes, void* stream);
/* Gradients of y = act(GroupNorm(x)) + add: dx, dgamma, dbeta, dadd (B, C) or NULL. */
int cd_op_group_norm_backward(const float* x, const float* gamma, const float* beta, const float* dy, float* dx, float* dgamma,
                              float* dbeta, float* dadd, int batch, int channels, int64_t voxels, int groups, int silu,
                              void* workspace, size_t workspace_bytes, void* stream);
size_t cd_op_scratch_bytes(int batch, int max_channels, int64_t max_voxels);

#ifdef __cplusplus
}
#endif
#endif /* CALODIFF_H */
