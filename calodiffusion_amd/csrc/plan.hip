// Host side of the C ABI (include/calodiff.h): plan construction, weight arena, and the launch sequences of
// CondUnet.forward / CaloDiffusion.denoise / DDim.__call__ / hybrid_weight loss on one HIP stream.
// Nothing here allocates or synchronises inside a compute call, so a sampler step is hipGraph-capturable.
#include "../../include/calodiff.h"
#include "cd_common.h"

#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace cd {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

// ------------------------------------------------------------------------------------------------------------
// weight registry
// ------------------------------------------------------------------------------------------------------------
enum PackKind { PK_NONE = 0, PK_CONV = 1, PK_CONVT = 2, PK_INIT = 3 };
struct WeightEntry {
  std::string name;
  int64_t numel = 0;
  size_t raw_off = 0;   // floats into the arena
  PackKind pack = PK_NONE;
  int cin = 0, cout = 0, taps = 0;
  size_t pk_off = 0;
  size_t pk3_off = 0;  // split-bf16 image of 3x3x3 convs (floats into the arena; 0 = none)
  size_t grad_off = 0; // floats into the flat gradient buffer of cd_train_step
  bool set = false;
  // input-gradient images of the training step (dgrad_images below), floats into CdPlan::dg_arena; dg_mode 0 = none
  int dg_mode = 0;
  size_t dg_pk_off = 0, dg_pk3_off = 0;
  bool dg_1x1 = false;  // a 1x1 conv kept raw for the forward (attention to_out: folded per sample) whose backward wants the image
};
// packed images a convolution's input gradient reads (conv_backward / conv_transpose_backward); null members: pack on the fly
struct DgImg {
  const float* pk = nullptr;
  const void* pk3 = nullptr;
};

struct ResW {
  int cin = 0, cout = 0;
  bool has_mlp = false, has_res = false;
  int c1w = -1, c1b = -1, n1g = -1, n1b = -1, c2w = -1, c2b = -1, n2g = -1, n2b = -1, mw = -1, mb = -1, rw = -1, rb = -1;
  int emb_off = 0;
};
struct AttnW {
  int c = 0;
  int ng = -1, nb = -1, qkv = -1, ow = -1, ob = -1, gg = -1, gb = -1;
};
struct LevelW {
  ResW r1, r2;
  AttnW attn;
  int sw = -1, sb = -1;  // down / up sampling conv
};

// ------------------------------------------------------------------------------------------------------------
// workspace allocator: deterministic first-fit over the caller's workspace, replayed identically by a dry run
// (to size the workspace) and by every real call (so captured graphs see stable addresses).
// ------------------------------------------------------------------------------------------------------------
class Arena {
 public:
  void reset(char* base, size_t cap, bool dry) {
    base_ = base; cap_ = cap; dry_ = dry; high_ = 0;
    blocks_.clear();
    blocks_.push_back({0, (size_t)1 << 60, true});
  }
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (size_t i = 0; i < blocks_.size(); ++i) {
      if (blocks_[i].free && blocks_[i].size >= bytes) {
        const size_t off = blocks_[i].off;
        if (blocks_[i].size > bytes) {
          Block rest{off + bytes, blocks_[i].size - bytes, true};
          blocks_[i].size = bytes;
          blocks_.insert(blocks_.begin() + i + 1, rest);
        }
        blocks_[i].free = false;
        if (off + bytes > high_) high_ = off + bytes;
        if (!dry_ && off + bytes > cap_) throw Fail{CD_EWORKSPACE, "workspace too small: call cd_plan_workspace_bytes for this batch size"};
        return dry_ ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base_ + off);
      }
    }
    throw Fail{CD_EWORKSPACE, "workspace allocator exhausted"};
  }
  template <typename T>
  T* get(size_t count) { return (T*)alloc(count * sizeof(T)); }
  void release(const void* p) {
    if (!p) return;
    const size_t off = dry_ ? (size_t)((uintptr_t)p - 0x1000) : (size_t)((const char*)p - base_);
    for (size_t i = 0; i < blocks_.size(); ++i) {
      if (blocks_[i].off == off && !blocks_[i].free) {
        blocks_[i].free = true;
        if (i + 1 < blocks_.size() && blocks_[i + 1].free) {
          blocks_[i].size += blocks_[i + 1].size;
          blocks_.erase(blocks_.begin() + i + 1);
        }
        if (i > 0 && blocks_[i - 1].free) {
          blocks_[i - 1].size += blocks_[i].size;
          blocks_.erase(blocks_.begin() + i);
        }
        return;
      }
    }
    throw Fail{CD_EINVAL, "internal: release of unknown workspace block"};
  }
  size_t high() const { return high_; }
  bool dry() const { return dry_; }

 private:
  struct Block { size_t off, size; bool free; };
  std::vector<Block> blocks_;
  char* base_ = nullptr;
  size_t cap_ = 0, high_ = 0;
  bool dry_ = false;
};

}  // namespace cd

using namespace cd;

struct CdPlan {
  CdUnetDesc desc{};
  int nres = 0;
  std::vector<Dims3> shapes;           // per level
  std::vector<int> up_kz;              // per up step (i = 0 .. nres-2)
  std::vector<Dims3> up_out;           // expected output dims of each up step
  std::vector<WeightEntry> weights;
  std::map<std::string, int> index;
  float* arena = nullptr;
  size_t arena_floats = 0;

  int init_w = -1, init_b = -1, head_w = -1, head_b = -1;
  int tw[3] = {-1, -1, -1}, tb[3] = {-1, -1, -1}, cw[3] = {-1, -1, -1}, cb[3] = {-1, -1, -1};
  std::vector<LevelW> downs, ups;
  ResW mid1, mid2, fin;
  AttnW mid_attn;
  int emb_ld = 0;
  EmbedLayer* d_embed_layers = nullptr;
  int n_embed_layers = 0;
  std::vector<std::pair<int, int>> embed_list;  // (weight idx of mlp w, emb offset) in ResW order

  float* d_coords = nullptr;  // r[W], z[D], phi[H]
  float* d_init_table = nullptr;  // (vox, C0): coordinate-channel part + bias of the init conv (refresh_init_table)
  bool coords_set = false;

  // sampler state (device): step table, counter, stepvals
  static constexpr int kMaxSteps = 4096;
  static constexpr int kEmbedChunk = 16;  // sampler steps whose embeddings one launch computes ahead (cd_ddim_sample)
  float* d_table = nullptr;
  int* d_counter = nullptr;
  float* d_stepvals = nullptr;
  // device word the f16x2 kernels of the current call OR their range flag into: d_counter + 2 (the sticky word cd_plan_status
  // reports) or, inside an entry point with its own bf16x3 fallback, d_counter + 3 (that call's private word)
  int* status_word = nullptr;

  // cached step graph; captured on a private stream (the caller's may be the legacy null stream, which cannot capture)
  hipStream_t cap_stream = nullptr;
  // job list of cd_plan_set_weights: host copy (with the callers' pointers of the last call) and device copy
  std::vector<PackJob> pack_jobs;
  PackJob* d_pack_jobs = nullptr;
  // training: the re-packed (channel-transposed, tap-flipped) weight images of every convolution's input gradient, made by ONE
  // job list per step (two launches) instead of two or three pack launches inside each conv_backward (118 launches per step)
  float* dg_arena = nullptr;
  // training: region for the weight gradients' per-workgroup partials while their reductions are queued (WgradReduceQueue); sized by
  // the first step's requests (that step reduces where it always did), re-sized if a later step asks for more (a larger batch)
  float* wq_region = nullptr;
  size_t wq_cap = 0;
  PackJob* d_dg_jobs = nullptr;
  int n_dg_jobs = 0;
  DgImg dg(int i) const {
    DgImg g;
    const WeightEntry& w = weights[i];
    if (dg_arena && w.dg_mode) {
      g.pk = dg_arena + w.dg_pk_off;
      if (w.dg_mode != 1) g.pk3 = dg_arena + w.dg_pk3_off;
    }
    return g;
  }
  hipGraphExec_t graph_exec = nullptr;
  hipGraphExec_t graph_exec_chunk = nullptr;  // kEmbedChunk consecutive steps as ONE graph (same key): no gap between their launches
  struct GraphKey {
    int batch = 0; const void* ws = nullptr; const void* cond = nullptr; const void* x = nullptr; int noisy = 0; uint64_t seed = 0, offset = 0;
    int precision = 0;  // the captured kernels are those of the convolution precision in force at capture time
    bool operator==(const GraphKey& o) const {
      return batch == o.batch && ws == o.ws && cond == o.cond && x == o.x && noisy == o.noisy && seed == o.seed && offset == o.offset &&
             precision == o.precision;
    }
  } graph_key;
  // cached step graph of a uniform sampler program (cd_sampler_run)
  hipGraphExec_t prog_exec = nullptr;
  struct ProgKey {
    int batch = 0, n_coef = 0, n_bufs = 0; const void* ws = nullptr; const void* cond = nullptr; const void* x = nullptr;
    const void* xs = nullptr; const void* x0s = nullptr; uint64_t ops_hash = 0; int precision = 0;
    bool operator==(const ProgKey& o) const {
      return precision == o.precision && batch == o.batch && n_coef == o.n_coef && n_bufs == o.n_bufs && ws == o.ws && cond == o.cond && x == o.x && xs == o.xs &&
             x0s == o.x0s && ops_hash == o.ops_hash;
    }
  } prog_key;

  // training: flat gradient layout and the device job list of the small Linear weight gradients
  size_t grad_floats = 0;
  std::vector<LinearWgradJob> lin_jobs_host;
  LinearWgradJob* d_lin_jobs = nullptr;

  Arena ws;

  const float* raw(int i) const { return arena + weights[i].raw_off; }
  const float* packed(int i) const { return arena + weights[i].pk_off; }
  const void* packed3(int i) const { return weights[i].pk3_off ? (const void*)(arena + weights[i].pk3_off) : nullptr; }
};

namespace {

int add_weight(CdPlan* p, const std::string& name, int64_t numel, PackKind pk = PK_NONE, int cin = 0, int cout = 0, int taps = 0) {
  WeightEntry e;
  e.name = name; e.numel = numel; e.pack = pk; e.cin = cin; e.cout = cout; e.taps = taps;
  p->weights.push_back(e);
  p->index[name] = (int)p->weights.size() - 1;
  return (int)p->weights.size() - 1;
}

ResW add_res(CdPlan* p, const std::string& pre, int cin, int cout, bool mlp) {
  ResW r;
  r.cin = cin; r.cout = cout; r.has_mlp = mlp; r.has_res = cin != cout;
  if (mlp) {
    r.mw = add_weight(p, pre + ".mlp.1.weight", (int64_t)cout * p->desc.cond_dim);
    r.mb = add_weight(p, pre + ".mlp.1.bias", cout);
    r.emb_off = p->emb_ld;
    p->emb_ld += cout;
    p->embed_list.push_back({r.mw, r.emb_off});
  }
  r.c1w = add_weight(p, pre + ".block1.proj.conv.weight", (int64_t)cout * cin * 27, PK_CONV, cin, cout, 27);
  r.c1b = add_weight(p, pre + ".block1.proj.conv.bias", cout);
  r.n1g = add_weight(p, pre + ".block1.norm.weight", cout);
  r.n1b = add_weight(p, pre + ".block1.norm.bias", cout);
  r.c2w = add_weight(p, pre + ".block2.proj.conv.weight", (int64_t)cout * cout * 27, PK_CONV, cout, cout, 27);
  r.c2b = add_weight(p, pre + ".block2.proj.conv.bias", cout);
  r.n2g = add_weight(p, pre + ".block2.norm.weight", cout);
  r.n2b = add_weight(p, pre + ".block2.norm.bias", cout);
  if (r.has_res) {
    r.rw = add_weight(p, pre + ".res_conv.conv.weight", (int64_t)cout * cin, PK_CONV, cin, cout, 1);
    r.rb = add_weight(p, pre + ".res_conv.conv.bias", cout);
  }
  return r;
}

AttnW add_attn(CdPlan* p, const std::string& pre, int c) {
  AttnW a;
  a.c = c;
  a.qkv = add_weight(p, pre + ".fn.fn.to_qkv.conv.weight", (int64_t)96 * c, PK_CONV, c, 96, 1);
  a.ow = add_weight(p, pre + ".fn.fn.to_out.0.conv.weight", (int64_t)c * 32);
  p->weights[a.ow].cin = 32; p->weights[a.ow].cout = c; p->weights[a.ow].taps = 1; p->weights[a.ow].dg_1x1 = true;
  a.ob = add_weight(p, pre + ".fn.fn.to_out.0.conv.bias", c);
  a.gg = add_weight(p, pre + ".fn.fn.to_out.1.weight", c);
  a.gb = add_weight(p, pre + ".fn.fn.to_out.1.bias", c);
  a.ng = add_weight(p, pre + ".fn.norm.weight", c);
  a.nb = add_weight(p, pre + ".fn.norm.bias", c);
  return a;
}

void check_channels(int c, const char* what) {
  if (c % 32 != 0 || c <= 0 || c > 256)
    throw Fail{CD_EINVAL, std::string(what) + ": channel widths must be multiples of 32 (<= 256) for the MFMA kernels"};
}

void build_plan(CdPlan* p) {
  const CdUnetDesc& d = p->desc;
  CD_REQUIRE(d.n_sizes >= 2 && d.n_sizes <= CD_MAX_SIZES, "LAYER_SIZE_UNET must have 2..8 entries");
  CD_REQUIRE(d.in_channels >= 1 && d.in_channels <= 4, "in_channels must be 1..4");
  CD_REQUIRE(d.cond_dim == 128 || (d.cond_dim % 4 == 0 && d.cond_dim <= 256), "COND_SIZE_UNET must be <= 256");
  CD_REQUIRE(d.cond_size >= 1 && d.cond_size <= 256, "cond_size must be 1..256");
  CD_REQUIRE(d.groups >= 1 && d.groups <= 64, "BLOCK_GROUPS must be 1..64");
  for (int i = 0; i < d.n_sizes; ++i) check_channels(d.layer_sizes[i], "LAYER_SIZE_UNET");
  CD_REQUIRE(d.layer_sizes[0] == 32, "the fused output head needs LAYER_SIZE_UNET[0] == 32");
  CD_REQUIRE(d.layer_sizes[0] == d.layer_sizes[1], "final_conv expects LAYER_SIZE_UNET[1] input channels (models.py:698): sizes 0 and 1 must agree");
  for (int i = 0; i < d.n_sizes; ++i)
    CD_REQUIRE(d.layer_sizes[i] % d.groups == 0 && (d.layer_sizes[i] / d.groups) % 4 == 0,
               "channels per GroupNorm group must be a multiple of 4");
  p->nres = d.n_sizes - 1;
  const int nres = p->nres;
  const int zs = d.compress_z ? 2 : 1;

  // level shapes and up-sampling geometry (CondUnet.__init__, models.py:619-635; Upsample, :335-348)
  Dims3 s{d.grid[0], d.grid[1], d.grid[2]};
  CD_REQUIRE(s.d > 0 && s.h > 0 && s.w > 0, "grid extents must be positive");
  p->shapes.push_back(s);
  std::vector<std::array<int, 3>> extras;
  for (int lv = 0; lv + 1 < nres; ++lv) {
    extras.push_back({(s.d + 1) % 2, s.h % 2, s.w % 2});
    const Dims3 n{d.compress_z ? (s.d + 1) / 2 : s.d, s.h / 2, s.w / 2};
    CD_REQUIRE(n.h >= 1 && n.w >= 1, "grid too small for the number of resolution levels");
    // the strided conv must actually produce that shape
    const int od = (s.d + 2 - 3) / zs + 1, oh = (s.h + 2 - 4) / 2 + 1, ow = (s.w + 2 - 4) / 2 + 1;
    CD_REQUIRE(od == n.d && oh == n.h && ow == n.w, "down-sampling output shape mismatch");
    s = n;
    p->shapes.push_back(s);
  }
  for (int i = 0; i + 1 < nres; ++i) {
    const auto e = extras[nres - 2 - i];
    const int kz = e[0] > 0 ? 4 : 3;
    const Dims3 in = p->shapes[nres - 1 - i];
    const Dims3 out{(in.d - 1) * zs - 2 + kz, 2 * in.h + e[1], 2 * in.w + e[2]};
    const Dims3 want = p->shapes[nres - 2 - i];
    CD_REQUIRE(out.d == want.d && out.h == want.h && out.w == want.w,
               "up-sampling output shape does not match the skip connection (the reference would fail in torch.cat)");
    p->up_kz.push_back(kz);
    p->up_out.push_back(out);
  }

  // weights, in CondUnet.state_dict() order
  const int half = d.cond_dim / 2, hidden = d.cond_size > half / 2 ? d.cond_size : half / 2;
  p->init_w = add_weight(p, "init_conv.conv.weight", (int64_t)d.layer_sizes[0] * d.in_channels * 27, PK_INIT, d.in_channels, d.layer_sizes[0], 27);
  p->init_b = add_weight(p, "init_conv.conv.bias", d.layer_sizes[0]);
  // Linear branch: time_mlp = [Unflatten, Linear(1, q), GELU, Linear(q, half), GELU, Linear(half, half)] (keys 1, 3, 5);
  // sinusoidal branch: [SinusoidalPositionEmbeddings(q), Linear(q, half), GELU, Linear(half, half)] (keys 1, 3).  The cond
  // MLP likewise (keys 0, 2, 4 resp. 1, 3; its sinusoidal form embeds a scalar condition, so hidden must equal q).
  CD_REQUIRE(!(d.time_sin || d.cond_sin) || (half / 2) % 2 == 0 && half / 2 >= 4, "sinusoidal embeddings need cond_dim / 4 even and >= 4");
  CD_REQUIRE(!d.cond_sin || hidden == half / 2,
             "cond_embed 'sin' embeds one scalar per sample into cond_dim/4 features: cond_size must not exceed cond_dim/4");
  const int tin[3] = {1, half / 2, half}, tout[3] = {half / 2, half, half};
  for (int i = d.time_sin ? 1 : 0; i < 3; ++i) {
    const std::string key = "time_mlp." + std::to_string(d.time_sin ? 2 * i - 1 : 2 * i + 1);
    p->tw[i] = add_weight(p, key + ".weight", (int64_t)tin[i] * tout[i]);
    p->tb[i] = add_weight(p, key + ".bias", tout[i]);
  }
  const int cin3[3] = {d.cond_size, hidden, half}, cout3[3] = {hidden, half, half};
  for (int i = d.cond_sin ? 1 : 0; i < 3; ++i) {
    const std::string key = "cond_mlp." + std::to_string(d.cond_sin ? 2 * i - 1 : 2 * i);
    p->cw[i] = add_weight(p, key + ".weight", (int64_t)cin3[i] * cout3[i]);
    p->cb[i] = add_weight(p, key + ".bias", cout3[i]);
  }
  p->downs.resize(nres);
  p->ups.resize(nres);
  for (int i = 0; i < nres; ++i) {
    const int ci = d.layer_sizes[i], co = d.layer_sizes[i + 1];
    const std::string pre = "downs." + std::to_string(i);
    p->downs[i].r1 = add_res(p, pre + ".0", ci, co, true);
    p->downs[i].r2 = add_res(p, pre + ".1", co, co, true);
    if (i + 1 < nres) {
      p->downs[i].sw = add_weight(p, pre + ".2.conv.weight", (int64_t)co * co * 48, PK_CONV, co, co, 48);
      p->downs[i].sb = add_weight(p, pre + ".2.conv.bias", co);
    }
  }
  for (int i = 0; i < nres; ++i) {
    const int lv = nres - 1 - i;
    const int ci = d.layer_sizes[lv], co = d.layer_sizes[lv + 1];
    const std::string pre = "ups." + std::to_string(i);
    p->ups[i].r1 = add_res(p, pre + ".0", co * 2, ci, true);
    p->ups[i].r2 = add_res(p, pre + ".1", ci, ci, true);
    if (i + 1 < nres) {
      const int kz = p->up_kz[i];
      p->ups[i].sw = add_weight(p, pre + ".2.convTrans.weight", (int64_t)ci * ci * kz * 16, PK_CONVT, ci, ci, kz * 16);
      p->ups[i].sb = add_weight(p, pre + ".2.convTrans.bias", ci);
    }
    check_channels(co * 2, "skip concat");
  }
  if (d.block_attn) {
    for (int i = 0; i < nres; ++i) p->downs[i].attn = add_attn(p, "downs_attn." + std::to_string(i), d.layer_sizes[i + 1]);
    for (int i = 0; i < nres; ++i) p->ups[i].attn = add_attn(p, "ups_attn." + std::to_string(i), d.layer_sizes[nres - 1 - i]);
  }
  const int mid = d.layer_sizes[nres];
  p->mid1 = add_res(p, "mid_block1", mid, mid, true);
  if (d.mid_attn) p->mid_attn = add_attn(p, "mid_attn", mid);
  p->mid2 = add_res(p, "mid_block2", mid, mid, true);
  p->fin = add_res(p, "final_conv.0", d.layer_sizes[1], d.layer_sizes[0], false);
  p->head_w = add_weight(p, "final_conv.1.conv.weight", d.layer_sizes[0]);
  p->head_b = add_weight(p, "final_conv.1.conv.bias", 1);

  // arena layout
  size_t off = 0;
  auto bump = [&](size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return o; };
  for (auto& w : p->weights) {
    w.raw_off = bump((size_t)w.numel);
    if (w.pack == PK_CONV || w.pack == PK_CONVT) w.pk_off = bump(packed_weight_floats(w.cin, w.cout, w.taps));
    // 16-bit split images: the 3x3x3 / strided convs and the attention's to_qkv (cout = 96)
    // (1x1: the attention's to_qkv and the ResnetBlocks' res_conv, which the deep-level kernel runs on the fp16 pipe)
    if (w.pack == PK_CONV && (w.taps == 27 || w.taps == 48 || w.taps == 1))
      w.pk3_off = bump(packed_split16_bytes(w.cin, w.cout, w.taps) / 4);
    else if (w.pack == PK_CONVT) w.pk3_off = bump(packed_f16x2_bytes(w.cin, w.cout, w.taps) / 4);  // f16x2 image only
    else if (w.pack == PK_INIT) w.pk_off = bump((size_t)w.numel);
  }
  p->arena_floats = off;
  size_t goff = 0;
  for (auto& w : p->weights) {
    w.grad_off = goff;
    goff += ((size_t)w.numel + 63) & ~(size_t)63;
  }
  p->grad_floats = goff;
  CD_HIP(hipMalloc((void**)&p->d_lin_jobs, sizeof(LinearWgradJob) * 64));
  CD_HIP(hipMalloc((void**)&p->arena, off * sizeof(float)));
  CD_HIP(hipMemset(p->arena, 0, off * sizeof(float)));

  // embedding projection descriptors
  std::vector<EmbedLayer> layers;
  for (auto& e : p->embed_list) {
    const WeightEntry& w = p->weights[e.first];
    EmbedLayer L;
    L.w = p->arena + w.raw_off;
    L.b = p->arena + p->weights[e.first + 1].raw_off;
    L.cout = (int)(w.numel / d.cond_dim);
    L.offset = e.second;
    layers.push_back(L);
  }
  p->n_embed_layers = (int)layers.size();
  CD_HIP(hipMalloc((void**)&p->d_embed_layers, sizeof(EmbedLayer) * (layers.size() + 1)));
  CD_HIP(hipMemcpy(p->d_embed_layers, layers.data(), sizeof(EmbedLayer) * layers.size(), hipMemcpyHostToDevice));
  if (p->emb_ld == 0) p->emb_ld = 4;

  CD_HIP(hipMalloc((void**)&p->d_coords, sizeof(float) * (size_t)(d.grid[0] + d.grid[1] + d.grid[2] + 4)));
  CD_HIP(hipMemset(p->d_coords, 0, sizeof(float) * (size_t)(d.grid[0] + d.grid[1] + d.grid[2] + 4)));
  CD_HIP(hipMalloc((void**)&p->d_init_table, sizeof(float) * (size_t)p->shapes[0].vox() * d.layer_sizes[0]));
  CD_HIP(hipMalloc((void**)&p->d_table, sizeof(float) * 4 * CdPlan::kMaxSteps));
  CD_HIP(hipMalloc((void**)&p->d_counter, sizeof(int) * 4));  // [0] sampler step counter, [2] range flags
  CD_HIP(hipMemset(p->d_counter, 0, sizeof(int) * 4));
  p->status_word = p->d_counter + 2;
  CD_HIP(hipMalloc((void**)&p->d_stepvals, sizeof(float) * 8));
}

// ------------------------------------------------------------------------------------------------------------
// launch sequences
// ------------------------------------------------------------------------------------------------------------
struct Run {
  Arena* ws;
  hipStream_t s;
  int B;
  int groups;
  int* status = nullptr;  // device word for sticky range flags (cd_plan_status), or null
  GnParamQueue* gq = nullptr;  // training step: the GroupNorm layers' parameter-gradient reductions, flushed once at the end
  bool dry() const { return ws->dry(); }
};

// weights of one block resolved to device pointers (conv weights in packed MFMA layout)
struct ResP {
  int cin = 0, cout = 0;
  bool has_res = false;
  const float *c1w = nullptr, *c1b = nullptr, *n1g = nullptr, *n1b = nullptr;
  const void *c1w3 = nullptr, *c2w3 = nullptr;  // split-bf16 images of the two 3x3x3 convs
  const float *c2w = nullptr, *c2b = nullptr, *n2g = nullptr, *n2b = nullptr;
  const float *rw = nullptr, *rb = nullptr;
  const void* rw16 = nullptr;  // f16x2 image of the 1x1 shortcut conv
  const float* emb = nullptr;  // (B, emb_ld) slice for this block, or null
  int emb_ld = 0;
};
struct AttnP {
  int c = 0;
  const float *ng = nullptr, *nb = nullptr, *qkv = nullptr, *ow = nullptr, *ob = nullptr, *gg = nullptr, *gb = nullptr;
  const void* qkv16 = nullptr;  // f16x2 image of to_qkv (fused attention kernels)
};

ResP resolve(const CdPlan* p, const ResW& w, const float* emb) {
  ResP r;
  r.cin = w.cin; r.cout = w.cout; r.has_res = w.has_res;
  r.c1w3 = p->packed3(w.c1w); r.c2w3 = p->packed3(w.c2w);
  r.c1w = p->packed(w.c1w); r.c1b = p->raw(w.c1b); r.n1g = p->raw(w.n1g); r.n1b = p->raw(w.n1b);
  r.c2w = p->packed(w.c2w); r.c2b = p->raw(w.c2b); r.n2g = p->raw(w.n2g); r.n2b = p->raw(w.n2b);
  if (w.has_res) {
    r.rw = p->packed(w.rw); r.rb = p->raw(w.rb);
    if (p->packed3(w.rw)) r.rw16 = (const char*)p->packed3(w.rw) + packed_bf16x3_bytes(w.cin, w.cout, 1);
  }
  if (w.has_mlp && emb) { r.emb = emb + w.emb_off; r.emb_ld = p->emb_ld; }
  return r;
}
AttnP resolve(const CdPlan* p, const AttnW& w) {
  AttnP a;
  a.c = w.c;
  a.ng = p->raw(w.ng); a.nb = p->raw(w.nb); a.qkv = p->packed(w.qkv); a.ow = p->raw(w.ow); a.ob = p->raw(w.ob);
  a.qkv16 = (const char*)p->packed3(w.qkv) + packed_bf16x3_bytes(w.c, 96, 1);
  a.gg = p->raw(w.gg); a.gb = p->raw(w.gb);
  return a;
}

// channel partials of a tensor from a standalone pass (producer without a stats epilogue)
float* stats_pass(Run& r, const float* x, int C, int64_t vox, int* units) {
  const int ns = gn_nsplit_for(vox, r.B);
  float* part = r.ws->get<float>((size_t)r.B * ns * C * 2);
  if (!r.dry()) launch_ch_stats(x, part, r.B, C, vox, ns, r.s);
  *units = ns;
  return part;
}

// conv + channel partials of its output (fused epilogue when the kernel supports it); input optionally normalised
// on the fly by `coef_in` (+SiLU).  Returns the partial buffer (caller releases) and sets *units.
// defer_in (optional, instead of coef_in): the input normalisation as partials + affine parameters, folded by the conv
// kernel itself; coef_buf is the [B][Cin][4] table a kernel without that prologue gets materialised.
float* conv3_with_stats(Run& r, const float* x0, int c0, const float* x1, int c1, const float* wpk, const void* wpk3,
                        const float* bias, float* out, int cout, Dims3 dims, const float* coef_in, int* units,
                        const GnDefer* defer_in = nullptr, float* coef_buf = nullptr, const ConvFusion::GnOut* gn_out = nullptr) {
  const int64_t vox = dims.vox();
  const int cap = (int)((vox + 31) / 32);
  float* part = r.ws->get<float>((size_t)r.B * cap * cout * 2);
  int u = 0;
  if (!r.dry()) {
    ConvGeom g{dims, dims, 3, 3, 3, 1, 1, 1};
    ConvFusion fu;
    fu.coef = coef_in; fu.act = 1; fu.ch_part = part; fu.units = &u; fu.wpk_bf16x3 = wpk3; fu.status = r.status;
    if (defer_in) { fu.defer = *defer_in; fu.coef_buf = coef_buf; fu.coef = nullptr; }
    if (gn_out) fu.gn_out = *gn_out;
    launch_conv_mfma(x0, c0, x1, c1, wpk, bias, out, r.B, cout, g, r.s, fu);
    if (gn_out && *gn_out->done) {
      // (the kernel normalised its own output: `out` is the block output, there are no partials of the conv output)
    } else if (u == 0) {  // kernel without a stats epilogue: separate pass, same buffer (nsplit <= cap)
      u = gn_nsplit_for(vox, r.B);
      if (u > cap) u = cap;
      launch_ch_stats(out, part, r.B, cout, vox, u, r.s);
    }
  }
  *units = u;
  return part;
}

// ResnetBlock.forward (models.py:191-200): block1 -> (+ mlp(cond)) -> block2 -> + res_conv(x).
//   conv1 (stats epilogue) -> finalize -> conv2 normalises h1 while staging it (stats epilogue) -> finalize ->
//   one elementwise pass: silu(gn(h2)) + shortcut.  `part_out`/`units_out` (optional): channel partials of the block
//   output for a following PreNorm.
// `lazy` (optional): leave the closing GroupNorm + SiLU + identity shortcut to the consumer (the head kernel).  If the block
// qualifies it returns its second conv's raw output, lazy->gn describes the normalisation, lazy->part (to be released by the
// caller) holds its partials and the shortcut is x0.
struct LazyClose {
  GnDefer gn;
  float* part = nullptr;
  bool on = false;
};
float* res_block(Run& r, const ResP& w, const float* x0, int c0, const float* x1, int c1, Dims3 dims,
                 float** part_out = nullptr, int* units_out = nullptr, LazyClose* lazy = nullptr) {
  Arena* ws = r.ws;
  CD_REQUIRE(c0 + c1 == w.cin, "internal: resnet block input width mismatch");
  const int64_t vox = dims.vox();
  const int G = r.groups;
  int u1 = 0, u2 = 0;
  static const bool defer_gn = getenv("CD_NO_GNDEFER") == nullptr;  // consumers fold the GroupNorm coefficients (gn_defer.h)
  float* h1 = ws->get<float>((size_t)r.B * vox * w.cout);
  // a 32-channel block on a grid of <= 128 voxels is ONE launch (kernels_conv_small.hip): decided below, once the shortcut exists
  const bool whole = vox <= 128 && w.cout == 32 && defer_gn && conv_precision() == PREC_F16X2 && w.c1w3 && w.c2w3;
  float* p1 = nullptr;
  if (!whole) p1 = conv3_with_stats(r, x0, c0, x1, c1, w.c1w, w.c1w3, w.c1b, h1, w.cout, dims, nullptr, &u1);
  else p1 = ws->get<float>((size_t)r.B * ((vox + 31) / 32) * w.cout * 2);  // (same block as conv3_with_stats would take)
  float* coef1 = ws->get<float>((size_t)r.B * w.cout * 4);
  GnDefer d1;
  d1.part = p1; d1.units = u1; d1.gamma = w.n1g; d1.beta = w.n1b; d1.add = w.emb; d1.add_ld = w.emb_ld; d1.C = w.cout; d1.groups = G;
  d1.vox = vox;
  if (!r.dry() && !defer_gn) launch_gn_finalize(p1, u1, w.n1g, w.n1b, w.emb, w.emb_ld, coef1, r.B, w.cout, G, vox, r.s);
  float* h2 = ws->get<float>((size_t)r.B * vox * w.cout);
  // Grids of at most 128 voxels (one workgroup sees a whole sample, kernels_conv_small.hip): the second conv closes the block
  // itself -- GroupNorm, SiLU, shortcut -- so the shortcut has to exist before it runs.
  const bool small = vox <= 128;
  float* po = nullptr;
  // A block whose shortcut is a 1x1 conv (models.py:200) is closed BY that conv (PointwiseArgs::gn_res): after the second conv it
  // computes shortcut + silu(gn(h2)) in one pass -- the shortcut tensor and the elementwise pass over the grid never exist.
  const bool no_pw_close = getenv("CD_NO_PW_CLOSE") != nullptr;  // (read per call: the parity test switches it in one process)
  const bool pw_close = w.has_res && !small && defer_gn && !no_pw_close && w.cout <= 128;
  // partials per sample of the block's output: those of whichever kernel closes it (1 if the second conv does)
  const int bps = pw_close ? pointwise_units(vox) : gn_apply_blocks_per_sample(r.B, w.cout, vox);
  if (part_out) {
    po = ws->get<float>((size_t)r.B * bps * w.cout * 2);
    *part_out = po;
  }
  float* res = nullptr;
  auto shortcut_conv = [&](const GnDefer* close = nullptr) {
    if (!close) res = ws->get<float>((size_t)r.B * vox * w.cout);
    if (!r.dry()) {
      PointwiseArgs a;
      a.in0 = x0; a.ld0 = c0; a.off0 = 0; a.c0 = c0; a.in1 = x1; a.ld1 = c1; a.c1 = c1;
      a.wpk = w.rw; a.bias = w.rb; a.out = close ? h2 : res; a.batch = r.B; a.cout = w.cout; a.vox = vox;
      if (conv_precision() == PREC_F16X2 && !getenv("CD_PW_F32")) { a.wpk16 = w.rw16; a.status = r.status; }  // (fp16 pipe, under the range fallback)
      if (close) { a.gn_res = h2; a.gn_defer = *close; a.ch_part = po; }
      launch_pointwise(a, r.s);
    }
  };
  if (small && w.has_res) shortcut_conv();
  int fused = 0;
  if (whole && !r.dry()) {
    const float* sc0 = w.has_res ? res : x0;
    const float* sc1 = w.has_res ? nullptr : (c1 ? x1 : nullptr);
    if (try_launch_res_block_small(x0, c0, x1, c1, (const char*)w.c1w3 + packed_bf16x3_bytes(c0 + c1, 32, 27), w.c1b, w.n1g, w.n1b,
                                   w.emb, w.emb_ld, (const char*)w.c2w3 + packed_bf16x3_bytes(32, 32, 27), w.c2b, w.n2g, w.n2b, G, sc0,
                                   sc1, w.has_res ? 0 : c0, h1, h2, po, r.B, w.cout, dims, r.status, r.s))
      fused = 2;
    else
      p1 = (ws->release(p1), conv3_with_stats(r, x0, c0, x1, c1, w.c1w, w.c1w3, w.c1b, h1, w.cout, dims, nullptr, &u1));
  }
  ConvFusion::GnOut go;
  if (small && defer_gn && fused != 2) {
    go.gamma = w.n2g; go.beta = w.n2b; go.groups = G; go.part_out = po; go.done = &fused;
    if (w.has_res) { go.res0 = res; }
    else { go.res0 = x0; go.res1 = c1 ? x1 : nullptr; go.res_c0 = c0; }
  }
  float* p2 = nullptr;
  if (fused == 2) p2 = ws->get<float>((size_t)r.B * ((vox + 31) / 32) * w.cout * 2);  // (conv2 ran inside the block launch)
  else p2 = conv3_with_stats(r, h1, w.cout, nullptr, 0, w.c2w, w.c2w3, w.c2b, h2, w.cout, dims, coef1, &u2, defer_gn ? &d1 : nullptr,
                             coef1, go.gamma ? &go : nullptr);
  ws->release(p1);
  ws->release(h1);
  ws->release(coef1);
  float* coef2 = ws->get<float>((size_t)r.B * w.cout * 4);
  GnDefer d2;
  d2.part = p2; d2.units = u2; d2.gamma = w.n2g; d2.beta = w.n2b; d2.C = w.cout; d2.groups = G; d2.vox = vox;
  if (!r.dry() && !defer_gn) launch_gn_finalize(p2, u2, w.n2g, w.n2b, nullptr, 0, coef2, r.B, w.cout, G, vox, r.s);
  const GnDefer* dp2 = defer_gn ? &d2 : nullptr;
  if (part_out) *units_out = fused ? 1 : bps;
  if (pw_close) {
    shortcut_conv(&d2);
  } else if (w.has_res) {
    if (!res) shortcut_conv();
    if (!r.dry() && !fused) launch_gn_apply(h2, h2, coef2, r.B, w.cout, vox, 1, res, nullptr, 0, po, r.s, dp2);
    ws->release(res);
  } else if (lazy && !small && defer_gn && c1 == 0 && w.cout == 32 && !part_out) {
    lazy->gn = d2;
    lazy->part = p2;
    lazy->on = true;
    ws->release(coef2);
    return h2;
  } else {
    // identity shortcut; for a concatenated input it is read from the two sources (models.py:200,741)
    if (!r.dry() && !fused) launch_gn_apply(h2, h2, coef2, r.B, w.cout, vox, 1, x0, c1 ? x1 : nullptr, c0, po, r.s, dp2);
  }
  ws->release(p2);
  ws->release(coef2);
  return h2;
}

// Residual(PreNorm(LinearAttention)) (models.py:111-117, 281-329).  xpart/xunits: channel partials of x if its producer
// emitted them (else a stats pass runs here).
float* attn_block(Run& r, const AttnP& w, const float* x, Dims3 dims, float* xpart = nullptr, int xunits = 0) {
  Arena* ws = r.ws;
  const int64_t vox = dims.vox();
  const int C = w.c;
  float* own = nullptr;
  if (!xpart) {
    own = stats_pass(r, x, C, vox, &xunits);
    xpart = own;
  }
  // The fused kernels run every product on the fp16 pipe (f16x2 splits: fp16 RANGE); the full-range precisions (bf16x3 / f32,
  // and with them the re-run of a range fallback) take the unfused form on the f32-input MFMA instead.
  static const bool no_fused_env = getenv("CD_NO_FUSED_ATTN") != nullptr;
  static const bool defer_env = getenv("CD_NO_GNDEFER") == nullptr;
  const bool no_fused = no_fused_env || conv_precision() != PREC_F16X2;
  const bool defer_gn = defer_env && !no_fused;  // consumers fold the coefficients (gn_defer.h)
  float* coefn = ws->get<float>((size_t)r.B * C * 4);
  GnDefer dn;
  dn.part = xpart; dn.units = xunits; dn.gamma = w.ng; dn.beta = w.nb; dn.C = C; dn.groups = 1; dn.vox = vox;
  const GnDefer* dnp = defer_gn ? &dn : nullptr;
  if (!r.dry() && !defer_gn) launch_gn_finalize(xpart, xunits, w.ng, w.nb, nullptr, 0, coefn, r.B, C, 1, vox, r.s);
  const int CT = (C + 31) / 32;
  float* y = nullptr;
  float* ypart = nullptr;
  int yu = 0;
  // grids of a few hundred voxels: the whole block -- both passes, the closing GroupNorm and the residual -- in one launch
  const bool single = !no_fused && defer_gn && attn_small_eligible(vox);
  bool moments = false;
  if (!no_fused) {
    // fused path (kernels_attn.hip): x -> {max, sum, context} partials -> per-sample folded W_out -> y; qkv never exists
    const int nsp = attn_fused_nsplit_for(vox, r.B);
    const int cap = single && nsp < 4 ? 4 : nsp;  // (the single-launch form may deal a sample to up to 4 co-operating workgroups)
    float* part = ws->get<float>(attn_partial_floats(r.B, cap));
    // Moment form (kernels_attn.hip): pass 1 also accumulates the moments of softmax(q), pass 2 then knows the closing GroupNorm's
    // statistics in closed form and writes gn(y) + x itself -- y is never written and the gn_apply pass below does not run
    // (read per call: the parity test switches it in one process)
    static const bool sep_combine = getenv("CD_ATTN_COMBINE_LAUNCH") != nullptr;  // A/B: the separate combine launch
    // It costs pass 1 ~40 % more per tile and both passes a few microseconds of prologue / epilogue, and saves a pass that moves
    // 3 B vox C floats: it pays from ~4 M elements per tensor (same-box A/B: Dataset-2 level 0, 13 M, +1.4 %; Dataset-3, 41 M,
    // +1.8 %; HGCal at batch 16, 3.6 M, -0.3 %)
    const char* mom_env = getenv("CD_ATTN_MOM_MIN");  // (read per call, like the switch: the parity test sets it)
    const int64_t mom_min = mom_env ? atoll(mom_env) : (4ll << 20);
    moments = !single && !sep_combine && defer_gn && attn_moments_eligible(C) && (vox * C) % 4 == 0 && (int64_t)r.B * vox * C >= mom_min &&
              getenv("CD_NO_ATTN_MOMENTS") == nullptr;
    float* momb = moments ? ws->get<float>(attn_moment_floats(r.B, nsp)) : nullptr;
    float* wpb = ws->get<float>((size_t)r.B * CT * 1024);
    y = ws->get<float>((size_t)r.B * vox * C);
    yu = nsp;
    ypart = ws->get<float>((size_t)r.B * cap * C * 2);
    if (!r.dry() && single) {
      launch_attn_small(x, C, coefn, w.qkv16, part, w.ow, 0.17677669529663689f /* 32^-1/2 */, w.ob, w.gg, w.gb, y, ypart, r.B, vox,
                        r.s, dnp, r.status, cap);
    } else if (!r.dry()) {
      launch_attn_kv_context(x, C, coefn, w.qkv16, part, r.B, vox, nsp, r.s, dnp, r.status, momb);
      if (sep_combine) {
        launch_attn_combine(part, nsp, w.ow, C, wpb, r.B, 0.17677669529663689f /* 32^-1/2 */, r.s, nullptr, nullptr, true);
        launch_attn_out(x, C, coefn, w.qkv16, wpb, w.ob, y, ypart, r.B, vox, nsp, r.s, dnp, nullptr, nullptr, 0.f, r.status);
      } else {
        launch_attn_out(x, C, coefn, w.qkv16, nullptr, w.ob, y, moments ? nullptr : ypart, r.B, vox, nsp, r.s, dnp, part, w.ow,
                        0.17677669529663689f, r.status, momb, w.gg, w.gb);
      }
    }
    if (momb) ws->release(momb);
    if (own) ws->release(own);
    own = nullptr;
    ws->release(coefn);
    ws->release(part);
    ws->release(wpb);
  } else {
    float* qkv = ws->get<float>((size_t)r.B * vox * 96);
    if (!r.dry()) {
      PointwiseArgs a;
      a.in0 = x; a.ld0 = C; a.c0 = C; a.wpk = w.qkv; a.out = qkv; a.batch = r.B; a.cout = 96; a.vox = vox;
      a.prologue = A_AFFINE; a.coef = coefn;
      launch_pointwise(a, r.s);
    }
    ws->release(coefn);
    const int nsp = attn_nsplit_for(vox, r.B);
    float* part = ws->get<float>(attn_partial_floats(r.B, nsp));
    float* wpb = ws->get<float>((size_t)r.B * CT * 1024);
    if (!r.dry()) {
      launch_attn_context(qkv, part, r.B, vox, nsp, r.s);
      launch_attn_combine(part, nsp, w.ow, C, wpb, r.B, 0.17677669529663689f /* 32^-1/2 */, r.s);
    }
    y = ws->get<float>((size_t)r.B * vox * C);
    yu = pointwise_units(vox);
    ypart = ws->get<float>((size_t)r.B * yu * C * 2);
    if (!r.dry()) {
      PointwiseArgs a;
      a.in0 = qkv; a.ld0 = 96; a.off0 = 0; a.c0 = 32; a.wpk = wpb; a.w_batch_stride = (int64_t)CT * 1024; a.bias = w.ob;
      a.out = y; a.batch = r.B; a.cout = C; a.vox = vox; a.prologue = A_SOFTMAX32; a.ch_part = ypart;
      launch_pointwise(a, r.s);
    }
    ws->release(part);
    ws->release(wpb);
    ws->release(qkv);
}
  if (own) ws->release(own);
  float* coefg = ws->get<float>((size_t)r.B * C * 4);
  if (!r.dry() && !single && !moments) {
    GnDefer dg;
    dg.part = ypart; dg.units = yu; dg.gamma = w.gg; dg.beta = w.gb; dg.C = C; dg.groups = 1; dg.vox = vox;
    if (!defer_gn) launch_gn_finalize(ypart, yu, w.gg, w.gb, nullptr, 0, coefg, r.B, C, 1, vox, r.s);
    launch_gn_apply(y, y, coefg, r.B, C, vox, 0, x, nullptr, 0, nullptr, r.s, defer_gn ? &dg : nullptr);
  }
  ws->release(ypart);
  ws->release(coefg);
  return y;
}

// ------------------------------------------------------------------------------------------------------------
// backward building blocks
// ------------------------------------------------------------------------------------------------------------
// per-channel sums of a (B, vox, C) tensor over batch and voxels -> db (bias gradients)
void bias_grad(Run& r, const float* dy, int C, int64_t vox, float* db) {
  int units = 0;
  float* part = stats_pass(r, dy, C, vox, &units);
  if (!r.dry()) launch_bias_grad(part, units, r.B, C, db, false, r.s);
  r.ws->release(part);
}

// Backward of a phi-periodic Conv3d y = conv(cat(x0, x1), w) + b  (3x3x3 stride 1, 1x1x1, or the (3,4,4) strided conv).
//   dx (optional): (B, vox_in, c0+c1) gradient of the concatenated input
//   dw: torch layout (cout, c0+c1, taps);  db: (cout) or null.   w_raw: torch-layout weights (device).
// img (optional): the input gradient's weight images already packed for this step (CdPlan::dg); without them they are packed here.
// xcoef (optional, single-source x0 only): the conv's input was silu(coef[0] x0 + coef[1]) + coef[2] (see launch_wgrad)
void conv_backward(Run& r, const float* x0, int c0, const float* x1, int c1, const float* w_raw, const float* dy, float* dx,
                   float* dw, float* db, int cout, const ConvGeom& g, const DgImg* img = nullptr, const float* xcoef = nullptr,
                   // dx = (input gradient) + dx_add, a tensor shaped like dx, where the kernel that runs can add it in its
                   // epilogue (3x3x3 stride 1 on the fp16 pipe): *dx_added says whether it did
                   const float* dx_add = nullptr, int* dx_added = nullptr) {
  Arena* ws = r.ws;
  const int cin = c0 + c1, T = g.kd * g.kh * g.kw;
  const bool pre = img && img->pk;
  if (dx) {
    if (T == 1) {
      float* wp = pre ? nullptr : ws->get<float>(packed_weight_floats(cout, cin, 1));
      if (!r.dry()) {
        if (!pre) launch_pack_weights(w_raw, wp, cin, cout, 1, true, r.s);
        PointwiseArgs a;
        a.in0 = dy; a.ld0 = cout; a.c0 = cout; a.wpk = pre ? img->pk : wp; a.out = dx; a.batch = r.B; a.cout = cin; a.vox = g.in.vox();
        launch_pointwise(a, r.s);
      }
      if (wp) ws->release(wp);
    } else if (g.sz == 1 && g.sh == 1 && g.sw == 1) {
      // dx = conv(dy, W^T flipped): the forward kernels with re-packed weights
      float* wp = pre ? nullptr : ws->get<float>(packed_weight_floats(cout, cin, T));
      float* wp3 = pre ? nullptr : ws->get<float>(packed_split16_bytes(cout, cin, T) / 4);
      if (!r.dry()) {
        if (!pre) {
          launch_pack_weights(w_raw, wp, cin, cout, T, true, r.s, true);
          launch_pack_weights_split16(w_raw, wp3, cin, cout, T, r.s, true, true);
        }
        ConvGeom gd{g.out, g.in, g.kd, g.kh, g.kw, 1, 1, 1};
        ConvFusion fu;
        fu.wpk_bf16x3 = pre ? img->pk3 : wp3;
        fu.in_absmax = launch_absmax_bits(dy, (size_t)r.B * g.out.vox() * cout, r.s);  // also serves the weight gradient below
        fu.add_src = dx_add; fu.add_done = dx_added;
        launch_conv_mfma(dy, cout, nullptr, 0, pre ? img->pk : wp, nullptr, dx, r.B, cin, gd, r.s, fu);
      }
      if (wp3) ws->release(wp3);
      if (wp) ws->release(wp);
    } else if (g.in.h & 1) {
      // odd phi ring: the circular halo breaks the parity classes of the gather kernel (see kernels_bwd.hip)
      if (!r.dry()) launch_strided_dgrad_naive(dy, w_raw, dx, r.B, cin, cout, g.in, g.out, g.kd, g.sz, r.s);
    } else {
      // strided conv: its adjoint is the transposed-conv gather kernel
      // (on the fp16 pipe like the forward up-conv, the tiny gradients rescaled by a power of two from their max)
      float* wp = pre ? nullptr : ws->get<float>(packed_weight_floats(cout, cin, T));
      float* wp16 = pre ? nullptr : ws->get<float>(packed_f16x2_bytes(cout, cin, T) / 4 + 64);
      if (!r.dry()) {
        if (!pre) {
          launch_pack_weights(w_raw, wp, cin, cout, T, true, r.s);
          launch_pack_weights_f16x2(w_raw, wp16, cin, cout, T, r.s, true, false);
        }
        const unsigned* amax = launch_absmax_bits(dy, (size_t)r.B * g.out.vox() * cout, r.s);
        launch_conv_transpose_mfma(dy, cout, pre ? img->pk : wp, nullptr, dx, r.B, cin, g.out, g.in, g.kd, g.sz, r.s,
                                   pre ? img->pk3 : wp16, r.status, amax);
      }
      if (wp16) ws->release(wp16);
      if (wp) ws->release(wp);
    }
  }
  float* part = ws->get<float>(wgrad_partial_floats(g.out.vox(), r.B, false, cout, c0 > c1 ? c0 : c1, T));
  if (!r.dry()) {
    CD_REQUIRE(!xcoef || !c1, "conv backward: a normalised input has one source");
    launch_wgrad(dy, cout, g.out, x0, c0, c0, 0, g.in, g.kd, g.kh, g.kw, g.sz, g.sh, r.B, false, part, dw, false, false, r.s, cin, 0, xcoef);
    if (c1) launch_wgrad(dy, cout, g.out, x1, c1, c1, 0, g.in, g.kd, g.kh, g.kw, g.sz, g.sh, r.B, false, part, dw, false, false, r.s, cin, c0);
    absmax_note_drop();  // max |dy| served this backward only
  }
  ws->release(part);
  if (db) bias_grad(r, dy, cout, g.out.vox(), db);
}

// Backward of the phi-periodic ConvTranspose3d (Upsample): y = convT(x, w) + b, w stored (cin, cout, kz, 4, 4)
void conv_transpose_backward(Run& r, const float* x, const float* w_raw, const float* dy, float* dx, float* dw, float* db, int c,
                             Dims3 din, Dims3 dout, int kz, int sz, const DgImg* img = nullptr) {
  Arena* ws = r.ws;
  const int T = kz * 16;
  const bool pre = img && img->pk;
  // Odd output phi extent (output_padding 1 along phi: Dataset-3 level 1, Dataset-1 grid): the forward's last phi row
  // duplicates row 0, so fold its gradient into row 0 and continue on the even ring (kernels_bwd.hip: fold_phi_kernel).
  float* folded = nullptr;
  const float* dy_full = dy;
  const Dims3 dout_full = dout;
  if (dout.h & 1) {
    folded = ws->get<float>((size_t)r.B * dout.d * (dout.h - 1) * dout.w * c);
    if (!r.dry()) launch_fold_phi(dy, folded, r.B, dout, c, r.s);
    dy = folded;
    dout.h -= 1;
  }
  if (dx) {
    // dx[i][ci] = sum_k dy[s*i + k - 1][co] w[ci][co][k]: a strided conv of dy with w viewed as (co' = ci, ci' = co)
    float* wp = pre ? nullptr : ws->get<float>(packed_weight_floats(c, c, T));
    float* wp3 = pre ? nullptr : ws->get<float>(packed_split16_bytes(c, c, T) / 4);
    if (!r.dry()) {
      if (!pre) {
        launch_pack_weights(w_raw, wp, c, c, T, false, r.s);
        launch_pack_weights_split16(w_raw, wp3, c, c, T, r.s, false, false);
      }
      ConvGeom gd{dout, din, kz, 4, 4, sz, 2, 2};
      ConvFusion fu;
      fu.wpk_bf16x3 = pre ? img->pk3 : wp3;
      fu.in_absmax = launch_absmax_bits(dy, (size_t)r.B * dout.vox() * c, r.s);
      launch_conv_mfma(dy, c, nullptr, 0, pre ? img->pk : wp, nullptr, dx, r.B, c, gd, r.s, fu);
    }
    if (wp3) ws->release(wp3);
    if (wp) ws->release(wp);
  }
  // dw[ci][co][k] = sum_i x[i][ci] * dy[s*i + k - 1][co]: the strided-conv weight gradient with the two tensors' roles swapped
  float* part = ws->get<float>(wgrad_partial_floats(din.vox(), r.B, false, c, c, T));
  if (!r.dry()) {
    launch_wgrad(x, c, din, dy, c, c, 0, dout, kz, 4, 4, sz, 2, r.B, false, part, dw, false, false, r.s);
    absmax_note_drop();
  }
  ws->release(part);
  if (db) bias_grad(r, dy_full, c, dout_full.vox(), db);
  if (folded) ws->release(folded);
}

// Descriptor of the deepest level for the one-launch form (kernels_deep.hip); false if the level does not qualify.
bool deep_level_desc(const CdPlan* p, const float* emb, DeepLevelDesc* out) {
  const CdUnetDesc& d = p->desc;
  const int nres = p->nres;
  if (conv_precision() != PREC_F16X2) return false;  // (the full-range precisions keep the per-op kernels)
  DeepLevelDesc L;
  L.dims = p->shapes[nres - 1];
  L.Ca = d.layer_sizes[nres - 1]; L.Cb = d.layer_sizes[nres]; L.groups = d.groups;
  const ResW* rw[6] = {&p->downs[nres - 1].r1, &p->downs[nres - 1].r2, &p->mid1, &p->mid2, &p->ups[0].r1, &p->ups[0].r2};
  for (int i = 0; i < 6; ++i) {
    const ResW& w = *rw[i];
    DeepLevelDesc::Res& r = L.res[i];
    const bool cat = i == 4;  // ups r1 reads cat(x, skip): two Cb-wide halves
    r.c0 = cat ? L.Cb : w.cin; r.c1 = cat ? w.cin - L.Cb : 0; r.cout = w.cout;
    if (!p->packed3(w.c1w) || !p->packed3(w.c2w)) return false;
    r.w1 = (const char*)p->packed3(w.c1w) + packed_bf16x3_bytes(w.cin, w.cout, 27);
    r.w2 = (const char*)p->packed3(w.c2w) + packed_bf16x3_bytes(w.cout, w.cout, 27);
    r.b1 = p->raw(w.c1b); r.b2 = p->raw(w.c2b); r.g1 = p->raw(w.n1g); r.be1 = p->raw(w.n1b); r.g2 = p->raw(w.n2g); r.be2 = p->raw(w.n2b);
    if (w.has_mlp && emb) { r.emb = emb + w.emb_off; r.emb_ld = p->emb_ld; }
    if (w.has_res) {
      if (!p->packed3(w.rw)) return false;
      r.wres = (const char*)p->packed3(w.rw) + packed_bf16x3_bytes(w.cin, w.cout, 1);
      r.bres = p->raw(w.rb);
    }
  }
  const AttnW* aw[3] = {&p->downs[nres - 1].attn, &p->mid_attn, &p->ups[0].attn};
  const bool on[3] = {d.block_attn != 0, d.mid_attn != 0, d.block_attn != 0};
  for (int i = 0; i < 3; ++i) {
    L.has_attn[i] = on[i] ? 1 : 0;
    if (!on[i]) continue;
    const AttnW& w = *aw[i];
    DeepLevelDesc::Attn& a = L.attn[i];
    a.C = w.c; a.ng = p->raw(w.ng); a.nb = p->raw(w.nb); a.wout = p->raw(w.ow); a.bout = p->raw(w.ob); a.gg = p->raw(w.gg); a.gb = p->raw(w.gb);
    if (!p->packed3(w.qkv)) return false;
    a.wqkv = (const char*)p->packed3(w.qkv) + packed_bf16x3_bytes(w.c, 96, 1);
  }
  if (L.res[0].c0 != L.Ca || L.res[0].cout != L.Cb || L.res[4].cout != L.Ca || L.res[4].c1 != L.Cb || L.res[5].cout != L.Ca) return false;
  if (!deep_level_eligible(L)) return false;
  *out = L;
  return true;
}

// CondUnet.forward after init_conv / embeddings (models.py:713-748). Takes ownership of h (a workspace block).
// `lazy`: see res_block -- when set on return, the result is the final block's raw conv output, *xin its (still allocated) input.
float* unet_body(CdPlan* p, Run& r, const float* emb, float* h, LazyClose* lazy = nullptr, float** xin = nullptr) {
  const CdUnetDesc& d = p->desc;
  const int nres = p->nres;
  const int zs = d.compress_z ? 2 : 1;
  std::vector<float*> skips(nres, nullptr);
  float* x = h;
  int cx = d.layer_sizes[0];
  // The deepest level (downs[-1], the mid blocks, ups[0] up to its transposed conv) as ONE launch where a sample is <= 128 voxels
  DeepLevelDesc deep;
  const bool deep_on = deep_level_desc(p, emb, &deep);
  for (int i = 0; i < nres; ++i) {
    const Dims3 dims = p->shapes[i];
    if (deep_on && i == nres - 1) {
      float* y = r.ws->get<float>((size_t)r.B * dims.vox() * cx);
      if (!r.dry()) launch_deep_level(deep, x, y, r.B, r.status, r.s);
      r.ws->release(x);
      x = y;
      break;
    }
    float* t = res_block(r, resolve(p, p->downs[i].r1, emb), x, cx, nullptr, 0, dims);
    r.ws->release(x);
    x = t; cx = p->downs[i].r1.cout;
    float* xp = nullptr;
    int xu = 0;
    t = res_block(r, resolve(p, p->downs[i].r2, emb), x, cx, nullptr, 0, dims, d.block_attn ? &xp : nullptr, &xu);
    r.ws->release(x);
    x = t;
    if (d.block_attn) {
      t = attn_block(r, resolve(p, p->downs[i].attn), x, dims, xp, xu);
      r.ws->release(xp);
      r.ws->release(x);
      x = t;
    }
    skips[i] = x;
    if (i + 1 < nres) {
      const Dims3 nd = p->shapes[i + 1];
      float* y = r.ws->get<float>((size_t)r.B * nd.vox() * cx);
      if (!r.dry()) {
        ConvGeom g{dims, nd, 3, 4, 4, zs, 2, 2};
        ConvFusion fu;
        fu.wpk_bf16x3 = p->packed3(p->downs[i].sw);
        launch_conv_mfma(x, cx, nullptr, 0, p->packed(p->downs[i].sw), p->raw(p->downs[i].sb), y, r.B, cx, g, r.s, fu);
      }
      x = y;
    } else {
      // Identity: the last level's skip and the running tensor are the same tensor (models.py:719-720)
      x = skips[i];
    }
  }
  const Dims3 md = p->shapes[nres - 1];
  float* t = nullptr;
  if (!deep_on) {
    float* mp = nullptr;
    int mu = 0;
    t = res_block(r, resolve(p, p->mid1, emb), x, cx, nullptr, 0, md, d.mid_attn ? &mp : nullptr, &mu);
    // x aliases skips[nres-1]: keep it alive for the concat
    x = t;
    if (d.mid_attn) {
      t = attn_block(r, resolve(p, p->mid_attn), x, md, mp, mu);
      r.ws->release(mp);
      r.ws->release(x);
      x = t;
    }
    t = res_block(r, resolve(p, p->mid2, emb), x, cx, nullptr, 0, md);
    r.ws->release(x);
    x = t;
  }

  for (int i = 0; i < nres; ++i) {
    const int lv = nres - 1 - i;
    const Dims3 dims = p->shapes[lv];
    const int cs = d.layer_sizes[lv + 1];  // width of the skip (and of x after the previous stage)
    if (!(deep_on && i == 0)) {  // (the deep-level launch already ran ups[0]'s blocks: x is their output, layer_sizes[lv] wide)
      CD_REQUIRE(cx == cs, "internal: up path width mismatch");
      t = res_block(r, resolve(p, p->ups[i].r1, emb), x, cx, skips[lv], cs, dims);
      r.ws->release(x);
      r.ws->release(skips[lv]);
      x = t; cx = p->ups[i].r1.cout;
      float* up = nullptr;
      int uu = 0;
      t = res_block(r, resolve(p, p->ups[i].r2, emb), x, cx, nullptr, 0, dims, d.block_attn ? &up : nullptr, &uu);
      r.ws->release(x);
      x = t;
      if (d.block_attn) {
        t = attn_block(r, resolve(p, p->ups[i].attn), x, dims, up, uu);
        r.ws->release(up);
        r.ws->release(x);
        x = t;
      }
    }
    if (i + 1 < nres) {
      const Dims3 od = p->up_out[i];
      float* y = r.ws->get<float>((size_t)r.B * od.vox() * cx);
      if (!r.dry())
        launch_conv_transpose_mfma(x, cx, p->packed(p->ups[i].sw), p->raw(p->ups[i].sb), y, r.B, cx, dims, od, p->up_kz[i], zs, r.s,
                                   p->packed3(p->ups[i].sw), r.status);
      r.ws->release(x);
      x = y;
    }
  }
  t = res_block(r, resolve(p, p->fin, nullptr), x, cx, nullptr, 0, p->shapes[0], nullptr, nullptr, lazy);
  if (lazy && lazy->on) *xin = x;  // (the head reads it as the shortcut; released by the caller)
  else r.ws->release(x);
  return t;
}

void check_ready(CdPlan* p, bool need_coords) {
  // need_coords = a denoise-based entry point: the reference's do_time_embed raises KeyError for TIME_EMBED 'sin'
  // (calodiffusion.py:148-152); only CondUnet.forward reaches the sinusoidal embeddings
  if (need_coords && (p->desc.time_sin || p->desc.cond_sin))
    throw Fail{CD_EINVAL, "sinusoidal time/cond embeddings are reachable through cd_unet_forward only (the reference's denoise "
                          "path raises KeyError for TIME_EMBED 'sin')"};
  for (auto& w : p->weights)
    if (!w.set) throw Fail{CD_EWEIGHTS, "weight '" + w.name + "' was never set (cd_plan_set_weight)"};
  if (need_coords && (p->desc.rz_input || p->desc.phi_input) && !p->coords_set)
    throw Fail{CD_EWEIGHTS, "coordinate profiles were never set (cd_plan_set_coords)"};
}

EmbedArgs embed_args(CdPlan* p, int B, const float* cond, const float* t, int kind, float* emb, float* scal) {
  const CdUnetDesc& d = p->desc;
  EmbedArgs e;
  e.cond = cond; e.time_or_sigma = t; e.time_kind = kind; e.sigma_data = d.sigma_data;
  e.cond_size = d.cond_size; e.half = d.cond_dim / 2;
  e.cond_hidden = d.cond_size > e.half / 2 ? d.cond_size : e.half / 2;
  e.time_sin = d.time_sin; e.cond_sin = d.cond_sin;
  e.tw1 = d.time_sin ? nullptr : p->raw(p->tw[0]); e.tb1 = d.time_sin ? nullptr : p->raw(p->tb[0]);
  e.tw2 = p->raw(p->tw[1]); e.tb2 = p->raw(p->tb[1]);
  e.tw3 = p->raw(p->tw[2]); e.tb3 = p->raw(p->tb[2]);
  e.cw1 = d.cond_sin ? nullptr : p->raw(p->cw[0]); e.cb1 = d.cond_sin ? nullptr : p->raw(p->cb[0]);
  e.cw2 = p->raw(p->cw[1]); e.cb2 = p->raw(p->cb[1]);
  e.cw3 = p->raw(p->cw[2]); e.cb3 = p->raw(p->cb[2]);
  e.layers = p->d_embed_layers; e.n_layers = p->n_embed_layers; e.emb = emb; e.emb_ld = p->emb_ld; e.scal = scal; e.batch = B;
  return e;
}

// The init conv's coordinate channels (R, Z, phi images) and bias contribute the same (vox, C0) tensor to every sample and
// step: kept in the plan, recomputed whenever the init conv's parameters or the coordinate profiles are (re)set.
static void init_conv_coord_args(CdPlan* p, InitConvArgs& a) {
  const CdUnetDesc& d = p->desc;
  a.cin = d.in_channels; a.wpk = p->packed(p->init_w); a.bias = p->raw(p->init_b); a.cout = d.layer_sizes[0]; a.dims = p->shapes[0];
  a.cx = 1; a.use_rz = d.rz_input; a.use_phi = d.phi_input;
  a.r_w = p->d_coords; a.z_d = p->d_coords + d.grid[2]; a.phi_h = p->d_coords + d.grid[2] + d.grid[0];
  a.coord_table = p->d_init_table;
}
static void refresh_init_table(CdPlan* p, hipStream_t s) {
  if (!p->weights[p->init_w].set || !p->weights[p->init_b].set) return;
  InitConvArgs a;
  init_conv_coord_args(p, a);
  launch_init_coord_table(a, s);
}

// shared by cd_unet_forward (raw = true) and cd_denoise; workspace must have been reset by the caller
// Options of the sampler loop (cd_ddim_sample): embeddings / scalings already in place (computed a chunk of steps ahead), and the
// sampler's update of the running sample fused into the head kernel.
struct FwdOpts {
  float* emb_pre = nullptr;   // (B, emb_ld) ready-made: no embedding launch
  float* scal_pre = nullptr;  // (B, 4)
  const HeadArgs* upd = nullptr;  // only the upd_* fields are read
};
void forward_impl(CdPlan* p, int B, const float* x, const float* cond, const float* t, float* out, bool raw, hipStream_t s,
                  const FwdOpts* opt = nullptr) {
  const CdUnetDesc& d = p->desc;
  const Dims3 dims = p->shapes[0];
  Run r{&p->ws, s, B, d.groups};
  r.status = p->status_word;
  const bool pre = opt && opt->emb_pre;
  float* emb = pre ? opt->emb_pre : p->ws.get<float>((size_t)B * p->emb_ld);
  float* scal = pre ? opt->scal_pre : p->ws.get<float>((size_t)B * 4);
  float* h = p->ws.get<float>((size_t)B * dims.vox() * d.layer_sizes[0]);
  if (!r.dry()) {
    // (running this launch beside the init conv on a second stream was measured: no gain inside the step graph)
    if (!pre) launch_embed(embed_args(p, B, cond, t, raw ? CD_TIME_RAW : d.time_embed_kind, emb, raw ? nullptr : scal), s);
    InitConvArgs a;
    a.x = x; a.cin = d.in_channels; a.wpk = p->packed(p->init_w); a.bias = p->raw(p->init_b); a.out = h; a.batch = B;
    a.cout = d.layer_sizes[0]; a.dims = dims;
    if (raw) {
      a.cx = d.in_channels;
    } else {
      a.cx = 1; a.sigma_b = t; a.sigma_data = d.sigma_data; a.use_rz = d.rz_input; a.use_phi = d.phi_input;
      a.r_w = p->d_coords; a.z_d = p->d_coords + d.grid[2]; a.phi_h = p->d_coords + d.grid[2] + d.grid[0];
      a.coord_table = p->d_init_table; a.table_ready = true; a.status = r.status;
    }
    launch_init_conv(a, s);
  }
  LazyClose lazy;
  float* xin = nullptr;
  static const bool head_fused = getenv("CD_NO_HEAD_GN") == nullptr;
  float* hf = unet_body(p, r, emb, h, head_fused ? &lazy : nullptr, &xin);
  if (!r.dry()) {
    HeadArgs ha;
    ha.h = hf; ha.w = p->raw(p->head_w); ha.bias = p->raw(p->head_b); ha.out = out; ha.batch = B; ha.vox = dims.vox();
    if (!raw) { ha.x = x; ha.scal = scal; ha.objective = d.objective; }
    if (lazy.on) { ha.defer = lazy.gn; ha.res = xin; }
    if (opt && opt->upd) {
      ha.upd_stepvals = opt->upd->upd_stepvals; ha.upd_noise = opt->upd->upd_noise; ha.upd_x_next = opt->upd->upd_x_next;
      ha.upd_xs = opt->upd->upd_xs; ha.upd_x0s = opt->upd->upd_x0s;
    }
    launch_head(ha, s);
  }
  if (lazy.on) {
    r.ws->release(lazy.part);
    r.ws->release(xin);
  }
  r.ws->release(hf);
  if (!pre) {
    r.ws->release(scal);
    r.ws->release(emb);
  }
}

#include "train.inc"

// Every C-ABI entry point runs inside guarded().  The launchers check `hipGetLastError()` after each launch, and that call
// reports the thread's LAST error whoever set it -- a HIP call that failed earlier in the process (another library's, the
// caller's own, or a previous entry point of this one) would otherwise fail the first unrelated launch here (round 3:
// a refused hipEventElapsedTime surfaced as "cd_randn: invalid resource handle").  So the error state is cleared on the way
// in, and again on the way out of a failed call.
template <typename F>
int guarded(F&& f) {
  (void)hipGetLastError();
  try {
    f();
    return CD_OK;
  } catch (const Fail& e) {
    (void)hipGetLastError();
    set_error(e.msg);
    return e.code;
  } catch (const std::exception& e) {
    (void)hipGetLastError();
    set_error(std::string("internal error: ") + e.what());
    return CD_EINVAL;
  }
}

void destroy_graph(CdPlan* p) {
  if (p->graph_exec) {
    hipGraphExecDestroy(p->graph_exec);
    p->graph_exec = nullptr;
  }
  if (p->graph_exec_chunk) {
    hipGraphExecDestroy(p->graph_exec_chunk);
    p->graph_exec_chunk = nullptr;
  }
}
void destroy_prog_graph(CdPlan* p) {
  if (p->prog_exec) {
    hipGraphExecDestroy(p->prog_exec);
    p->prog_exec = nullptr;
  }
}

// High-water mark of a dry run of the forward's allocation sequence (after `front`, the caller's own blocks).  The sequence
// depends on the convolution precision (whole-block launches, fused / unfused attention), and a range fallback re-runs a call
// in bf16x3 on the SAME workspace: the answer is the larger of the precision in force and the fallback's.
template <typename F>
size_t dry_forward_bytes(CdPlan* plan, int batch, F&& front) {
  struct Restore {
    ~Restore() { set_conv_precision_override(-1); }
  } restore;
  size_t need = 0;
  // (all three arithmetic modes, not only the one in force and the fallback's: cd_set_conv_precision may switch after the caller
  // sized -- and cached -- its workspace)
  for (int mode : {(int)PREC_F16X2, (int)PREC_BF16X3, (int)PREC_F32}) {
    set_conv_precision_override(mode);
    plan->ws.reset(nullptr, 0, true);
    front();
    forward_impl(plan, batch, nullptr, nullptr, nullptr, nullptr, false, nullptr);
    need = plan->ws.high() > need ? plan->ws.high() : need;
  }
  return need;
}

// Range fallback of the entry points that promise finite results (the samplers, cd_denoise_safe): `run(eager)` enqueues the
// whole call.  The f16x2 kernels of THIS call raise bit 0 of a private word (d_counter + 3, cleared first), so a flag left in
// the sticky word by an earlier, un-queried cd_denoise / cd_unet_forward / cd_train_step is neither mistaken for this call's
// overflow nor lost.  If the call left the fp16 range it is run again with the exact bf16x3 convolutions (full fp32 range;
// eagerly, a cached step graph holds the f16x2 kernels) -- the precision is overridden for THIS THREAD only, other plans /
// threads of the process keep their kernels -- and bit 1 is OR-ed into the sticky word.  Returns whether the fallback ran.
template <typename F>
bool run_with_range_fallback(CdPlan* plan, hipStream_t s, F&& run, bool report_sticky = true) {
  struct Restore {
    CdPlan* p;
    ~Restore() {
      p->status_word = p->d_counter + 2;
      set_conv_precision_override(-1);
    }
  } restore{plan};
  plan->status_word = plan->d_counter + 3;
  CD_HIP(hipMemsetAsync(plan->d_counter + 3, 0, sizeof(int), s));
  run(false);
  if (conv_precision() != PREC_F16X2) return false;
  int flags = 0;
  CD_HIP(hipMemcpyAsync(&flags, plan->d_counter + 3, sizeof(int), hipMemcpyDeviceToHost, s));
  CD_HIP(hipStreamSynchronize(s));
  if (!(flags & 1)) return false;
  set_conv_precision_override(PREC_BF16X3);
  run(true);
  if (report_sticky) launch_or_word(plan->d_counter + 2, 2, s);
  return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

const char* cd_last_error(void) { return g_last_error.c_str(); }
int cd_abi_version(void) { return CD_ABI_VERSION; }

int cd_device_check(char* name, int cap) {
  return guarded([&] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw Fail{CD_ENOGPU, "no HIP device visible"};
    int dev = 0;
    CD_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CD_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && cap > 0) {
      std::snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      throw Fail{CD_ENOGPU, std::string("this library is built for gfx950 only; found ") + prop.gcnArchName};
  });
}

int cd_plan_create(const CdUnetDesc* desc, CdPlan** plan) {
  return guarded([&] {
    CD_REQUIRE(desc && plan, "null argument");
    if (desc->struct_size != sizeof(CdUnetDesc))
      throw Fail{CD_EINVAL, "CdUnetDesc.struct_size is " + std::to_string(desc->struct_size) + ", this library expects " +
                                std::to_string(sizeof(CdUnetDesc)) + " (ABI version " + std::to_string(CD_ABI_VERSION) +
                                "): the binding was written against another calodiff.h"};
    std::unique_ptr<CdPlan> p(new CdPlan());
    p->desc = *desc;
    build_plan(p.get());
    *plan = p.release();
  });
}

int cd_plan_destroy(CdPlan* plan) {
  return guarded([&] {
    if (!plan) return;
    destroy_graph(plan);
    destroy_prog_graph(plan);
    if (plan->cap_stream) hipStreamDestroy(plan->cap_stream);
    if (plan->d_pack_jobs) hipFree(plan->d_pack_jobs);
    if (plan->d_dg_jobs) hipFree(plan->d_dg_jobs);
    if (plan->dg_arena) hipFree(plan->dg_arena);
    if (plan->wq_region) hipFree(plan->wq_region);
    if (plan->arena) hipFree(plan->arena);
    if (plan->d_embed_layers) hipFree(plan->d_embed_layers);
    if (plan->d_coords) hipFree(plan->d_coords);
    if (plan->d_init_table) hipFree(plan->d_init_table);
    if (plan->d_table) hipFree(plan->d_table);
    if (plan->d_counter) hipFree(plan->d_counter);
    if (plan->d_stepvals) hipFree(plan->d_stepvals);
    if (plan->d_lin_jobs) hipFree(plan->d_lin_jobs);
    delete plan;
  });
}

int cd_plan_num_weights(const CdPlan* plan, int* n) {
  return guarded([&] {
    CD_REQUIRE(plan && n, "null argument");
    *n = (int)plan->weights.size();
  });
}

int cd_plan_weight_name(const CdPlan* plan, int idx, char* name, int cap, int64_t* numel) {
  return guarded([&] {
    CD_REQUIRE(plan && idx >= 0 && idx < (int)plan->weights.size(), "weight index out of range");
    if (name && cap > 0) std::snprintf(name, cap, "%s", plan->weights[idx].name.c_str());
    if (numel) *numel = plan->weights[idx].numel;
  });
}

int cd_plan_set_weight(CdPlan* plan, const char* name, const float* dev_ptr, int64_t numel, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && name && dev_ptr, "null argument");
    auto it = plan->index.find(name);
    if (it == plan->index.end()) throw Fail{CD_EWEIGHTS, std::string("unknown weight name '") + name + "'"};
    WeightEntry& w = plan->weights[it->second];
    if (w.numel != numel)
      throw Fail{CD_EWEIGHTS, std::string("weight '") + name + "' has " + std::to_string(numel) + " elements, expected " + std::to_string(w.numel)};
    hipStream_t s = (hipStream_t)stream;
    CD_HIP(hipMemcpyAsync(plan->arena + w.raw_off, dev_ptr, sizeof(float) * (size_t)numel, hipMemcpyDeviceToDevice, s));
    if (w.pack == PK_CONV) launch_pack_weights(plan->arena + w.raw_off, plan->arena + w.pk_off, w.cout, w.cin, w.taps, false, s);
    if (w.pack == PK_CONVT) {
      launch_pack_weights(plan->arena + w.raw_off, plan->arena + w.pk_off, w.cout, w.cin, w.taps, true, s);
      launch_pack_weights_f16x2(plan->arena + w.raw_off, plan->arena + w.pk3_off, w.cout, w.cin, w.taps, s, true, false);
    } else if (w.pk3_off) {
      launch_pack_weights_split16(plan->arena + w.raw_off, plan->arena + w.pk3_off, w.cout, w.cin, w.taps, s);
    }
    else if (w.pack == PK_INIT) launch_pack_init_weights(plan->arena + w.raw_off, plan->arena + w.pk_off, w.cout, w.cin, s);
    w.set = true;
    if (it->second == plan->init_w || it->second == plan->init_b) refresh_init_table(plan, s);
  });
}

int cd_plan_set_weights(CdPlan* plan, int n, const float* const* dev_ptrs, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && dev_ptrs, "null argument");
    if (n != (int)plan->weights.size())
      throw Fail{CD_EWEIGHTS, "cd_plan_set_weights: " + std::to_string(n) + " tensors given, the plan has " +
                                  std::to_string(plan->weights.size())};
    hipStream_t s = (hipStream_t)stream;
    bool changed = plan->pack_jobs.size() != plan->weights.size();
    if (changed) plan->pack_jobs.assign(plan->weights.size(), PackJob{});
    for (int i = 0; i < n; ++i) {
      CD_REQUIRE(dev_ptrs[i], "cd_plan_set_weights: null tensor pointer");
      const WeightEntry& w = plan->weights[i];
      PackJob& j = plan->pack_jobs[i];
      if (j.src != dev_ptrs[i]) changed = true;
      j.src = dev_ptrs[i];
      j.raw = plan->arena + w.raw_off;
      j.cout = w.cout; j.cin = w.cin; j.taps = w.taps; j.kind = (int)w.pack;
      j.numel = (unsigned long long)w.numel;
      j.pk = nullptr; j.bf3 = nullptr; j.f16 = nullptr; j.n_pk = j.n_bf3 = j.n_f16 = 0;
      const unsigned long long n16 = w.pack == PK_CONV || w.pack == PK_CONVT
                                         ? (unsigned long long)(w.cin / 16) * w.taps * ((w.cout + 31) / 32) * 64 : 0ull;
      if (w.pack == PK_CONV || w.pack == PK_CONVT) {
        j.pk = plan->arena + w.pk_off;
        j.n_pk = packed_weight_floats(w.cin, w.cout, w.taps);
      } else if (w.pack == PK_INIT) {
        j.pk = plan->arena + w.pk_off;
        j.n_pk = (unsigned long long)w.cout * w.cin * 27;
      }
      if (w.pack == PK_CONVT) {  // (f16x2 image only, in transposed channel order)
        j.f16 = plan->arena + w.pk3_off;
        j.n_f16 = n16;
      } else if (w.pack == PK_CONV && w.pk3_off) {
        j.bf3 = plan->arena + w.pk3_off;
        j.f16 = (char*)(plan->arena + w.pk3_off) + packed_bf16x3_bytes(w.cin, w.cout, w.taps);
        j.n_bf3 = j.n_f16 = n16;
      }
    }
    if (!plan->d_pack_jobs) CD_HIP(hipMalloc((void**)&plan->d_pack_jobs, sizeof(PackJob) * plan->weights.size()));
    if (changed) {
      // (rare: torch keeps a parameter's storage across optimizer steps) the kernels of the previous call may still read the list
      CD_HIP(hipStreamSynchronize(s));
      CD_HIP(hipMemcpy(plan->d_pack_jobs, plan->pack_jobs.data(), sizeof(PackJob) * plan->pack_jobs.size(), hipMemcpyHostToDevice));
    }
    launch_pack_jobs(plan->d_pack_jobs, n, s);
    launch_pack_jobs_f16x2(plan->d_pack_jobs, n, s);
    for (auto& w : plan->weights) w.set = true;
    refresh_init_table(plan, s);
  });
}

int cd_plan_set_coords(CdPlan* plan, const float* r_w, const float* z_d, const float* phi_h, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && r_w && z_d && phi_h, "null argument");
    const CdUnetDesc& d = plan->desc;
    hipStream_t s = (hipStream_t)stream;
    CD_HIP(hipMemcpyAsync(plan->d_coords, r_w, sizeof(float) * d.grid[2], hipMemcpyHostToDevice, s));
    CD_HIP(hipMemcpyAsync(plan->d_coords + d.grid[2], z_d, sizeof(float) * d.grid[0], hipMemcpyHostToDevice, s));
    CD_HIP(hipMemcpyAsync(plan->d_coords + d.grid[2] + d.grid[0], phi_h, sizeof(float) * d.grid[1], hipMemcpyHostToDevice, s));
    CD_HIP(hipStreamSynchronize(s));  // the host arrays may be temporaries
    plan->coords_set = true;
    refresh_init_table(plan, s);
  });
}

int cd_plan_workspace_bytes(CdPlan* plan, int batch, size_t* bytes) {
  return guarded([&] {
    CD_REQUIRE(plan && bytes && batch > 0, "bad argument");
    const int64_t n = (int64_t)batch * plan->shapes[0].vox();
    *bytes = dry_forward_bytes(plan, batch, [&] {
      // superset of what any entry point allocates around forward_impl: x0 / noise / x_noisy, sigma, partials
      plan->ws.get<float>((size_t)n);
      plan->ws.get<float>((size_t)n);
      plan->ws.get<float>((size_t)batch + 64);
      plan->ws.get<double>((size_t)batch + 8);
      // cd_ddim_sample: this step's embeddings / scalings and the chunk computed ahead
      plan->ws.get<float>((size_t)batch * plan->emb_ld);
      plan->ws.get<float>((size_t)batch * 4);
      plan->ws.get<float>((size_t)CdPlan::kEmbedChunk * batch * plan->emb_ld);
      plan->ws.get<float>((size_t)CdPlan::kEmbedChunk * batch * 4);
    }) + 4096;
  });
}

int cd_unet_forward(CdPlan* plan, int batch, const float* x, const float* cond, const float* time, float* out,
                    void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && x && cond && time && out && workspace && batch > 0, "bad argument");
    check_ready(plan, false);
    plan->ws.reset((char*)workspace, workspace_bytes, false);
    forward_impl(plan, batch, x, cond, time, out, true, (hipStream_t)stream);
  });
}

int cd_denoise(CdPlan* plan, int batch, const float* x, const float* sigma, const float* cond, float* out,
               void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && x && sigma && cond && out && workspace && batch > 0, "bad argument");
    check_ready(plan, true);
    plan->ws.reset((char*)workspace, workspace_bytes, false);
    forward_impl(plan, batch, x, cond, sigma, out, false, (hipStream_t)stream);
  });
}

int cd_denoise_safe(CdPlan* plan, int batch, const float* x, const float* sigma, const float* cond, float* out,
                    void* workspace, size_t workspace_bytes, int* fell_back, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && x && sigma && cond && out && workspace && batch > 0, "bad argument");
    check_ready(plan, true);
    const bool fb = run_with_range_fallback(plan, (hipStream_t)stream, [&](bool) {
      plan->ws.reset((char*)workspace, workspace_bytes, false);
      forward_impl(plan, batch, x, cond, sigma, out, false, (hipStream_t)stream);
    }, /*report_sticky=*/fell_back == nullptr);
    if (fell_back) *fell_back = fb ? 1 : 0;
  });
}

int cd_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int64_t* numel, double lr, double beta1, double beta2, float eps, float weight_decay, int step, void* stream) {
  return guarded([&] {
    CD_REQUIRE(n >= 0 && (n == 0 || (params && grads && exp_avg && exp_avg_sq && numel)) && step >= 1, "bad argument");
    for (int i0 = 0; i0 < n; i0 += 48) {
      AdamChunk c{};
      const int k = n - i0 < 48 ? n - i0 : 48;
      int64_t mx = 0;
      for (int j = 0; j < k; ++j) {
        c.p[j] = params[i0 + j]; c.g[j] = grads[i0 + j]; c.m[j] = exp_avg[i0 + j]; c.v[j] = exp_avg_sq[i0 + j]; c.n[j] = numel[i0 + j];
        CD_REQUIRE(c.p[j] && c.g[j] && c.m[j] && c.v[j] && c.n[j] >= 0, "adam: null tensor pointer");
        if (c.n[j] > mx) mx = c.n[j];
      }
      launch_adam(c, k, mx, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
    }
  });
}

int cd_reverse_norm(const float* voxels, const float* energy, const float* layerE, float* out, int batch, const int32_t dims[3],
                    const float consts[6], float max_deposit, float ecut, void* stream) {
  return guarded([&] {
    CD_REQUIRE(voxels && energy && out && dims && consts && batch > 0, "bad argument");
    ReverseNormArgs a;
    a.voxels = voxels; a.energy = energy; a.layerE = layerE; a.out = out; a.batch = batch;
    a.D = dims[0]; a.H = dims[1]; a.W = dims[2]; a.layer_mode = layerE ? 1 : 0;
    a.logit_mean = consts[0]; a.logit_std = consts[1]; a.totalE_mean = consts[2]; a.totalE_std = consts[3];
    a.layers_mean = consts[4]; a.layers_std = consts[5]; a.max_deposit = max_deposit; a.ecut = ecut;
    launch_reverse_norm(a, (hipStream_t)stream);
  });
}

int cd_reverse_norm_staged(const float* voxels, const float* energy, const float* layerE, float* out, int batch,
                           const int32_t dims[3], const float consts[6], float max_deposit, float ecut, float alpha, float layer_eps,
                           int stage, void* stream) {
  return guarded([&] {
    CD_REQUIRE(voxels && out && dims && consts && batch > 0 && stage >= 0 && stage <= 2, "bad argument");
    CD_REQUIRE(stage == 1 || energy, "cd_reverse_norm_staged: stages 0 and 2 scale by the incident energies");
    ReverseNormArgs a;
    a.voxels = voxels; a.energy = energy; a.layerE = stage == 1 ? nullptr : layerE; a.out = out; a.batch = batch;
    a.D = dims[0]; a.H = dims[1]; a.W = dims[2]; a.layer_mode = a.layerE ? 1 : 0;
    a.logit_mean = consts[0]; a.logit_std = consts[1]; a.totalE_mean = consts[2]; a.totalE_std = consts[3];
    a.layers_mean = consts[4]; a.layers_std = consts[5]; a.max_deposit = max_deposit; a.ecut = ecut;
    a.stage = stage; a.alpha = alpha; a.layer_eps = layer_eps;
    launch_reverse_norm(a, (hipStream_t)stream);
  });
}

static void layer_mlp_call(const CdLayerMlpDesc* d, const float* const* weights, int n_weights, int batch, int mode,
                           const float* x, const float* cond, const float* tsig, const float* table, int n_steps,
                           const float* noise, float* out, float* xs, float* x0s, void* stream) {
  CD_REQUIRE(d && weights && x && cond && out && batch > 0, "bad argument");
  CD_REQUIRE(d->struct_size == sizeof(CdLayerMlpDesc), "CdLayerMlpDesc.struct_size does not match this library's calodiff.h");
  CD_REQUIRE(d->n_res >= 0 && d->n_res <= 8 && n_weights == 2 * (8 + 3 * d->n_res),
             "layer MLP: n_weights must be 2*(8 + 3*n_res) (time_mlp, cond_mlp, in_lay, blocks, out_lay)");
  CD_REQUIRE(d->time_embed_kind >= 0 && d->time_embed_kind <= 2 && d->objective >= 0 && d->objective <= 2, "bad descriptor");
  LayerMlpArgs a{};
  for (int i = 0; i < n_weights; ++i) {
    CD_REQUIRE(weights[i], "null weight pointer");
    a.w[i] = weights[i];
  }
  a.dim_in = d->dim_in; a.hidden = d->hidden; a.cond_emb = d->cond_emb; a.cond_size = d->cond_size; a.n_res = d->n_res;
  a.time_kind = d->time_embed_kind; a.objective = d->objective; a.mode = mode; a.batch = batch; a.n_steps = n_steps;
  a.sigma_data = d->sigma_data;
  a.x = x; a.cond = cond; a.tsig = tsig; a.table = table; a.noise = noise; a.out = out; a.xs = xs; a.x0s = x0s;
  launch_layer_mlp(a, (hipStream_t)stream);
}

int cd_layer_forward(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* x,
                     const float* cond, const float* time, float* out, void* stream) {
  return guarded([&] {
    CD_REQUIRE(time, "bad argument");
    layer_mlp_call(desc, weights, n_weights, batch, 0, x, cond, time, nullptr, 1, nullptr, out, nullptr, nullptr, stream);
  });
}
int cd_layer_denoise(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* x,
                     const float* sigma, const float* cond, float* out, void* stream) {
  return guarded([&] {
    CD_REQUIRE(sigma, "bad argument");
    layer_mlp_call(desc, weights, n_weights, batch, 1, x, cond, sigma, nullptr, 1, nullptr, out, nullptr, nullptr, stream);
  });
}
int cd_layer_sample(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* start,
                    const float* cond, const CdStep* steps_dev, int n_steps, const float* step_noise, float* x_out, float* xs,
                    float* x0s, void* stream) {
  return guarded([&] {
    CD_REQUIRE(steps_dev && n_steps > 0, "bad argument");
    layer_mlp_call(desc, weights, n_weights, batch, 2, start, cond, nullptr, (const float*)steps_dev, n_steps, step_noise,
                   x_out, xs, x0s, stream);
  });
}

static LayerMlpTrainArgs layer_train_args(const CdLayerMlpDesc* d, int batch) {
  CD_REQUIRE(d && batch > 0, "bad argument");
  CD_REQUIRE(d->struct_size == sizeof(CdLayerMlpDesc), "CdLayerMlpDesc.struct_size does not match this library's calodiff.h");
  CD_REQUIRE(d->n_res >= 0 && d->n_res <= 8 && d->time_embed_kind >= 0 && d->time_embed_kind <= 2, "bad descriptor");
  LayerMlpTrainArgs a{};
  a.dim_in = d->dim_in; a.hidden = d->hidden; a.cond_emb = d->cond_emb; a.cond_size = d->cond_size; a.n_res = d->n_res;
  a.time_kind = d->time_embed_kind; a.batch = batch; a.sigma_data = d->sigma_data;
  a.layout = layer_tape_layout(a.dim_in, a.hidden, a.cond_emb, a.cond_size, a.n_res);
  return a;
}
int cd_layer_train_workspace_bytes(const CdLayerMlpDesc* desc, int batch, size_t* bytes) {
  return guarded([&] {
    CD_REQUIRE(bytes, "bad argument");
    *bytes = layer_train_workspace_bytes(layer_train_args(desc, batch));
  });
}
int cd_layer_train_step(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* data,
                        const float* noise, const float* sigma, const float* cond, double* loss_out, float* grads,
                        void* workspace, size_t workspace_bytes, void* stream) {
  return cd_layer_train_step_loss(desc, weights, n_weights, batch, data, noise, sigma, cond, CD_LOSS_L2, loss_out, grads, workspace,
                                  workspace_bytes, stream);
}

int cd_layer_train_step_loss(const CdLayerMlpDesc* desc, const float* const* weights, int n_weights, int batch, const float* data,
                             const float* noise, const float* sigma, const float* cond, int loss_type, double* loss_out,
                             float* grads, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(weights && data && noise && sigma && cond && loss_out && grads && workspace, "bad argument");
    CD_REQUIRE(loss_type >= CD_LOSS_L2 && loss_type <= CD_LOSS_HUBER, "loss_type must be one of CD_LOSS_L2 / L1 / MSE / HUBER");
    LayerMlpTrainArgs a = layer_train_args(desc, batch);
    a.loss_type = loss_type;
    CD_REQUIRE(desc->objective == CD_OBJ_HYBRID, "cd_layer_train_step implements the hybrid_weight objective");
    CD_REQUIRE(n_weights == 2 * (8 + 3 * desc->n_res), "layer MLP: n_weights must be 2*(8 + 3*n_res)");
    CD_REQUIRE(workspace_bytes >= layer_train_workspace_bytes(a), "workspace too small: call cd_layer_train_workspace_bytes");
    for (int i = 0; i < n_weights; ++i) {
      CD_REQUIRE(weights[i], "null weight pointer");
      a.w[i] = weights[i];
    }
    a.data = data; a.noise = noise; a.sigma = sigma; a.cond = cond;
    launch_layer_mlp_train(a, grads, loss_out, workspace, (hipStream_t)stream);
  });
}

int cd_set_conv_precision(const char* mode) {
  return guarded([&] {
    CD_REQUIRE(mode, "null argument");
    if (!std::strcmp(mode, "f16x2")) set_conv_precision(PREC_F16X2);
    else if (!std::strcmp(mode, "bf16x3")) set_conv_precision(PREC_BF16X3);
    else if (!std::strcmp(mode, "f32")) set_conv_precision(PREC_F32);
    else throw Fail{CD_EINVAL, std::string("unknown convolution precision '") + mode + "' (f16x2, bf16x3, f32)"};
  });
}
const char* cd_get_conv_precision(void) {
  static const char* names[3] = {"f16x2", "bf16x3", "f32"};
  return names[conv_precision()];
}

int cd_profile_begin(void) {
  return guarded([&] { prof::begin(); });
}
int cd_profile_end(char* json, int cap) {
  return guarded([&] {
    CD_REQUIRE(json && cap > 2, "bad argument");
    if (prof::end(json, cap) < 0) throw Fail{CD_EINVAL, "profile buffer too small"};
  });
}

int cd_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  return guarded([&] {
    CD_REQUIRE(out && n >= 0, "bad argument");
    launch_randn(out, n, seed, offset, (hipStream_t)stream);
  });
}

int cd_ddim_sample(CdPlan* plan, int batch, const float* start, const float* cond, const CdStep* steps, int n_steps,
                   const float* step_noise, uint64_t seed, uint64_t offset, uint64_t noise_stride, float* x_out, float* xs,
                   float* x0s, int use_graph, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && start && cond && steps && x_out && workspace && batch > 0, "bad argument");
    CD_REQUIRE(n_steps >= 1 && n_steps <= CdPlan::kMaxSteps, "n_steps out of range (1..4096)");
    check_ready(plan, true);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)batch * plan->shapes[0].vox();
    bool noisy = false;
    for (int i = 0; i < n_steps; ++i) noisy |= steps[i].ddim_sigma != 0.f;

    static_assert(sizeof(CdStep) == 16, "CdStep must be 4 floats");
    CD_HIP(hipMemcpyAsync(plan->d_table, steps, sizeof(CdStep) * n_steps, hipMemcpyHostToDevice, s));

    plan->ws.reset((char*)workspace, workspace_bytes, false);
    float* x0 = plan->ws.get<float>((size_t)n);
    float* noise_buf = plan->ws.get<float>((size_t)n);
    float* sigma_b = plan->ws.get<float>((size_t)batch + 64);
    uint64_t* noise_dev = (uint64_t*)plan->ws.get<double>(4);  // {seed, base offset, stride}
    // Embeddings and EDM scalings depend on (sigma_step, cond) only -- not on x: one launch computes them for kEmbedChunk steps
    // ahead (the per-step embedding kernel was 34 us of pure latency in every 2 ms step), load_step hands each step its slice.
    const int K = CdPlan::kEmbedChunk;
    float* emb_cur = plan->ws.get<float>((size_t)batch * plan->emb_ld);
    float* scal_cur = plan->ws.get<float>((size_t)batch * 4);
    float* emb_chunk = plan->ws.get<float>((size_t)K * batch * plan->emb_ld);
    float* scal_chunk = plan->ws.get<float>((size_t)K * batch * 4);
    // Every sample of the batch runs at the step's sigma, and the ResnetBlock projections are linear in SiLU(cat(t, c))
    // (EmbedArgs::part): a chunk needs its K TIME rows only (16 rows instead of 16 x batch: the chunk's launch was 165 us at batch 64,
    // serial with the step graphs), the batch's CONDITION rows are computed once per call; load_step adds the two.  The regions
    // above are sized for the unsplit form (CD_NO_EMBED_SPLIT), the split form uses the front of them.
    static const bool no_split = getenv("CD_NO_EMBED_SPLIT") != nullptr;
    const bool split = !no_split && plan->emb_ld % 4 == 0 && batch >= 2;  // (batch 1: K + 1 rows do not fit the K-row region, nothing to gain)
    float* emb_cond = split ? emb_chunk + (size_t)K * plan->emb_ld : nullptr;  // [batch][emb_ld] behind the K time rows
    StepChunk chunk;
    chunk.emb_src = emb_chunk; chunk.emb_dst = emb_cur; chunk.emb_floats = split ? plan->emb_ld : batch * plan->emb_ld;
    chunk.scal_src = scal_chunk; chunk.scal_dst = scal_cur; chunk.scal_floats = split ? 4 : batch * 4; chunk.chunk_steps = K;
    chunk.emb_cond = emb_cond;
    auto embed_ahead = [&](hipStream_t st, int i0) {  // steps i0 .. i0 + K - 1 (slot = step % K; i0 is a multiple of K)
      const int nst = n_steps - i0 < K ? n_steps - i0 : K;
      if (split) {
        if (i0 == 0) {  // the condition rows, once
          EmbedArgs c = embed_args(plan, batch, cond, plan->d_table, plan->desc.time_embed_kind, emb_cond, nullptr);
          c.part = 2;
          launch_embed(c, st);
        }
        EmbedArgs e = embed_args(plan, nst, cond, plan->d_table + (size_t)i0 * 4, plan->desc.time_embed_kind, emb_chunk, scal_chunk);
        e.part = 1; e.time_stride = 4;
        launch_embed(e, st);
        return;
      }
      EmbedArgs e = embed_args(plan, nst * batch, cond, plan->d_table + (size_t)i0 * 4, plan->desc.time_embed_kind, emb_chunk, scal_chunk);
      e.cond_rows = batch; e.time_stride = 4;
      launch_embed(e, st);
    };
    if (noisy && !step_noise) {
      const uint64_t so[3] = {seed, offset, noise_stride ? noise_stride : (uint64_t)n};
      CD_HIP(hipMemcpyAsync(noise_dev, so, sizeof(so), hipMemcpyHostToDevice, s));
      CD_HIP(hipStreamSynchronize(s));  // `so` lives on this stack frame
    }
    // remaining workspace for the network: a nested arena view
    const size_t used = plan->ws.high();
    char* sub = (char*)workspace + used;
    const size_t sub_bytes = workspace_bytes > used ? workspace_bytes - used : 0;

    auto one_step = [&](hipStream_t st, int i, const float* noise_i, float* xs_i, float* x0s_i) {
      launch_load_step(plan->d_table, plan->d_counter, plan->d_stepvals, sigma_b, batch, st, &chunk);
      const float* nz = noise_i;
      if (!nz && noisy) {
        // stream position = offset + i * stride, read from device memory (the step counter is i + 1 after load_step): the same
        // launch serves every step, so stochastic samplers replay one captured graph as well
        launch_randn_step(noise_buf, n, noise_dev, plan->d_counter, st);
        nz = noise_buf;
      }
      // the update of the running sample (x_out, in place) happens in the network's head kernel
      HeadArgs upd;
      upd.upd_stepvals = plan->d_stepvals; upd.upd_noise = nz; upd.upd_x_next = x_out; upd.upd_xs = xs_i; upd.upd_x0s = x0s_i;
      FwdOpts fo;
      fo.emb_pre = emb_cur; fo.scal_pre = scal_cur; fo.upd = &upd;
      plan->ws.reset(sub, sub_bytes, false);
      forward_impl(plan, batch, x_out, cond, sigma_b, x0, false, st, &fo);
    };

    run_with_range_fallback(plan, s, [&](bool eager) {
      CD_HIP(hipMemsetAsync(plan->d_counter, 0, sizeof(int), s));
      // x = start * sigma_start (sample.py:62-66); x_out doubles as the running x
      launch_scale(start, x_out, plan->d_table, n, s);
      // A hipGraph of one step can be replayed only if nothing in it depends on the host-side step index: no trajectories and
      // no caller-supplied per-step noise (the device Philox noise of a stochastic sampler reads its stream position from the
      // step counter, see one_step).
      const bool graphable = use_graph && !eager && !step_noise && !xs && !x0s && !prof::enabled();
      if (graphable) {
        CdPlan::GraphKey key;
        key.batch = batch; key.ws = workspace; key.cond = cond; key.x = x_out; key.noisy = noisy ? 1 : 0;
        key.precision = conv_precision();
        // `count` consecutive steps as one graph: every step reads its index from the device counter, so the same capture serves any
        // position in the schedule
        auto capture = [&](int count) -> hipGraphExec_t {
          if (!plan->cap_stream) CD_HIP(hipStreamCreateWithFlags(&plan->cap_stream, hipStreamNonBlocking));
          hipStream_t cs = plan->cap_stream;
          hipGraph_t graph = nullptr;
          CD_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
          try {
            for (int k = 0; k < count; ++k) one_step(cs, 0, nullptr, nullptr, nullptr);
          } catch (...) {
            hipStreamEndCapture(cs, &graph);
            if (graph) hipGraphDestroy(graph);
            throw;
          }
          CD_HIP(hipStreamEndCapture(cs, &graph));
          hipGraphExec_t exec = nullptr;
          hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
          hipGraphDestroy(graph);
          if (e != hipSuccess) throw Fail{CD_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)};
          return exec;
        };
        if (!(plan->graph_exec && plan->graph_key == key)) {
          destroy_graph(plan);
          // one eager pass first: per-geometry kernel tuning (and lazy function attributes) cannot happen during capture.
          // It only writes x0 / scratch, which the replayed steps overwrite.
          launch_load_step(plan->d_table, plan->d_counter, plan->d_stepvals, sigma_b, batch, s);
          plan->ws.reset(sub, sub_bytes, false);
          forward_impl(plan, batch, x_out, cond, sigma_b, x0, false, s);
          CD_HIP(hipMemsetAsync(plan->d_counter, 0, sizeof(int), s));
          CD_HIP(hipStreamSynchronize(s));
          plan->graph_exec = capture(1);
          plan->graph_key = key;
        }
        // Schedules of at least one embedding chunk replay the chunk's kEmbedChunk steps as ONE graph (8.5 us of idle time sat
        // between two graph launches: profiles/r04_graph_gaps.txt); the tail, and short schedules, replay the one-step graph.
        static const bool no_chunk = getenv("CD_NO_CHUNK_GRAPH") != nullptr;
        if (!no_chunk && n_steps >= K && !plan->graph_exec_chunk) plan->graph_exec_chunk = capture(K);
        for (int i = 0; i < n_steps;) {
          if (i % K == 0) embed_ahead(s, i);
          if (!no_chunk && plan->graph_exec_chunk && i % K == 0 && i + K <= n_steps) {
            CD_HIP(hipGraphLaunch(plan->graph_exec_chunk, s));
            i += K;
          } else {
            CD_HIP(hipGraphLaunch(plan->graph_exec, s));
            i += 1;
          }
        }
      } else {
        for (int i = 0; i < n_steps; ++i) {
          if (i % K == 0) embed_ahead(s, i);
          one_step(s, i, step_noise ? step_noise + (size_t)i * n : nullptr, xs ? xs + (size_t)i * n : nullptr,
                   x0s ? x0s + (size_t)i * n : nullptr);
        }
      }
    });
  });
}

// workspace of cd_sampler_run: the buffers, the coefficient table, sigma / Philox words, and the network's own
static size_t sampler_front_bytes(CdPlan* plan, int batch, int n_bufs, size_t table_floats, float** bufs, float** table, float** sigma_b,
                                  uint64_t** noise_dev) {
  const int64_t n = (int64_t)batch * plan->shapes[0].vox();
  for (int k = 1; k < n_bufs; ++k) {
    float* b = plan->ws.get<float>((size_t)n);
    if (bufs) bufs[k] = b;
  }
  float* t = plan->ws.get<float>(table_floats + 64);
  float* sg = plan->ws.get<float>((size_t)batch + 64);
  uint64_t* nd = (uint64_t*)plan->ws.get<double>(4);
  if (table) *table = t;
  if (sigma_b) *sigma_b = sg;
  if (noise_dev) *noise_dev = nd;
  return plan->ws.high();
}

int cd_plan_sampler_workspace_bytes(CdPlan* plan, int batch, int n_bufs, int n_steps, int n_coef, size_t* bytes) {
  return guarded([&] {
    CD_REQUIRE(plan && bytes && batch > 0 && n_bufs >= 2 && n_bufs <= 16 && n_steps >= 1 && n_coef >= 1, "bad argument");
    plan->ws.reset(nullptr, 0, true);
    const size_t front = sampler_front_bytes(plan, batch, n_bufs, (size_t)n_steps * n_coef, nullptr, nullptr, nullptr, nullptr);
    *bytes = front + dry_forward_bytes(plan, batch, [] {}) + 8192;
  });
}

int cd_sampler_run(CdPlan* plan, int batch, const float* start, float start_scale, const float* cond, int n_bufs, int n_steps,
                   const CdSamplerOp* ops, int n_ops, const int32_t* op_begin, const float* coefs, int n_coef,
                   const float* step_noise, uint64_t seed, uint64_t offset, uint64_t noise_stride, float* x_out, float* xs,
                   float* x0s, int use_graph, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && start && cond && ops && coefs && x_out && workspace && batch > 0, "bad argument");
    CD_REQUIRE(n_bufs >= 2 && n_bufs <= 16 && n_steps >= 1 && n_steps <= 1 << 20 && n_ops >= 1 && n_coef >= 1, "bad program size");
    check_ready(plan, true);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)batch * plan->shapes[0].vox();
    const bool uniform = op_begin == nullptr;
    if (!uniform) {
      CD_REQUIRE(op_begin[0] == 0 && op_begin[n_steps] == n_ops, "op_begin must run from 0 to n_ops");
      for (int i = 0; i < n_steps; ++i) CD_REQUIRE(op_begin[i] <= op_begin[i + 1], "op_begin must be non-decreasing");
    }
    // validate the program before anything is enqueued: a bad buffer index would be a wild device pointer
    int randn_per_step = 0;
    for (int k = 0; k < n_ops; ++k) {
      const CdSamplerOp& o = ops[k];
      CD_REQUIRE(o.kind >= CD_SOP_LINCOMB && o.kind <= CD_SOP_LINDIV, "sampler op: unknown kind");
      const bool lin = o.kind == CD_SOP_LINCOMB || o.kind == CD_SOP_LINDIV;
      const int ns = lin ? o.nsrc : (o.kind == CD_SOP_RANDN ? 0 : 1);
      CD_REQUIRE(ns >= 0 && ns <= 6 && (!lin || ns >= 1), "sampler op: 1..6 sources");
      for (int j = 0; j < ns; ++j) CD_REQUIRE(o.src[j] >= 0 && o.src[j] < n_bufs, "sampler op: source buffer out of range");
      if (o.kind == CD_SOP_RECORD) CD_REQUIRE(o.dst == 0 || o.dst == 1, "record op: dst is 0 (xs) or 1 (x0s)");
      else CD_REQUIRE(o.dst >= 0 && o.dst < n_bufs, "sampler op: destination buffer out of range");
      if (lin) CD_REQUIRE(o.col >= 0 && o.col + ns + (o.kind == CD_SOP_LINDIV ? 1 : 0) <= n_coef, "lincomb op: coefficient columns out of range");
      if (o.kind == CD_SOP_DENOISE) {
        CD_REQUIRE(o.col >= 0 && o.col < n_coef, "denoise op: sigma column out of range");
        CD_REQUIRE(o.dst != o.src[0], "denoise op: output must not alias its input");
      }
      if (o.kind == CD_SOP_RANDN) ++randn_per_step;
    }

    plan->ws.reset((char*)workspace, workspace_bytes, false);
    float* bufs[16] = {nullptr};
    bufs[0] = x_out;
    float *table = nullptr, *sigma_b = nullptr;
    uint64_t* noise_dev = nullptr;
    const size_t used = sampler_front_bytes(plan, batch, n_bufs, (size_t)n_steps * n_coef, bufs, &table, &sigma_b, &noise_dev);
    CD_REQUIRE(used <= workspace_bytes, "workspace too small: call cd_plan_sampler_workspace_bytes");
    char* sub = (char*)workspace + used;
    const size_t sub_bytes = workspace_bytes - used;
    const uint64_t stride = noise_stride ? noise_stride : (uint64_t)n;
    CD_HIP(hipMemcpyAsync(table, coefs, sizeof(float) * (size_t)n_steps * n_coef, hipMemcpyHostToDevice, s));
    {
      const uint64_t so[3] = {seed, offset, stride};
      CD_HIP(hipMemcpyAsync(noise_dev, so, sizeof(so), hipMemcpyHostToDevice, s));
      CD_HIP(hipStreamSynchronize(s));  // `so` lives on this stack frame (and the caller's coefs may be a temporary)
    }
    int* counter = plan->d_counter;

    // one op; `draw` = running number of the RANDN op (eager), or -1 when the position comes from the device counter (graph)
    int64_t draws = 0;
    auto run_op = [&](hipStream_t st, const CdSamplerOp& o, int index_in_step, bool from_counter) {
      switch (o.kind) {
        case CD_SOP_LINCOMB: {
          const float* src[6];
          for (int j = 0; j < o.nsrc; ++j) src[j] = bufs[o.src[j]];
          launch_lincomb(bufs[o.dst], src, o.nsrc, table, n_coef, o.col, counter, n, st);
          break;
        }
        case CD_SOP_LINDIV: {
          const float* src[6];
          for (int j = 0; j < o.nsrc; ++j) src[j] = bufs[o.src[j]];
          launch_lincomb_div(bufs[o.dst], src, o.nsrc, table, n_coef, o.col, counter, n, st);
          break;
        }
        case CD_SOP_DENOISE:
          launch_fill_from_table(sigma_b, batch, table, n_coef, o.col, counter, st);
          plan->ws.reset(sub, sub_bytes, false);
          forward_impl(plan, batch, bufs[o.src[0]], cond, sigma_b, bufs[o.dst], false, st);
          break;
        case CD_SOP_RANDN:
          if (step_noise) CD_HIP(hipMemcpyAsync(bufs[o.dst], step_noise + (size_t)draws * n, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
          else if (from_counter) launch_randn_step(bufs[o.dst], n, noise_dev, counter, st, randn_per_step, index_in_step);
          else launch_randn(bufs[o.dst], n, seed, offset + (uint64_t)draws * stride, st);
          ++draws;
          break;
        case CD_SOP_RECORD: {
          float* traj = o.dst == 0 ? xs : x0s;
          if (traj) launch_record_step(traj, bufs[o.src[0]], counter, n, st);
          break;
        }
      }
    };
    auto run_step = [&](hipStream_t st, int i, bool from_counter) {
      launch_step_advance(counter, st);
      const int b = uniform ? 0 : op_begin[i], e = uniform ? n_ops : op_begin[i + 1];
      int ri = 0;
      for (int k = b; k < e; ++k) {
        run_op(st, ops[k], ri, from_counter);
        if (ops[k].kind == CD_SOP_RANDN) ++ri;
      }
    };

    run_with_range_fallback(plan, s, [&](bool eager) {
      draws = 0;
      CD_HIP(hipMemsetAsync(counter, 0, sizeof(int), s));
      for (int k = 1; k < n_bufs; ++k) CD_HIP(hipMemsetAsync(bufs[k], 0, sizeof(float) * n, s));
      launch_scale_imm(start, x_out, start_scale, n, s);
      const bool graphable = use_graph && !eager && uniform && !step_noise && !prof::enabled();
      if (graphable) {
        CdPlan::ProgKey key;
        key.batch = batch; key.n_coef = n_coef; key.n_bufs = n_bufs; key.ws = workspace; key.cond = cond; key.x = x_out;
        key.xs = xs; key.x0s = x0s; key.precision = conv_precision();
        uint64_t h = 1469598103934665603ull;  // FNV-1a over the op list
        for (size_t b = 0; b < sizeof(CdSamplerOp) * (size_t)n_ops; ++b) h = (h ^ ((const unsigned char*)ops)[b]) * 1099511628211ull;
        key.ops_hash = h ^ ((uint64_t)n_steps << 40) ^ (uint64_t)n_ops;
        if (!(plan->prog_exec && plan->prog_key == key)) {
          destroy_prog_graph(plan);
          // one eager denoise first (kernel tuning / lazy function attributes cannot happen during capture): x -> buffer 1
          for (int k = 0; k < n_ops; ++k)
            if (ops[k].kind == CD_SOP_DENOISE) {
              launch_step_advance(counter, s);
              launch_fill_from_table(sigma_b, batch, table, n_coef, ops[k].col, counter, s);
              plan->ws.reset(sub, sub_bytes, false);
              forward_impl(plan, batch, x_out, cond, sigma_b, bufs[1], false, s);
              CD_HIP(hipMemsetAsync(counter, 0, sizeof(int), s));
              CD_HIP(hipMemsetAsync(bufs[1], 0, sizeof(float) * n, s));
              break;
            }
          CD_HIP(hipStreamSynchronize(s));
          if (!plan->cap_stream) CD_HIP(hipStreamCreateWithFlags(&plan->cap_stream, hipStreamNonBlocking));
          hipStream_t cs = plan->cap_stream;
          hipGraph_t graph = nullptr;
          CD_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
          try {
            run_step(cs, 0, true);
          } catch (...) {
            hipStreamEndCapture(cs, &graph);
            if (graph) hipGraphDestroy(graph);
            throw;
          }
          CD_HIP(hipStreamEndCapture(cs, &graph));
          hipError_t e = hipGraphInstantiate(&plan->prog_exec, graph, nullptr, nullptr, 0);
          hipGraphDestroy(graph);
          if (e != hipSuccess) {
            plan->prog_exec = nullptr;
            throw Fail{CD_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)};
          }
          plan->prog_key = key;
        }
        for (int i = 0; i < n_steps; ++i) CD_HIP(hipGraphLaunch(plan->prog_exec, s));
      } else {
        for (int i = 0; i < n_steps; ++i) run_step(s, i, false);
      }
    });
  });
}

int cd_plan_status(CdPlan* plan, int* flags, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && flags, "null argument");
    hipStream_t s = (hipStream_t)stream;
    CD_HIP(hipMemcpyAsync(flags, plan->d_counter + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    CD_HIP(hipMemsetAsync(plan->d_counter + 2, 0, sizeof(int), s));
    CD_HIP(hipStreamSynchronize(s));
  });
}

int cd_plan_grad_layout(const CdPlan* plan, int idx, int64_t* offset, int64_t* total_floats) {
  return guarded([&] {
    CD_REQUIRE(plan, "null argument");
    if (total_floats) *total_floats = (int64_t)plan->grad_floats;
    if (offset) {
      CD_REQUIRE(idx >= 0 && idx < (int)plan->weights.size(), "weight index out of range");
      *offset = (int64_t)plan->weights[idx].grad_off;
    }
  });
}

// The weight images every convolution's INPUT gradient reads (the forward kernels run on channel-transposed, tap-flipped weights:
// conv_backward / conv_transpose_backward), laid out once in a plan-owned arena and described by one job list whose sources are
// the plan's own raw copies of the tensors -- stable pointers, so the list never changes; train_step_impl launches it once per
// step.  dg_mode: 1 = 1x1 conv (f32 image, transposed), 2 = 3x3x3 stride 1 (f32 + split16 images, transposed + flipped),
// 3 = strided down conv (f32 + f16x2 images, transposed: its adjoint is the up-conv gather kernel), 4 = up conv (f32 + split16
// images of the tensor read as a plain conv: its adjoint is the strided conv).
static void dgrad_images(CdPlan* p) {
  if (p->dg_arena) return;
  size_t off = 0;
  auto bump = [&](size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return o; };
  std::vector<PackJob> jobs;
  for (auto& w : p->weights) {
    if (w.pack != PK_CONV && w.pack != PK_CONVT && !w.dg_1x1) continue;
    PackJob j{};
    j.kind = 1;
    j.taps = w.taps;
    if (w.pack == PK_CONVT) {
      w.dg_mode = 4;
      j.cout = w.cout; j.cin = w.cin;
    } else {
      w.dg_mode = w.taps == 1 ? 1 : (w.taps == 27 ? 2 : 3);
      j.cout = w.cin; j.cin = w.cout;  // the gradient's convolution maps the conv's output channels back to its input channels
      j.tr = 1;
      j.flip = w.dg_mode == 2 ? 1 : 0;
    }
    if (j.cin % 32 || j.cout % 32) {  // (the init conv's 3 / 4 input channels never need an input gradient)
      w.dg_mode = 0;
      continue;
    }
    w.dg_pk_off = bump(packed_weight_floats(j.cin, j.cout, w.taps));
    const unsigned long long n16 = (unsigned long long)(j.cin / 16) * w.taps * ((j.cout + 31) / 32) * 64;
    if (w.dg_mode == 2 || w.dg_mode == 4) w.dg_pk3_off = bump(packed_split16_bytes(j.cin, j.cout, w.taps) / 4);
    else if (w.dg_mode == 3) w.dg_pk3_off = bump(packed_f16x2_bytes(j.cin, j.cout, w.taps) / 4 + 64);
    j.n_pk = packed_weight_floats(j.cin, j.cout, w.taps);
    if (w.dg_mode == 2 || w.dg_mode == 4) j.n_bf3 = j.n_f16 = n16;
    else if (w.dg_mode == 3) j.n_f16 = n16;
    jobs.push_back(j);
  }
  CD_HIP(hipMalloc((void**)&p->dg_arena, (off + 64) * sizeof(float)));
  CD_HIP(hipMemset(p->dg_arena, 0, (off + 64) * sizeof(float)));
  size_t k = 0;
  for (auto& w : p->weights) {
    if (!w.dg_mode) continue;
    PackJob& j = jobs[k++];
    j.src = p->arena + w.raw_off;
    j.pk = p->dg_arena + w.dg_pk_off;
    if (w.dg_mode == 2 || w.dg_mode == 4) {
      j.bf3 = p->dg_arena + w.dg_pk3_off;
      j.f16 = (char*)(p->dg_arena + w.dg_pk3_off) + packed_bf16x3_bytes(j.cin, j.cout, w.taps);
    } else if (w.dg_mode == 3) {
      j.f16 = p->dg_arena + w.dg_pk3_off;
    }
  }
  p->n_dg_jobs = (int)jobs.size();
  CD_HIP(hipMalloc((void**)&p->d_dg_jobs, sizeof(PackJob) * (jobs.size() + 1)));
  CD_HIP(hipMemcpy(p->d_dg_jobs, jobs.data(), sizeof(PackJob) * jobs.size(), hipMemcpyHostToDevice));
}

int cd_plan_train_workspace_bytes(CdPlan* plan, int batch, size_t* bytes) {
  return guarded([&] {
    CD_REQUIRE(plan && bytes && batch > 0, "bad argument");
    CD_REQUIRE(!plan->desc.time_sin && !plan->desc.cond_sin, "the training step needs the Linear time/cond embeddings");
    dgrad_images(plan);
    plan->ws.reset(nullptr, 0, true);
    train_step_impl(plan, batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    *bytes = plan->ws.high() + 4096;
  });
}

int cd_train_step(CdPlan* plan, int batch, const float* data, const float* noise, const float* sigma, const float* cond,
                  int loss_type, double* loss_out, float* grads, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && data && noise && sigma && cond && loss_out && grads && workspace && batch > 0, "bad argument");
    CD_REQUIRE(loss_type >= CD_LOSS_L2 && loss_type <= CD_LOSS_HUBER, "loss_type must be one of CD_LOSS_L2 / L1 / MSE / HUBER");
    check_ready(plan, true);
    dgrad_images(plan);
    plan->ws.reset((char*)workspace, workspace_bytes, false);
    train_step_impl(plan, batch, data, noise, sigma, cond, loss_out, grads, (hipStream_t)stream, loss_type);
  });
}

int cd_loss_hybrid_l2(CdPlan* plan, int batch, const float* data, const float* noise, const float* sigma,
                      const float* cond, double* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
  return cd_loss_hybrid(plan, batch, data, noise, sigma, cond, CD_LOSS_L2, loss_out, workspace, workspace_bytes, stream);
}

int cd_loss_hybrid(CdPlan* plan, int batch, const float* data, const float* noise, const float* sigma, const float* cond,
                   int loss_type, double* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(plan && data && noise && sigma && cond && loss_out && workspace && batch > 0, "bad argument");
    CD_REQUIRE(loss_type >= CD_LOSS_L2 && loss_type <= CD_LOSS_HUBER, "loss_type must be one of CD_LOSS_L2 / L1 / MSE / HUBER");
    check_ready(plan, true);
    hipStream_t s = (hipStream_t)stream;
    const int64_t per = plan->shapes[0].vox();
    const int64_t n = (int64_t)batch * per;
    plan->ws.reset((char*)workspace, workspace_bytes, false);
    float* xn = plan->ws.get<float>((size_t)n);
    float* x0 = plan->ws.get<float>((size_t)n);
    double* part = plan->ws.get<double>((size_t)batch + 8);
    const size_t used = plan->ws.high();
    launch_axpy_sigma(data, noise, sigma, xn, batch, per, s);
    plan->ws.reset((char*)workspace + used, workspace_bytes > used ? workspace_bytes - used : 0, false);
    forward_impl(plan, batch, xn, cond, sigma, x0, false, s);
    launch_loss_partial(x0, data, noise, sigma, part, batch, per, s, loss_type, plan->desc.objective);
    launch_loss_final(part, sigma, loss_out, batch, per, s, loss_type, plan->desc.objective);
  });
}

// ---- primitives -----------------------------------------------------------------------------------------------
size_t cd_op_scratch_bytes(int batch, int max_channels, int64_t max_voxels) {
  // packed weights of the largest supported conv (256 x 256 x 64 taps) + norm partials + one activation
  return (size_t)256 * 256 * 64 * 4 * 4 + (size_t)batch * 64 * 64 * 16 + (size_t)batch * max_channels * max_voxels * 4 + (1 << 20);
}

int cd_op_to_channels_last(const float* ncdhw, float* ndhwc, int batch, int channels, int64_t voxels, void* stream) {
  return guarded([&] { launch_transpose_to_cl(ncdhw, ndhwc, batch, channels, voxels, (hipStream_t)stream); });
}
int cd_op_to_ncdhw(const float* ndhwc, float* ncdhw, int batch, int channels, int64_t voxels, void* stream) {
  return guarded([&] { launch_transpose_to_planar(ndhwc, ncdhw, batch, channels, voxels, (hipStream_t)stream); });
}

int cd_op_cyl_conv(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias, float* y,
                   int batch, int cout, const int32_t dims_in[3], const int32_t kernel[3], const int32_t stride[3],
                   void* scratch, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x0 && w && y && scratch, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int taps = kernel[0] * kernel[1] * kernel[2];
    float* wpk = (float*)scratch;
    launch_pack_weights(w, wpk, cout, c0 + c1, taps, false, s);
    const Dims3 din{dims_in[0], dims_in[1], dims_in[2]};
    if (taps == 1) {
      PointwiseArgs a;
      a.in0 = x0; a.ld0 = c0; a.c0 = c0; a.in1 = x1; a.ld1 = c1; a.c1 = c1; a.wpk = wpk; a.bias = bias; a.out = y;
      a.batch = batch; a.cout = cout; a.vox = din.vox();
      launch_pointwise(a, s);
    } else {
      CD_REQUIRE(cout % 32 == 0, "cout must be a multiple of 32");
      ConvGeom g;
      g.in = din;
      g.kd = kernel[0]; g.kh = kernel[1]; g.kw = kernel[2]; g.sz = stride[0]; g.sh = stride[1]; g.sw = stride[2];
      g.out = Dims3{(din.d + 2 - g.kd) / g.sz + 1, (din.h + 2 - g.kh) / g.sh + 1, (din.w + 2 - g.kw) / g.sw + 1};
      ConvFusion fu;
      if (taps == 27 || taps == 48) {
        float* w3 = wpk + packed_weight_floats(c0 + c1, cout, taps);
        launch_pack_weights_split16(w, w3, cout, c0 + c1, taps, s);
        fu.wpk_bf16x3 = w3;
      }
      launch_conv_mfma(x0, c0, x1, c1, wpk, bias, y, batch, cout, g, s, fu);
    }
  });
}

int cd_op_cyl_conv_transpose(const float* x, const float* w, const float* bias, float* y, int batch, int channels,
                             const int32_t dims_in[3], int kernel_z, int stride_z, const int32_t out_pad[3],
                             void* scratch, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x && w && y && scratch, "null argument");
    CD_REQUIRE(out_pad[0] == 0, "z output padding is always 0 (models.py:339)");
    hipStream_t s = (hipStream_t)stream;
    float* wpk = (float*)scratch;
    launch_pack_weights(w, wpk, channels, channels, kernel_z * 16, true, s);
    float* wpk16 = wpk + ((packed_weight_floats(channels, channels, kernel_z * 16) + 63) & ~(size_t)63);
    launch_pack_weights_f16x2(w, wpk16, channels, channels, kernel_z * 16, s, true, false);
    const Dims3 din{dims_in[0], dims_in[1], dims_in[2]};
    const Dims3 dout{(din.d - 1) * stride_z - 2 + kernel_z, 2 * din.h + out_pad[1], 2 * din.w + out_pad[2]};
    launch_conv_transpose_mfma(x, channels, wpk, bias, y, batch, channels, din, dout, kernel_z, stride_z, s, wpk16, nullptr);
  });
}

int cd_op_init_conv(const float* x_ncdhw, const float* w, const float* bias, float* y, int batch, int cin, int cout,
                    const int32_t dims[3], void* scratch, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x_ncdhw && w && y && scratch, "null argument");
    hipStream_t s = (hipStream_t)stream;
    float* wpk = (float*)scratch;
    launch_pack_init_weights(w, wpk, cout, cin, s);
    InitConvArgs a;
    a.x = x_ncdhw; a.cx = cin; a.cin = cin; a.wpk = wpk; a.bias = bias; a.out = y; a.batch = batch; a.cout = cout;
    a.dims = Dims3{dims[0], dims[1], dims[2]};
    launch_init_conv(a, s);
  });
}

int cd_op_group_norm(const float* x, float* y, const float* gamma, const float* beta, int batch, int channels,
                     int64_t voxels, int groups, int silu, const float* add_bc, const float* residual,
                     void* scratch, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x && y && gamma && beta && scratch, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int ns = gn_nsplit_for(voxels, batch);
    float* part = (float*)scratch;
    float* coef = part + (size_t)batch * ns * channels * 2;
    launch_ch_stats(x, part, batch, channels, voxels, ns, s);
    launch_gn_finalize(part, ns, gamma, beta, add_bc, channels, coef, batch, channels, groups, voxels, s);
    launch_gn_apply(x, y, coef, batch, channels, voxels, silu, residual, nullptr, 0, nullptr, s);
  });
}

int cd_op_resnet_block(const float* x0, int c0, const float* x1, int c1, const float* const* w, const float* cond, float* y,
                       int batch, int cout, const int32_t dims[3], int groups, void* workspace, size_t workspace_bytes,
                       void* stream) {
  return guarded([&] {
    CD_REQUIRE(x0 && w && y && workspace, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int cin = c0 + c1;
    Arena ws;
    ws.reset((char*)workspace, workspace_bytes, false);
    float* p1 = ws.get<float>(packed_weight_floats(cin, cout, 27));
    float* p2 = ws.get<float>(packed_weight_floats(cout, cout, 27));
    launch_pack_weights(w[0], p1, cout, cin, 27, false, s);
    launch_pack_weights(w[4], p2, cout, cout, 27, false, s);
    float* q1 = ws.get<float>(packed_split16_bytes(cin, cout, 27) / 4);
    float* q2 = ws.get<float>(packed_split16_bytes(cout, cout, 27) / 4);
    launch_pack_weights_split16(w[0], q1, cout, cin, 27, s);
    launch_pack_weights_split16(w[4], q2, cout, cout, 27, s);
    ResP r;
    r.cin = cin; r.cout = cout; r.has_res = w[10] != nullptr;
    r.c1w3 = q1; r.c2w3 = q2;
    r.c1w = p1; r.c1b = w[1]; r.n1g = w[2]; r.n1b = w[3]; r.c2w = p2; r.c2b = w[5]; r.n2g = w[6]; r.n2b = w[7];
    if (r.has_res) {
      float* p3 = ws.get<float>(packed_weight_floats(cin, cout, 1));
      launch_pack_weights(w[10], p3, cout, cin, 1, false, s);
      r.rw = p3; r.rb = w[11];
    }
    if (w[8] && cond) {
      float* emb = ws.get<float>((size_t)batch * cout);
      launch_silu_linear(cond, w[8], w[9], emb, batch, 128, cout, s);
      r.emb = emb; r.emb_ld = cout;
    }
    Run run{&ws, s, batch, groups};
    const Dims3 d{dims[0], dims[1], dims[2]};
    float* out = res_block(run, r, x0, c0, x1, c1, d);
    CD_HIP(hipMemcpyAsync(y, out, sizeof(float) * (size_t)batch * d.vox() * cout, hipMemcpyDeviceToDevice, s));
  });
}

int cd_op_linear_attention(const float* x, const float* const* w, float* y, int batch, int channels, const int32_t dims[3],
                           void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x && w && y && workspace, "null argument");
    hipStream_t s = (hipStream_t)stream;
    Arena ws;
    ws.reset((char*)workspace, workspace_bytes, false);
    float* pq = ws.get<float>(packed_weight_floats(channels, 96, 1));
    launch_pack_weights(w[2], pq, 96, channels, 1, false, s);
    float* pq16 = ws.get<float>(packed_f16x2_bytes(channels, 96, 1) / 4);
    launch_pack_weights_f16x2(w[2], pq16, 96, channels, 1, s);
    AttnP a;
    a.c = channels; a.ng = w[0]; a.nb = w[1]; a.qkv = pq; a.ow = w[3]; a.ob = w[4]; a.gg = w[5]; a.gb = w[6];
    a.qkv16 = pq16;
    Run run{&ws, s, batch, 8};
    const Dims3 d{dims[0], dims[1], dims[2]};
    float* out = attn_block(run, a, x, d);
    CD_HIP(hipMemcpyAsync(y, out, sizeof(float) * (size_t)batch * d.vox() * channels, hipMemcpyDeviceToDevice, s));
  });
}

int cd_op_conv_backward(const float* x0, int c0, const float* x1, int c1, const float* w, const float* dy, float* dx, float* dw,
                        float* db, int batch, int cout, const int32_t dims_in[3], const int32_t kernel[3],
                        const int32_t stride[3], void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x0 && w && dy && dw && workspace, "null argument");
    Arena ws;
    ws.reset((char*)workspace, workspace_bytes, false);
    Run run{&ws, (hipStream_t)stream, batch, 8};
    ConvGeom g;
    g.in = Dims3{dims_in[0], dims_in[1], dims_in[2]};
    g.kd = kernel[0]; g.kh = kernel[1]; g.kw = kernel[2]; g.sz = stride[0]; g.sh = stride[1]; g.sw = stride[2];
    if (g.kd * g.kh * g.kw == 1) g.out = g.in;
    else g.out = Dims3{(g.in.d + 2 - g.kd) / g.sz + 1, (g.in.h + 2 - g.kh) / g.sh + 1, (g.in.w + 2 - g.kw) / g.sw + 1};
    conv_backward(run, x0, c0, x1, c1, w, dy, dx, dw, db, cout, g);
  });
}

int cd_op_conv_transpose_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int batch,
                                  int channels, const int32_t dims_in[3], int kernel_z, int stride_z, const int32_t out_pad[3],
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x && w && dy && dw && workspace, "null argument");
    Arena ws;
    ws.reset((char*)workspace, workspace_bytes, false);
    Run run{&ws, (hipStream_t)stream, batch, 8};
    const Dims3 din{dims_in[0], dims_in[1], dims_in[2]};
    const Dims3 dout{(din.d - 1) * stride_z - 2 + kernel_z, 2 * din.h + out_pad[1], 2 * din.w + out_pad[2]};
    conv_transpose_backward(run, x, w, dy, dx, dw, db, channels, din, dout, kernel_z, stride_z);
  });
}

int cd_op_group_norm_backward(const float* x, const float* gamma, const float* beta, const float* dy, float* dx, float* dgamma,
                              float* dbeta, float* dadd, int batch, int channels, int64_t voxels, int groups, int silu,
                              void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    CD_REQUIRE(x && gamma && beta && dy && dx && dgamma && dbeta && workspace, "null argument");
    hipStream_t s = (hipStream_t)stream;
    Arena ws;
    ws.reset((char*)workspace, workspace_bytes, false);
    Run run{&ws, s, batch, groups};
    int units = 0;
    float* part = stats_pass(run, x, channels, voxels, &units);
    float* coef = ws.get<float>((size_t)batch * channels * 4);
    float* stat = ws.get<float>((size_t)batch * groups * 2);
    launch_gn_finalize(part, units, gamma, beta, nullptr, 0, coef, batch, channels, groups, voxels, s, stat);
    float* scratch = ws.get<float>(gn_backward_scratch_floats(batch, channels, voxels));
    launch_gn_backward(dy, x, coef, stat, gamma, dx, dgamma, dbeta, dadd, channels, batch, channels, voxels, groups, silu, scratch,
                       false, s);
  });
}

}  // extern "C"
