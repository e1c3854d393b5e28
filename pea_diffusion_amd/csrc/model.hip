// UNet op tape (build / forward / reverse data-gradient pass), weight store, PEA adapter and
// the fused KD training step.  Mirrors, op for op, the diffusers-0.23 UNet2DConditionModel
// graph that train_sdxl_zh.py:397,415 executes (restated on CPU in oracle/unet_ref.py), with
// diffusers state-dict key names for every weight.
#include "model.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
int pea_zero_page(const bf16** out);

#define RC(x)                \
  do {                       \
    int rc__ = (x);          \
    if (rc__ != PEA_OK) return rc__; \
  } while (0)

// ============================================================================ graph construction
namespace {
// The UNet's upsampler convs run in their sub-pixel form (Builder::conv, ups == 2) unless PEA_UPCONV_SUBPIXEL=0 (A/B switch:
// the nearest-2x upsample folded into a 3 x 3 gather over the virtual image, 2.25 x the tap products).
static bool upconv_subpixel() {
  static const bool on = !(getenv("PEA_UPCONV_SUBPIXEL") && atoi(getenv("PEA_UPCONV_SUBPIXEL")) == 0);
  return on;
}

struct Builder {
  Tape& u;
  explicit Builder(Tape& un) : u(un) {}
  int T(long long rows, int cols, int B = 0, int H = 0, int W = 0) {
    Tn t;
    t.rows = rows; t.cols = cols; t.B = B; t.H = H; t.W = W;
    u.tn.push_back(t);
    return (int)u.tn.size() - 1;
  }
  int slot(const std::string& name, int kind, int d0, int d1, long long numel) {
    auto it = u.slot_by_name.find(name);
    if (it != u.slot_by_name.end()) return it->second;
    WSlot s;
    s.name = name; s.kind = kind; s.d0 = d0; s.d1 = d1; s.numel = numel;
    u.slots.push_back(s);
    const int id = (int)u.slots.size() - 1;
    u.slot_by_name[name] = id;
    return id;
  }
  int vec(const std::string& n, int d) { return slot(n, W_VEC, d, 0, d); }
  // N, K: stored dims; pad_mode/d/dp describe how they relate to the torch tensor (see WSlot)
  int lin(const std::string& n, int N, int K, int pad_mode = 0, int d = 0, int dp = 0) {
    int Nt = N, Kt = K;
    if (pad_mode == 1) Nt = N / dp * d;
    if (pad_mode == 2) Kt = K / dp * d;
    const int id = slot(n, W_LINEAR, Nt, Kt, (long long)Nt * Kt);
    WSlot& s = u.slots[id];
    s.st_n = N; s.st_k = K; s.pad_mode = pad_mode; s.pad_d = d; s.pad_dp = dp;
    return id;
  }
  int conv3(const std::string& n, int Co, int Ci) { return slot(n, W_CONV3, Co, Ci, 9LL * Co * Ci); }

  Op& push(int kind) {
    Op o;
    o.kind = kind;
    u.ops.push_back(o);
    return u.ops.back();
  }
  int linear(int x, const std::string& pfx, int N, bool bias, int res = -1, int pad_mode = 0, int d = 0, int dp = 0) {
    const int K = u.tn[x].cols;
    const int w = lin(pfx + ".weight", N, K, pad_mode, d, dp);
    const int b = bias ? vec(pfx + ".bias", N) : -1;
    const int out = T(u.tn[x].rows, N, u.tn[x].B, u.tn[x].H, u.tn[x].W);
    Op& o = push(OP_LINEAR);
    o.a = x; o.w = w; o.bias = b; o.out = out; o.res = res;
    return out;
  }
  // several Linear layers over the same input, stacked along N (one GEMM)
  int fused_linear(int x, const std::vector<std::string>& pfx, const std::vector<int>& Ns, bool bias,
                   const std::vector<int>* pad_d = nullptr, const std::vector<int>* pad_dp = nullptr) {
    const int K = u.tn[x].cols;
    FusedMat f;
    f.K = K; f.has_bias = bias;
    u.fused.push_back(f);
    const int fi = (int)u.fused.size() - 1;
    int off = 0;
    for (size_t i = 0; i < pfx.size(); ++i) {
      const bool padded = pad_d && (*pad_d)[i] != (*pad_dp)[i];
      const int w = padded ? lin(pfx[i] + ".weight", Ns[i], K, 1, (*pad_d)[i], (*pad_dp)[i]) : lin(pfx[i] + ".weight", Ns[i], K);
      u.slots[w].fused_parent = fi; u.slots[w].row_off = off;
      if (bias) {
        const int b = vec(pfx[i] + ".bias", Ns[i]);
        u.slots[b].fused_parent = fi; u.slots[b].row_off = off;
      }
      off += Ns[i];
    }
    u.fused[fi].N = off;
    const int out = T(u.tn[x].rows, off, u.tn[x].B, u.tn[x].H, u.tn[x].W);
    Op& o = push(OP_LINEAR);
    o.a = x; o.fused = fi; o.out = out;
    return out;
  }
  int gn(int x, const std::string& pfx, bool silu, float eps) {
    const int C = u.tn[x].cols;
    const int out = T(u.tn[x].rows, C, u.tn[x].B, u.tn[x].H, u.tn[x].W);
    Op& o = push(OP_GN);
    o.a = x; o.out = out; o.w = vec(pfx + ".weight", C); o.bias = vec(pfx + ".bias", C);
    o.p0 = silu; o.f0 = eps; o.aux_bytes = sizeof(float) * 2 * u.B * u.cfg.groups;
    return out;
  }
  int ln(int x, const std::string& pfx, float eps = 1e-5f) {
    const int C = u.tn[x].cols;
    const int out = T(u.tn[x].rows, C, u.tn[x].B, u.tn[x].H, u.tn[x].W);
    Op& o = push(OP_LN);
    o.a = x; o.out = out; o.w = vec(pfx + ".weight", C); o.bias = vec(pfx + ".bias", C);
    o.f0 = eps; o.aux_bytes = sizeof(float) * 2 * u.tn[x].rows;
    return out;
  }
  int rms(int x, const std::string& name, float eps) {        // T5LayerNorm: scale only, no mean subtraction
    const int C = u.tn[x].cols;
    const int out = T(u.tn[x].rows, C, u.tn[x].B, u.tn[x].H, u.tn[x].W);
    Op& o = push(OP_LN);
    o.a = x; o.out = out; o.w = vec(name, C); o.p0 = 1; o.f0 = eps;
    return out;
  }
  int silu(int x) {
    const int out = T(u.tn[x].rows, u.tn[x].cols, u.tn[x].B, u.tn[x].H, u.tn[x].W);
    Op& o = push(OP_SILU);
    o.a = x; o.out = out;
    return out;
  }
  // ups: 1 = nearest-2x upsample folded into the 3 x 3 gather;  2 = the same conv in its sub-pixel form (four 2 x 2 kernels of
  // summed taps, one per output parity: 16 tap products per source pixel instead of 36) -- the output tensor is stored
  // depth-to-space (Tn::d2s), which only concat() and the feature taps may read
  int conv(int x, const std::string& pfx, int Cout, int stride, int ups, int rv = -1, int rv_off = 0, int res = -1) {
    const Tn& t = u.tn[x];
    const int Hv = ups ? t.H * 2 : t.H, Wv = ups ? t.W * 2 : t.W;
    const int Ho = stride == 2 ? (Hv + 1) / 2 : Hv, Wo = stride == 2 ? (Wv + 1) / 2 : Wv;
    const int w = conv3(pfx + ".weight", Cout, t.cols);
    const int b = vec(pfx + ".bias", Cout);
    const int out = T((long long)t.B * Ho * Wo, Cout, t.B, Ho, Wo);
    if (ups == 2) { u.slots[w].subpix = true; u.tn[out].d2s = true; }
    Op& o = push(OP_CONV3);
    o.a = x; o.w = w; o.bias = b; o.out = out; o.p0 = stride; o.p1 = ups; o.rv = rv; o.rv_off = rv_off; o.res = res;
    return out;
  }
  // 3x3 conv whose input tensor is stored wider than the layer's real Cin (zero-padded channels, so the implicit GEMM's
  // K = 9 * stored width stays a multiple of 64) and whose output goes into a tensor `width` >= Cout columns wide
  int conv_padded(int x, const std::string& pfx, int cin_real, int Cout, int width, int stride, int act, int res = -1) {
    const Tn t = u.tn[x];
    const int Ho = stride == 2 ? (t.H + 1) / 2 : t.H, Wo = stride == 2 ? (t.W + 1) / 2 : t.W;
    const int w = conv3(pfx + ".weight", Cout, cin_real);
    u.slots[w].pad_dp = t.cols;
    const int b = vec(pfx + ".bias", Cout);
    const int out = T((long long)t.B * Ho * Wo, width, t.B, Ho, Wo);
    u.tn[out].zero_init = width != Cout;
    Op& o = push(OP_CONV3);
    o.a = x; o.w = w; o.bias = b; o.out = out; o.p0 = stride; o.p3 = act; o.res = res;
    return out;
  }
  // AutoencoderKL pieces (diffusers 0.23 [ext]; call site train_sdxl_zh.py:306-309)
  int resnet_plain(int x, const std::string& pfx, int cout) {          // ResnetBlock2D with temb_channels=None
    const int cin = u.tn[x].cols;
    const float eps = u.cfg.eps;
    int h = gn(x, pfx + ".norm1", true, eps);
    h = conv(h, pfx + ".conv1", cout, 1, 0);
    h = gn(h, pfx + ".norm2", true, eps);
    int sc = x;
    if (cin != cout) sc = linear(x, pfx + ".conv_shortcut", cout, true);
    return conv(h, pfx + ".conv2", cout, 1, 0, -1, 0, sc);
  }
  int attention_mat(int x, const std::string& pfx) {                   // Attention(heads=1, residual_connection=True)
    const Tn t0 = u.tn[x];
    const int C = t0.cols;
    int n = gn(x, pfx + ".group_norm", false, u.cfg.eps);
    int q = linear(n, pfx + ".to_q", C, true);
    int k = linear(n, pfx + ".to_k", C, true);
    int v = linear(n, pfx + ".to_v", C, true);
    const int o = T(t0.rows, C, t0.B, t0.H, t0.W);
    {
      Op& op = push(OP_ATTN_MAT);
      op.a = q; op.b = k; op.c = v; op.out = o; op.f0 = 1.0f / sqrtf((float)C);
    }
    return linear(o, pfx + ".to_out.0", C, true, x);
  }
  int concat(int a, int b) {
    const int out = T(u.tn[a].rows, u.tn[a].cols + u.tn[b].cols, u.tn[a].B, u.tn[a].H, u.tn[a].W);
    Op& o = push(OP_CONCAT);
    o.a = a; o.b = b; o.out = out;
    return out;
  }

  // fold the LayerNorm pushed as op `ln_idx` into the Linear pushed as op `lin_idx` (see LnFold).  Opt-in (PEA_LN_FOLD=1):
  // measured in the SDXL step (B = 4, same box, profiles/r02_ln_fold_ab.txt) the LayerNorm family drops 6.22 -> 5.28 ms but
  // the three consuming GEMMs per block pay 2.8 ms more for the heavier tile transition (s[n] quads + row statistics + one
  // more FMA per element while the matrix pipes wait), so the separate kernel stays the default.
  void fold_ln(int ln_idx, int lin_idx) {
    static const bool off = !(getenv("PEA_LN_FOLD") && atoi(getenv("PEA_LN_FOLD")) == 1);
    if (off || !(u.graph == 0 || u.graph == 2)) return;
    Op& l = u.ops[ln_idx];
    Op& g = u.ops[lin_idx];
    if (l.kind != OP_LN || g.kind != OP_LINEAR || g.a != l.out || g.res >= 0) return;
    LnFold f;
    f.ln_op = ln_idx; f.lin_op = lin_idx; f.gamma = l.w; f.beta = l.bias;
    f.K = u.tn[l.a].cols;
    if (g.fused >= 0) { f.fused = g.fused; f.N = u.fused[g.fused].N; }
    else { f.w_slot = g.w; f.bias_slot = g.bias; f.N = u.slots[g.w].st_n ? u.slots[g.w].st_n : u.slots[g.w].d0; }
    if (f.N % 16 != 0 || f.K % 64 != 0) return;
    u.folds.push_back(f);
    l.fold = g.fold = (int)u.folds.size() - 1;
  }

  int tproj_off = 0, kv_off = 0;
  int resnet(int x, const std::string& pfx, int cout) {
    const int cin = u.tn[x].cols;
    const float eps = u.cfg.eps;
    int h = gn(x, pfx + ".norm1", true, eps);
    h = conv(h, pfx + ".conv1", cout, 1, 0, u.t_tproj, tproj_off);
    tproj_off += cout;
    h = gn(h, pfx + ".norm2", true, eps);
    int sc = x;
    if (cin != cout) sc = linear(x, pfx + ".conv_shortcut", cout, true);
    return conv(h, pfx + ".conv2", cout, 1, 0, -1, 0, sc);
  }
  int transformer(int x, const std::string& pfx, int heads, int depth) {
    const int C = u.tn[x].cols;
    const Tn t0 = u.tn[x];
    const int S = t0.H * t0.W;
    const int d = C / heads, nd = (d + 63) / 64, dp = 64 * nd, Cp = heads * dp;   // heads stored dp wide (zero padded)
    const bool padded = dp != d;
    const std::vector<int> vd3{d, d, d}, vdp3{dp, dp, dp};
    int h = gn(x, pfx + ".norm", false, 1e-6f);
    h = linear(h, pfx + ".proj_in", C, true);
    for (int i = 0; i < depth; ++i) {
      const std::string bp = pfx + ".transformer_blocks." + std::to_string(i);
      int n1 = ln(h, bp + ".norm1");
      const int ln1_idx = (int)u.ops.size() - 1;
      int qkv = fused_linear(n1, {bp + ".attn1.to_q", bp + ".attn1.to_k", bp + ".attn1.to_v"}, {Cp, Cp, Cp}, false, &vd3,
                             &vdp3);
      fold_ln(ln1_idx, (int)u.ops.size() - 1);
      int a1 = T(t0.rows, Cp, t0.B, t0.H, t0.W);
      {
        Op& o = push(OP_ATTN);
        o.a = qkv; o.acol = 0; o.b = qkv; o.bcol = Cp; o.c = qkv; o.ccol = 2 * Cp; o.out = a1;
        o.p0 = heads; o.p1 = S; o.p2 = S; o.p3 = nd; o.f0 = 1.0f / sqrtf((float)d);
        o.aux_bytes = sizeof(float) * t0.B * heads * S;
      }
      h = linear(a1, bp + ".attn1.to_out.0", C, true, h, padded ? 2 : 0, d, dp);
      int n2 = ln(h, bp + ".norm2");
      const int ln2_idx = (int)u.ops.size() - 1;
      int q2 = linear(n2, bp + ".attn2.to_q", Cp, false, -1, padded ? 1 : 0, d, dp);
      fold_ln(ln2_idx, (int)u.ops.size() - 1);
      // K|V of every cross-attention layer come from ONE GEMM over encoder_hidden_states (u.t_kvall)
      const int kv = u.t_kvall, kvo = kv_off;
      kv_off += 2 * Cp;
      int a2 = T(t0.rows, Cp, t0.B, t0.H, t0.W);
      {
        Op& o = push(OP_ATTN);
        o.a = q2; o.acol = 0; o.b = kv; o.bcol = kvo; o.c = kv; o.ccol = kvo + Cp; o.out = a2;
        o.p0 = heads; o.p1 = S; o.p2 = u.L; o.p3 = nd; o.f0 = 1.0f / sqrtf((float)d);
        o.aux_bytes = sizeof(float) * t0.B * heads * S;
      }
      h = linear(a2, bp + ".attn2.to_out.0", C, true, h, padded ? 2 : 0, d, dp);
      int n3 = ln(h, bp + ".norm3");
      const int ln3_idx = (int)u.ops.size() - 1;
      // FF projection with GEGLU fused into the GEMM epilogue: weight rows interleaved (h_i, gate_i) at load time;
      // `g` = h * gelu(gate); the [rows][8C] pre-activation (op.c) is kept only when a backward pass will need it
      int g = T(t0.rows, 4 * C, t0.B, t0.H, t0.W);
      {
        const int w = lin(bp + ".ff.net.0.proj.weight", 8 * C, C, 3, 0, 0);
        const int bsl = vec(bp + ".ff.net.0.proj.bias", 8 * C);
        u.slots[bsl].pad_mode = 3;
        const int hg = u.needs_grad ? T(t0.rows, 8 * C, t0.B, t0.H, t0.W) : -1;
        Op& o = push(OP_LINEAR);
        o.a = n3; o.w = w; o.bias = bsl; o.out = g; o.c = hg; o.p3 = 3;
      }
      fold_ln(ln3_idx, (int)u.ops.size() - 1);
      h = linear(g, bp + ".ff.net.2", C, true, h);
    }
    return linear(h, pfx + ".proj_out", C, true, x);
  }
};

// (prefix, C) of every BasicTransformerBlock in creation order (to build the stacked K|V projection)
struct CrossAttnInfo { std::string pfx; int C, heads; };
std::vector<CrossAttnInfo> enumerate_cross_attn(const PeaUnetCfg& c, bool include_up = true) {
  std::vector<CrossAttnInfo> r;
  const int n = c.n_levels;
  int cur_heads = 0;
  auto add = [&](const std::string& pfx, int C, int depth) {
    for (int k = 0; k < depth; ++k) r.push_back({pfx + ".transformer_blocks." + std::to_string(k) + ".attn2", C, cur_heads});
  };
  for (int i = 0; i < n; ++i)
    if (c.down_cross[i])
      for (int j = 0; j < c.layers_per_block; ++j) {
        cur_heads = c.heads[i];
        add("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), c.block_out[i], c.depth_down[i][j]);
      }
  cur_heads = c.heads[n - 1];
  if (c.depth_mid >= 0) add("mid_block.attentions.0", c.block_out[n - 1], c.depth_mid);
  for (int i = 0; i < n && include_up; ++i)
    if (c.up_cross[i])
      for (int j = 0; j < c.layers_per_block + 1; ++j) {
        cur_heads = c.heads[n - 1 - i];
        add("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), c.block_out[n - 1 - i],
            c.depth_up[i][j]);
      }
  return r;
}

// cout of every ResnetBlock2D in creation order (to size the fused time_emb_proj matrix)
std::vector<std::pair<std::string, int>> enumerate_resnets(const PeaUnetCfg& c, bool include_up = true) {
  std::vector<std::pair<std::string, int>> r;
  const int n = c.n_levels;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < c.layers_per_block; ++j)
      r.push_back({"down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), c.block_out[i]});
  if (c.depth_mid >= 0) {
    r.push_back({"mid_block.resnets.0", c.block_out[n - 1]});
    r.push_back({"mid_block.resnets.1", c.block_out[n - 1]});
  }
  for (int i = 0; i < n && include_up; ++i)
    for (int j = 0; j < c.layers_per_block + 1; ++j)
      r.push_back({"up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), c.block_out[n - 1 - i]});
  return r;
}
}  // namespace

// uniform per-level depths (depth[]) -> the per-position tables every builder reads
void normalize_depths(PeaUnetCfg& c) {
  if (c.per_layer_depth) return;
  const int n = c.n_levels;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      c.depth_down[i][j] = i < n ? c.depth[i] : 0;
      c.depth_up[i][j] = i < n ? c.depth[n - 1 - i] : 0;
    }
  c.depth_mid = n >= 1 ? c.depth[n - 1] : 0;
  c.per_layer_depth = 1;
}

int Tape::build_vae_encoder() {
  const PeaUnetCfg& c = cfg;
  SHAPECHK(c.n_levels >= 2 && c.n_levels <= 4, "vae: n_levels=%d", c.n_levels);
  SHAPECHK(!needs_grad && !residual_inputs, "vae encoder: inference graph only");
  SHAPECHK(c.out_channels <= 8 && c.out_channels % 2 == 0, "vae: %d moment channels", c.out_channels);
  const int f = 1 << (c.n_levels - 1);
  SHAPECHK(H % f == 0 && W % f == 0 && ((H / f) * (W / f)) % 64 == 0, "vae: image %dx%d (latent tokens must be a multiple of 64)", H, W);
  for (int i = 0; i < c.n_levels; ++i) SHAPECHK(c.block_out[i] % 64 == 0, "vae: block_out_channels[%d]=%d", i, c.block_out[i]);
  Builder bd(*this);
  int x = bd.T((long long)B * H * W, c.block_out[0], B, H, W);
  {
    Op& o = bd.push(OP_CONV_IN);
    o.out = x;
    o.w = bd.slot("encoder.conv_in.weight", W_CONV_IN, c.block_out[0], c.in_channels, 9LL * c.block_out[0] * c.in_channels);
    o.bias = bd.vec("encoder.conv_in.bias", c.block_out[0]);
  }
  const int n = c.n_levels;
  for (int i = 0; i < n; ++i) {
    const std::string p = "encoder.down_blocks." + std::to_string(i);
    for (int j = 0; j < c.layers_per_block; ++j) x = bd.resnet_plain(x, p + ".resnets." + std::to_string(j), c.block_out[i]);
    if (i != n - 1) {
      x = bd.conv(x, p + ".downsamplers.0.conv", c.block_out[i], 2, 0);
      ops.back().p2 = 1;                       // Downsample2D(padding=0): F.pad (0,1,0,1) then a stride-2 conv
    }
  }
  x = bd.resnet_plain(x, "encoder.mid_block.resnets.0", c.block_out[n - 1]);
  x = bd.attention_mat(x, "encoder.mid_block.attentions.0");
  x = bd.resnet_plain(x, "encoder.mid_block.resnets.1", c.block_out[n - 1]);
  x = bd.gn(x, "encoder.conv_norm_out", true, c.eps);
  t_out_in = x;
  {
    Op& o = bd.push(OP_CONV_OUT);
    o.a = x;
    o.w = bd.slot("encoder.conv_out.weight", W_CONV_OUT, c.out_channels, c.block_out[n - 1], 9LL * c.out_channels * c.block_out[n - 1]);
    o.bias = bd.vec("encoder.conv_out.bias", c.out_channels);
  }
  w_quant = bd.vec("quant_conv.weight", c.out_channels * c.out_channels);   // [C2][C2][1][1]
  b_quant = bd.vec("quant_conv.bias", c.out_channels);
  return PEA_OK;
}

// AutoencoderKL.decode (tests/test_sdxl_zh.py:430: `self.vae.decode(latents / scaling_factor)`): post_quant_conv (1x1,
// applied with the 1/scaling division in a pointwise kernel before the tape) -> conv_in -> mid block (resnet, single-head
// attention, resnet) -> UpDecoderBlock2D x n (layers_per_block + 1 resnets, nearest-2x + conv folded into one implicit
// GEMM) -> GroupNorm + SiLU -> conv_out.  cfg: in_channels = latent channels (4), out_channels = image channels (3),
// block_out = the ENCODER's block_out_channels (the decoder walks them reversed), H x W = LATENT size.
int Tape::build_vae_decoder() {
  const PeaUnetCfg& c = cfg;
  SHAPECHK(c.n_levels >= 2 && c.n_levels <= 4, "vae: n_levels=%d", c.n_levels);
  SHAPECHK(!needs_grad && !residual_inputs, "vae decoder: inference graph only");
  SHAPECHK(c.out_channels <= 8, "vae decoder: %d image channels", c.out_channels);
  SHAPECHK((H * W) % 64 == 0, "vae decoder: latent %dx%d (tokens must be a multiple of 64)", H, W);
  for (int i = 0; i < c.n_levels; ++i) SHAPECHK(c.block_out[i] % 64 == 0, "vae: block_out_channels[%d]=%d", i, c.block_out[i]);
  Builder bd(*this);
  const int n = c.n_levels;
  const int top = c.block_out[n - 1];
  int x = bd.T((long long)B * H * W, top, B, H, W);
  {
    Op& o = bd.push(OP_CONV_IN);
    o.out = x;
    o.w = bd.slot("decoder.conv_in.weight", W_CONV_IN, top, c.in_channels, 9LL * top * c.in_channels);
    o.bias = bd.vec("decoder.conv_in.bias", top);
  }
  x = bd.resnet_plain(x, "decoder.mid_block.resnets.0", top);
  x = bd.attention_mat(x, "decoder.mid_block.attentions.0");
  x = bd.resnet_plain(x, "decoder.mid_block.resnets.1", top);
  for (int i = 0; i < n; ++i) {
    const std::string p = "decoder.up_blocks." + std::to_string(i);
    const int co = c.block_out[n - 1 - i];
    for (int j = 0; j < c.layers_per_block + 1; ++j) x = bd.resnet_plain(x, p + ".resnets." + std::to_string(j), co);
    if (i != n - 1) x = bd.conv(x, p + ".upsamplers.0.conv", co, 1, 1);
  }
  x = bd.gn(x, "decoder.conv_norm_out", true, c.eps);
  t_out_in = x;
  {
    Op& o = bd.push(OP_CONV_OUT);
    o.a = x;
    o.w = bd.slot("decoder.conv_out.weight", W_CONV_OUT, c.out_channels, c.block_out[0], 9LL * c.out_channels * c.block_out[0]);
    o.bias = bd.vec("decoder.conv_out.bias", c.out_channels);
  }
  w_quant = bd.vec("post_quant_conv.weight", c.in_channels * c.in_channels);
  b_quant = bd.vec("post_quant_conv.bias", c.in_channels);
  return PEA_OK;
}

// Text encoders in front of the step.  flavor 0: CLIPTextModel[WithProjection] (the teacher's two
// encoders, train_sdxl_zh.py:147-150,170-285; HF transformers keys `text_model.*`, `text_projection.weight`): token +
// position embeddings, pre-LN blocks with causal attention, final LayerNorm, pooled = final[EOS] @ text_projection.
// flavor 1: BERT (the Chinese-CLIP text tower, train_sdxl_zh.py:103-107,327-329; keys `embeddings.*`,
// `encoder.layer.N.*`): word + position + token-type embeddings -> LN, post-LN blocks, key-padding mask.
int Tape::build_text() {
  const PeaTextCfg& c = tcfg;
  SHAPECHK(!needs_grad, "text encoder: inference graph only");
  if (c.flavor == 2) return build_text_t5();
  SHAPECHK(c.width % 64 == 0 && c.heads > 0 && c.width / c.heads == 64 && c.width % c.heads == 0,
           "text encoder: width %d / heads %d (head_dim must be 64)", c.width, c.heads);
  SHAPECHK(c.intermediate % 64 == 0 && c.layers >= 1 && L + c.pos_offset <= c.max_pos && c.pos_offset >= 0 && c.proj_dim % 4 == 0,
           "text encoder: dims");
  Builder bd(*this);
  const bool bert = c.flavor == 1;
  const std::string emb = bert ? "embeddings." : "text_model.embeddings.";
  const int W = c.width;
  int x = bd.T((long long)B * L, W, B, 1, L);
  {
    Op& o = bd.push(OP_EMBED);
    o.out = x;
    o.w = bd.lin(emb + (bert ? "word_embeddings.weight" : "token_embedding.weight"), c.vocab, W);
    o.bias = bd.lin(emb + (bert ? "position_embeddings.weight" : "position_embedding.weight"), c.max_pos, W);
    if (bert) o.c = bd.lin(emb + "token_type_embeddings.weight", c.pos_offset ? 1 : 2, W);   // RoBERTa family: one type row
  }
  if (bert) x = bd.ln(x, emb + "LayerNorm", c.eps);
  hidden.push_back(x);
  const float scale = 0.125f;
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = (bert ? "encoder.layer." : "text_model.encoder.layers.") + std::to_string(i);
    const int a_in = bert ? x : bd.ln(x, p + ".layer_norm1", c.eps);
    const int qkv = bert ? bd.fused_linear(a_in, {p + ".attention.self.query", p + ".attention.self.key", p + ".attention.self.value"}, {W, W, W}, true)
                         : bd.fused_linear(a_in, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, {W, W, W}, true);
    const int att = bd.T((long long)B * L, W, B, 1, L);
    {
      Op& o = bd.push(OP_ATTN);
      o.a = qkv; o.acol = 0; o.b = qkv; o.bcol = W; o.c = qkv; o.ccol = 2 * W; o.out = att;
      o.p0 = c.heads; o.p1 = L; o.p2 = L; o.p3 = 1; o.f0 = scale; o.mask = bert ? 2 : 1;
    }
    if (bert) {
      int y = bd.linear(att, p + ".attention.output.dense", W, true, x);
      x = bd.ln(y, p + ".attention.output.LayerNorm", c.eps);
      int f = bd.linear(x, p + ".intermediate.dense", c.intermediate, true);
      ops.back().p2 = c.act;
      y = bd.linear(f, p + ".output.dense", W, true, x);
      x = bd.ln(y, p + ".output.LayerNorm", c.eps);
    } else {
      x = bd.linear(att, p + ".self_attn.out_proj", W, true, x);
      const int n2 = bd.ln(x, p + ".layer_norm2", c.eps);
      int f = bd.linear(n2, p + ".mlp.fc1", c.intermediate, true);
      ops.back().p2 = c.act;
      x = bd.linear(f, p + ".mlp.fc2", W, true, x);
    }
    hidden.push_back(x);
  }
  t_final = x;
  if (!bert) {
    t_final = bd.ln(x, "text_model.final_layer_norm", c.eps);
    const int eos = bd.T(B, W, B);
    { Op& o = bd.push(OP_GATHER_EOS); o.a = t_final; o.out = eos; }
    t_pooled = c.proj_dim ? bd.linear(eos, "text_projection", c.proj_dim, false) : eos;
  }
  return PEA_OK;
}

// flavor 2: the T5 v1.1 encoder stack (the mT5 student option, train_sdxl_zh.py:108-112,331-345; HF transformers
// T5EncoderModel keys `shared.weight`, `encoder.block.N.layer.0.{SelfAttention.{q,k,v,o},layer_norm}`,
// `encoder.block.N.layer.1.{DenseReluDense.{wi_0,wi_1,wo},layer_norm}`, `encoder.final_layer_norm.weight`):
//   x = shared[ids];  per block:  x += o(attn(q, k, v of rms(x)))  with scores = q.k + bias[h][i][j] (no 1/sqrt(d)),
//   x += wo(gelu_new(wi_0 n) * wi_1 n), n = rms(x);  output = rms_final(x).
// The position bias comes from block 0's `relative_attention_bias` table ([buckets][heads]) through T5's bidirectional
// log-spaced buckets and is shared by all blocks; padded keys (ids == pad, right padding) are masked through kv_len.
// The gated FF runs as ONE GEMM over (wi_1_i, wi_0_i) row-interleaved weights with the h * gelu(gate) epilogue.
int Tape::build_text_t5() {
  const PeaTextCfg& c = tcfg;
  const int W = c.width, I = c.heads * 64, F = c.intermediate;
  SHAPECHK(W % 64 == 0 && c.heads > 0 && F % 64 == 0 && c.layers >= 1 && c.rel_buckets >= 2 && c.rel_buckets % 2 == 0 &&
           c.rel_max_dist > c.rel_buckets / 2, "t5 encoder: dims");
  Builder bd(*this);
  int x = bd.T((long long)B * L, W, B, 1, L);
  {
    Op& o = bd.push(OP_EMBED);
    o.out = x;
    o.w = bd.lin("shared.weight", c.vocab, W);
  }
  hidden.push_back(x);
  w_rel = bd.vec("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", c.rel_buckets * c.heads);
  for (int i = 0; i < c.layers; ++i) {
    const std::string p = "encoder.block." + std::to_string(i) + ".layer.";
    const int n1 = bd.rms(x, p + "0.layer_norm.weight", c.eps);
    const int qkv = bd.fused_linear(n1, {p + "0.SelfAttention.q", p + "0.SelfAttention.k", p + "0.SelfAttention.v"}, {I, I, I}, false);
    const int att = bd.T((long long)B * L, I, B, 1, L);
    {
      Op& o = bd.push(OP_ATTN);
      o.a = qkv; o.acol = 0; o.b = qkv; o.bcol = I; o.c = qkv; o.ccol = 2 * I; o.out = att;
      o.p0 = c.heads; o.p1 = L; o.p2 = L; o.p3 = 1; o.f0 = 1.0f; o.mask = 2 | 4;
    }
    x = bd.linear(att, p + "0.SelfAttention.o", W, false, x);
    const int n2 = bd.rms(x, p + "1.layer_norm.weight", c.eps);
    const int g = bd.T((long long)B * L, F, B, 1, L);
    {
      FusedMat f;
      f.K = W; f.N = 2 * F; f.has_bias = false;
      fused.push_back(f);
      const int fi = (int)fused.size() - 1;
      const int wh = bd.lin(p + "1.DenseReluDense.wi_1.weight", F, W);      // linear half -> even rows
      const int wg = bd.lin(p + "1.DenseReluDense.wi_0.weight", F, W);      // gated half  -> odd rows
      slots[wh].fused_parent = fi; slots[wh].row_off = 0; slots[wh].row_step = 2;
      slots[wg].fused_parent = fi; slots[wg].row_off = 1; slots[wg].row_step = 2;
      Op& o = bd.push(OP_LINEAR);
      o.a = n2; o.fused = fi; o.out = g; o.p3 = 3; o.p1 = 1;                // p1: gelu_new (tanh form) in the gate
    }
    x = bd.linear(g, p + "1.DenseReluDense.wo", W, false, x);
    hidden.push_back(x);
  }
  t_final = bd.rms(x, "encoder.final_layer_norm.weight", c.eps);
  return PEA_OK;
}

int Tape::build() {
  if (graph == 4) return build_text();
  if (graph == 1) return build_vae_encoder();
  if (graph == 3) return build_vae_decoder();
  normalize_depths(cfg);
  const PeaUnetCfg& c = cfg;
  SHAPECHK(c.n_levels >= 2 && c.n_levels <= 4, "unet: n_levels=%d", c.n_levels);
  SHAPECHK(c.layers_per_block >= 1 && c.layers_per_block <= 3, "unet: layers_per_block=%d", c.layers_per_block);
  SHAPECHK(c.depth_mid >= 0 || graph == 0, "controlnet: a mid block is required");
  for (int i = 0; i < c.n_levels; ++i) {
    SHAPECHK(c.block_out[i] % 64 == 0, "unet: block_out_channels[%d]=%d must be a multiple of 64", i, c.block_out[i]);
    if (c.down_cross[i] || c.up_cross[c.n_levels - 1 - i] || i == c.n_levels - 1)
      SHAPECHK(c.heads[i] > 0 && c.block_out[i] % c.heads[i] == 0 && c.block_out[i] / c.heads[i] <= 192 &&
                   (c.block_out[i] / c.heads[i]) % 8 == 0,
               "unet: level %d has %d heads over %d channels; head_dim must be a multiple of 8 and <= 192", i,
               c.heads[i], c.block_out[i]);
  }
  SHAPECHK(c.cross_dim % 64 == 0, "unet: cross_attention_dim %% 64");
  SHAPECHK((H % (1 << (c.n_levels - 1))) == 0 && (W % (1 << (c.n_levels - 1))) == 0, "unet: latent %dx%d", H, W);
  Builder bd(*this);
  const int temb_dim = c.block_out[0] * 4;
  // conditioning inputs
  t_ehs = bd.T((long long)B * L, c.cross_dim, B, 1, L);
  tn[t_ehs].rg = needs_grad;
  if (c.text_time) {
    const int pooled = c.proj_in_dim - 6 * c.add_time_dim;
    SHAPECHK(pooled > 0 && pooled % 8 == 0 && c.proj_in_dim % 64 == 0, "unet: projection dims");
    t_text = bd.T(B, pooled, B);
    tn[t_text].rg = needs_grad;
  }
  // time embedding
  int te = bd.T(B, c.block_out[0], B);
  { Op& o = bd.push(OP_TEMB); o.out = te; o.src = 0; o.p0 = c.block_out[0]; }
  int emb = bd.linear(te, "time_embedding.linear_1", temb_dim, true);
  emb = bd.silu(emb);
  emb = bd.linear(emb, "time_embedding.linear_2", temb_dim, true);
  if (c.text_time) {
    int tid = bd.T(B, 6 * c.add_time_dim, B);
    { Op& o = bd.push(OP_TEMB); o.out = tid; o.src = 1; o.p0 = c.add_time_dim; }
    int add = bd.concat(t_text, tid);
    int a = bd.linear(add, "add_embedding.linear_1", temb_dim, true);
    a = bd.silu(a);
    emb = bd.linear(a, "add_embedding.linear_2", temb_dim, true, emb);
  }
  int semb = bd.silu(emb);
  {   // all ResnetBlock2D.time_emb_proj stacked into one GEMM over silu(emb)
    auto rs = enumerate_resnets(c, graph != 2);
    std::vector<std::string> names;
    std::vector<int> ns;
    for (auto& r : rs) { names.push_back(r.first + ".time_emb_proj"); ns.push_back(r.second); }
    t_tproj = bd.fused_linear(semb, names, ns, true);
    tproj_total = tn[t_tproj].cols;
    ops.back().p3 = 1;   // its gradient arrives through the fp32 column-sum scratch
  }
  {   // every attn2.to_k / attn2.to_v stacked into one GEMM over encoder_hidden_states
    auto ca = enumerate_cross_attn(c, graph != 2);
    std::vector<std::string> names;
    std::vector<int> ns, pd, pdp;
    for (auto& r : ca) {
      const int d = r.C / r.heads, dp = (d + 63) / 64 * 64, Cp = r.heads * dp;
      for (const char* nm : {".to_k", ".to_v"}) {
        names.push_back(r.pfx + nm); ns.push_back(Cp); pd.push_back(d); pdp.push_back(dp);
      }
    }
    t_kvall = bd.fused_linear(t_ehs, names, ns, false, &pd, &pdp);
    ops.back().p3 = 2;                      // backward: split-K dgrad (few rows, very deep K)
    kvall_total = tn[t_kvall].cols;
  }
  // conv_in
  int x = bd.T((long long)B * H * W, c.block_out[0], B, H, W);
  {
    Op& o = bd.push(OP_CONV_IN);
    o.out = x;
    o.w = bd.slot("conv_in.weight", W_CONV_IN, c.block_out[0], c.in_channels, 9LL * c.block_out[0] * c.in_channels);
    o.bias = bd.vec("conv_in.bias", c.block_out[0]);
  }
  if (graph == 2) {
    // ControlNetConditioningEmbedding (diffusers 0.23 [ext]; call site tests/test_sdxl_zh_controlnet.py:510-519):
    // conv_in 3->16 + SiLU, then (16->16, 16->32 /2, 32->32, 32->96 /2, 96->96, 96->256 /2) each + SiLU at 8x the latent
    // resolution, then conv_out 256->block_out[0] added to conv_in(sample).  Widths are stored padded to 64 / 128.
    SHAPECHK(!needs_grad && !residual_inputs, "controlnet: inference graph only");
    SHAPECHK(c.n_levels == 3 || c.n_levels == 4, "controlnet: n_levels=%d", c.n_levels);
    const int f = cond_scale_f;
    static const int ch[4] = {16, 32, 96, 256}, wd[4] = {64, 64, 128, 256};
    ce_begin = (int)ops.size();
    int e = bd.T((long long)B * H * f * W * f, wd[0], B, H * f, W * f);
    tn[e].zero_init = true;
    {
      Op& o = bd.push(OP_CONV_IN);
      o.out = e; o.src = 1; o.p0 = ch[0]; o.p3 = 2;
      o.w = bd.slot("controlnet_cond_embedding.conv_in.weight", W_CONV_IN, ch[0], 3, 9LL * ch[0] * 3);
      o.bias = bd.vec("controlnet_cond_embedding.conv_in.bias", ch[0]);
    }
    for (int i = 0; i < 3; ++i) {
      const std::string p = "controlnet_cond_embedding.blocks.";
      e = bd.conv_padded(e, p + std::to_string(2 * i), ch[i], ch[i], wd[i], 1, 2);
      e = bd.conv_padded(e, p + std::to_string(2 * i + 1), ch[i], ch[i + 1], wd[i + 1], 2, 2);
    }
    ce_end = (int)ops.size();
    SHAPECHK(tn[e].H == H && tn[e].W == W, "controlnet: conditioning image must be 8x the latent size");
    x = bd.conv_padded(e, "controlnet_cond_embedding.conv_out", ch[3], c.block_out[0], c.block_out[0], 1, 0, x);
  }
  std::vector<int> skips{x};
  const int n = c.n_levels;
  for (int i = 0; i < n; ++i) {
    const std::string p = "down_blocks." + std::to_string(i);
    for (int j = 0; j < c.layers_per_block; ++j) {
      x = bd.resnet(x, p + ".resnets." + std::to_string(j), c.block_out[i]);
      if (c.down_cross[i]) x = bd.transformer(x, p + ".attentions." + std::to_string(j), c.heads[i], c.depth_down[i][j]);
      skips.push_back(x);
    }
    if (i != n - 1) {
      x = bd.conv(x, p + ".downsamplers.0.conv", c.block_out[i], 2, 0);
      skips.push_back(x);
    }
    taps.push_back(x);
    tap_names.push_back("d" + std::to_string(i));
  }
  auto add_external = [&](int t) {              // t + (externally supplied residual, zero until set)
    const Tn& a = tn[t];
    const int e = bd.T(a.rows, a.cols, a.B, a.H, a.W);
    const int y = bd.T(a.rows, a.cols, a.B, a.H, a.W);
    ext_res.push_back(e);
    Op& o = bd.push(OP_ADD);
    o.a = t; o.b = e; o.out = y;
    return y;
  };
  if (residual_inputs) {
    // diffusers 0.23 [ext]: the ControlNet residuals are added to the skip tensors after the down path ran, i.e.
    // only the copies the up blocks consume change (tests/test_sdxl_zh_controlnet.py:534)
    SHAPECHK(!needs_grad, "unet: residual inputs are an inference feature (no backward through them)");
    for (int& sk : skips) sk = add_external(sk);
  }
  if (c.depth_mid >= 0) {                       // mid_block_type == null (SSD-1B-style pruned UNets): no mid block at all
    x = bd.resnet(x, "mid_block.resnets.0", c.block_out[n - 1]);
    x = bd.transformer(x, "mid_block.attentions.0", c.heads[n - 1], c.depth_mid);
    x = bd.resnet(x, "mid_block.resnets.1", c.block_out[n - 1]);
    taps.push_back(x);
    tap_names.push_back("m");
  }
  if (residual_inputs) x = add_external(x);     // mid_block_additional_residual (:535)
  if (graph == 2) {
    // zero-convs: one 1x1 conv per skip tensor and one for the mid block; their outputs ARE the residuals
    for (size_t k = 0; k < skips.size(); ++k)
      cn_out.push_back(bd.linear(skips[k], "controlnet_down_blocks." + std::to_string(k), tn[skips[k]].cols, true));
    cn_out.push_back(bd.linear(x, "controlnet_mid_block", tn[x].cols, true));
    SHAPECHK(bd.kv_off == kvall_total, "controlnet: stacked K|V projection layout mismatch (%d vs %d)", bd.kv_off, kvall_total);
    return PEA_OK;
  }
  for (int i = 0; i < n; ++i) {
    const std::string p = "up_blocks." + std::to_string(i);
    const int lvl = n - 1 - i;
    for (int j = 0; j < c.layers_per_block + 1; ++j) {
      const int sk = skips.back();
      skips.pop_back();
      x = bd.concat(x, sk);
      x = bd.resnet(x, p + ".resnets." + std::to_string(j), c.block_out[lvl]);
      if (c.up_cross[i]) x = bd.transformer(x, p + ".attentions." + std::to_string(j), c.heads[lvl], c.depth_up[i][j]);
    }
    if (i != n - 1) x = bd.conv(x, p + ".upsamplers.0.conv", c.block_out[lvl], 1, upconv_subpixel() ? 2 : 1);
    taps.push_back(x);
    tap_names.push_back("u" + std::to_string(i));
  }
  SHAPECHK(skips.empty(), "unet: skip stack not consumed (%d left)", (int)skips.size());
  SHAPECHK(bd.kv_off == kvall_total, "unet: stacked K|V projection layout mismatch (%d vs %d)", bd.kv_off, kvall_total);
  x = bd.gn(x, "conv_norm_out", true, c.eps);
  t_out_in = x;
  {
    Op& o = bd.push(OP_CONV_OUT);
    o.a = x;
    o.w = bd.slot("conv_out.weight", W_CONV_OUT, c.out_channels, c.block_out[0], 9LL * c.out_channels * c.block_out[0]);
    o.bias = bd.vec("conv_out.bias", c.out_channels);
  }
  // requires-grad propagation + which weights need a dgrad layout
  for (Op& o : ops) {
    bool rg = false;
    for (int t : {o.a, o.b, o.kind == OP_LINEAR ? -1 : o.c, o.res, o.rv})
      if (t >= 0 && tn[t].rg) rg = true;
    if (o.out >= 0) tn[o.out].rg = tn[o.out].rg || rg;
    if ((o.kind == OP_LINEAR || o.kind == OP_CONV3) && o.a >= 0 && tn[o.a].rg) {
      if (o.fused >= 0) fused[o.fused].need_wt = true;
      else slots[o.w].need_wt = true;
    }
  }
  return PEA_OK;
}

// Every flash-attention op takes its Q from the projection right in front of it (fused Q|K|V: columns [0, C); to_q: the
// whole output) and nothing else reads that block: the projection's epilogue multiplies it by softmax_scale * log2(e)
// (GemmP::qscale: one rounding, from the fp32 accumulator) and the attention kernels run in the log2 domain without a
// per-score multiply (AttnP::q_prescaled).  The attention backward returns the gradient w.r.t. the unscaled q, so the
// projection's data-gradient GEMM is unchanged.
void Tape::tag_q_prescale() {
  static const bool off = getenv("PEA_ATTN_NO_PRESCALE") != nullptr;          // A/B switch
  n_attn = n_attn_pre = 0;
  for (const Op& a : ops)
    if (a.kind == OP_ATTN) { ++n_attn; n_attn_pre += a.pre ? 1 : 0; }
  if (off) return;
  // An attention op that fails a condition below keeps a plain Q: the kernels then round Q * scale * log2(e) to bf16 themselves
  // (attention.hip: scale_frag), one more rounding than the tagged path.  Nothing on the product's graphs may take that path
  // silently: the census (pea_tape_attention_census) is asserted by the tests for every graph, and a miss is logged once here.
  struct Census {
    Tape* t;
    ~Census() {
      t->n_attn_pre = 0;
      for (const Op& a : t->ops) t->n_attn_pre += (a.kind == OP_ATTN && a.pre) ? 1 : 0;
      if (t->n_attn_pre != t->n_attn && !t->plan_only)
        fprintf(stderr, "pea: graph %d: %d of %d attention ops run on a plain (not prescaled) Q\n", t->graph,
                t->n_attn - t->n_attn_pre, t->n_attn);
    }
  } census{this};
  for (size_t i = 0; i < ops.size(); ++i) {
    Op& a = ops[i];
    if (a.kind != OP_ATTN || a.pre || a.acol != 0) continue;
    int prod = -1, readers = 0;
    for (size_t j = 0; j < ops.size(); ++j) {
      const Op& o = ops[j];
      if (o.out == a.a && j < i) prod = (int)j;
      if (j != i && (o.a == a.a || o.res == a.a || o.rv == a.a || (o.kind != OP_LINEAR && o.kind != OP_EMBED && (o.b == a.a || o.c == a.a)))) {   // (OP_EMBED's b / c are weight slots, not tensors)
        // the attention itself may read K / V from the same tensor (fused Q|K|V): other columns, not a reader of the Q block
        ++readers;
      }
    }
    if (prod < 0 || readers) continue;
    Op& l = ops[prod];
    const int qcols = 64 * a.p3 * a.p0;                                       // heads x padded head width
    if (l.kind != OP_LINEAR || l.p2 != 0 || l.p3 == 3 || l.p3 == 2 || l.res >= 0 || l.qs_cols || qcols % 16 || qcols > tn[l.out].cols) continue;
    l.qs_cols = qcols;
    l.qs = a.f0 * 1.4426950408889634f;
    a.pre = 1;
  }
}

int Tape::alloc() {
  tag_q_prescale();
  // a depth-to-space tensor (sub-pixel upsampler output) is only understood by concat's first operand and the feature taps
  for (const Op& o : ops) {
    const int in[5] = {o.a, o.kind == OP_LINEAR || o.kind == OP_EMBED ? -1 : o.b, o.kind == OP_LINEAR || o.kind == OP_EMBED ? -1 : o.c, o.res, o.rv};
    for (int k = 0; k < 5; ++k) {
      if (in[k] < 0 || in[k] >= (int)tn.size() || !tn[in[k]].d2s) continue;
      SHAPECHK(o.kind == OP_CONCAT && k == 0 && o.a != o.b, "tape: op kind %d reads a depth-to-space tensor", o.kind);
    }
  }
  // ---- weights
  size_t off = 0;
  size_t max_numel = 0;
  if (owns_weights) {
    for (FusedMat& f : fused) {
      f.off_w = off; off += al256((size_t)f.N * f.K * 2);
      if (f.need_wt) { f.off_wt = off; off += al256((size_t)f.N * f.K * 2); }
      if (f.has_bias) { f.off_bias = off; off += al256((size_t)f.N * 4); }
    }
    for (WSlot& s : slots) {
      max_numel = std::max(max_numel, (size_t)s.numel);
      if (s.fused_parent >= 0) continue;
      if (s.kind == W_VEC || s.kind == W_CONV_IN || s.kind == W_CONV_OUT) { s.off_f32 = off; off += al256(s.numel * 4); }
      else {
        const size_t st = s.kind == W_LINEAR ? (size_t)s.st_n * s.st_k
                          : (s.kind == W_CONV3 && s.subpix ? (size_t)16 * s.d0 * s.d1
                          : (s.kind == W_CONV3 && s.pad_dp ? (size_t)s.d0 * 9 * s.pad_dp : (size_t)s.numel));
        s.off_w = off; off += al256(st * 2);
        if (s.need_wt) { s.off_wt = off; off += al256(st * 2); }
      }
    }
    for (LnFold& f : folds) {
      f.off_wf = off; off += al256((size_t)f.N * f.K * 2);
      f.off_s = off; off += al256((size_t)f.N * 4);
      f.off_t = off; off += al256((size_t)f.N * 4);
    }
    wbytes = off;
    if (!plan_only) {
    HIPCHK(hipMalloc((void**)&warena, wbytes));
    for (FusedMat& f : fused) {
      f.w = (bf16*)(warena + f.off_w);
      if (f.need_wt) f.wt = (bf16*)(warena + f.off_wt);
      if (f.has_bias) f.bias = (float*)(warena + f.off_bias);
    }
    for (WSlot& s : slots) {
      if (s.fused_parent >= 0) {
        FusedMat& f = fused[s.fused_parent];
        if (s.kind == W_VEC) s.f32 = f.bias + s.row_off;
        else {
          s.w = f.w + (size_t)s.row_off * f.K; s.ldw = f.K * s.row_step;
          if (f.need_wt) { s.wt = f.wt + s.row_off; s.ldwt = f.N; s.need_wt = true; }
        }
        continue;
      }
      if (s.off_f32 != (size_t)-1) s.f32 = (float*)(warena + s.off_f32);
      if (s.off_w != (size_t)-1) {
        s.w = (bf16*)(warena + s.off_w);
        s.ldw = s.kind == W_CONV3 ? (s.subpix ? 4 * s.d1 : 9 * (s.pad_dp ? s.pad_dp : s.d1)) : s.st_k;
      }
      if (s.off_wt != (size_t)-1) {
        s.wt = (bf16*)(warena + s.off_wt);
        s.ldwt = s.kind == W_CONV3 ? (s.subpix ? 16 : 9) * s.d0 : s.st_n;
      }
    }
    for (LnFold& f : folds) {
      f.wf = (bf16*)(warena + f.off_wf); f.s = (float*)(warena + f.off_s); f.t = (float*)(warena + f.off_t);
    }
    tmp_f32_elems = max_numel;
    HIPCHK(hipMalloc((void**)&tmp_f32, tmp_f32_elems * 4));
    }
  }
  // ---- activations (+ per-op aux), gradients
  size_t ao = 0, go = 0;
  for (Tn& t : tn) {
    t.off_d = ao; ao += al256((size_t)t.rows * t.cols * 2);
    // gradients exist only for the differentiated samples (merged passes: the leading bwd_batch of B; every tensor is
    // batch-major and the backward pass works on rb(t) = rows / B * bwd_batch leading rows)
    if (needs_grad && t.rg) { t.off_g = go; go += al256((size_t)(bwd_batch > 0 ? t.rows / B * bwd_batch : t.rows) * t.cols * 2); }
  }
  for (Op& o : ops)
    if (o.aux_bytes) { o.aux_off = ao; ao += al256(o.aux_bytes); }
  abytes = ao; gbytes = go;
  return PEA_OK;
}

// Bytes of every scratch buffer of this graph (host only; order: GroupNorm partials, per-sample column sums, attention row
// constants, upsample-conv gradient, FF d(pre-activation), attention dK/dV split partials, stacked K|V split-K partials,
// fp32 time_emb_proj gradient).  Also sets kv_nsplit.
void Tape::scratch_needs(size_t need[8]) {
  size_t delta_elems = 0, ups_elems = 0, part_bytes = 0, geglu_elems = 0;
  for (Op& o : ops) {
    if (o.kind == OP_ATTN) {
      delta_elems = std::max(delta_elems, (size_t)B * o.p0 * o.p1);
      part_bytes = std::max(part_bytes, attention_bwd_scratch_bytes(B, o.p0, o.p1, o.p2, o.p3));
      // the backward runs on the leading bwd_batch samples, and a SMALLER batch can choose MORE dK / dV splits (fewer heads per
      // round): size for that launch too (a 12-sample merged pass differentiating 8 needs 8 splits x 160 heads = 50 MB where the
      // 12-sample count gives 4 x 240 = 38 MB -- a memory fault in round 4's dead-row contexts at batch 8)
      if (bwd_batch > 0) part_bytes = std::max(part_bytes, attention_bwd_scratch_bytes(bwd_batch, o.p0, o.p1, o.p2, o.p3));
    }
    if (o.kind == OP_LINEAR && o.p3 == 3 && o.c >= 0) geglu_elems = std::max(geglu_elems, (size_t)tn[o.c].rows * tn[o.c].cols);
    if (o.kind == OP_CONV3 && o.p1 == 1 && tn[o.a].rg)
      ups_elems = std::max(ups_elems, (size_t)tn[o.out].rows * tn[o.a].cols);
  }
  size_t gn_bytes = 256, cs_bytes = 256;
  for (Op& o : ops) {
    if (o.kind == OP_GN) gn_bytes = std::max(gn_bytes, groupnorm_scratch_bytes(tn[o.a].B, tn[o.a].H * tn[o.a].W, tn[o.a].cols, cfg.groups));
    if (o.kind == OP_CONV3 && o.rv >= 0) cs_bytes = std::max(cs_bytes, colsum_batched_scratch_bytes(tn[o.out].B, tn[o.out].H * tn[o.out].W, tn[o.out].cols));
  }
  for (int i = 0; i < 8; ++i) need[i] = 0;
  need[0] = gn_bytes;
  if (!needs_grad) return;
  need[1] = cs_bytes;
  need[2] = delta_elems * 4 * 2;               // two row constants per (b, h, q): -delta, -lse*log2e
  need[3] = ups_elems * 2;
  need[4] = geglu_elems * 2;
  need[5] = 2 * part_bytes;                    // two halves: a launch writes one while the previous layer's half is being reduced
  part_half_elems = part_bytes / sizeof(float);
  if (t_ehs >= 0 && kvall_total > 0) {
    const int ksteps = kvall_total / 64;
    kv_nsplit = std::max(1, std::min(32, ksteps / 128));
    need[6] = sizeof(float) * kv_nsplit * (size_t)tn[t_ehs].rows * tn[t_ehs].cols;
  }
  need[7] = sizeof(float) * (size_t)B * tproj_total;
}
size_t Tape::scratch_own_bytes() const {
  const size_t sz[8] = {sc_gn, sc_cs, sc_delta, sc_ups, sc_geglu, sc_part, sc_kv, sc_tproj};
  size_t t = 0;
  for (int i = 0; i < 8; ++i)
    if (!(scratch_borrowed & (1u << i))) t += sz[i];
  return t;
}

// Activation / gradient arenas and scratch are allocated on first use, not at creation: a trainer that runs merged
// passes never touches the activations of the student and teacher contexts it was given (they only carry the weights),
// which is half of the resident HBM at the benchmark size.
int Tape::ensure_acts() {
  if (aarena) return PEA_OK;
  if (arena_donor) {
    RC(arena_donor->ensure_acts());
    SHAPECHK(abytes <= arena_donor->abytes && gbytes <= arena_donor->gbytes, "tape: borrowed arenas are too small");
    aarena = arena_donor->aarena;
    garena = gbytes ? arena_donor->garena : nullptr;
    arena_borrowed = true;
  } else {
    HIPCHK(hipMalloc((void**)&aarena, abytes));
    if (gbytes) HIPCHK(hipMalloc((void**)&garena, gbytes));
  }
  for (Tn& t : tn) {
    t.d = (bf16*)(aarena + t.off_d);
    if (needs_grad && t.rg) t.g = (bf16*)(garena + t.off_g);
  }
  for (Op& o : ops)
    if (o.aux_bytes) o.aux = (float*)(aarena + o.aux_off);
  for (int e : ext_res) HIPCHK(hipMemset(tn[e].d, 0, (size_t)tn[e].rows * tn[e].cols * 2));   // "no residual" = zeros
  for (Tn& t : tn)
    if (t.zero_init) HIPCHK(hipMemset(t.d, 0, (size_t)t.rows * t.cols * 2));
  // ---- scratch
  for (Op& o : ops)
    if (o.kind == OP_ATTN_MAT && !am_scores) {
      const size_t HW = (size_t)tn[o.a].H * tn[o.a].W;
      HIPCHK(hipMalloc((void**)&am_scores, HW * HW * 2));
      HIPCHK(hipMalloc((void**)&am_vt, HW * tn[o.a].cols * 2));
    }
  size_t need[8];
  scratch_needs(need);
  // A context that borrows its arenas (dead-row contexts of a trainer: one per live-row count, never live together with the
  // donor) borrows the donor's scratch buffers too wherever they are large enough, instead of holding ~0.5 GB of its own each
  // (the FF scratch alone is 0.5 GB at 12 samples, 1024 x 1024): only what the donor cannot cover is allocated here.
  auto want = [&](void** ptr, void* const* donor_ptr, size_t* have, const size_t* donor_have, size_t need, int bit) -> int {
    *have = need;
    if (!need) return PEA_OK;
    if (arena_donor && donor_ptr && *donor_ptr && *donor_have >= need) { *ptr = *donor_ptr; scratch_borrowed |= 1u << bit; return PEA_OK; }
    HIPCHK(hipMalloc(ptr, need));
    return PEA_OK;
  };
  Tape* dn = arena_donor;
#define WANT(field, szf, need, bit) RC(want((void**)&field, dn ? (void* const*)&dn->field : nullptr, &szf, dn ? &dn->szf : nullptr, (need), bit))
  WANT(gn_scratch, sc_gn, need[0], 0);
  WANT(cs_scratch, sc_cs, need[1], 1);
  WANT(delta, sc_delta, need[2], 2);
  WANT(ups_tmp, sc_ups, need[3], 3);
  WANT(geglu_tmp, sc_geglu, need[4], 4);
  WANT(attn_part, sc_part, need[5], 5);
  WANT(kv_part, sc_kv, need[6], 6);
  WANT(tproj_grad, sc_tproj, need[7], 7);
#undef WANT
  if (graph == 1) {
    const Tn& t = tn[t_out_in];
    HIPCHK(hipMalloc((void**)&vae_h, sizeof(float) * (size_t)B * cfg.out_channels * t.H * t.W));
  }
  if (graph == 4) HIPCHK(hipMalloc((void**)&kvlen, sizeof(int) * B));
  if (graph == 4 && tcfg.flavor == 2) {
    const size_t n = (size_t)tcfg.heads * L * (((L + 63) >> 6) << 6);
    HIPCHK(hipMalloc((void**)&rel_bias, n * sizeof(float)));
    HIPCHK(hipMemset(rel_bias, 0, n * sizeof(float)));
    // |key - query| -> sub-bucket (T5Attention._relative_position_bucket): exact below nb/2, then log-spaced up to
    // max_distance; fp32 arithmetic in the order the reference evaluates it
    const int nb = tcfg.rel_buckets / 2, max_exact = nb / 2;
    std::vector<int> tab(L);
    for (int d = 0; d < L; ++d) {
      if (d < max_exact) { tab[d] = d; continue; }
      const float lg = logf((float)d / (float)max_exact) / (float)log((double)tcfg.rel_max_dist / (double)max_exact) * (float)(nb - max_exact);
      tab[d] = std::min(max_exact + (int)lg, nb - 1);
    }
    HIPCHK(hipMalloc((void**)&rel_bucket, sizeof(int) * L));
    HIPCHK(hipMemcpy(rel_bucket, tab.data(), sizeof(int) * L, hipMemcpyHostToDevice));
  }
  if (graph == 3) HIPCHK(hipMalloc((void**)&vae_h, sizeof(float) * (size_t)B * cfg.in_channels * H * W));   // post_quant_conv(z / s)
  RC(pea_zero_page(&zeros));
  return PEA_OK;
}

// Give the activation / gradient arenas and the scratch buffers back (weights stay).  The next forward allocates them
// again (ensure_acts).  For trainers that hold one context per aspect-ratio bucket (utils/custom_dataset_sdxl.py:30) and
// keep only the recently used ones resident.
int Tape::release_acts() {
  HIPCHK(hipDeviceSynchronize());
  if (arena_borrowed) { aarena = nullptr; garena = nullptr; arena_borrowed = false; }    // the donor frees them
  drop_borrowed_scratch();
  void** bufs[] = {(void**)&aarena, (void**)&garena, (void**)&gn_scratch, (void**)&delta, (void**)&ups_tmp, (void**)&tproj_grad,
                   (void**)&cs_scratch, (void**)&attn_part, (void**)&kv_part, (void**)&geglu_tmp, (void**)&am_scores,
                   (void**)&am_vt, (void**)&vae_h, (void**)&kvlen, (void**)&rel_bias, (void**)&rel_bucket};
  for (void** b : bufs)
    if (*b) { HIPCHK(hipFree(*b)); *b = nullptr; }
  for (Tn& t : tn) { t.d = nullptr; t.g = nullptr; }
  for (Op& o : ops) o.aux = nullptr;
  ce_valid = false;
  return PEA_OK;
}

// scratch pointers taken from the arena donor (ensure_acts) are the donor's to free
void Tape::drop_borrowed_scratch() {
  void** sc[] = {(void**)&gn_scratch, (void**)&cs_scratch, (void**)&delta, (void**)&ups_tmp, (void**)&geglu_tmp, (void**)&attn_part,
                 (void**)&kv_part, (void**)&tproj_grad};
  for (int i = 0; i < 8; ++i)
    if (scratch_borrowed & (1u << i)) *sc[i] = nullptr;
  scratch_borrowed = 0;
}

Tape::~Tape() {
  drop_borrowed_scratch();
  if (owns_weights && warena) (void)hipFree(warena);
  if (tmp_f32) (void)hipFree(tmp_f32);
  if (aarena && !arena_borrowed) (void)hipFree(aarena);
  if (garena && !arena_borrowed) (void)hipFree(garena);
  if (gn_scratch) (void)hipFree(gn_scratch);
  if (delta) (void)hipFree(delta);
  if (ups_tmp) (void)hipFree(ups_tmp);
  if (tproj_grad) (void)hipFree(tproj_grad);
  if (cs_scratch) (void)hipFree(cs_scratch);
  if (attn_part) (void)hipFree(attn_part);
  if (kv_part) (void)hipFree(kv_part);
  if (geglu_tmp) (void)hipFree(geglu_tmp);
  if (am_scores) (void)hipFree(am_scores);
  if (am_vt) (void)hipFree(am_vt);
  if (vae_h) (void)hipFree(vae_h);
  if (kvlen) (void)hipFree(kvlen);
  if (rel_bias) (void)hipFree(rel_bias);
  if (rel_bucket) (void)hipFree(rel_bucket);
  if (cross_kvlen) (void)hipFree(cross_kvlen);
}

int Tape::load_weight(const char* name, const float* src, long long numel, hipStream_t s) {
  auto it = slot_by_name.find(name);
  if (it == slot_by_name.end()) {
    pea_set_error("unet: unknown weight '%s'", name);
    return PEA_E_NOTFOUND;
  }
  WSlot& w = slots[it->second];
  SHAPECHK(numel == w.numel, "unet: weight '%s' has %lld elements, expected %lld", name, numel, w.numel);
  switch (w.kind) {
    case W_VEC:
      if (w.pad_mode == 3) { RC(launch_permute_geglu_vec(src, w.f32, (int)(numel / 2), s)); break; }
      HIPCHK(hipMemcpyAsync(w.f32, src, numel * 4, hipMemcpyDeviceToDevice, s));
      break;
    case W_CONV_IN:
      HIPCHK(hipMemcpyAsync(w.f32, src, numel * 4, hipMemcpyDeviceToDevice, s));
      break;
    case W_CONV_OUT:
      RC(launch_pack_conv_out(src, w.f32, w.d0, w.d1, s));
      break;
    case W_LINEAR:
      if (w.pad_mode) {
        RC(launch_pad_gather(src, w.d0, w.d1, w.pad_mode, w.pad_d, w.pad_dp, w.w, w.ldw, w.wt, w.ldwt, w.st_n, w.st_k, s));
      } else if (w.row_step > 1) {      // row-interleaved member of a fused matrix (T5 gated FF)
        RC(launch_pad_gather(src, w.d0, w.d1, 0, 0, 0, w.w, w.ldw, nullptr, 0, w.d0, w.d1, s));
      } else {
        RC(launch_cast_f32_bf16(src, w.w, numel, s));
        if (w.wt) RC(launch_transpose_f32_bf16(src, w.wt, w.d0, w.d1, w.ldwt, s));
      }
      break;
    case W_CONV3:
      if (w.subpix) {
        RC(launch_pack_conv_subpix(src, w.w, w.d0, w.d1, 0, s));
        if (w.wt) RC(launch_pack_conv_subpix(src, w.wt, w.d0, w.d1, 1, s));
        break;
      }
      RC(launch_pack_conv_fwd(src, w.w, w.d0, w.d1, s, w.pad_dp));
      if (w.wt) RC(launch_pack_conv_dgrad(src, w.wt, w.d0, w.d1, s));
      break;
  }
  w.loaded = true;
  fold_dirty = true;
  return PEA_OK;
}

// (re)compute W' / s / t of every folded LayerNorm from the current weights; blocks until they are in place, because
// contexts that share these weights may read them from other streams
int Tape::ensure_folded(hipStream_t s) {
  if (weights_owner) return weights_owner->ensure_folded(s);
  if (!fold_dirty || folds.empty()) { fold_dirty = false; return PEA_OK; }
  for (LnFold& f : folds) {
    const bf16* W; int ldw; const float* bias = nullptr;
    if (f.fused >= 0) { W = fused[f.fused].w; ldw = fused[f.fused].K; if (fused[f.fused].has_bias) bias = fused[f.fused].bias; }
    else { W = slots[f.w_slot].w; ldw = slots[f.w_slot].ldw; if (f.bias_slot >= 0) bias = slots[f.bias_slot].f32; }
    RC(launch_ln_fold(W, ldw, slots[f.gamma].f32, slots[f.beta].f32, bias, f.wf, f.s, f.t, f.N, f.K, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  fold_dirty = false;
  return PEA_OK;
}

static bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int Tape::init_random(unsigned long long seed, hipStream_t s) {
  SHAPECHK(owns_weights, "unet: init_random on a context that shares weights");
  unsigned long long i = 0;
  for (WSlot& w : slots) {
    ++i;
    float scale = 0.02f, offset = 0.f;
    if (w.kind == W_LINEAR) scale = 1.0f / sqrtf((float)w.d1);
    else if (w.kind == W_CONV3 || w.kind == W_CONV_IN || w.kind == W_CONV_OUT) scale = 1.0f / sqrtf(9.0f * w.d1);
    else if (ends_with(w.name, ".weight")) { scale = 0.f; offset = 1.f; }    // norm gammas
    RC(launch_fill_random_f32(tmp_f32, w.numel, seed * 1000003ull + i, scale, offset, s));
    RC(load_weight(w.name.c_str(), tmp_f32, w.numel, s));
  }
  return PEA_OK;
}

int Tape::share_weights_from(const Tape& src) {
  SHAPECHK(!owns_weights, "unet: share_weights_from needs a context created without its own weights");
  SHAPECHK(src.slots.size() == slots.size() && src.fused.size() == fused.size(), "unet: configs differ");
  for (size_t i = 0; i < slots.size(); ++i) {
    const WSlot& a = src.slots[i];
    WSlot& b = slots[i];
    SHAPECHK(a.name == b.name && a.numel == b.numel, "unet: weight tables differ at %s", a.name.c_str());
    SHAPECHK(!b.need_wt || a.wt, "unet: source lacks the dgrad layout of %s", a.name.c_str());
    // the packed layout of a conv slot is decided when its tape is built (sub-pixel form of the upsampler convs: [4][Co][4 Ci]
    // against [Co][9 Ci]; padded head / channel widths): a borrower built with the other form would read the other layout
    SHAPECHK(a.subpix == b.subpix && a.pad_dp == b.pad_dp && a.pad_d == b.pad_d && a.pad_mode == b.pad_mode,
             "unet: packed weight layout of %s differs between the two contexts (sub-pixel %d/%d, padding %d/%d)", a.name.c_str(),
             (int)a.subpix, (int)b.subpix, a.pad_dp, b.pad_dp);
    b.f32 = a.f32; b.w = a.w; b.ldw = a.ldw; b.wt = a.wt; b.ldwt = a.ldwt; b.loaded = a.loaded;
  }
  for (size_t i = 0; i < fused.size(); ++i) {
    fused[i].w = src.fused[i].w; fused[i].wt = src.fused[i].wt; fused[i].bias = src.fused[i].bias;
  }
  SHAPECHK(src.folds.size() == folds.size(), "unet: LayerNorm folds differ");
  for (size_t i = 0; i < folds.size(); ++i) { folds[i].wf = src.folds[i].wf; folds[i].s = src.folds[i].s; folds[i].t = src.folds[i].t; }
  weights_owner = src.weights_owner ? src.weights_owner : const_cast<Tape*>(&src);
  wseq_drop();          // the recorded weight sequences name the OLD tables' matrices: the next pass records again
  return PEA_OK;
}

int Tape::all_loaded(std::string* missing) const {
  for (const WSlot& w : slots)
    if (!w.loaded) {
      if (missing) *missing = w.name;
      return 0;
    }
  return 1;
}

// ============================================================================ forward
static void fill_gemm(GemmP& p) { memset(&p, 0, sizeof(p)); p.alpha = 1.f; p.rows_per_batch = 1; }
// What a fused-GEGLU projection (op.p3 == 3 with a stash tensor op.c) leaves in its stash: 1 = (gelu(gate), h * gelu'(gate)),
// the two factors of the backward (GemmP::stash_grad); 0 = the raw (h, gate) pre-activation -- the tanh form (no backward
// exists for it) and the PEA_GEGLU_UNFUSED experiment, whose separate GEGLU kernel reads the pre-activation.
static bool g_geglu_unfused = getenv("PEA_GEGLU_UNFUSED") != nullptr;
static bool g_geglu_stash_raw = getenv("PEA_GEGLU_STASH_RAW") != nullptr;      // A/B switch: stash (h, gate) as in round 2
static int geglu_stash_form(const Op& o) {
  return (o.p3 == 3 && o.c >= 0 && o.p1 == 0 && !g_geglu_stash_raw && !(g_geglu_unfused && o.fold < 0)) ? 1 : 0;
}

int Tape::forward(const float* x, const float* t, const void* ehs, int ehs_dtype, const void* text, int text_dtype,
                  const float* time_ids, float* eps, hipStream_t s) {
  std::string miss;
  if (!all_loaded(&miss)) {
    pea_set_error("unet: weight '%s' was never loaded", miss.c_str());
    return PEA_E_STATE;
  }
  RC(ensure_acts());
  RC(ensure_folded(s));
  x_in = x; t_in = t; tid_in = time_ids; eps_out = eps;
  if (graph == 0 || graph == 2) {
    Tn& e = tn[t_ehs];
    const long long n = e.rows * e.cols;
    if ((const void*)e.d != ehs) {
      if (ehs_dtype == 0) RC(launch_cast_f32_bf16((const float*)ehs, e.d, n, s));
      else HIPCHK(hipMemcpyAsync(e.d, ehs, n * 2, hipMemcpyDeviceToDevice, s));
    }
    if (t_text >= 0) {
      Tn& q = tn[t_text];
      SHAPECHK(text != nullptr && time_ids != nullptr, "unet: text_embeds/time_ids required (text_time)");
      const long long m = q.rows * q.cols;
      if ((const void*)q.d != text) {
        if (text_dtype == 0) RC(launch_cast_f32_bf16((const float*)text, q.d, m, s));
        else HIPCHK(hipMemcpyAsync(q.d, text, m * 2, hipMemcpyDeviceToDevice, s));
      }
    }
  }
  if (graph == 0) wseq_begin(wseq_fwd);                              // (the UNet of the step / of the denoise loop)
  const int rc_ops = exec_ops(0, ops.size(), true, s);
  if (rc_ops == PEA_OK) wseq_end();
  else wseq_cur = nullptr;
  RC(rc_ops);
  if (graph == 2) ce_valid = true;
  return PEA_OK;
}

int Tape::gemm(GemmP& p, hipStream_t s) {
  if (WSeq* q = wseq_cur) {
    // contiguous weight matrices only (every Linear / conv / fused matrix of the tapes: ldw == K)
    const long long bytes = (p.ldw == p.K && p.ksplit <= 1) ? (long long)p.N * p.K * 2 : 0;
    if (!q->ready) q->w.push_back({(const void*)p.W, bytes});
    else {
      const size_t i = q->pos++;
      if (i >= q->w.size() || q->w[i].first != (const void*)p.W) { q->ready = false; q->w.clear(); wseq_cur = nullptr; }
      else {
        static const int dist = getenv("PEA_GEMM_PF_DIST") ? atoi(getenv("PEA_GEMM_PF_DIST")) : 1;     // experiment: launches ahead
        if (i + dist < q->w.size()) {
          // armed only for a target inside the weights owner's arena: a recorded pointer that no longer is (a weights owner
          // re-created behind a borrower's back) would send the DMA waves' touch loads to unmapped memory -- a GPU page fault
          const Tape* ow = weights_owner ? weights_owner : this;
          const char* t = (const char*)q->w[i + dist].first;
          const long long nb = q->w[i + dist].second;
          if (ow->warena && t >= ow->warena && t + nb <= ow->warena + ow->wbytes) { p.pf_ptr = t; p.pf_bytes = nb; }
          else { q->ready = false; q->w.clear(); wseq_cur = nullptr; }
        }
      }
    }
  }
  return launch_gemm(p, s);
}

// ops [begin, end) of the tape in order; skip_cached: leave out the ControlNet conditioning embedding when it is valid
int Tape::exec_ops(size_t begin, size_t end, bool skip_cached, hipStream_t s) {
  RC(ensure_acts());
  for (size_t oi = begin; oi < end; ++oi) {
    Op& o = ops[oi];
    if (skip_cached && ce_valid && (int)oi >= ce_begin && (int)oi < ce_end) continue;   // cached conditioning embedding
    switch (o.kind) {
      case OP_TEMB: {
        Tn& out = tn[o.out];
        if (o.src == 0) RC(launch_timestep_embed(t_in, out.d, B, o.p0, s));
        else RC(launch_timestep_embed(tid_in, out.d, B * 6, o.p0, s));
        break;
      }
      case OP_LINEAR: {
        Tn &a = tn[o.a], &out = tn[o.out];
        GemmP p; fill_gemm(p);
        p.A = a.d; p.lda = a.cols; p.M = (int)a.rows; p.K = a.cols; p.N = out.cols;
        if (o.fused >= 0) { FusedMat& f = fused[o.fused]; p.W = f.w; p.ldw = f.K; p.bias = f.bias; }
        else { WSlot& w = slots[o.w]; p.W = w.w; p.ldw = w.ldw; p.bias = o.bias >= 0 ? slots[o.bias].f32 : nullptr; }
        p.C = out.d; p.ldc = out.cols; p.act = o.p2;
        p.qscale_cols = o.qs_cols; p.qscale = o.qs;
        if (o.p3 == 3) {                  // fused GEGLU: N = 8C interleaved, y -> out, pre-activation -> op.c (student only)
          p.N = 2 * out.cols; p.geglu_y = out.d; p.ldy = out.cols; p.geglu_tanh = o.p1;
          p.C = o.c >= 0 ? tn[o.c].d : nullptr; p.ldc = 2 * out.cols;
          if (bwd_batch > 0) p.stash_rows = (int)(out.rows / B * bwd_batch);   // only the differentiated samples are stashed
          p.stash_grad = o.stash_form = geglu_stash_form(o);        // the form travels with the stash (Op::stash_form)
        }
        if (o.res >= 0) { p.res = tn[o.res].d; p.ldres = tn[o.res].cols; }
        if (o.fold >= 0) {                // folded LayerNorm: the GEMM reads the un-normalised rows
          const LnFold& f = folds[o.fold];
          p.A = tn[ops[f.ln_op].a].d; p.W = f.wf; p.ldw = f.K; p.bias = f.t;
          p.ln_stats = ops[f.ln_op].aux; p.ln_s = f.s;
        }
        if (o.p3 == 3 && g_geglu_unfused && o.c >= 0 && o.fold < 0) {      // A/B switch for experiments
          p.geglu_y = nullptr; p.stash_rows = 0;
          RC(gemm(p, s));
          RC(launch_geglu_fwd_il(tn[o.c].d, out.d, out.rows, out.cols, s));
          break;
        }
        RC(gemm(p, s));
        break;
      }
      case OP_ATTN_MAT: {              // one head over all H*W tokens: S = Q K^T, row softmax, O = P V per image
        Tn &q = tn[o.a], &k = tn[o.b], &v = tn[o.c], &out = tn[o.out];
        const int HW = q.H * q.W, C = q.cols;
        for (int b = 0; b < B; ++b) {
          const long long off = (long long)b * HW * C;
          GemmP p; fill_gemm(p);
          p.A = q.d + off; p.lda = C; p.M = HW; p.K = C; p.W = k.d + off; p.ldw = C; p.N = HW; p.C = am_scores; p.ldc = HW;
          RC(launch_gemm(p, s));
          RC(launch_softmax_rows(am_scores, HW, HW, HW, o.f0, s));
          RC(launch_transpose_bf16(v.d + off, am_vt, HW, C, HW, s));
          GemmP r; fill_gemm(r);
          r.A = am_scores; r.lda = HW; r.M = HW; r.K = HW; r.W = am_vt; r.ldw = HW; r.N = C; r.C = out.d + off; r.ldc = C;
          RC(launch_gemm(r, s));
        }
        break;
      }
      case OP_EMBED: {
        SHAPECHK(ids_in != nullptr, "text encoder: no input ids");
        Tn& out = tn[o.out];
        RC(launch_embed_tokens(ids_in, slots[o.w].w,
                               o.bias >= 0 ? slots[o.bias].w + (long long)tcfg.pos_offset * out.cols : nullptr,
                               o.c >= 0 ? slots[o.c].w : nullptr, out.d, B, L, out.cols, tcfg.vocab, s));
        break;
      }
      case OP_GATHER_EOS:
        RC(launch_gather_eos(ids_in, tn[o.a].d, tn[o.out].d, B, L, tn[o.a].cols, tcfg.eos_id, s));
        break;
      case OP_ADD:
        RC(launch_add(tn[o.a].d, tn[o.b].d, tn[o.out].d, tn[o.out].rows * tn[o.out].cols, s));
        break;
      case OP_SILU:
        RC(launch_silu_fwd(tn[o.a].d, tn[o.out].d, tn[o.a].rows * tn[o.a].cols, s));
        break;
      case OP_CONCAT:
        RC(launch_concat2(tn[o.a].d, tn[o.a].cols, tn[o.b].d, tn[o.b].cols, tn[o.out].d, tn[o.a].rows, s,
                          tn[o.a].d2s ? tn[o.a].H : 0, tn[o.a].d2s ? tn[o.a].W : 0));
        break;
      case OP_CONV_IN:
        if (o.src == 1) {              // ControlNet conditioning image -> first (channel-padded) embedding tensor
          SHAPECHK(cond_in != nullptr, "controlnet: no conditioning image set");
          RC(launch_conv_in(cond_in, slots[o.w].f32, slots[o.bias].f32, tn[o.out].d, B, 3, tn[o.out].H, tn[o.out].W,
                            o.p0, s, tn[o.out].cols, o.p3 == 2));
          break;
        }
        RC(launch_conv_in(x_in, slots[o.w].f32, slots[o.bias].f32, tn[o.out].d, B, cfg.in_channels, H, W,
                          tn[o.out].cols, s));
        break;
      case OP_CONV3: {
        Tn &a = tn[o.a], &out = tn[o.out];
        GemmP p; fill_gemm(p);
        p.mode = 1; p.A = a.d; p.W = slots[o.w].w; p.ldw = slots[o.w].ldw; p.C = out.d; p.ldc = out.cols;
        p.Hs = a.H; p.Ws = a.W; p.Cin = a.cols; p.Ho = out.H; p.Wo = out.W; p.stride = o.p0; p.shift = o.p1 ? 1 : 0;
        p.M = (int)out.rows; p.N = slots[o.w].d0; p.K = 9 * a.cols; p.bias = slots[o.bias].f32; p.zeros = zeros;
        p.act = o.p3;
        p.rows_per_batch = out.H * out.W; p.pad_off = o.p2;
        if (o.rv >= 0) { p.rowvec = tn[o.rv].d + o.rv_off; p.ldrv = tn[o.rv].cols; }
        if (o.res >= 0) { p.res = tn[o.res].d; p.ldres = tn[o.res].cols; }
        if (o.p1 == 2) {
          // sub-pixel form: one 2 x 2 conv over the SOURCE per output parity, written into that parity's channel block of the
          // depth-to-space output (row stride 4 Cout).  pack_conv_subpix_kernel has the tap sums.
          SHAPECHK(o.rv < 0 && o.res < 0 && !o.p3 && !o.p2 && o.p0 == 1, "unet: sub-pixel upsampler conv with an epilogue operand");
          const int Cout = slots[o.w].d0;
          p.shift = 0; p.kside = 2; p.Ho = a.H; p.Wo = a.W; p.M = (int)a.rows; p.K = 4 * a.cols; p.ldc = 4 * Cout;
          p.rows_per_batch = a.H * a.W;
          for (int pl = 0; pl < 4; ++pl) {
            p.W = slots[o.w].w + (size_t)pl * Cout * 4 * a.cols;
            p.C = out.d + (size_t)pl * Cout;
            p.pad_off = pl >> 1; p.pad_dx = (pl & 1) - (pl >> 1);
            RC(gemm(p, s));
          }
          break;
        }
        RC(gemm(p, s));
        break;
      }
      case OP_GN: {
        Tn& a = tn[o.a];
        RC(launch_groupnorm_fwd(a.d, slots[o.w].f32, slots[o.bias].f32, tn[o.out].d, o.aux, gn_scratch, a.B,
                                a.H * a.W, a.cols, cfg.groups, o.f0, o.p0, s));
        break;
      }
      case OP_LN: {
        Tn& a = tn[o.a];
        if (o.fold >= 0) {                // folded into its consumer: statistics only
          RC(launch_layernorm_stats(a.d, o.aux, (int)a.rows, a.cols, o.f0, s));
          break;
        }
        if (o.p0 == 1) {                  // T5LayerNorm
          RC(launch_rmsnorm_fwd(a.d, slots[o.w].f32, tn[o.out].d, (int)a.rows, a.cols, o.f0, s));
          break;
        }
        RC(launch_layernorm_fwd(a.d, slots[o.w].f32, slots[o.bias].f32, tn[o.out].d, o.aux, (int)a.rows, a.cols, o.f0,
                                s));
        break;
      }
      case OP_ATTN: {
        AttnP p; memset(&p, 0, sizeof(p));
        p.Q = tn[o.a].d + o.acol; p.ldq = tn[o.a].cols; p.K = tn[o.b].d + o.bcol; p.ldk = tn[o.b].cols;
        p.V = tn[o.c].d + o.ccol; p.ldv = tn[o.c].cols; p.O = tn[o.out].d; p.ldo = tn[o.out].cols; p.lse = o.aux;
        p.B = B; p.H = o.p0; p.Sq = o.p1; p.Skv = o.p2; p.scale = o.f0; p.nd = o.p3;
        p.causal = o.mask & 1; p.kv_len = (o.mask & 2) ? kvlen : nullptr;
        p.q_prescaled = o.pre;
        if (o.mask & 4) p.bias = rel_bias;
        if (cross_kvlen && o.b == t_kvall) p.kv_len = cross_kvlen;
        RC(launch_attention_fwd(p, s));
        break;
      }
      case OP_GEGLU:
        RC(launch_geglu_fwd(tn[o.a].d, tn[o.out].d, tn[o.a].rows, tn[o.out].cols, s));
        break;
      case OP_CONV_OUT:
        RC(launch_conv_out(tn[o.a].d, slots[o.w].f32, slots[o.bias].f32, eps_out, B, tn[o.a].cols, tn[o.a].H, tn[o.a].W,
                           cfg.out_channels, s));
        break;
    }
  }
  return PEA_OK;
}

// ============================================================================ backward
// 1 (default): the GEGLU backward runs in the epilogue of the FF output projection's dgrad GEMM; 0: as its own kernel (A/B,
// and the cross-check of tests/test_model_gpu.py).  PEA_GEGLU_BWD_UNFUSED in the environment starts with 0.
static int g_geglu_bwd_fused = getenv("PEA_GEGLU_BWD_UNFUSED") ? 0 : 1;
extern "C" void pea_debug_set_geglu_bwd_fused(int v) { g_geglu_bwd_fused = v; }
void Tape::begin_backward() {
  for (Tn& t : tn) { t.gw = false; t.gpend = nullptr; }
  pend_red_valid = false;                      // (a pass that failed half-way leaves nothing behind)
}

int Tape::backward(const float* deps, hipStream_t s) {
  SHAPECHK(needs_grad, "unet: created without gradient support");
  RC(ensure_acts());
  // bwd_batch < B: the forward ran on B samples (student rows first, teacher rows behind them: merged passes of a
  // trainer whose teacher IS the student checkpoint) and only the first bwd_batch samples are differentiated.
  // Every tensor is batch-major, so the restricted pass is the same tape on the leading rows of every tensor.
  const int Bb = bwd_batch > 0 ? bwd_batch : B;
  auto rb = [&](const Tn& t) -> long long { return t.rows / B * Bb; };
  HIPCHK(hipMemsetAsync(tproj_grad, 0, sizeof(float) * Bb * tproj_total, s));
  struct WSeqScope {                       // record / replay the pass's weight sequence (next-op prefetch); an error path drops it
    Tape* t; bool ok = false;
    ~WSeqScope() { if (ok) t->wseq_end(); else t->wseq_cur = nullptr; }
  } wscope{this};
  if (graph == 0) wseq_begin(wseq_bwd);
  // A residual connection hands its gradient on unchanged.  Instead of copying / adding it into the skip tensor's
  // buffer at once, the skip tensor remembers it as a PENDING alias (Tn::gpend) and the next kernel that writes that
  // tensor's gradient (LayerNorm / GroupNorm backward, dgrad GEMM) takes it as its addend: one launch and one
  // read+write of the tensor less per residual.  Writers without an addend input materialise the alias first.
  auto addend = [](Tn& t) -> const bf16* {       // addend for a kernel about to write t.g (consumes a pending alias)
    if (t.gw) return t.g;
    const bf16* p = t.gpend;
    t.gpend = nullptr;
    return p;
  };
  auto materialize = [&](Tn& t) -> int {         // for writers that can only accumulate in place
    if (!t.gw && t.gpend) {
      RC(launch_accum(t.gpend, t.g, rb(t) * t.cols, 0, s));
      t.gpend = nullptr;
      t.gw = true;
    }
    return PEA_OK;
  };
  auto pass_on = [&](Tn& from, Tn& r) -> int {   // residual: r.g += from.g, deferred when r has no gradient yet
    if (!r.gw && !r.gpend) { r.gpend = from.g; return PEA_OK; }
    RC(materialize(r));
    RC(launch_accum(from.g, r.g, rb(r) * r.cols, 1, s));
    return PEA_OK;
  };
  int dpre_of = -1;          // tensor whose gradient currently lives in geglu_tmp as d(pre-activation) (fused GEGLU backward)
  for (int oi = (int)ops.size() - 1; oi >= 0; --oi) {
    Op& o = ops[oi];
    if (o.kind == OP_CONV_OUT) {
      if (!deps) continue;
      Tn& a = tn[o.a];
      SHAPECHK(!a.gw, "unet: conv_out input gradient already written");
      RC(launch_conv_out_dgrad(deps, slots[o.w].f32, a.g, Bb, a.cols, H, W, cfg.out_channels, s));
      a.gw = true;
      continue;
    }
    if (o.out < 0) continue;
    Tn& out = tn[o.out];
    if (!out.rg) continue;
    if (o.kind == OP_LINEAR && o.p3 == 1) {   // fused time_emb_proj: gradient collected in fp32
      RC(launch_cast_f32_bf16(tproj_grad, out.g, (long long)Bb * tproj_total, s));
      out.gw = true;
    }
    RC(materialize(out));
    if (!out.gw) continue;                     // no consumer produced a gradient for this tensor
    switch (o.kind) {
      case OP_LINEAR: {
        Tn& a = tn[o.a];
        if (a.rg && o.p3 == 2) {          // stacked K|V projection: M = B*L rows, K = sum(2C) -> split-K + ordered reduce
          RC(flush_pending_reduce(s));     // the last cross-attention layer's dK / dV partials -> out.g
          FusedMat& f = fused[o.fused];
          GemmP p; fill_gemm(p);
          p.A = out.g; p.lda = out.cols; p.M = (int)rb(out); p.K = out.cols; p.N = a.cols;
          p.W = f.wt; p.ldw = f.N; p.C = kv_part; p.ldc = a.cols; p.out_f32 = 1;
          p.ksplit = kv_nsplit; p.split_stride = rb(out) * a.cols;
          RC(gemm(p, s));
          RC(materialize(a));
          RC(launch_splitk_reduce(kv_part, kv_nsplit, p.split_stride, a.g, a.cols, (int)rb(a), a.cols, a.gw, s));
          a.gw = true;
          break;
        }
        if (a.rg && o.p3 == 3) {          // fused GEGLU: d(pre-activation) into scratch, then the dgrad GEMM over K = 8C
          Tn& hg = tn[o.c];
          SHAPECHK(o.stash_form >= 0, "unet: GEGLU op %d has no stash from a forward pass", oi);
          if (dpre_of != o.out) RC(launch_geglu_bwd_il(hg.d, out.g, geglu_tmp, rb(out), out.cols, s, o.stash_form));
          dpre_of = -1;
          WSlot& w = slots[o.w];
          GemmP p; fill_gemm(p);
          p.A = geglu_tmp; p.lda = hg.cols; p.M = (int)rb(out); p.K = hg.cols; p.N = a.cols;
          p.W = w.wt; p.ldw = w.ldwt; p.C = a.g; p.ldc = a.cols;
          SHAPECHK(p.W != nullptr, "unet: dgrad weights missing for op %d", oi);
          if (const bf16* ad = addend(a)) { p.res = ad; p.ldres = a.cols; }
          RC(gemm(p, s));
          a.gw = true;
          break;
        }
        if (a.rg) {
          GemmP p; fill_gemm(p);
          p.A = out.g; p.lda = out.cols; p.M = (int)rb(out); p.K = out.cols; p.N = a.cols;
          if (o.fused >= 0) { FusedMat& f = fused[o.fused]; p.W = f.wt; p.ldw = f.N; }
          else { WSlot& w = slots[o.w]; p.W = w.wt; p.ldw = w.ldwt; }
          SHAPECHK(p.W != nullptr, "unet: dgrad weights missing for op %d", oi);
          // The FF output projection right behind a fused-GEGLU projection: its input gradient d y is only ever consumed
          // by the GEGLU backward, so that runs in this GEMM's epilogue and d(pre-activation) lands in the scratch the
          // next (GEGLU) op's dgrad reads -- d y is never written, one launch and a read + write of it less per block.
          const Op* prev = oi > 0 ? &ops[oi - 1] : nullptr;
          if (g_geglu_bwd_fused && prev && prev->kind == OP_LINEAR && prev->p3 == 3 && prev->out == o.a && prev->c >= 0 &&
              tn[prev->a].rg && !a.gw && !a.gpend && geglu_tmp && a.cols % 16 == 0) {
            Tn& hg = tn[prev->c];
            SHAPECHK(prev->stash_form >= 0, "unet: GEGLU op %d has no stash from a forward pass", oi - 1);
            p.gbwd_pre = hg.d; p.ldgp = hg.cols; p.gbwd_form = prev->stash_form;
            p.C = geglu_tmp; p.ldc = hg.cols;
            RC(gemm(p, s));
            a.gw = true;
            dpre_of = o.a;
          } else {
            p.C = a.g; p.ldc = a.cols;
            if (const bf16* ad = addend(a)) { p.res = ad; p.ldres = a.cols; }
            RC(gemm(p, s));
            a.gw = true;
          }
        }
        if (o.res >= 0 && tn[o.res].rg) RC(pass_on(out, tn[o.res]));
        break;
      }
      case OP_SILU: {
        Tn& a = tn[o.a];
        if (a.rg) { RC(materialize(a)); RC(launch_silu_bwd(a.d, out.g, a.g, rb(a) * a.cols, a.gw, s)); a.gw = true; }
        break;
      }
      case OP_CONCAT: {
        Tn &a = tn[o.a], &b = tn[o.b];
        if (o.a == o.b) {                 // cat(x, x): a UNet without a mid block feeds the last down output to the first
          if (a.rg) {                     // up resnet both as hidden state and as skip -> two ordered passes into one buffer
            RC(materialize(a));
            RC(launch_split2(out.g, a.cols, b.cols, a.g, a.gw, nullptr, false, rb(a), s));
            RC(launch_split2(out.g, a.cols, b.cols, nullptr, false, a.g, true, rb(a), s));
            a.gw = true;
          }
          break;
        }
        if (a.rg) RC(materialize(a));
        if (b.rg) RC(materialize(b));
        RC(launch_split2(out.g, a.cols, b.cols, a.rg ? a.g : nullptr, a.gw, b.rg ? b.g : nullptr, b.gw, rb(a), s,
                         a.d2s ? a.H : 0, a.d2s ? a.W : 0));
        if (a.rg) a.gw = true;
        if (b.rg) b.gw = true;
        break;
      }
      case OP_CONV3: {
        Tn& a = tn[o.a];
        if (a.rg) {
          WSlot& w = slots[o.w];
          SHAPECHK(w.wt != nullptr, "unet: conv dgrad weights missing for %s", w.name.c_str());
          GemmP p; fill_gemm(p);
          p.mode = 1; p.A = out.g; p.W = w.wt; p.ldw = w.ldwt; p.Hs = out.H; p.Ws = out.W; p.Cin = out.cols;
          p.N = a.cols; p.K = 9 * out.cols; p.zeros = zeros; p.stride = 1;
          if (o.p1 == 2) {       // sub-pixel form: 16 taps (4 parity blocks x 2 x 2) over the depth-to-space d out, one launch
            p.W = w.wt; p.ldw = w.ldwt; p.Hs = a.H; p.Ws = a.W; p.pix = 4 * out.cols; p.kside = 4; p.K = 16 * out.cols;
            p.pad_off = 1; p.Ho = a.H; p.Wo = a.W; p.M = (int)rb(a); p.C = a.g; p.ldc = a.cols;
            if (const bf16* ad = addend(a)) { p.res = ad; p.ldres = a.cols; }
            RC(gemm(p, s));
          } else if (o.p1) {     // upsample-folded conv: gradient at the upsampled resolution, then 2x2 sum
            p.Ho = out.H; p.Wo = out.W; p.M = (int)rb(out); p.C = ups_tmp; p.ldc = a.cols;
            RC(gemm(p, s));
            RC(materialize(a));
            RC(launch_sumpool2(ups_tmp, a.g, Bb, a.H, a.W, a.cols, a.gw, s));
          } else {
            if (o.p0 == 2) { p.shift = 1; p.parity = 1; }
            p.Ho = a.H; p.Wo = a.W; p.M = (int)rb(a); p.C = a.g; p.ldc = a.cols;
            if (const bf16* ad = addend(a)) { p.res = ad; p.ldres = a.cols; }
            RC(gemm(p, s));
          }
          a.gw = true;
        }
        if (o.rv >= 0 && tn[o.rv].rg)
          RC(launch_colsum_batched(out.g, tproj_grad + o.rv_off, Bb, out.H * out.W, out.cols, tproj_total, cs_scratch, s));
        if (o.res >= 0 && tn[o.res].rg) RC(pass_on(out, tn[o.res]));
        break;
      }
      case OP_GN: {
        Tn& a = tn[o.a];
        if (a.rg) {
          RC(launch_groupnorm_bwd(a.d, out.g, slots[o.w].f32, slots[o.bias].f32, o.aux, a.g, gn_scratch, Bb,
                                  a.H * a.W, a.cols, cfg.groups, o.p0, addend(a), s));
          a.gw = true;
        }
        break;
      }
      case OP_LN: {
        Tn& a = tn[o.a];
        if (a.rg) {
          RC(launch_layernorm_bwd(a.d, out.g, slots[o.w].f32, o.aux, a.g, nullptr, nullptr, (int)rb(a), a.cols,
                                  addend(a), s));
          a.gw = true;
        }
        break;
      }
      case OP_ATTN: {
        Tn &q = tn[o.a], &k = tn[o.b], &v = tn[o.c];
        AttnP p; memset(&p, 0, sizeof(p));
        p.Q = q.d + o.acol; p.ldq = q.cols; p.K = k.d + o.bcol; p.ldk = k.cols; p.V = v.d + o.ccol; p.ldv = v.cols;
        p.O = out.d; p.ldo = out.cols; p.lse = o.aux; p.B = Bb; p.H = o.p0; p.Sq = o.p1; p.Skv = o.p2; p.scale = o.f0;
        p.nd = o.p3; p.q_prescaled = o.pre;
        p.dO = out.g; p.lddo = out.cols; p.delta = delta; p.dkv_part = attn_part;
        if (cross_kvlen && o.b == t_kvall) p.kv_len = cross_kvlen;
        SHAPECHK(!q.gw && !q.gpend && !k.gpend && (!k.gw || o.b == t_kvall), "unet: attention operand gradient written twice");
        if (q.rg) { p.dQ = q.g + o.acol; p.lddq = q.cols; }
        if (k.rg) { p.dK = k.g + o.bcol; p.lddk = k.cols; p.dV = v.g + o.ccol; p.lddv = v.cols; }
        if (o.b == t_kvall && attention_bwd_defers(p)) {
          // cross-attention layer on the specialised-wave kernel: its split reduce rides in the NEXT such launch's prologue
          // (d(K|V) of t_kvall is read by nothing before the stacked K|V dgrad GEMM at the end of the pass)
          p.dkv_part = attn_part + (part_toggle ? part_half_elems : 0);
          p.defer_reduce = 1;
          if (pend_red_valid) attention_set_deferred(p, pend_red);
          RC(launch_attention_bwd(p, s));
          pend_red = p;
          pend_red_valid = true;
          part_toggle ^= 1;
        } else {
          RC(launch_attention_bwd(p, s));
        }
        if (q.rg) q.gw = true;
        if (k.rg) { k.gw = true; v.gw = true; }
        break;
      }
      case OP_GEGLU: {
        Tn& a = tn[o.a];
        if (a.rg) {
          SHAPECHK(!a.gw, "unet: geglu input gradient written twice");
          RC(launch_geglu_bwd(a.d, out.g, a.g, rb(a), out.cols, s));
          a.gw = true;
        }
        break;
      }
      default:
        break;
    }
  }
  RC(flush_pending_reduce(s));                 // (a graph without the stacked projection op)
  for (Tn& t : tn)
    if (t.rg) RC(materialize(t));              // graph inputs that only ever received a passed-on gradient
  wscope.ok = true;
  return PEA_OK;
}

int Tape::flush_pending_reduce(hipStream_t s) {
  if (!pend_red_valid) return PEA_OK;
  pend_red_valid = false;
  return launch_attention_dkv_reduce(pend_red, s);
}

// ============================================================================ adapter
// MLP of train_sdxl_zh.py:43-67 (SD1.5: train_sd_zh.py:41-56): LayerNorm -> 3 x (Linear, no bias)
// with GELU(erf) between -> { GELU -> Linear+bias = tokens ; mean over tokens = pooled }.
// Forward GEMMs use the fused GELU epilogue (pre-activation stashed for the backward);
// backward = dgrad (transposed bf16 copies) + wgrad (fp32 accumulation into the flat grad buffer).
Adapter::~Adapter() {
  if (arena) (void)hipFree(arena);
}

int Adapter::prepare(int B2_, int L_) {
  SHAPECHK(in_dim % 64 == 0 && hidden % 64 == 0 && out_dim % 64 == 0 && (out1 == 0 || out1 % 64 == 0),
           "adapter: dims must be multiples of 64 (in=%d hidden=%d out=%d out1=%d)", in_dim, hidden, out_dim, out1);
  SHAPECHK(!use_residual || in_dim == out_dim, "adapter: use_residual needs in_dim == out_dim");
  B2 = B2_; L = L_; R = B2 * L; Rpad = (R + 63) / 64 * 64;
  off_lnw = 0; off_lnb = in_dim; off_w0 = off_lnb + in_dim; off_w1 = off_w0 + (long long)hidden * in_dim;
  off_w2 = off_w1 + (long long)hidden * hidden; off_fcw = off_w2 + (long long)out_dim * hidden;
  off_fcb = off_fcw + (long long)out1 * out_dim; nparam = off_fcb + out1;
  if (arena) { (void)hipFree(arena); arena = nullptr; }
  size_t off = 0;
  std::vector<std::pair<bf16**, size_t>> req;
  auto want = [&](bf16** p, size_t elems) { req.push_back({p, off}); off += al256(elems * 2); };
  const size_t mx = (size_t)std::max(std::max(in_dim, hidden), std::max(out_dim, std::max(out1, 64)));
  want(&w0, (size_t)hidden * in_dim); want(&w1, (size_t)hidden * hidden); want(&w2, (size_t)out_dim * hidden);
  want(&wfc, (size_t)out1 * out_dim + 64);
  want(&w0t, (size_t)hidden * in_dim); want(&w1t, (size_t)hidden * hidden); want(&w2t, (size_t)out_dim * hidden);
  want(&wfct, (size_t)out1 * out_dim + 64);
  want(&x, (size_t)R * in_dim); want(&xn, (size_t)R * in_dim);
  want(&z0, (size_t)R * hidden); want(&a0, (size_t)R * hidden); want(&z1, (size_t)R * hidden); want(&a1, (size_t)R * hidden);
  want(&z2, (size_t)R * out_dim); want(&a2, (size_t)R * out_dim); want(&tok, (size_t)R * (out1 ? out1 : 64));
  want(&pooled, (size_t)B2 * out_dim);
  want(&dtok, (size_t)R * (out1 ? out1 : 64)); want(&da2, (size_t)R * out_dim); want(&dz2, (size_t)R * out_dim);
  want(&da1, (size_t)R * hidden); want(&dz1, (size_t)R * hidden); want(&da0, (size_t)R * hidden);
  want(&dz0, (size_t)R * hidden); want(&dxn, (size_t)R * in_dim); want(&dpool, (size_t)B2 * out_dim);
  want(&tA, mx * Rpad); want(&tB, mx * Rpad);
  const size_t stats_off = off;
  off += al256((size_t)R * 2 * 4);
  HIPCHK(hipMalloc((void**)&arena, off));
  HIPCHK(hipMemset(arena, 0, off));
  for (auto& r : req) *r.first = (bf16*)(arena + r.second);
  ln_stats = (float*)(arena + stats_off);
  return PEA_OK;
}

int Adapter::sync_weights(hipStream_t s) {
  SHAPECHK(params != nullptr, "adapter: parameter buffer not set");
  RC(launch_cast_f32_bf16(params + off_w0, w0, (long long)hidden * in_dim, s));
  RC(launch_cast_f32_bf16(params + off_w1, w1, (long long)hidden * hidden, s));
  RC(launch_cast_f32_bf16(params + off_w2, w2, (long long)out_dim * hidden, s));
  RC(launch_transpose_f32_bf16(params + off_w0, w0t, hidden, in_dim, hidden, s));
  RC(launch_transpose_f32_bf16(params + off_w1, w1t, hidden, hidden, hidden, s));
  RC(launch_transpose_f32_bf16(params + off_w2, w2t, out_dim, hidden, out_dim, s));
  if (out1) {
    RC(launch_cast_f32_bf16(params + off_fcw, wfc, (long long)out1 * out_dim, s));
    RC(launch_transpose_f32_bf16(params + off_fcw, wfct, out1, out_dim, out1, s));
  }
  return PEA_OK;
}

static int agemm(const bf16* A, int lda, const bf16* W, int ldw, void* C, int ldc, int M, int N, int K, int act,
                 bf16* pre, const float* bias, int out_f32, int accum, hipStream_t s) {
  GemmP p; fill_gemm(p);
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.act = act;
  p.preact = pre; p.ldpre = N; p.bias = bias; p.out_f32 = out_f32; p.accum_f32 = accum;
  return launch_gemm(p, s);
}

int Adapter::forward(const void* enc, const void* enc2, int dtype, hipStream_t s) {
  SHAPECHK(arena && params, "adapter: prepare()/bind() first");
  // enc2 == nullptr: enc holds all R rows; otherwise enc / enc2 hold R/2 rows each (cond | uncond)
  const long long n1 = (enc2 ? (long long)(R / 2) : (long long)R) * in_dim;
  if (dtype == 0) RC(launch_cast_f32_bf16((const float*)enc, x, n1, s));
  else HIPCHK(hipMemcpyAsync(x, enc, (size_t)n1 * 2, hipMemcpyDeviceToDevice, s));
  if (enc2) {
    if (dtype == 0) RC(launch_cast_f32_bf16((const float*)enc2, x + n1, n1, s));
    else HIPCHK(hipMemcpyAsync(x + n1, enc2, (size_t)n1 * 2, hipMemcpyDeviceToDevice, s));
  }
  RC(launch_layernorm_fwd(x, params + off_lnw, params + off_lnb, xn, ln_stats, R, in_dim, 1e-5f, s));
  RC(agemm(xn, in_dim, w0, in_dim, a0, hidden, R, hidden, in_dim, 1, z0, nullptr, 0, 0, s));
  RC(agemm(a0, hidden, w1, hidden, a1, hidden, R, hidden, hidden, 1, z1, nullptr, 0, 0, s));
  if (out1) {
    RC(agemm(a1, hidden, w2, hidden, a2, out_dim, R, out_dim, hidden, 1, z2, nullptr, 0, 0, s));
    RC(agemm(a2, out_dim, wfc, out_dim, tok, out1, R, out1, out_dim, 0, nullptr, params + off_fcb, 0, 0, s));
    if (use_residual) {
      RC(launch_add(z2, x, dz2, (long long)R * out_dim, s));   // dz2 is free during the forward pass
      RC(launch_mean_tokens(dz2, pooled, B2, L, out_dim, s));
    } else {
      RC(launch_mean_tokens(z2, pooled, B2, L, out_dim, s));
    }
  } else {
    RC(agemm(a1, hidden, w2, hidden, z2, out_dim, R, out_dim, hidden, 0, nullptr, nullptr, 0, 0, s));
  }
  return PEA_OK;
}

// dW[N][K] (+)= dZ[R][N]^T . X[R][K]  via transposed operands (contraction padded to Rpad with zeros)
static int wgrad(Adapter& a, const bf16* dZ, int N, const bf16* X, int K, float* dW, int accum, hipStream_t s) {
  RC(launch_transpose_bf16(dZ, a.tA, a.R, N, a.Rpad, s));
  RC(launch_transpose_bf16(X, a.tB, a.R, K, a.Rpad, s));
  return agemm(a.tA, a.Rpad, a.tB, a.Rpad, dW, K, N, K, a.Rpad, 0, nullptr, nullptr, 1, accum, s);
}

int Adapter::backward(float* g, int accumulate, hipStream_t s) {
  // inputs: dtok [R][out1] (SD1.5: dz2 [R][out]) and dpool [B2][out] already filled by the caller
  if (!accumulate) HIPCHK(hipMemsetAsync(g, 0, nparam * 4, s));
  if (out1) {
    RC(agemm(dtok, out1, wfct, out1, da2, out_dim, R, out_dim, out1, 0, nullptr, nullptr, 0, 0, s));
    RC(wgrad(*this, dtok, out1, a2, out_dim, g + off_fcw, 1, s));
    RC(launch_colsum(dtok, g + off_fcb, R, out1, 1, s));
    RC(launch_gelu_bwd(z2, da2, dz2, (long long)R * out_dim, 0, s));
    RC(launch_mean_tokens_bwd(dpool, dz2, B2, L, out_dim, 1, s));
  }
  RC(agemm(dz2, out_dim, w2t, out_dim, da1, hidden, R, hidden, out_dim, 0, nullptr, nullptr, 0, 0, s));
  RC(wgrad(*this, dz2, out_dim, a1, hidden, g + off_w2, 1, s));
  RC(launch_gelu_bwd(z1, da1, dz1, (long long)R * hidden, 0, s));
  RC(agemm(dz1, hidden, w1t, hidden, da0, hidden, R, hidden, hidden, 0, nullptr, nullptr, 0, 0, s));
  RC(wgrad(*this, dz1, hidden, a0, hidden, g + off_w1, 1, s));
  RC(launch_gelu_bwd(z0, da0, dz0, (long long)R * hidden, 0, s));
  RC(agemm(dz0, hidden, w0t, hidden, dxn, in_dim, R, in_dim, hidden, 0, nullptr, nullptr, 0, 0, s));
  RC(wgrad(*this, dz0, hidden, xn, in_dim, g + off_w0, 1, s));
  RC(launch_layernorm_bwd(x, dxn, params + off_lnw, ln_stats, nullptr /*no input gradient needed*/, g + off_lnw, g + off_lnb, R,
                          in_dim, 0, s));
  return PEA_OK;
}

// ============================================================================ trainer
Trainer::~Trainer() {
  if (side) { (void)hipStreamDestroy(side); (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); }
  for (void* p : {(void*)xt, (void*)eps_s, (void*)eps_t, (void*)deps, (void*)ac, (void*)t_ehs_sel, (void*)dehs_full, (void*)t_f32,
                  (void*)losses, (void*)kd_ws, (void*)tehs_c, (void*)tehs_n, (void*)xt2, (void*)eps2, (void*)t2, (void*)tid2})
    if (p) (void)hipFree(p);
  for (auto& kv : merged_n) delete kv.second;           // (their arenas are borrowed from `merged`: freed below)
  if (tmap_d) (void)hipFree(tmap_d);
  if (tpool_c) (void)hipFree(tpool_c);
  delete merged;
}

int Trainer::prepare() {
  if (const char* e = getenv("PEA_TWO_STREAM")) two_stream = atoi(e);
  if (const char* e = getenv("PEA_MERGE_PASSES")) merge_passes = atoi(e);
  Tape& S = *student;
  Tape& Tt = *teacher;
  SHAPECHK(S.needs_grad, "trainer: student context needs gradient support");
  SHAPECHK(S.B == Tt.B && S.H == Tt.H && S.W == Tt.W, "trainer: student/teacher shapes differ");
  // feature taps are paired by hook name (d0.., m, u0..: cast_hook, train_sdxl_zh.py:79-84).  A student without a mid
  // block (SSD-1B-style, mid_block_type null) has no 'm' tap: that term leaves the feature loss (the reference's
  // cast_hook would fail on `unet.mid_block is None`; every other tap must exist on both sides).
  tap_pairs.clear();
  for (size_t k = 0; k < S.taps.size(); ++k) {
    int found = -1;
    for (size_t j = 0; j < Tt.taps.size(); ++j)
      if (Tt.tap_names[j] == S.tap_names[k]) found = (int)j;
    SHAPECHK(found >= 0, "trainer: the teacher has no '%s' tap", S.tap_names[k].c_str());
    SHAPECHK(S.tn[S.taps[k]].rows == Tt.tn[Tt.taps[found]].rows && S.tn[S.taps[k]].cols == Tt.tn[Tt.taps[found]].cols,
             "trainer: tap '%s' shapes differ", S.tap_names[k].c_str());
    tap_pairs.push_back({(int)k, found});
  }
  for (size_t j = 0; j < Tt.taps.size(); ++j) {
    bool used = false;
    for (auto& pr : tap_pairs) used = used || pr.second == (int)j;
    SHAPECHK(used || Tt.tap_names[j] == "m", "trainer: the student has no '%s' tap", Tt.tap_names[j].c_str());
  }
  SHAPECHK(tap_pairs.size() <= 9, "trainer: %d taps", (int)tap_pairs.size());
  SHAPECHK(ad->B2 == 2 * S.B && ad->L == S.L, "trainer: adapter prepared for %d x %d, need %d x %d", ad->B2, ad->L,
           2 * S.B, S.L);
  const int tok_dim = ad->out1 ? ad->out1 : ad->out_dim;
  SHAPECHK(tok_dim == S.cfg.cross_dim, "trainer: adapter token dim %d != student cross_attention_dim %d", tok_dim,
           S.cfg.cross_dim);
  const size_t n = (size_t)S.B * S.cfg.in_channels * S.H * S.W;
  HIPCHK(hipMalloc((void**)&xt, n * 4));
  HIPCHK(hipMalloc((void**)&eps_s, n * 4));
  HIPCHK(hipMalloc((void**)&eps_t, n * 4));
  HIPCHK(hipMalloc((void**)&deps, n * 4));
  HIPCHK(hipMalloc((void**)&losses, 16));
  {
    std::vector<long long> per;
    for (auto& pr : tap_pairs) per.push_back(S.tn[S.taps[pr.first]].rows / S.B * S.tn[S.taps[pr.first]].cols);
    HIPCHK(hipMalloc((void**)&kd_ws, kd_loss_workspace_bytes((int)per.size(), per.data(), (long long)S.cfg.in_channels * S.H * S.W, S.B)));
  }
  HIPCHK(hipMalloc((void**)&tehs_c, (size_t)Tt.B * Tt.L * Tt.cfg.cross_dim * 2));
  HIPCHK(hipMalloc((void**)&tehs_n, (size_t)Tt.B * Tt.L * Tt.cfg.cross_dim * 2));
  // DDPM alphas_cumprod, scaled_linear betas (train_sdxl_zh.py:140)
  std::vector<float> acv(1000);
  {
    const float b0 = sqrtf(0.00085f), b1 = sqrtf(0.012f);
    float prod = 1.f;
    for (int i = 0; i < 1000; ++i) {
      const float sb = b0 + (b1 - b0) * (float)i / 999.0f;
      prod *= 1.0f - sb * sb;
      acv[i] = prod;
    }
  }
  HIPCHK(hipMalloc((void**)&ac, 4000));
  HIPCHK(hipMemcpy(ac, acv.data(), 4000, hipMemcpyHostToDevice));
  return PEA_OK;
}

// One KD training step on one batch (train_sdxl_zh.py:311-441 after the frozen encoders):
// add_noise -> adapter (cond | uncond) -> CFG-dropout select -> student UNet (taps) -> teacher UNet
// -> fused KD loss + gradient seeds -> student data-gradient pass -> adapter dgrad + wgrad.
int Trainer::step(const float* latents, const float* noise, const long long* timesteps, const float* enc,
                  const float* enc_uncond, const unsigned char* prompt_mask, const long long* zh,
                  const float* teacher_ehs, const float* teacher_neg, const float* teacher_pooled,
                  const float* time_ids, float grad_scale, float* grads, int accumulate, float* losses_out,
                  hipStream_t s) {
  Tape& S = *student;
  Tape& Tt = *teacher;
  Adapter& A = *ad;
  const int B = S.B;
  SHAPECHK(A.B2 == 2 * B && A.L == S.L, "trainer: adapter is prepared for %d x %d rows, the step needs %d x %d", A.B2, A.L,
           2 * B, S.L);
  if (merge_passes && merge_state == 0) {
    // eligible: the teacher context shares the student's weight arena (same checkpoint, train_sdxl_zh.py:138,151 load
    // the same model_path) and both see the same context length
    // (merge_passes 1: for per-GPU batches <= 8 -- end of round 2, one box: 109.2 vs 114.0 ms at B = 4, 203.5 vs 205.2 ms
    // at B = 8 against the two-stream path; 2: always when eligible)
    // A student context shorter than the teacher's (the reference's default: Chinese-CLIP emits 52 tokens,
    // utils/custom_dataset_sdxl.py:352-353, the teacher's CLIP towers 77) merges too: the merged context is Tt.L tokens
    // long, the student rows carry S.L tokens + zero padding and a per-sample key count masks the padding in the
    // cross-attention forward / backward kernels (head_dim 64 instances only).
    bool all_nd1 = true;
    for (const Op& o : S.ops)
      if (o.kind == OP_ATTN && o.p3 != 1) all_nd1 = false;
    bool ok = (merge_passes >= 2 || B <= 8) && !Tt.owns_weights && Tt.slots.size() == S.slots.size() &&
              (S.L == Tt.L || (S.L < Tt.L && all_nd1)) &&
              S.graph == 0 && Tt.graph == 0 &&
              memcmp(&S.cfg, &Tt.cfg, sizeof(PeaUnetCfg)) == 0;
    for (size_t i = 0; ok && i < S.slots.size(); ++i)
      ok = S.slots[i].w == Tt.slots[i].w && S.slots[i].f32 == Tt.slots[i].f32;
    merge_state = ok ? 1 : -1;
    if (ok) {
      merged = new Tape();
      merged->cfg = S.cfg;
      merged->B = 2 * B; merged->H = S.H; merged->W = S.W; merged->L = Tt.L;
      merged->needs_grad = true; merged->owns_weights = false; merged->bwd_batch = B;
      RC(merged->build());
      RC(merged->share_weights_from(S));
      RC(merged->alloc());
      if (S.L != Tt.L) {
        std::vector<int> kl(2 * B, Tt.L);
        for (int i = 0; i < B; ++i) kl[i] = S.L;
        HIPCHK(hipMalloc((void**)&merged->cross_kvlen, sizeof(int) * 2 * B));
        HIPCHK(hipMemcpy(merged->cross_kvlen, kl.data(), sizeof(int) * 2 * B, hipMemcpyHostToDevice));
        merged->tn[merged->t_ehs].zero_init = true;     // the padding rows of the student samples stay zero
      }
      const size_t n = (size_t)B * S.cfg.in_channels * S.H * S.W;
      HIPCHK(hipMalloc((void**)&xt2, 2 * n * 4));
      HIPCHK(hipMalloc((void**)&eps2, 2 * n * 4));
      HIPCHK(hipMalloc((void**)&t2, 2 * B * 4));
      HIPCHK(hipMalloc((void**)&tid2, 2 * B * 6 * 4));
    }
  }
  if (merge_passes && merge_state == 1)
    return step_merged(latents, noise, timesteps, enc, enc_uncond, prompt_mask, zh, teacher_ehs, teacher_neg,
                       teacher_pooled, time_ids, grad_scale, grads, accumulate, losses_out, s);
  RC(S.ensure_acts());
  RC(Tt.ensure_acts());
  const long long per_img = (long long)S.cfg.in_channels * S.H * S.W;
  if (!t_f32) HIPCHK(hipMalloc((void**)&t_f32, sizeof(float) * B));
  RC(launch_add_noise(latents, noise, timesteps, ac, xt, B, per_img, s));
  RC(launch_cast_i64_f32(timesteps, t_f32, B, s));
  // The teacher forward (no_grad, :410-415) is independent of the adapter and of the student forward: it runs
  // on a side HIP stream so the two UNet passes fill each other's under-occupied launches.
  hipStream_t ts = s;
  if (two_stream) {
    if (!side) {
      HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(ev_fork, s));
    HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
    ts = side;
  }
  const long long per_tt = (long long)Tt.L * Tt.cfg.cross_dim;
  RC(launch_cast_f32_bf16(teacher_ehs, tehs_c, B * per_tt, ts));
  RC(launch_cast_f32_bf16(teacher_neg, tehs_n, B * per_tt, ts));
  Tn& tehs = Tt.tn[Tt.t_ehs];
  RC(launch_select_rows(tehs_c, tehs_n, prompt_mask, tehs.d, B, per_tt, ts));                    // :413
  RC(Tt.forward(xt, t_f32, tehs.d, 1, teacher_pooled, 0, time_ids, eps_t, ts));
  if (two_stream) HIPCHK(hipEventRecord(ev_join, side));
  // adapter on (cond | uncond) rows; train_sdxl_zh.py:383-384
  RC(A.forward(enc, enc_uncond, 0, s));
  const long long per_tok = (long long)S.L * S.cfg.cross_dim;
  const bf16* tokens = A.out1 ? A.tok : A.z2;
  Tn& ehs = S.tn[S.t_ehs];
  RC(launch_select_rows(tokens, tokens + B * per_tok, prompt_mask, ehs.d, B, per_tok, s));      // :395
  RC(S.forward(xt, t_f32, ehs.d, 1, S.t_text >= 0 ? (const void*)A.pooled : nullptr, 1, time_ids, eps_s, s));
  if (two_stream) HIPCHK(hipStreamWaitEvent(s, ev_join, 0));
  // fused KD loss + seeds; :399-441
  KdLossP kp;
  memset(&kp, 0, sizeof(kp));
  kp.kd_samples_hint = kd_samples_hint;
  kp.ntaps = (int)tap_pairs.size();
  S.begin_backward();
  for (int k = 0; k < kp.ntaps; ++k) {
    Tn& ts = S.tn[S.taps[tap_pairs[k].first]];
    kp.fs[k] = ts.d; kp.ft[k] = Tt.tn[Tt.taps[tap_pairs[k].second]].d; kp.dfs[k] = ts.g;
    kp.per[k] = ts.rows / B * ts.cols;
    ts.gw = true;
  }
  kp.eps_s = eps_s; kp.eps = noise; kp.eps_t = eps_t; kp.deps_s = deps; kp.per_eps = per_img; kp.zh = zh; kp.B = B;
  kp.feat_weight = feat_weight; kp.nan_guard = nan_guard; kp.grad_scale = grad_scale; kp.losses = losses;
  kp.partial = (float*)kd_ws;
  RC(launch_kd_loss(kp, s));
  if (losses_out) HIPCHK(hipMemcpyAsync(losses_out, losses, 16, hipMemcpyDeviceToDevice, s));
  // student data-gradient pass
  RC(S.backward(deps, s));
  SHAPECHK(ehs.gw, "trainer: no gradient reached encoder_hidden_states");
  // route d(ehs) to the cond / uncond adapter rows; pooled gradient only to the cond half (:384,390)
  bf16* dtokens = A.out1 ? A.dtok : A.dz2;
  RC(launch_select_rows_bwd(ehs.g, prompt_mask, dtokens, dtokens + B * per_tok, B, per_tok, s));
  if (A.out1) {
    HIPCHK(hipMemsetAsync(A.dpool, 0, (size_t)A.B2 * A.out_dim * 2, s));
    if (S.t_text >= 0 && S.tn[S.t_text].gw)
      HIPCHK(hipMemcpyAsync(A.dpool, S.tn[S.t_text].g, (size_t)B * A.out_dim * 2, hipMemcpyDeviceToDevice, s));
  }
  RC(A.backward(grads, accumulate, s));
  return PEA_OK;
}


// The same step with the two UNet forwards merged into one pass over 2B samples (rows [0, B): student conditioning,
// rows [B, 2B): teacher conditioning, same noisy latents and timesteps), possible when the teacher context shares the
// student's weights.  Every GEMM / conv / attention launch then works on twice the rows -- at B = 4 that is the
// difference between one and two 128-row tiles per CU in most launches -- and the backward pass walks the tape on
// the leading B samples only (Tape::bwd_batch).  The teacher half runs without `no_grad` bookkeeping differences:
// nothing in the forward depends on whether a gradient will be taken.
// merged-pass context for B student rows + nt live teacher rows (nt < B); see Trainer::live_teacher_mask
int Trainer::context_for(int nt, Tape** out) {
  Tape& S = *student;
  Tape& Tt = *teacher;
  const int B = S.B;
  if (nt >= B) { *out = merged; return PEA_OK; }
  auto it = merged_n.find(nt);
  if (it != merged_n.end()) { *out = it->second; return PEA_OK; }
  Tape* m = new Tape();
  m->cfg = S.cfg;
  m->B = B + nt; m->H = S.H; m->W = S.W; m->L = Tt.L;
  m->needs_grad = true; m->owns_weights = false; m->bwd_batch = B;
  int rc = m->build();
  if (rc == PEA_OK) rc = m->share_weights_from(S);
  if (rc == PEA_OK) rc = m->alloc();
  if (rc != PEA_OK) { delete m; return rc; }
  m->arena_donor = merged;
  if (S.L != Tt.L) {
    std::vector<int> kl(B + nt, Tt.L);
    for (int i = 0; i < B; ++i) kl[i] = S.L;
    HIPCHK(hipMalloc((void**)&m->cross_kvlen, sizeof(int) * (B + nt)));
    HIPCHK(hipMemcpy(m->cross_kvlen, kl.data(), sizeof(int) * (B + nt), hipMemcpyHostToDevice));
    m->tn[m->t_ehs].zero_init = true;
  }
  merged_n[nt] = m;
  *out = m;
  return PEA_OK;
}

int Trainer::step_merged(const float* latents, const float* noise, const long long* timesteps, const float* enc,
                         const float* enc_uncond, const unsigned char* prompt_mask, const long long* zh,
                         const float* teacher_ehs, const float* teacher_neg, const float* teacher_pooled,
                         const float* time_ids, float grad_scale, float* grads, int accumulate, float* losses_out,
                         hipStream_t s) {
  const int B = student->B;
  // live teacher rows (dead-row elimination, model.h): idx[j] = the sample whose teacher row is merged row B + j
  // (both tables are sized from B: a merged pass may run at any batch, merge_passes = 2 with the SD1.5 micro-batch of 40,
  // train_sd_zh.sh:18; the elimination itself needs the mask's bits, hence B <= 30)
  int nt = 0;
  std::vector<int> idx((size_t)B);
  tmap_h.assign((size_t)B, 0);
  const bool dre = live_teacher_mask >= 0 && B <= 30 && (live_teacher_mask & ((1 << B) - 1)) != ((1 << B) - 1);
  for (int i = 0; i < B; ++i) {
    const bool live = !dre || ((live_teacher_mask >> i) & 1);
    tmap_h[i] = live ? nt : -1;
    if (live) idx[nt++] = i;
  }
  Tape* Mp = merged;
  if (dre) RC(context_for(nt, &Mp));
  Tape& M = *Mp;
  SHAPECHK(M.t_text < 0 || (teacher_pooled != nullptr && time_ids != nullptr),
           "trainer: teacher_pooled / time_ids are required for a text_time UNet (added_cond_kwargs, train_sdxl_zh.py:386-396)");
  RC(M.ensure_acts());
  if (last_ctx != Mp && (dre || last_ctx != nullptr)) {
    // the contexts share one pair of arenas with different layouts: whatever must read as zero is cleared again
    for (Tn& t : M.tn)
      if (t.zero_init) HIPCHK(hipMemsetAsync(t.d, 0, (size_t)t.rows * t.cols * 2, s));
    for (int e : M.ext_res) HIPCHK(hipMemsetAsync(M.tn[e].d, 0, (size_t)M.tn[e].rows * M.tn[e].cols * 2, s));
  }
  last_ctx = Mp;
  Adapter& A = *ad;
  const long long per_img = (long long)M.cfg.in_channels * M.H * M.W;
  RC(launch_add_noise(latents, noise, timesteps, ac, xt2, B, per_img, s));
  RC(launch_cast_i64_f32(timesteps, t2, B, s));
  if (!dre) {
    HIPCHK(hipMemcpyAsync(xt2 + B * per_img, xt2, (size_t)B * per_img * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(t2 + B, t2, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  } else {
    for (int j = 0; j < nt; ++j) {
      HIPCHK(hipMemcpyAsync(xt2 + (B + j) * per_img, xt2 + idx[j] * per_img, (size_t)per_img * 4, hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(t2 + B + j, t2 + idx[j], 4, hipMemcpyDeviceToDevice, s));
    }
  }
  if (time_ids) {
    HIPCHK(hipMemcpyAsync(tid2, time_ids, (size_t)B * 6 * 4, hipMemcpyDeviceToDevice, s));
    if (!dre) HIPCHK(hipMemcpyAsync(tid2 + B * 6, time_ids, (size_t)B * 6 * 4, hipMemcpyDeviceToDevice, s));
    else
      for (int j = 0; j < nt; ++j)
        HIPCHK(hipMemcpyAsync(tid2 + (B + j) * 6, time_ids + idx[j] * 6, 6 * 4, hipMemcpyDeviceToDevice, s));
  }
  const long long per_tok = (long long)M.L * M.cfg.cross_dim;               // merged context: the teacher's length
  const long long per_stok = (long long)student->L * M.cfg.cross_dim;       // tokens the adapter emits per sample
  Tn& ehs = M.tn[M.t_ehs];
  // teacher rows: where(prompt_mask, negative, prompt) (:413)
  RC(launch_cast_f32_bf16(teacher_ehs, tehs_c, B * per_tok, s));
  RC(launch_cast_f32_bf16(teacher_neg, tehs_n, B * per_tok, s));
  if (!dre) RC(launch_select_rows(tehs_c, tehs_n, prompt_mask, ehs.d + B * per_tok, B, per_tok, s));
  else {
    // every live row is selected straight into its compacted place behind the student rows (the kernel's pointers are
    // __restrict__: no in-place select)
    for (int j = 0; j < nt; ++j)
      RC(launch_select_rows(tehs_c + idx[j] * per_tok, tehs_n + idx[j] * per_tok, prompt_mask + idx[j], ehs.d + (B + j) * per_tok, 1,
                            per_tok, s));
  }
  // student rows: adapter on (cond | uncond), CFG-dropout select (:383-395); a shorter student context leaves the
  // sample's tail rows at their zero padding (masked by Tape::cross_kvlen)
  RC(A.forward(enc, enc_uncond, 0, s));
  const bf16* tokens = A.out1 ? A.tok : A.z2;
  RC(launch_select_rows(tokens, tokens + B * per_stok, prompt_mask, ehs.d, B, per_stok, s, per_tok));
  const void* text = nullptr;
  if (M.t_text >= 0) {
    Tn& q = M.tn[M.t_text];
    HIPCHK(hipMemcpyAsync(q.d, A.pooled, (size_t)B * q.cols * 2, hipMemcpyDeviceToDevice, s));
    if (!dre) RC(launch_cast_f32_bf16(teacher_pooled, q.d + (long long)B * q.cols, (long long)B * q.cols, s));
    else {
      if (!tpool_c) HIPCHK(hipMalloc((void**)&tpool_c, (size_t)B * q.cols * 2));
      RC(launch_cast_f32_bf16(teacher_pooled, tpool_c, (long long)B * q.cols, s));
      for (int j = 0; j < nt; ++j)
        HIPCHK(hipMemcpyAsync(q.d + (long long)(B + j) * q.cols, tpool_c + (long long)idx[j] * q.cols, (size_t)q.cols * 2,
                              hipMemcpyDeviceToDevice, s));
    }
    text = q.d;
  }
  if (dre) {
    if (!tmap_d) HIPCHK(hipMalloc((void**)&tmap_d, sizeof(int) * 32));
    HIPCHK(hipMemcpyAsync(tmap_d, tmap_h.data(), sizeof(int) * B, hipMemcpyHostToDevice, s));
  }
  RC(M.forward(xt2, t2, ehs.d, 1, text, 1, tid2, eps2, s));
  KdLossP kp;
  memset(&kp, 0, sizeof(kp));
  kp.kd_samples_hint = kd_samples_hint;
  kp.ntaps = (int)M.taps.size();
  M.begin_backward();
  for (int k = 0; k < kp.ntaps; ++k) {
    Tn& tp = M.tn[M.taps[k]];
    const long long half = tp.rows / M.B * B * tp.cols;                      // the B student samples come first
    kp.fs[k] = tp.d; kp.ft[k] = tp.d + half; kp.dfs[k] = tp.g;
    kp.per[k] = tp.rows / M.B * tp.cols;
    tp.gw = true;
  }
  kp.eps_s = eps2; kp.eps = noise; kp.eps_t = eps2 + B * per_img; kp.deps_s = deps; kp.per_eps = per_img; kp.zh = zh;
  kp.B = B; kp.feat_weight = feat_weight; kp.nan_guard = nan_guard; kp.grad_scale = grad_scale; kp.losses = losses;
  kp.partial = (float*)kd_ws;
  kp.tmap = dre ? tmap_d : nullptr;
  RC(launch_kd_loss(kp, s));
  if (losses_out) HIPCHK(hipMemcpyAsync(losses_out, losses, 16, hipMemcpyDeviceToDevice, s));
  RC(M.backward(deps, s));
  SHAPECHK(ehs.gw, "trainer: no gradient reached encoder_hidden_states");
  bf16* dtokens = A.out1 ? A.dtok : A.dz2;
  RC(launch_select_rows_bwd(ehs.g, prompt_mask, dtokens, dtokens + B * per_stok, B, per_stok, s, per_tok));
  if (A.out1) {
    HIPCHK(hipMemsetAsync(A.dpool, 0, (size_t)A.B2 * A.out_dim * 2, s));
    if (M.t_text >= 0 && M.tn[M.t_text].gw)
      HIPCHK(hipMemcpyAsync(A.dpool, M.tn[M.t_text].g, (size_t)B * A.out_dim * 2, hipMemcpyDeviceToDevice, s));
  }
  RC(A.backward(grads, accumulate, s));
  return PEA_OK;
}
