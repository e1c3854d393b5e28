// C-ABI, model level (include/pea_hip.h): UNet contexts, the PEA adapter, the KD trainer.
#include <string.h>

#include "../../include/pea_hip.h"
#include "model.h"

#define RCX(x)                        \
  do {                                \
    int rc__ = (x);                   \
    if (rc__ != PEA_OK) return rc__;  \
  } while (0)
#define NOTNULL(p, what)                         \
  do {                                           \
    if (!(p)) {                                  \
      pea_set_error("%s: null handle", what);    \
      return PEA_E_INVALID;                      \
    }                                            \
  } while (0)

static_assert(sizeof(pea_unet_config) == sizeof(PeaUnetCfg), "config struct mismatch");

// shared tail of every pea_*_create: build the graph, plan and allocate the weights
static int finish_create(Tape* u, void** out) {
  int rc = u->build();
  if (rc == PEA_OK) rc = u->alloc();
  if (rc != PEA_OK) {
    delete u;
    return rc;
  }
  *out = u;
  return PEA_OK;
}
static int require_device(const char* what) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    pea_set_error("%s: no HIP device (there is no CPU fallback)", what);
    return PEA_E_HIP;
  }
  return PEA_OK;
}

extern "C" {

int pea_unet_create(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int own_weights,
                    void** out) {
  NOTNULL(cfg, "pea_unet_create");
  NOTNULL(out, "pea_unet_create");
  RCX(require_device("pea_unet_create"));
  Tape* u = new Tape();
  memcpy(&u->cfg, cfg, sizeof(PeaUnetCfg));
  u->B = B; u->H = H; u->W = W; u->L = L; u->needs_grad = (flags & 1) != 0; u->residual_inputs = (flags & 2) != 0;
  u->owns_weights = own_weights != 0;
  return finish_create(u, out);
}
int pea_unet_plan(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int* n_ops, int* n_weights,
                  long long* n_params, long long* weight_bytes, long long* act_bytes, long long* grad_bytes) {
  NOTNULL(cfg, "pea_unet_plan");
  Tape u;
  memcpy(&u.cfg, cfg, sizeof(PeaUnetCfg));
  u.B = B; u.H = H; u.W = W; u.L = L; u.needs_grad = (flags & 1) != 0; u.residual_inputs = (flags & 2) != 0;
  u.plan_only = true;
  int rc = u.build();
  if (rc == PEA_OK) rc = u.alloc();
  if (rc != PEA_OK) return rc;
  long long np = 0;
  for (const WSlot& s : u.slots) np += s.numel;
  if (n_ops) *n_ops = (int)u.ops.size();
  if (n_weights) *n_weights = (int)u.slots.size();
  if (n_params) *n_params = np;
  if (weight_bytes) *weight_bytes = (long long)u.wbytes;
  if (act_bytes) *act_bytes = (long long)u.abytes;
  if (grad_bytes) *grad_bytes = (long long)u.gbytes;
  return PEA_OK;
}
/* scratch buffers the same context allocates beside its arenas on first use (host only).  bwd_batch > 0: a merged-pass context
 * that differentiates only its leading bwd_batch samples (Tape::bwd_batch) */
int pea_unet_plan_scratch(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int bwd_batch, long long* scratch_bytes) {
  NOTNULL(cfg, "pea_unet_plan_scratch");
  NOTNULL(scratch_bytes, "pea_unet_plan_scratch");
  Tape u;
  memcpy(&u.cfg, cfg, sizeof(PeaUnetCfg));
  u.B = B; u.H = H; u.W = W; u.L = L; u.needs_grad = (flags & 1) != 0; u.residual_inputs = (flags & 2) != 0;
  u.bwd_batch = bwd_batch;
  u.plan_only = true;
  int rc = u.build();
  if (rc == PEA_OK) rc = u.alloc();
  if (rc != PEA_OK) return rc;
  size_t need[8];
  u.scratch_needs(need);
  long long t = 0;
  for (int i = 0; i < 8; ++i) t += (long long)need[i];
  *scratch_bytes = t;
  return PEA_OK;
}
/* attention ops of a graph and how many of them receive a Q already multiplied by scale * log2(e) by the producing
 * projection's epilogue (Tape::tag_q_prescale); every graph the product builds is expected to have the two equal */
int pea_tape_attention_census(void* h, int* n_attn, int* n_prescaled) {
  NOTNULL(h, "pea_tape_attention_census");
  const Tape* u = (const Tape*)h;
  if (n_attn) *n_attn = u->n_attn;
  if (n_prescaled) *n_prescaled = u->n_attn_pre;
  return PEA_OK;
}
int pea_unet_plan_attention(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int* n_attn, int* n_prescaled) {
  NOTNULL(cfg, "pea_unet_plan_attention");
  Tape u;
  memcpy(&u.cfg, cfg, sizeof(PeaUnetCfg));
  u.B = B; u.H = H; u.W = W; u.L = L; u.needs_grad = (flags & 1) != 0; u.residual_inputs = (flags & 2) != 0;
  u.plan_only = true;
  int rc = u.build();
  if (rc == PEA_OK) rc = u.alloc();
  if (rc != PEA_OK) return rc;
  if (n_attn) *n_attn = u.n_attn;
  if (n_prescaled) *n_prescaled = u.n_attn_pre;
  return PEA_OK;
}
static int plan_census(Tape& u, int* n_attn, int* n_prescaled) {
  u.plan_only = true;
  int rc = u.build();
  if (rc == PEA_OK) rc = u.alloc();
  if (rc != PEA_OK) return rc;
  if (n_attn) *n_attn = u.n_attn;
  if (n_prescaled) *n_prescaled = u.n_attn_pre;
  return PEA_OK;
}
/* graph: 1 VAE encoder, 2 ControlNet, 3 VAE decoder (the values of Tape::graph) */
int pea_graph_plan_attention(int graph, const pea_unet_config* cfg, int B, int H, int W, int L, int* n_attn, int* n_prescaled) {
  NOTNULL(cfg, "pea_graph_plan_attention");
  if (graph < 1 || graph > 3) { pea_set_error("pea_graph_plan_attention: graph %d (1 VAE encoder, 2 ControlNet, 3 VAE decoder)", graph); return PEA_E_INVALID; }
  Tape u;
  memcpy(&u.cfg, cfg, sizeof(PeaUnetCfg));
  u.graph = graph;
  u.B = B; u.H = H; u.W = W; u.L = graph == 2 ? L : 0; u.needs_grad = false; u.owns_weights = true;
  return plan_census(u, n_attn, n_prescaled);
}
int pea_text_plan_attention(const pea_text_config* cfg, int B, int L, int* n_attn, int* n_prescaled) {
  NOTNULL(cfg, "pea_text_plan_attention");
  Tape u;
  memset(&u.cfg, 0, sizeof(PeaUnetCfg));
  memcpy(&u.tcfg, cfg, sizeof(PeaTextCfg));
  u.graph = 4;
  u.B = B; u.H = 1; u.W = L; u.L = L; u.needs_grad = false; u.owns_weights = true;
  return plan_census(u, n_attn, n_prescaled);
}
int pea_controlnet_create(const pea_unet_config* cfg, int B, int H, int W, int L, void** out) {
  NOTNULL(cfg, "pea_controlnet_create");
  NOTNULL(out, "pea_controlnet_create");
  RCX(require_device("pea_controlnet_create"));
  Tape* u = new Tape();
  memcpy(&u->cfg, cfg, sizeof(PeaUnetCfg));
  u->graph = 2;
  u->B = B; u->H = H; u->W = W; u->L = L; u->needs_grad = false; u->owns_weights = true;
  return finish_create(u, out);
}
#define CN_HANDLE(h, what)                                                        \
  NOTNULL(h, what);                                                               \
  Tape* u = (Tape*)h;                                                             \
  if (u->graph != 2) { pea_set_error("%s: not a ControlNet handle", what); return PEA_E_INVALID; }
int pea_controlnet_set_cond(void* h, const float* image, void* stream) {
  CN_HANDLE(h, "pea_controlnet_set_cond");
  NOTNULL(image, "pea_controlnet_set_cond");
  std::string miss;
  if (!u->all_loaded(&miss)) {
    pea_set_error("controlnet: weight '%s' was never loaded", miss.c_str());
    return PEA_E_STATE;
  }
  u->cond_in = image;
  u->ce_valid = false;
  int rc = u->exec_ops(u->ce_begin, u->ce_end, false, (hipStream_t)stream);
  if (rc != PEA_OK) return rc;
  u->ce_valid = true;
  u->cond_in = nullptr;            // the image is consumed here; the embedding stays valid until the next set_cond
  return PEA_OK;
}
int pea_controlnet_forward(void* h, const float* x, const float* t, const void* ehs, int ehs_dtype, const void* text,
                           int text_dtype, const float* time_ids, void* stream) {
  CN_HANDLE(h, "pea_controlnet_forward");
  if (!u->ce_valid) { pea_set_error("pea_controlnet_forward: call pea_controlnet_set_cond first"); return PEA_E_STATE; }
  return u->forward(x, t, ehs, ehs_dtype, text, text_dtype, time_ids, nullptr, (hipStream_t)stream);
}
int pea_controlnet_num_outputs(void* h) { return h ? (int)((Tape*)h)->cn_out.size() : 0; }
int pea_controlnet_output(void* h, int i, void** ptr, int* C, int* H, int* W) {
  CN_HANDLE(h, "pea_controlnet_output");
  if (i < 0 || i >= (int)u->cn_out.size()) {
    pea_set_error("pea_controlnet_output: index %d out of range (%d outputs)", i, (int)u->cn_out.size());
    return PEA_E_INVALID;
  }
  { int rc = u->ensure_acts(); if (rc != PEA_OK) return rc; }
  const Tn& t = u->tn[u->cn_out[i]];
  if (ptr) *ptr = t.d;
  if (C) *C = t.cols;
  if (H) *H = t.H;
  if (W) *W = t.W;
  return PEA_OK;
}
int pea_controlnet_export_nchw(void* h, int i, float* dst, void* stream) {
  CN_HANDLE(h, "pea_controlnet_export_nchw");
  NOTNULL(dst, "pea_controlnet_export_nchw");
  if (i < 0 || i >= (int)u->cn_out.size()) {
    pea_set_error("pea_controlnet_export_nchw: index %d out of range", i);
    return PEA_E_INVALID;
  }
  const Tn& t = u->tn[u->cn_out[i]];
  return launch_nhwc_to_nchw_f32(t.d, dst, u->B, t.H * t.W, t.cols, (hipStream_t)stream);
}
int pea_vae_encoder_create(const pea_unet_config* cfg, int B, int H, int W, void** out) {
  NOTNULL(cfg, "pea_vae_encoder_create");
  NOTNULL(out, "pea_vae_encoder_create");
  RCX(require_device("pea_vae_encoder_create"));
  Tape* u = new Tape();
  memcpy(&u->cfg, cfg, sizeof(PeaUnetCfg));
  u->graph = 1;
  u->B = B; u->H = H; u->W = W; u->L = 0; u->needs_grad = false; u->owns_weights = true;
  return finish_create(u, out);
}
int pea_vae_latent_shape(void* h, int* C, int* H, int* W) {
  NOTNULL(h, "pea_vae_latent_shape");
  Tape* u = (Tape*)h;
  if (u->graph != 1) { pea_set_error("pea_vae_latent_shape: not a VAE encoder handle"); return PEA_E_INVALID; }
  const Tn& t = u->tn[u->t_out_in];
  if (C) *C = u->cfg.out_channels / 2;
  if (H) *H = t.H;
  if (W) *W = t.W;
  return PEA_OK;
}
int pea_vae_encode(void* h, const float* pixels, const float* noise, float scaling, float* moments, float* latents,
                   void* stream) {
  NOTNULL(h, "pea_vae_encode");
  NOTNULL(pixels, "pea_vae_encode");
  Tape* u = (Tape*)h;
  if (u->graph != 1) { pea_set_error("pea_vae_encode: not a VAE encoder handle"); return PEA_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  { int rc0 = u->ensure_acts(); if (rc0 != PEA_OK) return rc0; }
  int rc = u->forward(pixels, nullptr, nullptr, 0, nullptr, 0, nullptr, u->vae_h, s);
  if (rc != PEA_OK) return rc;
  const Tn& t = u->tn[u->t_out_in];
  return launch_vae_posterior(u->vae_h, u->slots[u->w_quant].f32, u->slots[u->b_quant].f32, noise, moments, latents, u->B,
                              u->cfg.out_channels, (long long)t.H * t.W, scaling, s);
}
int pea_vae_decoder_create(const pea_unet_config* cfg, int B, int H, int W, void** out) {
  NOTNULL(cfg, "pea_vae_decoder_create");
  NOTNULL(out, "pea_vae_decoder_create");
  RCX(require_device("pea_vae_decoder_create"));
  Tape* u = new Tape();
  memcpy(&u->cfg, cfg, sizeof(PeaUnetCfg));
  u->graph = 3;
  u->B = B; u->H = H; u->W = W; u->L = 0; u->needs_grad = false; u->owns_weights = true;
  return finish_create(u, out);
}
int pea_vae_decode(void* h, const float* latents, float inv_scaling, float* image, void* stream) {
  NOTNULL(h, "pea_vae_decode");
  NOTNULL(latents, "pea_vae_decode");
  NOTNULL(image, "pea_vae_decode");
  Tape* u = (Tape*)h;
  if (u->graph != 3) { pea_set_error("pea_vae_decode: not a VAE decoder handle"); return PEA_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  std::string miss;
  if (!u->all_loaded(&miss)) {
    pea_set_error("vae decoder: weight '%s' was never loaded", miss.c_str());
    return PEA_E_STATE;
  }
  { int rc0 = u->ensure_acts(); if (rc0 != PEA_OK) return rc0; }
  int rc = launch_vae_post_quant(latents, u->slots[u->w_quant].f32, u->slots[u->b_quant].f32, u->vae_h, u->B,
                                 u->cfg.in_channels, (long long)u->H * u->W, inv_scaling, s);
  if (rc != PEA_OK) return rc;
  return u->forward(u->vae_h, nullptr, nullptr, 0, nullptr, 0, nullptr, image, s);
}
static_assert(sizeof(pea_text_config) == sizeof(PeaTextCfg), "text config struct mismatch");
int pea_text_create(const pea_text_config* cfg, int B, int L, void** out) {
  NOTNULL(cfg, "pea_text_create");
  NOTNULL(out, "pea_text_create");
  RCX(require_device("pea_text_create"));
  Tape* u = new Tape();
  memset(&u->cfg, 0, sizeof(PeaUnetCfg));
  memcpy(&u->tcfg, cfg, sizeof(PeaTextCfg));
  u->graph = 4;
  u->B = B; u->H = 1; u->W = L; u->L = L; u->needs_grad = false; u->owns_weights = true;
  return finish_create(u, out);
}
int pea_text_forward(void* h, const long long* ids, int hidden_index, float* hidden_out, float* pooled_out, void* stream) {
  NOTNULL(h, "pea_text_forward");
  NOTNULL(ids, "pea_text_forward");
  Tape* u = (Tape*)h;
  if (u->graph != 4) { pea_set_error("pea_text_forward: not a text-encoder handle"); return PEA_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  std::string miss;
  if (!u->all_loaded(&miss)) {
    pea_set_error("text encoder: weight '%s' was never loaded", miss.c_str());
    return PEA_E_STATE;
  }
  { int rc0 = u->ensure_acts(); if (rc0 != PEA_OK) return rc0; }
  u->ids_in = ids;
  if (u->tcfg.flavor >= 1) {
    int rc = launch_kv_len(ids, u->kvlen, u->B, u->L, u->tcfg.eos_id, s);
    if (rc != PEA_OK) return rc;
  }
  if (u->tcfg.flavor == 2) {          // block 0's bucket table -> [heads][L][pitch] additive score bias (cheap; weights may have changed)
    int rc = launch_t5_rel_bias(u->slots[u->w_rel].f32, u->rel_bucket, u->rel_bias, u->tcfg.heads, u->L, ((u->L + 63) >> 6) << 6,
                                u->tcfg.rel_buckets / 2, s);
    if (rc != PEA_OK) return rc;
  }
  int rc = u->exec_ops(0, u->ops.size(), false, s);
  if (rc != PEA_OK) return rc;
  if (hidden_out) {
    const int nh = (int)u->hidden.size();
    int t = u->t_final;
    if (hidden_index != -1) {
      const int k = hidden_index < 0 ? nh + hidden_index : hidden_index;
      if (k < 0 || k >= nh) { pea_set_error("pea_text_forward: hidden_index %d out of range (%d states)", hidden_index, nh); return PEA_E_INVALID; }
      t = u->hidden[k];
    }
    rc = launch_cast_bf16_f32(u->tn[t].d, hidden_out, u->tn[t].rows * u->tn[t].cols, s);
    if (rc != PEA_OK) return rc;
  }
  if (pooled_out) {
    if (u->t_pooled < 0) { pea_set_error("pea_text_forward: this encoder has no pooled output"); return PEA_E_INVALID; }
    const Tn& t = u->tn[u->t_pooled];
    rc = launch_cast_bf16_f32(t.d, pooled_out, t.rows * t.cols, s);
    if (rc != PEA_OK) return rc;
  }
  return PEA_OK;
}
int pea_text_rel_bias(void* h, float* bias_out, void* stream) {
  NOTNULL(h, "pea_text_rel_bias");
  NOTNULL(bias_out, "pea_text_rel_bias");
  Tape* u = (Tape*)h;
  if (u->graph != 4 || u->tcfg.flavor != 2) { pea_set_error("pea_text_rel_bias: not a T5 encoder handle"); return PEA_E_INVALID; }
  if (!u->slots[u->w_rel].loaded) { pea_set_error("pea_text_rel_bias: the bias table was never loaded"); return PEA_E_STATE; }
  hipStream_t s = (hipStream_t)stream;
  { int rc0 = u->ensure_acts(); if (rc0 != PEA_OK) return rc0; }
  const int L = u->L, pitch = ((L + 63) >> 6) << 6;
  int rc = launch_t5_rel_bias(u->slots[u->w_rel].f32, u->rel_bucket, u->rel_bias, u->tcfg.heads, L, pitch, u->tcfg.rel_buckets / 2, s);
  if (rc != PEA_OK) return rc;
  HIPCHK(hipMemcpy2DAsync(bias_out, sizeof(float) * L, u->rel_bias, sizeof(float) * pitch, sizeof(float) * L,
                          (size_t)u->tcfg.heads * L, hipMemcpyDeviceToDevice, s));
  return PEA_OK;
}
int pea_unet_destroy(void* h) {
  delete (Tape*)h;
  return PEA_OK;
}
int pea_unet_num_residuals(void* h) { return h ? (int)((Tape*)h)->ext_res.size() : 0; }
int pea_unet_residual_info(void* h, int i, int* C, int* H, int* W) {
  NOTNULL(h, "pea_unet_residual_info");
  Tape* u = (Tape*)h;
  if (i < 0 || i >= (int)u->ext_res.size()) {
    pea_set_error("pea_unet_residual_info: index %d out of range (%d residual inputs)", i, (int)u->ext_res.size());
    return PEA_E_INVALID;
  }
  const Tn& t = u->tn[u->ext_res[i]];
  if (C) *C = t.cols;
  if (H) *H = t.H;
  if (W) *W = t.W;
  return PEA_OK;
}
int pea_unet_set_residuals(void* h, int n, const void* const* ptrs, int dtype, float scale, void* stream) {
  NOTNULL(h, "pea_unet_set_residuals");
  Tape* u = (Tape*)h;
  if (!u->residual_inputs) {
    pea_set_error("pea_unet_set_residuals: context created without PEA_UNET_RESIDUAL_INPUTS");
    return PEA_E_STATE;
  }
  if (n != (int)u->ext_res.size() || !ptrs) {
    pea_set_error("pea_unet_set_residuals: %d pointers given, the graph has %d residual inputs (mid last)", n,
                  (int)u->ext_res.size());
    return PEA_E_SHAPE;
  }
  { int rc = u->ensure_acts(); if (rc != PEA_OK) return rc; }
  for (int i = 0; i < n; ++i) {
    Tn& t = u->tn[u->ext_res[i]];
    if (!ptrs[i]) {
      HIPCHK(hipMemsetAsync(t.d, 0, (size_t)t.rows * t.cols * 2, (hipStream_t)stream));
      continue;
    }
    int rc = launch_residual_import(ptrs[i], dtype, t.d, u->B, t.cols, (long long)t.H * t.W, scale, (hipStream_t)stream);
    if (rc != PEA_OK) return rc;
  }
  return PEA_OK;
}
int pea_unet_num_weights(void* h) { return h ? (int)((Tape*)h)->slots.size() : 0; }
int pea_unet_weight_info(void* h, int i, char* name, int name_len, long long* numel, int* kind, int* d0, int* d1) {
  NOTNULL(h, "pea_unet_weight_info");
  Tape* u = (Tape*)h;
  if (i < 0 || i >= (int)u->slots.size()) {
    pea_set_error("pea_unet_weight_info: index %d out of range", i);
    return PEA_E_INVALID;
  }
  const WSlot& s = u->slots[i];
  if (name && name_len > 0) {
    strncpy(name, s.name.c_str(), name_len - 1);
    name[name_len - 1] = 0;
  }
  if (numel) *numel = s.numel;
  if (kind) *kind = s.kind;
  if (d0) *d0 = s.d0;
  if (d1) *d1 = s.d1;
  return PEA_OK;
}
int pea_unet_load_weight(void* h, const char* name, const float* src, long long numel, void* stream) {
  NOTNULL(h, "pea_unet_load_weight");
  Tape* u = (Tape*)h;
  if (!u->owns_weights) {
    pea_set_error("pea_unet_load_weight: context borrows its weights");
    return PEA_E_STATE;
  }
  return u->load_weight(name, src, numel, (hipStream_t)stream);
}
int pea_unet_init_random(void* h, unsigned long long seed, void* stream) {
  NOTNULL(h, "pea_unet_init_random");
  return ((Tape*)h)->init_random(seed, (hipStream_t)stream);
}
int pea_unet_share_weights(void* dst, void* src) {
  NOTNULL(dst, "pea_unet_share_weights");
  NOTNULL(src, "pea_unet_share_weights");
  return ((Tape*)dst)->share_weights_from(*(Tape*)src);
}
int pea_unet_forward(void* h, const float* x, const float* t, const void* ehs, int ehs_dtype, const void* text,
                     int text_dtype, const float* time_ids, float* eps_out, void* stream) {
  NOTNULL(h, "pea_unet_forward");
  return ((Tape*)h)->forward(x, t, ehs, ehs_dtype, text, text_dtype, time_ids, eps_out, (hipStream_t)stream);
}
int pea_unet_num_taps(void* h) { return h ? (int)((Tape*)h)->taps.size() : 0; }
int pea_unet_tap_name(void* h, int k, char* name, int name_len) {
  NOTNULL(h, "pea_unet_tap_name");
  NOTNULL(name, "pea_unet_tap_name");
  Tape* u = (Tape*)h;
  if (k < 0 || k >= (int)u->taps.size() || name_len < 2) {
    pea_set_error("pea_unet_tap_name: tap %d out of range", k);
    return PEA_E_INVALID;
  }
  snprintf(name, name_len, "%s", k < (int)u->tap_names.size() ? u->tap_names[k].c_str() : "");
  return PEA_OK;
}
int pea_unet_tap_info(void* h, int k, void** data, void** grad, int* B, int* H, int* W, int* C) {
  NOTNULL(h, "pea_unet_tap_info");
  Tape* u = (Tape*)h;
  if (k < 0 || k >= (int)u->taps.size()) {
    pea_set_error("pea_unet_tap_info: tap %d out of range", k);
    return PEA_E_INVALID;
  }
  { int rc = u->ensure_acts(); if (rc != PEA_OK) return rc; }
  const Tn& t = u->tn[u->taps[k]];
  // raw pointers of a depth-to-space tap: B / H / W / C do not reveal the storage order, so a caller that has never asked
  // for the layout would read (and seed) permuted pixels without any error -- refuse until it has
  if ((data || grad) && t.d2s && !u->tap_layout_queried) {
    pea_set_error("pea_unet_tap_info: tap %d is stored depth-to-space ([B][H/2][W/2][(y&1)*2+(x&1)][C]); call "
                  "pea_unet_tap_layout first (or use pea_unet_tap_export_nchw / pea_unet_tap_import_grad_nchw)", k);
    return PEA_E_STATE;
  }
  if (data) *data = t.d;
  if (grad) *grad = t.g;
  if (B) *B = t.B;
  if (H) *H = t.H;
  if (W) *W = t.W;
  if (C) *C = t.cols;
  return PEA_OK;
}
int pea_unet_tap_export_nchw(void* h, int k, int grad, float* out, void* stream) {
  NOTNULL(h, "pea_unet_tap_export_nchw");
  Tape* u = (Tape*)h;
  if (k < 0 || k >= (int)u->taps.size()) {
    pea_set_error("pea_unet_tap_export_nchw: tap %d out of range", k);
    return PEA_E_INVALID;
  }
  { int rc = u->ensure_acts(); if (rc != PEA_OK) return rc; }
  const Tn& t = u->tn[u->taps[k]];
  const bf16* src = grad ? t.g : t.d;
  NOTNULL(src, "pea_unet_tap_export_nchw(grad)");
  const int nb = (grad && u->bwd_batch > 0) ? u->bwd_batch : t.B;    // gradients exist for the differentiated samples only
  return launch_nhwc_to_nchw_f32(src, out, nb, t.H * t.W, t.cols, (hipStream_t)stream, t.d2s ? t.H : 0, t.d2s ? t.W : 0);
}
int pea_unet_tap_layout(void* h, int k) {
  Tape* u = (Tape*)h;
  if (!u || k < 0 || k >= (int)u->taps.size()) return -1;
  u->tap_layout_queried = true;
  return u->tn[u->taps[k]].d2s ? 1 : 0;
}
int pea_unet_tap_import_grad_nchw(void* h, int k, const float* src, void* stream) {
  NOTNULL(h, "pea_unet_tap_import_grad_nchw");
  NOTNULL(src, "pea_unet_tap_import_grad_nchw");
  Tape* u = (Tape*)h;
  if (k < 0 || k >= (int)u->taps.size()) {
    pea_set_error("pea_unet_tap_import_grad_nchw: tap %d out of range", k);
    return PEA_E_INVALID;
  }
  { int rc = u->ensure_acts(); if (rc != PEA_OK) return rc; }
  const Tn& t = u->tn[u->taps[k]];
  NOTNULL(t.g, "pea_unet_tap_import_grad_nchw(no gradient buffer)");
  const int nb = u->bwd_batch > 0 ? u->bwd_batch : t.B;
  return launch_nchw_f32_to_nhwc(src, t.g, nb, t.H * t.W, t.cols, (hipStream_t)stream, t.d2s ? t.H : 0, t.d2s ? t.W : 0);
}
int pea_unet_backward(void* h, const float* deps, unsigned tap_seed_mask, void* stream) {
  NOTNULL(h, "pea_unet_backward");
  Tape* u = (Tape*)h;
  u->begin_backward();
  for (size_t k = 0; k < u->taps.size(); ++k)
    if (tap_seed_mask & (1u << k)) u->tn[u->taps[k]].gw = true;
  return u->backward(deps, (hipStream_t)stream);
}
int pea_unet_input_grads(void* h, void** d_ehs, void** d_text) {
  NOTNULL(h, "pea_unet_input_grads");
  Tape* u = (Tape*)h;
  if (d_ehs) *d_ehs = u->tn[u->t_ehs].gw ? u->tn[u->t_ehs].g : nullptr;
  if (d_text) *d_text = (u->t_text >= 0 && u->tn[u->t_text].gw) ? u->tn[u->t_text].g : nullptr;
  return PEA_OK;
}
/* Parity instrumentation: the gradients of the two STACKED projections after the last backward pass, per layer.
 * which 0: every cross-attention K|V projection (ONE GEMM over encoder_hidden_states): d(K|V) as fp32 [rows][cols], rows =
 * differentiated samples x context length; which 1: every ResnetBlock2D.time_emb_proj (ONE GEMM over silu(emb)): the fp32
 * per-sample column sums [differentiated samples][cols].  out may be NULL (sizes only).  pea_unet_stacked_layout names the
 * column block of member i (diffusers weight key, e.g. `...attn2.to_k.weight`), PEA_E_NOTFOUND behind the last member. */
static int stacked_op(Tape* u, int which, const Op** op) {
  const int t = which == 0 ? u->t_kvall : u->t_tproj;
  if (t < 0) { pea_set_error("pea_unet_stacked_*: this graph has no such projection"); return PEA_E_INVALID; }
  for (const Op& o : u->ops)
    if (o.out == t && o.fused >= 0) { *op = &o; return PEA_OK; }
  pea_set_error("pea_unet_stacked_*: producer not found");
  return PEA_E_STATE;
}
int pea_unet_stacked_grad(void* h, int which, float* out, long long* rows, int* cols, void* stream) {
  NOTNULL(h, "pea_unet_stacked_grad");
  Tape* u = (Tape*)h;
  if (which != 0 && which != 1) { pea_set_error("pea_unet_stacked_grad: which=%d", which); return PEA_E_INVALID; }
  const Op* op = nullptr;
  { int rc = stacked_op(u, which, &op); if (rc != PEA_OK) return rc; }
  const Tn& t = u->tn[op->out];
  const int Bb = u->bwd_batch > 0 ? u->bwd_batch : u->B;
  const long long r = t.rows / u->B * Bb;
  if (rows) *rows = r;
  if (cols) *cols = t.cols;
  if (!out) return PEA_OK;
  if (which == 1 && !t.rg) {           /* e.g. SD1.5: no text_time conditioning, the adapter's outputs never reach the time embedding */
    pea_set_error("pea_unet_stacked_grad: the time embedding receives no gradient on this graph");
    return PEA_E_NOTFOUND;
  }
  if (!u->needs_grad || !t.g || !u->tproj_grad) { pea_set_error("pea_unet_stacked_grad: no backward pass has run on this context"); return PEA_E_STATE; }
  if (which == 0) return launch_cast_bf16_f32(t.g, out, r * t.cols, (hipStream_t)stream);
  HIPCHK(hipMemcpyAsync(out, u->tproj_grad, sizeof(float) * (size_t)r * t.cols, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PEA_OK;
}
int pea_unet_stacked_layout(void* h, int which, int i, char* name, int name_len, int* col_off, int* cols) {
  NOTNULL(h, "pea_unet_stacked_layout");
  Tape* u = (Tape*)h;
  const Op* op = nullptr;
  { int rc = stacked_op(u, which, &op); if (rc != PEA_OK) return rc; }
  int k = 0;
  for (const WSlot& s : u->slots) {
    if (s.fused_parent != op->fused || s.kind != W_LINEAR) continue;
    if (k++ != i) continue;
    if (name && name_len > 1) snprintf(name, name_len, "%s", s.name.c_str());
    if (col_off) *col_off = s.row_off;
    if (cols) *cols = s.st_n ? s.st_n : s.d0;
    return PEA_OK;
  }
  return PEA_E_NOTFOUND;
}
int pea_unet_memory(void* h, long long* weight_bytes, long long* act_bytes, long long* grad_bytes, int* n_ops) {
  NOTNULL(h, "pea_unet_memory");
  Tape* u = (Tape*)h;
  if (weight_bytes) *weight_bytes = (long long)u->wbytes;
  /* activations / gradients are allocated on first use: report what is resident */
  if (act_bytes) *act_bytes = u->aarena ? (long long)u->abytes : 0;
  if (grad_bytes) *grad_bytes = u->garena ? (long long)u->gbytes : 0;
  if (n_ops) *n_ops = (int)u->ops.size();
  return PEA_OK;
}

int pea_unet_release_activations(void* h) {
  NOTNULL(h, "pea_unet_release_activations");
  return ((Tape*)h)->release_acts();
}

// ------------------------------------------------------------------------------ adapter
int pea_adapter_create(int in_dim, int out_dim, int hidden_dim, int out_dim1, int use_residual, void** out) {
  NOTNULL(out, "pea_adapter_create");
  Adapter* a = new Adapter();
  a->in_dim = in_dim; a->out_dim = out_dim; a->hidden = hidden_dim; a->out1 = out_dim1; a->use_residual = use_residual;
  a->nparam = 2LL * in_dim + (long long)hidden_dim * in_dim + (long long)hidden_dim * hidden_dim +
              (long long)out_dim * hidden_dim + (long long)out_dim1 * out_dim + out_dim1;
  *out = a;
  return PEA_OK;
}
int pea_adapter_destroy(void* h) {
  delete (Adapter*)h;
  return PEA_OK;
}
long long pea_adapter_num_params(void* h) { return h ? ((Adapter*)h)->nparam : 0; }
int pea_adapter_bind(void* h, float* flat_params) {
  NOTNULL(h, "pea_adapter_bind");
  ((Adapter*)h)->params = flat_params;
  return PEA_OK;
}
int pea_adapter_prepare(void* h, int batch, int L) {
  NOTNULL(h, "pea_adapter_prepare");
  RCX(require_device("pea_adapter_prepare"));
  return ((Adapter*)h)->prepare(batch, L);
}
int pea_adapter_sync(void* h, void* stream) {
  NOTNULL(h, "pea_adapter_sync");
  return ((Adapter*)h)->sync_weights((hipStream_t)stream);
}
int pea_adapter_forward(void* h, const void* enc, int dtype, float* pooled, float* tokens, void* stream) {
  NOTNULL(h, "pea_adapter_forward");
  Adapter* a = (Adapter*)h;
  hipStream_t s = (hipStream_t)stream;
  RCX(a->forward(enc, nullptr, dtype, s));
  if (a->out1) {
    if (pooled) RCX(launch_cast_bf16_f32(a->pooled, pooled, (long long)a->B2 * a->out_dim, s));
    if (tokens) RCX(launch_cast_bf16_f32(a->tok, tokens, (long long)a->R * a->out1, s));
  } else if (tokens) {
    RCX(launch_cast_bf16_f32(a->z2, tokens, (long long)a->R * a->out_dim, s));
  }
  return PEA_OK;
}
int pea_adapter_backward(void* h, const float* d_pooled, const float* d_tokens, float* grads, int accumulate,
                         void* stream) {
  NOTNULL(h, "pea_adapter_backward");
  Adapter* a = (Adapter*)h;
  hipStream_t s = (hipStream_t)stream;
  if (a->out1) {
    if (d_tokens) RCX(launch_cast_f32_bf16(d_tokens, a->dtok, (long long)a->R * a->out1, s));
    else HIPCHK(hipMemsetAsync(a->dtok, 0, (size_t)a->R * a->out1 * 2, s));
    if (d_pooled) RCX(launch_cast_f32_bf16(d_pooled, a->dpool, (long long)a->B2 * a->out_dim, s));
    else HIPCHK(hipMemsetAsync(a->dpool, 0, (size_t)a->B2 * a->out_dim * 2, s));
  } else {
    NOTNULL(d_tokens, "pea_adapter_backward(d_tokens)");
    RCX(launch_cast_f32_bf16(d_tokens, a->dz2, (long long)a->R * a->out_dim, s));
  }
  return a->backward(grads, accumulate, s);
}

// ------------------------------------------------------------------------------ trainer
int pea_trainer_create(void* adapter, void* student, void* teacher, float feat_weight, int nan_guard,
                       const float* alphas_cumprod, void** out) {
  NOTNULL(adapter, "pea_trainer_create");
  NOTNULL(student, "pea_trainer_create");
  NOTNULL(teacher, "pea_trainer_create");
  Trainer* t = new Trainer();
  t->ad = (Adapter*)adapter; t->student = (Tape*)student; t->teacher = (Tape*)teacher;
  t->feat_weight = feat_weight; t->nan_guard = nan_guard;
  int rc = t->prepare();
  if (rc == PEA_OK && alphas_cumprod)
    if (hipMemcpy(t->ac, alphas_cumprod, 4000, hipMemcpyDeviceToDevice) != hipSuccess) rc = PEA_E_HIP;
  if (rc != PEA_OK) {
    delete t;
    return rc;
  }
  *out = t;
  return PEA_OK;
}
int pea_trainer_destroy(void* h) {
  delete (Trainer*)h;
  return PEA_OK;
}
int pea_train_step(void* h, const float* latents, const float* noise, const long long* timesteps, const float* enc,
                   const float* enc_uncond, const unsigned char* prompt_mask, const long long* zh_or_not,
                   const float* teacher_ehs, const float* teacher_neg, const float* teacher_pooled,
                   const float* time_ids, float grad_scale, float* grads, int accumulate, float* losses,
                   void* stream) {
  NOTNULL(h, "pea_train_step");
  return ((Trainer*)h)->step(latents, noise, timesteps, enc, enc_uncond, prompt_mask, zh_or_not, teacher_ehs,
                             teacher_neg, teacher_pooled, time_ids, grad_scale, grads, accumulate, losses,
                             (hipStream_t)stream);
}
int pea_trainer_set_option(void* h, const char* name, int value) {
  NOTNULL(h, "pea_trainer_set_option");
  Trainer* t = (Trainer*)h;
  if (!strcmp(name, "two_stream")) t->two_stream = value;
  else if (!strcmp(name, "merge_passes")) t->merge_passes = value;
  else if (!strcmp(name, "nan_guard")) t->nan_guard = value;
  else if (!strcmp(name, "live_teacher_mask")) t->live_teacher_mask = value;   /* dead-row elimination: bit i = compute sample i's teacher row; -1 = all */
  else if (!strcmp(name, "kd_samples_hint")) t->kd_samples_hint = value;    /* profiling only: samples with zh_or_not == 0 */
  else {
    pea_set_error("pea_trainer_set_option: unknown option '%s'", name);
    return PEA_E_INVALID;
  }
  return PEA_OK;
}
int pea_trainer_release_activations(void* h) {
  NOTNULL(h, "pea_trainer_release_activations");
  Trainer* t = (Trainer*)h;
  RCX(t->student->release_acts());
  RCX(t->teacher->release_acts());
  for (auto& kv : t->merged_n) RCX(kv.second->release_acts());     /* borrowers first */
  if (t->merged) RCX(t->merged->release_acts());
  t->last_ctx = nullptr;
  return PEA_OK;
}
int pea_trainer_get_option(void* h, const char* name) {
  if (!h || !name) return PEA_E_INVALID;
  Trainer* t = (Trainer*)h;
  if (!strcmp(name, "two_stream")) return t->two_stream;
  if (!strcmp(name, "merge_passes")) return t->merge_passes;
  if (!strcmp(name, "merge_state")) return t->merge_state;     /* 0 undecided, 1 merged, -1 not eligible */
  if (!strcmp(name, "nan_guard")) return t->nan_guard;
  if (!strcmp(name, "kd_samples_hint")) return t->kd_samples_hint;
  if (!strcmp(name, "live_teacher_mask")) return t->live_teacher_mask;
  if (!strcmp(name, "merged_rows")) return t->last_ctx ? t->last_ctx->B : (t->merged ? t->merged->B : 0);   /* samples in the last merged pass */
  if (!strcmp(name, "merged_mib"))                                /* activations + gradients of the merged-pass context */
  {
    if (!t->merged || !t->merged->aarena) return 0;
    size_t by = t->merged->abytes + t->merged->gbytes + t->merged->scratch_own_bytes();
    for (auto& kv : t->merged_n)                                  /* dead-row contexts: arenas and most scratch are borrowed */
      if (kv.second->aarena) by += kv.second->scratch_own_bytes();
    return (int)(by >> 20);
  }
  return PEA_E_INVALID;
}
/* the UNet context whose backward pass ran in the last step: the merged-pass context (teacher == student checkpoint), else the
 * student's own (parity instrumentation: pea_unet_stacked_grad) */
int pea_trainer_backward_context(void* h, void** unet) {
  NOTNULL(h, "pea_trainer_backward_context");
  NOTNULL(unet, "pea_trainer_backward_context");
  Trainer* t = (Trainer*)h;
  const bool mg = t->merge_passes && t->merge_state == 1;
  *unet = mg ? (void*)(t->last_ctx ? t->last_ctx : t->merged) : (void*)t->student;
  return PEA_OK;
}
int pea_trainer_export(void* h, int which, float* out, void* stream) {
  NOTNULL(h, "pea_trainer_export");
  Trainer* t = (Trainer*)h;
  const size_t n = (size_t)t->student->B * t->student->cfg.in_channels * t->student->H * t->student->W;
  const bool mg = t->merge_passes && t->merge_state == 1;
  const float* src = mg ? (which == 0 ? t->xt2 : which == 1 ? t->eps2 : t->eps2 + n)
                        : (which == 0 ? t->xt : which == 1 ? t->eps_s : t->eps_t);
  const int B = t->student->B;
  if (mg && which == 2 && t->last_ctx && t->last_ctx->B != 2 * B) {
    /* dead-row elimination: the teacher rows are compacted; rows that were not computed come back as NaN */
    const size_t per = n / B;
    HIPCHK(hipMemsetAsync(out, 0xff, n * 4, (hipStream_t)stream));
    for (int i = 0; i < B; ++i)
      if (i < (int)t->tmap_h.size() && t->tmap_h[i] >= 0)
        HIPCHK(hipMemcpyAsync(out + i * per, src + (size_t)t->tmap_h[i] * per, per * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PEA_OK;
  }
  HIPCHK(hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PEA_OK;
}

}  // extern "C"
