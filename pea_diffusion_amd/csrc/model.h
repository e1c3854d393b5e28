// Host runtime: the UNet as a static op tape (forward schedule + reverse data-gradient
// schedule), the PEA adapter, and the fused KD training step.  No autograd, no tracing compiler:
// the tape is built once per (config, batch, resolution, context length) and every launch is an
// explicit kernel from pea_kernels.h.
#pragma once
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "pea_kernels.h"

struct PeaUnetCfg {          // mirrors `pea_unet_config` of include/pea_hip.h
  int in_channels, out_channels;
  int n_levels;
  int block_out[4];
  int down_cross[4];         // CrossAttnDownBlock2D ?
  int up_cross[4];           // CrossAttnUpBlock2D ?
  int layers_per_block;
  int depth[4];              // transformer_layers_per_block (down order)
  int heads[4];              // attention heads per level (down order); head_dim must be 64
  int cross_dim;
  int linear_proj;           // use_linear_projection (SDXL 1, SD1.5 0 -> 1x1 conv weights [C][C][1][1])
  int groups;
  float eps;
  int text_time;             // addition_embed_type == "text_time"
  int add_time_dim;          // addition_time_embed_dim
  int proj_in_dim;           // projection_class_embeddings_input_dim
  int per_layer_depth;       // 0: fill the arrays below from depth[] (normalize_depths)
  int depth_down[4][4];      // transformer_layers_per_block[i][j] (down order, j < layers_per_block)
  int depth_up[4][4];        // reverse_transformer_layers_per_block[i][j] (UP order, j <= layers_per_block)
  int depth_mid;             // mid block transformer layers; -1: no mid block
};
void normalize_depths(PeaUnetCfg& c);

struct PeaTextCfg {          // mirrors `pea_text_config` of include/pea_hip.h
  int vocab, max_pos, width, heads, layers, intermediate;
  int act;                   // GEMM epilogue activation of the MLP: 1 GELU(erf), 3 quick-GELU
  int flavor;                // 0: CLIP text model (pre-LN, causal, final LN, EOS pooling), 1: BERT (post-LN, key padding), 2: T5 encoder
  int proj_dim;              // CLIP text_projection width (0: none)
  float eps;
  int pos_offset;            // position row = token index + pos_offset (RoBERTa / XLM-R: 2, else 0)
  long long eos_id;          // CLIP: EOS token id (< 0: argmax of the ids);  BERT / T5: pad token id
  int rel_buckets;           // T5: relative_attention_num_buckets
  int rel_max_dist;          // T5: relative_attention_max_distance
};

enum WKind { W_VEC, W_LINEAR, W_CONV3, W_CONV_IN, W_CONV_OUT };

struct WSlot {
  std::string name;
  int kind;
  int d0 = 0, d1 = 0;        // LINEAR: [N=d0][K=d1]; CONV3: [Co=d0][Ci=d1][3][3]; VEC: [d0]
  long long numel = 0;
  // device copies (pointers into the weight arena; fused matrices share a base with offsets)
  float* f32 = nullptr;      // VEC / CONV_IN raw / CONV_OUT packed
  bf16* w = nullptr; int ldw = 0;        // forward layout  [N][K] (conv: K = 9*Ci, (ky,kx,ci))
  bf16* wt = nullptr; int ldwt = 0;      // dgrad layout    [K][N] (conv: [Ci][9*Co] flipped)
  bool need_wt = false;
  bool loaded = false;
  bool subpix = false;       // CONV3 of an upsampler in its sub-pixel form: w = [4 parities][Co][4 Ci], wt = [Ci][16 Co] (launch_pack_conv_subpix)
  // arena bookkeeping
  size_t off_f32 = (size_t)-1, off_w = (size_t)-1, off_wt = (size_t)-1;
  int fused_parent = -1;     // index of the fused matrix this slot is a row block of
  int row_off = 0;
  int row_step = 1;          // > 1: this slot's rows are interleaved with its siblings' (T5 gated FF: wi_1 even, wi_0 odd)
  // head padding (heads whose width d is not a multiple of 64 are stored dp = 64*ceil(d/64) wide, zero filled)
  int pad_mode = 0, pad_d = 0, pad_dp = 0;   // 1: rows (to_q/k/v), 2: columns (to_out), 3: GEGLU (h_i, gate_i) row interleave
  int st_n = 0, st_k = 0;    // stored (padded) dims of a LINEAR weight
};

struct FusedMat {            // several Linear weights stacked along N sharing one input
  int N = 0, K = 0;
  bf16* w = nullptr; bf16* wt = nullptr; float* bias = nullptr;
  size_t off_w = (size_t)-1, off_wt = (size_t)-1, off_bias = (size_t)-1;
  bool need_wt = false, has_bias = false;
};

// A LayerNorm whose only consumer is a Linear is folded into it (UNet / ControlNet transformer blocks: norm1 -> fused QKV,
// norm2 -> attn2.to_q, norm3 -> FF projection):  LN(x) . W^T + b = rstd (x . W'^T - mean s) + t  with W' = W . gamma,
// s[n] = sum_k W'[n][k], t[n] = sum_k beta[k] W[n][k] + b[n].  Forward: a statistics-only pass over x and the GEMM on the
// un-normalised rows (the normalised tensor is never written); backward: unchanged (dgrad with the original W^T, then the
// LayerNorm backward from x and the statistics).
struct LnFold {
  int ln_op = -1, lin_op = -1;
  int gamma = -1, beta = -1;         // LayerNorm weight slots
  int w_slot = -1, fused = -1;       // the Linear's weight: a slot or a fused matrix
  int bias_slot = -1;
  int N = 0, K = 0;
  bf16* wf = nullptr; float* s = nullptr; float* t = nullptr;
  size_t off_wf = 0, off_s = 0, off_t = 0;
};

struct Tn {
  long long rows = 0; int cols = 0;
  int B = 0, H = 0, W = 0;
  bf16* d = nullptr; bf16* g = nullptr;
  bool rg = false;           // requires grad
  bool zero_init = false;    // channel-padded tensor: the padding columns are never written and must read as zeros
  bool gw = false;           // gradient already written in the current backward pass
  const bf16* gpend = nullptr;   // gradient passed on by a residual, not yet added into g (Tape::backward)
  size_t off_d = 0, off_g = 0;
  bool external = false;     // no buffer of its own
  bool d2s = false;          // stored depth-to-space, [B][H/2][W/2][(y&1)*2 + (x&1)][cols]: the output of a sub-pixel upsampler conv
                             // (OP_CONV3 p1 == 2).  Readers: OP_CONCAT's first operand, the KD-loss taps (elementwise per sample)
};

enum OpKind { OP_CONV_IN, OP_CONV3, OP_LINEAR, OP_GN, OP_LN, OP_ATTN, OP_GEGLU, OP_CONCAT, OP_SILU, OP_TEMB,
              OP_CONV_OUT, OP_ADD, OP_ATTN_MAT, OP_EMBED, OP_GATHER_EOS };

struct Op {
  int kind;
  int a = -1, b = -1, c = -1;     // input tensors
  int acol = 0, bcol = 0, ccol = 0;
  int out = -1;
  int res = -1;                   // residual tensor (epilogue add)
  int rv = -1, rv_off = 0;        // per-sample row vector tensor + column offset (conv1 + temb)
  int w = -1, bias = -1;          // weight slots (GN/LN: gamma, beta); fused: index into fused list (w = -2 - idx)
  int fused = -1;
  int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  float f0 = 0.f;
  float* aux = nullptr; size_t aux_off = 0, aux_bytes = 0;   // GN/LN stats, attention lse
  int src = 0;                    // OP_TEMB: 0 = timesteps, 1 = time_ids
  int mask = 0;                   // OP_ATTN: 1 causal, 2 per-sample key count (text encoders), 4 additive T5 position bias
  int fold = -1;                  // OP_LN / OP_LINEAR: index into Tape::folds (LayerNorm folded into the consuming Linear)
  int qs_cols = 0; float qs = 1.f; // OP_LINEAR: output columns [0, qs_cols) leave multiplied by qs (the Q block of an attention's projection)
  int pre = 0;                    // OP_ATTN: Q arrives multiplied by scale * log2(e) (Tape::tag_q_prescale)
  int stash_form = -1;            // fused-GEGLU OP_LINEAR: what the last FORWARD left in its stash (0: (h, gate), 1: the backward's
                                  // factors); written by Tape::forward, read by Tape::backward -- never re-derived there
};

// The static op tape: tensors, weight slots and ops of ONE graph, built once per (config, batch, size).  graph selects the
// builder: the UNet of the KD step (0), AutoencoderKL encoder / decoder (1 / 3), ControlNet (2), a text encoder (4).
struct Tape {
  PeaUnetCfg cfg;
  int B, H, W, L;                 // batch, latent H/W, context length
  bool needs_grad;
  int bwd_batch = 0;                 // > 0: backward() differentiates only the first bwd_batch samples (merged passes)
  int graph = 0;                     // 0: UNet2DConditionModel, 1: AutoencoderKL encoder, 2: ControlNetModel, 3: AutoencoderKL decoder, 4: text encoder (1-4: inference only)
  std::vector<int> cn_out;           // ControlNet: output tensors (down residuals in diffusers order, mid last)
  int ce_begin = -1, ce_end = -1;    // ControlNet: op range of the conditioning embedding (constant over a generation)
  bool ce_valid = false;             // ... already computed for the current conditioning image
  const float* cond_in = nullptr;    // ControlNet: conditioning image fp32 [B][3][8H][8W]
  int cond_scale_f = 8;
  int w_quant = -1, b_quant = -1;    // VAE: quant_conv (1x1) slots
  bf16* am_scores = nullptr;         // VAE mid attention: materialised [HW][HW] scores of one image, V^T of one image
  bf16* am_vt = nullptr;
  float* vae_h = nullptr;            // VAE: encoder output before quant_conv, fp32 [B][C2][h][w]
  bool residual_inputs = false;      // ControlNet: down_block_additional_residuals / mid_block_additional_residual
  std::vector<int> ext_res;          // their tensors, diffusers order (conv_in, down blocks..., then mid last)
  std::deque<WSlot> slots;
  std::map<std::string, int> slot_by_name;
  std::deque<FusedMat> fused;
  std::vector<Tn> tn;
  std::vector<Op> ops;
  std::vector<LnFold> folds;      // LayerNorm -> Linear folds; their W' / s / t live in the weight arena of the owner
  bool fold_dirty = true;         // a weight was (re)loaded since the folds were last computed (owner only)
  Tape* weights_owner = nullptr;  // set by share_weights_from
  bool tap_layout_queried = false;   // pea_unet_tap_layout was called: raw pointers of depth-to-space taps may be handed out
  int ensure_folded(hipStream_t s);
  std::vector<int> taps;          // tensor ids: d0.., m, u0..
  std::vector<std::string> tap_names;   // "d0".., "m" (absent without a mid block), "u0"..
  int t_ehs = -1, t_text = -1, t_tproj = -1, t_out_in = -1, t_kvall = -1;
  int tproj_total = 0, kvall_total = 0, kv_nsplit = 1;
  float* kv_part = nullptr;
  bf16* geglu_tmp = nullptr;
  // arenas
  char* warena = nullptr; size_t wbytes = 0; bool owns_weights = true;
  bool plan_only = false;          // alloc() computes the layout (wbytes / abytes / gbytes) without touching the device
  char* aarena = nullptr; size_t abytes = 0;
  char* garena = nullptr; size_t gbytes = 0;
  // scratch
  double* gn_scratch = nullptr; float* cs_scratch = nullptr; float* attn_part = nullptr; float* delta = nullptr; bf16* ups_tmp = nullptr; float* tproj_grad = nullptr;
  float* tmp_f32 = nullptr; size_t tmp_f32_elems = 0;   // load-time staging
  const bf16* zeros = nullptr;
  // per-call externals
  const float* x_in = nullptr; const float* t_in = nullptr; const float* tid_in = nullptr;
  float* eps_out = nullptr; const float* deps_in = nullptr;

  int build();
  int build_vae_encoder();
  int build_vae_decoder();
  int release_acts();
  int build_text();
  int build_text_t5();
  PeaTextCfg tcfg{};                 // graph 4: text encoder
  std::vector<int> hidden;           // graph 4: hidden_states[0..layers] tensor ids; t_final = final LN output, t_pooled
  int t_final = -1, t_pooled = -1;
  const long long* ids_in = nullptr;
  int* kvlen = nullptr;              // graph 4 (BERT): per-sample valid token count
  float* rel_bias = nullptr;         // graph 4 (T5): [heads][L][64*ceil(L/64)] relative-position bias in the log2 domain
  int* rel_bucket = nullptr;         // ... |key - query| -> sub-bucket table [L]
  int w_rel = -1;                    // ... its [buckets][heads] weight slot
  int* cross_kvlen = nullptr;        // graph 0: per-sample valid context tokens of the cross-attention (merged passes with a shorter
                                     // student context; rows beyond it in t_ehs are zero padding), null = all L
  int exec_ops(size_t begin, size_t end, bool skip_cached, hipStream_t s);
  // Next-op weight prefetch (GemmP::pf_ptr).  The GEMM / conv launches of a pass and their weight matrices form a fixed
  // sequence per tape: the first forward (backward) pass records it, every later pass hands launch j the matrix of launch
  // j + 1 (a launch whose weights do not match the record drops it; the next pass records again).
  struct WSeq { std::vector<std::pair<const void*, long long>> w; size_t pos = 0; bool ready = false; };
  WSeq wseq_fwd, wseq_bwd;
  WSeq* wseq_cur = nullptr;
  void wseq_begin(WSeq& q) { wseq_cur = &q; q.pos = 0; if (!q.ready) q.w.clear(); }
  void wseq_end() { if (wseq_cur && !wseq_cur->ready && !wseq_cur->w.empty()) wseq_cur->ready = true; wseq_cur = nullptr; }
  void wseq_drop() { for (WSeq* q : {&wseq_fwd, &wseq_bwd}) { q->ready = false; q->w.clear(); q->pos = 0; } wseq_cur = nullptr; }
  int gemm(GemmP& p, hipStream_t s);   // launch_gemm with the prefetch target filled in
  // Deferred split reduce of the cross-attention backward (AttnP::defer_reduce): the partials of the last deferrable launch wait
  // in one half of attn_part for the next cross-attention launch (whose workgroups add them up in their prologue) or for the
  // flush in front of the stacked K|V dgrad GEMM / at the end of the pass
  AttnP pend_red{};
  bool pend_red_valid = false;
  int part_toggle = 0;
  size_t part_half_elems = 0;        // floats per half of attn_part (scratch_needs doubles the partial buffer)
  int flush_pending_reduce(hipStream_t s);
  Tape* arena_donor = nullptr;       // activation / gradient arenas borrowed from this (larger) tape: the two are never live at once
  bool arena_borrowed = false;
  // bytes of each scratch buffer (ensure_acts) and which of them are the donor's (bit order: gn, cs, delta, ups, geglu, attn_part,
  // kv_part, tproj_grad): a borrower takes the donor's buffer when it is large enough
  size_t sc_gn = 0, sc_cs = 0, sc_delta = 0, sc_ups = 0, sc_geglu = 0, sc_part = 0, sc_kv = 0, sc_tproj = 0;
  unsigned scratch_borrowed = 0;
  void drop_borrowed_scratch();
  void scratch_needs(size_t need[8]);
  size_t scratch_own_bytes() const;  // scratch this context allocated itself (not borrowed)
  int n_attn = 0, n_attn_pre = 0;    // attention ops on the tape / of those, fed a prescaled Q (tag_q_prescale)
  void tag_q_prescale();
  int ensure_acts();                 // lazy allocation of the activation / gradient arenas and scratch
  int alloc();
  int load_weight(const char* name, const float* dev_ptr, long long numel, hipStream_t s);
  int init_random(unsigned long long seed, hipStream_t s);
  int share_weights_from(const Tape& src);
  int forward(const float* x, const float* t, const void* ehs, int ehs_dtype, const void* text, int text_dtype,
              const float* time_ids, float* eps, hipStream_t s);
  int backward(const float* deps, hipStream_t s);   // tap grads must have been seeded (or zero-flagged) first
  void begin_backward();
  int all_loaded(std::string* missing) const;
  ~Tape();
};

struct Adapter {
  int in_dim, out_dim, hidden, out1, use_residual;   // out1 == 0: SD1.5 variant (tokens only)
  int R = 0, Rpad = 0, B2 = 0, L = 0;                // rows = B2 * L
  long long nparam = 0;
  // flat fp32 parameter order == state_dict order: layernorm.weight, layernorm.bias,
  // projector.0.weight, projector.2.weight, projector.4.weight, fc.weight, fc.bias
  long long off_lnw, off_lnb, off_w0, off_w1, off_w2, off_fcw, off_fcb;
  float* params = nullptr;        // caller-owned flat fp32 (device)
  char* arena = nullptr;
  bf16 *w0, *w1, *w2, *wfc, *w0t, *w1t, *w2t, *wfct;            // bf16 working copies
  bf16 *x, *xn, *z0, *a0, *z1, *a1, *z2, *a2, *tok, *pooled;     // activations
  bf16 *dtok, *da2, *dz2, *da1, *dz1, *da0, *dz0, *dxn, *dpool;  // gradients
  bf16 *tA, *tB;                                                 // transposed operands for wgrad
  float* ln_stats;
  int prepare(int B2, int L);
  int sync_weights(hipStream_t s);                               // refresh bf16 copies after an optimizer step
  int forward(const void* enc, const void* enc2, int dtype, hipStream_t s);   // enc [B2*L][in] or two halves
  int backward(float* grads, int accumulate, hipStream_t s);     // consumes dtok / dpool
  ~Adapter();
};

struct Trainer {
  Adapter* ad; Tape* student; Tape* teacher;
  float feat_weight = 0.1f; int nan_guard = 0;
  float* xt = nullptr; float* eps_s = nullptr; float* eps_t = nullptr; float* deps = nullptr; float* ac = nullptr;
  bf16* t_ehs_sel = nullptr; bf16* dehs_full = nullptr;
  float* losses = nullptr; double* kd_ws = nullptr;
  bf16 *tehs_c = nullptr, *tehs_n = nullptr;
  int prepare();
  int step(const float* latents, const float* noise, const long long* timesteps, const float* enc,
           const float* enc_uncond, const unsigned char* prompt_mask, const long long* zh, const float* teacher_ehs,
           const float* teacher_neg, const float* teacher_pooled, const float* time_ids, float grad_scale,
           float* grads, int accumulate, float* losses_out, hipStream_t s);
  float* t_f32 = nullptr;
  std::vector<std::pair<int, int>> tap_pairs;   // (student tap, teacher tap) with the same hook name
  int two_stream = 1; hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // merged passes: when the teacher IS the student checkpoint (shared weights, reference default) both forwards run
  // as ONE pass over 2B samples (student rows first) and the backward differentiates the first B only
  int kd_samples_hint = -1;          // profiling only (KdLossP::kd_samples_hint)
  int merge_passes = 1; int merge_state = 0;   // state: 0 undecided, 1 merged, -1 not eligible
  Tape* merged = nullptr;
  // Dead-row elimination (opt-in, option "live_teacher_mask"): the KD terms of a sample carry the weight (1 - zh_or_not)
  // (train_sdxl_zh.py:402-441), so the teacher row of a sample with zh_or_not == 1 is multiplied by zero and never read by the
  // loss kernel -- with the mask known on the host that row is not computed: the merged pass runs over B + n_t samples (student
  // rows first, then the n_t live teacher rows, compacted).  Bit i set = sample i's teacher row is computed; -1 = all (default).
  int live_teacher_mask = -1;
  std::map<int, Tape*> merged_n;        // contexts for B + n_t rows, n_t < B: built on first use, arenas borrowed from `merged`
  Tape* last_ctx = nullptr;
  int* tmap_d = nullptr;                // device int[B]: teacher row of sample b (relative to the first teacher row) or -1
  std::vector<int> tmap_h;              // host copy of the last step's map, one entry per sample (export)
  bf16* tpool_c = nullptr;              // [B][pooled] bf16: teacher pooled embeds of all samples, before compaction
  int context_for(int nt, Tape** out);
  float *xt2 = nullptr, *eps2 = nullptr, *t2 = nullptr, *tid2 = nullptr;
  int step_merged(const float* latents, const float* noise, const long long* timesteps, const float* enc,
                  const float* enc_uncond, const unsigned char* prompt_mask, const long long* zh, const float* teacher_ehs,
                  const float* teacher_neg, const float* teacher_pooled, const float* time_ids, float grad_scale,
                  float* grads, int accumulate, float* losses_out, hipStream_t s);
  ~Trainer();
};
