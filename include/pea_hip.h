/* pea_hip.h -- C ABI of libpea_hip.so, the MI355X (gfx950) PEA-Diffusion training-step library.
 *
 * The reference (OPPO-Mente-Lab/PEA-Diffusion) is pure Python and defines no FFI; its plug-in
 * surfaces are three Python call signatures:
 *   - the adapter      `proj(x) -> (pooled, tokens)`           train_sdxl_zh.py:43-67, :383-384
 *   - the UNet call    `unet(x_t, t, ehs, added_cond_kwargs)`  train_sdxl_zh.py:397,415
 *   - the train step   `training_step(batch, i) -> {"loss"}`   train_sdxl_zh.py:305-449
 * Each entry point below cites the reference lines whose work it replaces.  Conventions:
 *   - every function returns 0 on success or a negative PEA_E_* code; pea_last_error() returns a
 *     thread-local message.  No exception crosses the boundary.
 *   - the caller owns all tensor memory: raw DEVICE pointers (e.g. torch `tensor.data_ptr()`),
 *     contiguous row-major.  Images cross the boundary as NCHW fp32; inside, activations are
 *     token-major ("NHWC") bf16.  `bf16` arguments are `void*` to 2-byte brain-float data.
 *   - all work is enqueued on the `stream` argument (a hipStream_t passed as void*; NULL = the
 *     default stream); nothing synchronises unless documented as blocking.
 *   - a context is driven by one host thread at a time.
 * There is NO CPU fallback: without a gfx950 device every compute entry point fails.
 */
#ifndef PEA_HIP_H
#define PEA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PEA_OK 0
#define PEA_E_INVALID (-1)
#define PEA_E_HIP (-2)
#define PEA_E_SHAPE (-3)
#define PEA_E_STATE (-4)
#define PEA_E_NOTFOUND (-5)
#define PEA_E_TIMEOUT (-6)

const char* pea_last_error(void);
int pea_version(void);
/* number of visible HIP devices (0 when none); never initialises a context on its own */
int pea_device_count(void);

/* ======================================================================== operator level ====
 * One entry point per hand-written kernel family; used by the parity tests and by callers that
 * want a single op.  act: 0 none, 1 GELU(erf), 2 SiLU.                                         */

/* C[M][N] = act(alpha * A[M][K] . W[N][K]^T + bias[n] + rowvec[m / rows_per_batch][n]) + res[m][n]
 * (torch.nn.Linear / F.linear inside the UNet and the adapter, train_sdxl_zh.py:48-55,62-64).
 * A, W, rowvec, res, preact: bf16.  C: bf16, or fp32 when out_f32 (accum_f32: C += ...).
 * K % 64 == 0, N % 4 == 0.  NULL pointers switch the corresponding term off.                    */
int pea_op_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, float alpha,
                const float* bias, const void* rowvec, int ldrv, int rows_per_batch, int act, void* preact,
                int ldpre, const void* res, int ldres, int out_f32, int accum_f32, void* stream);

/* C = A . W^T + bias with columns n < qscale_cols multiplied by qscale (qscale_cols % 16 == 0): the fused Q|K|V projection
 * of an attention layer, whose Q block leaves multiplied by softmax_scale * log2(e) (diffusers Attention.to_q/to_k/to_v).   */
int pea_op_gemm_qscale(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* bias,
                       int qscale_cols, float qscale, void* stream);

/* The FF projection with GEGLU in its epilogue (diffusers GEGLU.forward: hidden, gate = proj(x).chunk(2); hidden *
 * gelu(gate); W rows interleaved (h_i, gate_i)):  y[m][n] = h * gelu(gate)  (bf16 [M][N/2]); stash (optional, bf16 [M][N],
 * rows < stash_rows when stash_rows > 0): stash_grad 0: the pre-activation pair (h, gate); 1: the pair the BACKWARD
 * multiplies by, (gelu(gate), h * gelu'(gate)) -- no weight gradients exist on this path, so nothing else reads it.  */
int pea_op_gemm_geglu(const void* A, int lda, const void* W, int ldw, const float* bias, void* y, void* stash, int M, int N,
                      int K, int stash_grad, int stash_rows, void* stream);

/* The data-gradient GEMM of the FF output projection with the GEGLU backward in its epilogue (diffusers GEGLU:
 * y = h * gelu(gate); train_sdxl_zh.py's student backward runs it in every transformer block):
 *   dy = A[M][K] . W[N][K]^T;  form 0: pre[m][2n] = h, pre[m][2n+1] = gate (the pre-activation stash of the forward);
 *   form 1: pre[m][2n] = gelu(gate), pre[m][2n+1] = h * gelu'(gate) (a stash_grad forward);
 *   C[m][2n] = dy * gelu(gate),  C[m][2n+1] = dy * h * gelu'(gate)      (C: bf16 [M][2N], ldc >= 2N)
 * dy itself is never stored.  K % 64 == 0, N % 16 == 0, ldc % 8 == 0, ldpre % 8 == 0.                              */
int pea_op_gemm_geglu_bwd(const void* A, int lda, const void* W, int ldw, const void* pre, int ldpre, void* C, int ldc,
                          int M, int N, int K, int form, void* stream);

/* LayerNorm folded into the Linear that consumes it (UNet transformer blocks: norm1 -> to_q|k|v, norm2 -> attn2.to_q,
 * norm3 -> ff.net.0.proj):  y = LN(x; gamma, beta, eps) . W^T + bias  computed as  rstd (x . W'^T - mean s) + t  with
 * W' = W . gamma, s[n] = sum_k W'[n][k], t[n] = sum_k beta[k] W[n][k] + bias[n] -- one statistics pass over x and one
 * GEMM on the un-normalised rows.  x bf16 [M][K]; W bf16 [N][K]; y bf16 [M][N]; geglu_y (optional) bf16 [M][N/2] =
 * h * gelu(gate) for interleaved (h_i, gate_i) weight rows (y then receives the pre-activation, may be NULL).
 * Caller-provided scratch: Wf bf16 [N][K], svec / tvec fp32 [N], stats fp32 [M][2].                      */
int pea_op_ln_linear(const void* x, const float* gamma, const float* beta, const void* W, const float* bias, void* y,
                     void* geglu_y, int M, int N, int K, float eps, void* Wf, float* svec, float* tvec, float* stats,
                     void* stream);

/* 3x3 convolution, padding 1, as implicit GEMM over an NHWC bf16 tensor x[B][Hs][Ws][Cin] with
 * packed weights w[Cout][(ky,kx,ci)] (see pea_op_pack_conv).  (ResnetBlock2D conv1/conv2,
 * Downsample2D, Upsample2D of the UNet called at train_sdxl_zh.py:397,415.)
 * stride 1|2; upsample2x: nearest-2x upsample folded into the gather; transposed2: zero-stuffed
 * input (data-gradient of a stride-2 conv).  Epilogue as pea_op_gemm.                            */
int pea_op_conv3x3(const void* x, const void* w, void* y, int B, int Hs, int Ws, int Cin, int Cout, int stride,
                   int upsample2x, int transposed2, const float* bias, const void* rowvec, int ldrv,
                   const void* res, void* stream);
/* torch conv weight [Co][Ci][3][3] fp32 -> bf16 packed; dgrad=1 gives the flipped/transposed
 * weights w'[ci][(2-ky,2-kx,co)] whose forward conv is the data gradient.                        */
int pea_op_pack_conv(const float* w, void* out, int Co, int Ci, int dgrad, void* stream);
/* conv3x3(interpolate(x, 2x nearest)) -- diffusers Upsample2D, the UNet's up_blocks[i].upsamplers[0] -- in its sub-pixel
 * form: per output parity (y&1, x&1) a 2 x 2 kernel of summed taps over the SOURCE (16 tap products per source pixel and
 * channel pair instead of the 36 of a 3 x 3 gather over the upsampled image).
 *   pea_op_pack_conv_subpixel: [Co][Ci][3][3] fp32 -> bf16 [4][Co][4 Ci] (dgrad = 0) or [Ci][16 Co] (dgrad = 1), 16 Co Ci elements
 *   pea_op_upconv_subpixel:    x NHWC [B][Hs][Ws][Cin] -> y depth-to-space [B][Hs][Ws][(y&1)*2+(x&1)][Cout]  (+ bias)
 *   pea_op_upconv_subpixel_dgrad: dy (that layout) -> dx NHWC [B][Hs][Ws][Cin] (+ res, may alias dx)                     */
int pea_op_pack_conv_subpixel(const float* w, void* out, int Co, int Ci, int dgrad, void* stream);
/* torch.cat([a, b], dim=1) on NHWC rows (unet_2d_blocks.py: the skip concatenation of every up-block resnet) and its backward
 * (da (+)= dy[:, :C1], db (+)= dy[:, C1:]; NULL = skip).  aH, aW != 0: a / da are stored depth-to-space at full resolution
 * aH x aW (an upsampler output in its sub-pixel form); the result rows are plain NHWC.                                   */
int pea_op_concat2(const void* a, int C1, const void* b, int C2, void* y, long long rows, int aH, int aW, void* stream);
int pea_op_split2(const void* dy, int C1, int C2, void* da, int accum_a, void* db, int accum_b, long long rows, int aH,
                  int aW, void* stream);
int pea_op_upconv_subpixel(const void* x, const void* w, void* y, int B, int Hs, int Ws, int Cin, int Cout,
                           const float* bias, void* stream);
int pea_op_upconv_subpixel_dgrad(const void* dy, const void* wt, void* dx, int B, int Hs, int Ws, int Cin, int Cout,
                                 const void* res, void* stream);
int pea_op_conv_in(const float* x_nchw, const float* w, const float* bias, void* y_nhwc, int B, int Cin, int H,
                   int W, int Cout, void* stream);
int pea_op_conv_out(const void* x_nhwc, const float* w_packed, const float* bias, float* y_nchw, int B, int Cin,
                    int H, int W, int Cout, void* stream);
int pea_op_conv_out_dgrad(const float* dy_nchw, const float* w_packed, void* dx_nhwc, int B, int Cin, int H, int W,
                          int Cout, void* stream);
int pea_op_pack_conv_out(const float* w, float* out, int Co, int Ci, void* stream);

/* GroupNorm (+ optional SiLU) over x[B][HW][C] bf16; stats fp32 [B][groups][2]; scratch: device bytes from
 * pea_op_groupnorm_scratch_bytes (per-block partial sums: the reduction order is fixed, results are
 * bit-reproducible)                                                                               */
long long pea_op_groupnorm_scratch_bytes(int B, int HW, int C, int groups);
int pea_op_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, void* scratch,
                         int B, int HW, int C, int groups, float eps, int silu, void* stream);
int pea_op_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats,
                         void* dx, void* scratch, int B, int HW, int C, int groups, int silu, int accum,
                         void* stream);
/* LayerNorm over rows; stats fp32 [R][2]; dgamma/dbeta (fp32, accumulated) may be NULL */
int pea_op_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int R, int C,
                         float eps, void* stream);
int pea_op_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* stats, void* dx,
                         float* dgamma, float* dbeta, int R, int C, int accum, void* stream);

/* softmax(scale Q K^T) V (diffusers AttnProcessor2_0 -> SDPA).  Q/K/V/O bf16 with row strides ld* (elements);
 * head h occupies columns [64*nd*h, 64*nd*(h+1)) where nd = ceil(head_dim / 64) in {1,2,3}; a head narrower than
 * 64*nd is zero padded (SD1.5: 40 -> 64, 80 -> 128, 160 -> 192) and `scale` stays head_dim^-0.5.  lse fp32 [B][H][Sq]. */
int pea_op_attention_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                         float* lse, int B, int H, int Sq, int Skv, float scale, int nd, void* stream);
/* ... on a Q that already carries scale * log2(e) (the product path: the Q|K|V / to_q projection applies the factor to its fp32
 * accumulator, pea_op_gemm_qscale, so nothing is rounded twice); `scale` is still the softmax scale */
int pea_op_attention_fwd_prescaled(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                                   float* lse, int B, int H, int Sq, int Skv, float scale, int nd, void* stream);
/* text-encoder attention (head_dim 64, forward only): `causal` = key index <= query index (CLIP text model), kv_len =
 * device int[B] of valid key counts (BERT right padding) or NULL */
int pea_op_attention_fwd_masked(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                                float* lse, int B, int H, int Sq, int Skv, float scale, int causal, const int* kv_len,
                                void* stream);
/* dQ/dK/dV (dQ may be NULL; dK and dV together); delta: fp32 scratch [2][B][H][Sq] (row constants of the backward kernels); scratch: optional device
 * buffer of pea_op_attention_bwd_scratch_bytes(...) bytes enabling the query-split dK/dV form used when the
 * key count is small (cross-attention); NULL = single pass                                            */
long long pea_op_attention_bwd_scratch_bytes(int B, int H, int Sq, int Skv, int nd);
int pea_op_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                         int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                         void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                         int accum_dq, int accum_dkv, int nd, void* scratch, void* stream);

/* the backward on a prescaled Q (see pea_op_attention_fwd_prescaled); dQ is the gradient w.r.t. the UNSCALED q, so the
 * projection's data-gradient GEMM needs no factor */
int pea_op_attention_bwd_prescaled(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                                   int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                                   void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                                   int accum_dq, int accum_dkv, int nd, void* scratch, void* stream);

int pea_op_geglu_fwd(const void* hg, void* y, long long rows, int inner, void* stream);
int pea_op_geglu_bwd(const void* hg, const void* dy, void* dhg, long long rows, int inner, void* stream);
int pea_op_sumpool2(const void* x, void* y, int B, int H, int W, int C, int accum, void* stream);
int pea_op_timestep_embed(const float* t, void* y, int n, int dim, void* stream);
/* DDPM add_noise (train_sdxl_zh.py:322); ac = alphas_cumprod fp32[1000] on device */
int pea_op_add_noise(const float* x0, const float* eps, const long long* t, const float* ac, float* xt, int B,
                     long long per, void* stream);
int pea_op_cast_f32_bf16(const float* x, void* y, long long n, void* stream);
int pea_op_cast_bf16_f32(const void* x, float* y, long long n, void* stream);

/* Inference denoise loop glue (tests/test_sdxl_zh.py:376-406).
 * cfg_combine: eps2 = fp32 [2B][per] (unconditional half first, :394); out[B][per] = u + g*(t-u) (:395), then when
 *   guidance_rescale > 0 `rescale_noise_cfg` (:44-56, unbiased per-sample std); workspace from *_workspace_bytes(B).
 * dpm_update: one DPMSolverMultistepScheduler.step (:406; diffusers 0.23 [ext], dpmsolver++ midpoint) with host-side
 *   coefficients: x0 = (sample - sigma_s*eps)/alpha_s; sample <- c_s*sample + c_0*x0 + c_1*x0_prev; x0_prev <- x0. */
long long pea_op_cfg_combine_workspace_bytes(int B);
int pea_op_cfg_combine(const float* eps2, float* out, int B, long long per, float guidance_scale, float guidance_rescale,
                       void* workspace, void* stream);
int pea_op_dpm_update(float* sample, const float* eps, float* x0_prev, long long n, float alpha_s, float sigma_s,
                      float c_s, float c_0, float c_1, void* stream);

/* Fused KD loss of train_sdxl_zh.py:399-441 (SD1.5: train_sd_zh.py:217-276, nan_guard=1).
 * taps_s/taps_t/dtaps: HOST arrays of ntaps device pointers (bf16, elementwise-paired layouts);
 * per: HOST array of per-sample element counts.  eps_*: fp32 [B][per_eps].  zh: int64 [B] device.
 * losses: fp32[4] device = (loss, train_loss, train_loss_logits, train_loss_features).
 * dtaps[k] / deps_s receive dL/d(student tap) / dL/d(eps_s) (may be NULL).
 * workspace: device scratch of pea_op_kd_loss_workspace_bytes(...) bytes (per-block partials).           */
long long pea_op_kd_loss_workspace_bytes(int ntaps, const long long* per, long long per_eps, int B);
int pea_op_kd_loss(int ntaps, const void* const* taps_s, const void* const* taps_t, void* const* dtaps,
                   const long long* per, const float* eps_s, const float* eps, const float* eps_t, float* deps_s,
                   long long per_eps, const long long* zh, int B, float feat_weight, int nan_guard,
                   float grad_scale, float* losses, void* workspace, void* stream);

/* touch `bytes` of a read-only buffer so it becomes resident in the Infinity Cache (side-stream prefetch) */
int pea_op_prefetch(const void* p, long long bytes, void* stream);

/* fused AdamW over a flat fp32 buffer (DeepSpeed FusedAdam(adam_w_mode=True), utils/model_utils.py:64-67) */
int pea_op_adamw(float* w, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, void* stream);


/* ========================================================================== model level ====
 * The three plug-in surfaces of the reference, as opaque handles.                               */

/* UNet2DConditionModel config (diffusers 0.23 names; `heads` is what diffusers calls
 * `attention_head_dim` for SDXL).  Arrays are in DOWN-block order, n_levels entries used.        */
typedef struct pea_unet_config {
  int in_channels, out_channels;
  int n_levels;
  int block_out[4];
  int down_cross[4];          /* 1 = CrossAttnDownBlock2D, 0 = DownBlock2D */
  int up_cross[4];            /* 1 = CrossAttnUpBlock2D,   0 = UpBlock2D  (UP order) */
  int layers_per_block;
  int depth[4];               /* transformer_layers_per_block */
  int heads[4];               /* attention heads; channels / heads must be 64 */
  int cross_dim;
  int linear_proj;            /* use_linear_projection */
  int groups;
  float eps;
  int text_time;              /* addition_embed_type == "text_time" */
  int add_time_dim;
  int proj_in_dim;
  /* per-position transformer depths (diffusers >= 0.22 configs such as SSD-1B, loaded as a downstream UNet at
   * tests/test_sdxl_zh.py:449-454).  per_layer_depth = 0: the arrays below are ignored and filled from depth[]
   * (every attention of a level has depth[level], the mid block depth[n_levels - 1], the up path mirrors the down path). */
  int per_layer_depth;
  int depth_down[4][4];       /* transformer_layers_per_block[i][j], DOWN order, j < layers_per_block */
  int depth_up[4][4];         /* reverse_transformer_layers_per_block[i][j], UP order, j <= layers_per_block */
  int depth_mid;              /* mid block transformer layers; -1: mid_block_type == null (no mid block at all) */
} pea_unet_config;

/* Builds the static op tape for a fixed (batch B, latent H x W, context length L) and allocates
 * its activations (all kept resident: 288 GB HBM) -- replaces `UNet2DConditionModel.from_pretrained`
 * (train_sdxl_zh.py:138,151).  needs_grad=1 adds the reverse data-gradient tape (student);
 * own_weights=0 creates a context that must borrow weights via pea_unet_share_weights (the
 * reference loads teacher and student from the same checkpoint, train_sdxl_zh.py:138 vs :151).    */
/* flags: PEA_UNET_GRAD (backward support) | PEA_UNET_RESIDUAL_INPUTS (ControlNet residual inputs, inference only) */
#define PEA_UNET_GRAD 1
#define PEA_UNET_RESIDUAL_INPUTS 2
int pea_unet_create(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int own_weights,
                    void** out);
/* Host-only planning pass of pea_unet_create (no device needed, nothing allocated): builds the op tape for the same
 * arguments and reports its size -- ops, weight tensors, parameters (torch numel), and the bytes the weight / activation
 * / gradient arenas will take in HBM.  Use it to size a configuration against 288 GB before creating it.   */
int pea_unet_plan(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int* n_ops, int* n_weights,
                  long long* n_params, long long* weight_bytes, long long* act_bytes, long long* grad_bytes);
/* ... and the scratch buffers the same context allocates beside those arenas on first use (GroupNorm partials, attention
 * row constants and dK/dV split partials, the FF d(pre-activation) buffer, split-K partials of the stacked K|V projection,
 * fp32 time-embedding gradients).  bwd_batch > 0: a merged-pass context that differentiates its leading bwd_batch samples. */
int pea_unet_plan_scratch(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int bwd_batch,
                          long long* scratch_bytes);
/* Attention ops on a graph (any handle of this library that owns an op tape: UNet, ControlNet, VAE, text encoder) and how
 * many of them are fed a Q the producing projection already multiplied by softmax_scale * log2(e) (the accurate path: one
 * rounding, from the fp32 accumulator).  An op outside that count runs the operator-level plain-Q path, which rounds the
 * scaled operand to bf16 once more.  pea_unet_plan_attention: the same census from the host-only planning pass. */
int pea_tape_attention_census(void* graph, int* n_attn, int* n_prescaled);
int pea_unet_plan_attention(const pea_unet_config* cfg, int B, int H, int W, int L, int flags, int* n_attn, int* n_prescaled);
int pea_graph_plan_attention(int graph, const pea_unet_config* cfg, int B, int H, int W, int L, int* n_attn,
                             int* n_prescaled);          /* graph: 1 VAE encoder, 2 ControlNet, 3 VAE decoder */
/* ControlNet extras of the UNet call (tests/test_sdxl_zh_controlnet.py:534-535): `down_block_additional_residuals`
 * (conv_in output, then every down-block resnet/attention output and downsampler output, in diffusers order) followed
 * by `mid_block_additional_residual` LAST.  ptrs: HOST array of n device pointers ([B,C,H,W]; NULL entry = zero);
 * dtype 0 fp32 NCHW, 1 bf16 NCHW, 2 bf16 NHWC; values are multiplied by `scale` (conditioning_scale) on import and
 * stay in effect for every following pea_unet_forward until set again. */
int pea_unet_num_residuals(void* unet);
int pea_unet_residual_info(void* unet, int i, int* C, int* H, int* W);
int pea_unet_set_residuals(void* unet, int n, const void* const* ptrs, int dtype, float scale, void* stream);
int pea_unet_destroy(void* unet);

/* ControlNet (`self.controlnet(control_model_input, t, encoder_hidden_states=..., controlnet_cond=image,
 * conditioning_scale=..., guess_mode=False, added_cond_kwargs=..., return_dict=False)`,
 * tests/test_sdxl_zh_controlnet.py:510-519; diffusers 0.23 ControlNetModel [ext]) on the same op tape: the UNet's
 * conditioning, conv_in, down blocks and mid block + `controlnet_cond_embedding.*`, `controlnet_down_blocks.*`,
 * `controlnet_mid_block.*` (diffusers keys; handle works with the pea_unet_* weight functions and destroy).
 * set_cond: conditioning image fp32 [B,3,8H,8W]; its embedding is computed here, once per generation (the image does
 *   not change over the denoise steps), and reused by every forward.
 * forward: same conditioning arguments as pea_unet_forward; results stay in the context as bf16 NHWC tensors
 *   (pea_controlnet_output: down residuals in diffusers order, mid LAST) -- hand them to
 *   pea_unet_set_residuals(unet, n, ptrs, 2, conditioning_scale, stream). */
int pea_controlnet_create(const pea_unet_config* cfg, int B, int H, int W, int L, void** out);
int pea_controlnet_set_cond(void* cn, const float* image, void* stream);
int pea_controlnet_forward(void* cn, const float* x, const float* t, const void* ehs, int ehs_dtype, const void* text,
                           int text_dtype, const float* time_ids, void* stream);
int pea_controlnet_num_outputs(void* cn);
int pea_controlnet_output(void* cn, int i, void** ptr, int* C, int* H, int* W);
int pea_controlnet_export_nchw(void* cn, int i, float* dst, void* stream);   /* fp32 [B,C,H,W] copy of output i */

/* Text encoders in front of the step, on the same op tape; handle works with the pea_unet_* weight
 * functions (HF transformers keys).  flavor 0 = CLIPTextModel[WithProjection] -- the teacher's CLIP-L and OpenCLIP-bigG
 * (train_sdxl_zh.py:147-150; encode_prompt :170-285 takes `hidden_states[-2]` of both and the pooled output of the
 * second): pre-LN, causal mask, final LayerNorm, pooled = final[EOS position] @ text_projection.  flavor 1 = BERT -- the
 * Chinese-CLIP text tower (train_sdxl_zh.py:103-107; `self.text_encoder.encode_text(batch["input_ids"])` :327-329 returns
 * the per-token states): post-LN, right-padding mask from pad id `eos_id`.  head_dim must be 64 (true for all three).
 * forward: ids int64 [B,L] device; hidden_index -1 = last state (CLIP: after the final LayerNorm), -2 = hidden_states[-2],
 * k >= 0 = hidden_states[k]; hidden_out fp32 [B,L,width] and / or pooled_out fp32 [B,proj_dim] (CLIP only).
 * flavor 2 = T5 encoder stack -- the mT5 student option (`T5EncoderModel.from_pretrained('mt5-xl')`, train_sdxl_zh.py:108-112;
 * `self.text_encoder.encoder(ids, attention_mask=ids.ne(pad))[0]` :337-345; HF keys `shared.weight`, `encoder.block.N.*`,
 * `encoder.final_layer_norm.weight`): token embedding only, pre-RMSNorm blocks, scores = q.k (unscaled) + bucketed
 * relative-position bias of block 0, right-padding mask from pad id `eos_id`, gated FF wo(gelu_new(wi_0 x) * wi_1 x);
 * `intermediate` = d_ff; inner attention width = heads * 64 (d_kv must be 64).  hidden_index -1 = after the final RMSNorm. */
typedef struct pea_text_config {
  int vocab, max_pos, width, heads, layers, intermediate;
  int act;          /* 1 GELU(erf), 3 quick-GELU */
  int flavor;       /* 0 CLIP, 1 BERT, 2 T5 encoder (mT5: RMSNorm, relative position bias, unscaled scores, gated gelu_new FF) */
  int proj_dim;     /* CLIP text_projection width, 0 = none */
  float eps;
  int pos_offset;   /* position row = token index + pos_offset: 2 for RoBERTa / XLM-R towers (mul_clip, alt_clip), else 0 */
  long long eos_id; /* CLIP: EOS id (< 0: argmax of the ids); BERT / T5: pad id */
  int rel_buckets;  /* T5: relative_attention_num_buckets (32) */
  int rel_max_dist; /* T5: relative_attention_max_distance (128) */
} pea_text_config;
int pea_text_create(const pea_text_config* cfg, int B, int L, void** out);
int pea_text_plan_attention(const pea_text_config* cfg, int B, int L, int* n_attn, int* n_prescaled);   /* see pea_tape_attention_census */
int pea_text_forward(void* enc, const long long* ids, int hidden_index, float* hidden_out, float* pooled_out, void* stream);
/* T5 flavour only: the additive attention bias the encoder uses, bias_out fp32 [heads][L][L] =
 * relative_attention_bias[bucket(k - q)][h] * log2(e) (HF T5Attention.compute_bias; the attention kernels work in the
 * log2 domain).  Diagnostic read-back for the bucket parity test. */
int pea_text_rel_bias(void* enc, float* bias_out, void* stream);

/* VAE encoder (AutoencoderKL.encode, train_sdxl_zh.py:306-309; train_sd_zh.py:188-189) on the same op tape: cfg uses
 * in_channels (3), out_channels (2 * latent channels = 8), n_levels, block_out, layers_per_block, groups, eps.
 * The handle works with pea_unet_num_weights / weight_info / load_weight / init_random / memory / destroy (diffusers
 * keys `encoder.*`, `quant_conv.*`).  pea_vae_encode: pixels fp32 [B,3,H,W] -> moments = quant_conv(encoder(x))
 * fp32 [B,2L,h,w] (optional) and latents = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scaling fp32 [B,L,h,w]
 * (`.latent_dist.sample() * vae.config.scaling_factor`; noise NULL = `.mode()`). */
int pea_vae_encoder_create(const pea_unet_config* cfg, int B, int H, int W, void** out);
int pea_vae_latent_shape(void* vae, int* C, int* H, int* W);
int pea_vae_encode(void* vae, const float* pixels, const float* noise, float scaling, float* moments, float* latents,
                   void* stream);
/* VAE decoder (`image = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]`,
 * tests/test_sdxl_zh.py:430): cfg.in_channels = latent channels, out_channels = image channels, block_out = the
 * AutoencoderKL block_out_channels; H x W = LATENT size; diffusers keys `decoder.*`, `post_quant_conv.*`.
 * latents fp32 [B,4,h,w] (multiplied by inv_scaling first) -> image fp32 [B,3,8h,8w]. */
int pea_vae_decoder_create(const pea_unet_config* cfg, int B, int H, int W, void** out);
int pea_vae_decode(void* vae, const float* latents, float inv_scaling, float* image, void* stream);
int pea_unet_num_weights(void* unet);
/* diffusers state-dict key + torch shape (d0,d1; conv adds [3][3]) of weight i; kind: 0 vector,
 * 1 linear [d0][d1] (1x1 convs included), 2 conv3x3 [d0][d1][3][3], 3 conv_in, 4 conv_out         */
int pea_unet_weight_info(void* unet, int i, char* name, int name_len, long long* numel, int* kind, int* d0, int* d1);
/* src: DEVICE fp32, torch layout, `numel` elements; converted to the internal bf16 layouts      */
int pea_unet_load_weight(void* unet, const char* name, const float* src, long long numel, void* stream);
/* random weights of this architecture (no checkpoints exist in the image)                        */
int pea_unet_init_random(void* unet, unsigned long long seed, void* stream);
int pea_unet_share_weights(void* dst, void* src);
/* unet(sample, t, encoder_hidden_states, added_cond_kwargs={text_embeds,time_ids})[0]
 * (train_sdxl_zh.py:397,415; tests/test_sdxl_zh.py:384-391).  x, eps_out: fp32 NCHW [B][4][H][W];
 * t: fp32 [B]; ehs: [B][L][cross_dim], text: [B][pooled], dtype 0 = fp32, 1 = bf16; time_ids fp32 [B][6]. */
int pea_unet_forward(void* unet, const float* x, const float* t, const void* ehs, int ehs_dtype, const void* text,
                     int text_dtype, const float* time_ids, float* eps_out, void* stream);
/* feature taps in the order of the reference's cast_hook (train_sdxl_zh.py:79-84): d0.., m, u0..  */
int pea_unet_num_taps(void* unet);
/* hook name of tap k: "d<i>" (down_blocks[i], the hidden state of its (hidden, res_samples) tuple), "m" (mid_block;
 * absent when the config has no mid block), "u<i>" (up_blocks[i])                                     */
int pea_unet_tap_name(void* unet, int k, char* name, int name_len);
/* shape of tap k and (optionally) its raw storage pointers.  A tap that is stored depth-to-space (pea_unet_tap_layout == 1)
 * hands out raw pointers only after pea_unet_tap_layout has been called on this context (PEA_E_STATE otherwise): the
 * shape alone does not reveal the storage order.                                                                   */
int pea_unet_tap_info(void* unet, int k, void** data, void** grad, int* B, int* H, int* W, int* C);
int pea_unet_tap_export_nchw(void* unet, int k, int grad, float* out, void* stream);
/* Storage layout behind pea_unet_tap_info's pointers: 0 = NHWC [B][H][W][C]; 1 = depth-to-space [B][H/2][W/2][(y&1)*2+(x&1)][C]
 * (the output of an upsampler conv in its sub-pixel form -- Upsample2D = interpolate(2x nearest) + conv, diffusers
 * resnet.py; reference call site: the u<i> hooks of train_sdxl_zh.py:79-84).  -1 = bad handle / index.  The NCHW export above
 * and the import below hide the layout; only callers that touch the raw pointers need it.                                  */
int pea_unet_tap_layout(void* unet, int k);
/* write a gradient seed for tap k (fp32 NCHW [B or bwd_batch][C][H][W]) into the tap's gradient buffer in its storage layout;
 * then pass bit k in pea_unet_backward's tap_seed_mask                                                                   */
int pea_unet_tap_import_grad_nchw(void* unet, int k, const float* src, void* stream);
/* reverse pass: d(loss)/d(eps) in `deps` (fp32 NCHW, may be NULL) plus tap gradient seeds already
 * written into the tap grad buffers for every k with bit k set in tap_seed_mask.  Results:
 * pea_unet_input_grads -> bf16 d(ehs) [B][L][cross], d(text_embeds) [B][pooled].                 */
int pea_unet_backward(void* unet, const float* deps, unsigned tap_seed_mask, void* stream);
int pea_unet_input_grads(void* unet, void** d_ehs, void** d_text);
/* resident bytes: weights; activations and gradients (allocated on the first forward / backward, 0 before) */
/* Parity instrumentation: gradients of the two STACKED projections after the last backward pass, per layer.  which 0: all
 * cross-attention to_k / to_v projections (one GEMM over encoder_hidden_states; diffusers Attention.to_k/to_v of every
 * BasicTransformerBlock under train_sdxl_zh.py:397): d(K|V) fp32 [rows][cols], rows = differentiated samples x context length;
 * which 1: all ResnetBlock2D.time_emb_proj layers (one GEMM over silu(emb)): fp32 [differentiated samples][cols].  out may be
 * NULL (sizes only; PEA_E_NOTFOUND for which 1 with a buffer on a graph whose time embedding receives no gradient, e.g. SD1.5).
 * pea_unet_stacked_layout: diffusers weight key and column block of member i; PEA_E_NOTFOUND past the end. */
int pea_unet_stacked_grad(void* unet, int which, float* out, long long* rows, int* cols, void* stream);
int pea_unet_stacked_layout(void* unet, int which, int i, char* name, int name_len, int* col_off, int* cols);
int pea_unet_memory(void* unet, long long* weight_bytes, long long* act_bytes, long long* grad_bytes, int* n_ops);
/* Free the activation / gradient arenas and scratch of a context (weights stay); the next forward allocates them again.
 * Pointers handed out by pea_unet_tap_info / pea_unet_input_grads become invalid.  Synchronises the device. */
int pea_unet_release_activations(void* unet);

/* The PEA adapter `MLP(in_dim, out_dim, hidden_dim, out_dim1, use_residual)` (train_sdxl_zh.py:43-67);
 * out_dim1 = 0 selects the SD1.5 variant (train_sd_zh.py:41-56).  Parameters live in ONE flat fp32
 * device buffer owned by the caller, in state_dict order (layernorm.weight, layernorm.bias,
 * projector.0.weight, projector.2.weight, projector.4.weight, fc.weight, fc.bias).               */
int pea_adapter_create(int in_dim, int out_dim, int hidden_dim, int out_dim1, int use_residual, void** out);
int pea_adapter_destroy(void* ad);
long long pea_adapter_num_params(void* ad);
int pea_adapter_bind(void* ad, float* flat_params);
int pea_adapter_prepare(void* ad, int batch, int L);     /* rows = batch * L */
int pea_adapter_sync(void* ad, void* stream);            /* refresh bf16 working copies after an update */
/* x1, x2 = proj(x): enc [batch][L][in] (dtype 0 fp32 / 1 bf16) -> pooled fp32 [batch][out] (may be NULL),
 * tokens fp32 [batch][L][out1]                                                                   */
int pea_adapter_forward(void* ad, const void* enc, int dtype, float* pooled, float* tokens, void* stream);
/* grads (flat fp32, same layout as the parameters) (+)= backward of the last forward             */
int pea_adapter_backward(void* ad, const float* d_pooled, const float* d_tokens, float* grads, int accumulate,
                         void* stream);

/* The fused KD training step (train_sdxl_zh.py:311-441 downstream of the frozen encoders).       */
int pea_trainer_create(void* adapter, void* student, void* teacher, float feat_weight, int nan_guard,
                       const float* alphas_cumprod, void** out);
int pea_trainer_destroy(void* tr);
/* latents/noise fp32 [B][4][H][W]; timesteps int64 [B]; enc/enc_uncond fp32 [B][L][in]; prompt_mask uint8 [B];
 * zh_or_not int64 [B]; teacher_ehs/teacher_neg fp32 [B][Lt][cross]; teacher_pooled fp32 [B][pooled] or NULL;
 * time_ids fp32 [B][6] or NULL.  grads: flat fp32 adapter gradients; losses fp32[4] device =
 * (loss, train_loss, train_loss_logits, train_loss_features).                                    */
int pea_train_step(void* tr, const float* latents, const float* noise, const long long* timesteps,
                   const float* enc, const float* enc_uncond, const unsigned char* prompt_mask,
                   const long long* zh_or_not, const float* teacher_ehs, const float* teacher_neg,
                   const float* teacher_pooled, const float* time_ids, float grad_scale, float* grads,
                   int accumulate, float* losses, void* stream);
/* options: "two_stream" (1 = teacher forward on a side HIP stream, default), "nan_guard", "merge_passes" (default 1:
 * when the teacher context shares the student's weights -- the reference default, train_sdxl_zh.py:138,151 -- and the
 * context lengths agree, both UNet forwards run as ONE pass over 2B samples and the backward differentiates the
 * first B; otherwise the two-stream path is used);
 * "live_teacher_mask" (dead-row elimination, opt-in, merged passes at B <= 30): bit i set = sample i's teacher row is
 * computed, -1 = all (default).  CONTRACT: bit i may only be cleared for a sample whose zh_or_not[i] != 0 (its KD weight
 * 1 - zh_or_not is zero, train_sdxl_zh.py:402-441).  The mask is host state and zh_or_not is device memory, so the library
 * cannot compare them before the step; a sample that carries KD weight without a teacher row makes the loss kernel emit
 * NaN for the losses and for that sample's gradient seeds (never another sample's row). */
int pea_trainer_set_option(void* tr, const char* name, int value);
/* the UNet context whose backward pass ran in the last step (the merged-pass context when the teacher is the student
 * checkpoint, else the student's): handle for pea_unet_stacked_grad.  Owned by the trainer. */
int pea_trainer_backward_context(void* trainer, void** unet);
int pea_trainer_get_option(void* trainer, const char* name);   /* also "merge_state": 0 undecided, 1 merged, -1 n/a; "kd_samples_hint" (set: profiling only -- samples with zh_or_not == 0, for the KD-loss kernel's byte count) */
/* pea_unet_release_activations on the trainer's student, teacher and merged-pass contexts.  The reference trains over
 * nine aspect-ratio buckets (utils/custom_dataset_sdxl.py:30, one bucket per batch): a caller keeps one trainer per
 * bucket -- all sharing one set of weights and one adapter -- and releases the least recently used when HBM runs short
 * (pea_diffusion_amd/train.py: BucketedTrainer). */
int pea_trainer_release_activations(void* trainer);
/* intermediate results of the last step (fp32 NCHW [B][4][H][W]): which 0 = x_t, 1 = eps_student, 2 = eps_teacher */
int pea_trainer_export(void* tr, int which, float* out, void* stream);

/* Data-parallel collective (replaces DeepSpeed ZeRO-1's gradient all-reduce + parameter
 * all-gather, train_sdxl_zh.sh:22,87 and utils/model_utils.py:57-67): one process per GPU, ONE RCCL all-reduce (sum, then
 * x 1/world) over the flat fp32 adapter gradient per step, on a dedicated HIP stream owned by the communicator.
 *   pea_comm_unique_id: rank 0 creates the 128-byte ncclUniqueId; the caller ships it to the other ranks (TCP store / file).
 *   pea_comm_init:      BLOCKING rendezvous (ncclCommInitRank) on the calling thread's current HIP device, bounded:
 *                       pea_comm_init_timeout waits at most `timeout_s` seconds (<= 0: forever) for all `world` ranks and
 *                       returns PEA_E_TIMEOUT past that (the reference's deadline is torch.distributed.run's rendezvous,
 *                       train_sdxl_zh.sh:108-113); the rendezvous thread cannot be cancelled, so after PEA_E_TIMEOUT the process
 *                       must exit (non-zero).  pea_comm_init = the same with PEA_COMM_TIMEOUT_S from the environment (600).
 *                       A process holds ONE trainer / communicator at a time: the RCCL binding and the profiler tables are
 *                       process-global.
 *   pea_allreduce_grads: asynchronous.  The comm stream first waits for everything enqueued on `compute_stream` so far
 *                        (the adapter wgrad), then all-reduces `grads` in place and scales by 1/world.
 *   pea_comm_join:      makes `stream` wait for the last all-reduce (call before the optimizer reads `grads`).
 *   pea_comm_last_ms:   BLOCKING; device time of the last all-reduce + scale in milliseconds.
 *   pea_comm_last_exposed_ms: BLOCKING; how long the first stream that joined the last all-reduce stood still for it
 *                       (0 when the collective had finished before the stream reached the join).
 *   pea_comm_broadcast: parameter broadcast from `root` (identical replicas at start), enqueued on `stream`.      */
int pea_comm_unique_id(void* out128);
int pea_comm_init(int rank, int world, const void* unique_id128, void** comm_out);
int pea_comm_init_timeout(int rank, int world, const void* unique_id128, double timeout_s, void** comm_out);
int pea_comm_destroy(void* comm);
int pea_comm_world(void* comm);
int pea_comm_rank(void* comm);
int pea_allreduce_grads(void* comm, float* grads, long long n, void* compute_stream);
int pea_comm_join(void* comm, void* stream);
int pea_comm_last_ms(void* comm, float* ms);
int pea_comm_last_exposed_ms(void* comm, float* ms);
int pea_comm_broadcast(void* comm, float* buf, long long n, int root, void* stream);

/* per-launch HIP-event timing by kernel family (bench.py roofline leg); families 0..7:
 * gemm<plain>, gemm<conv3x3>, attn_fwd, attn_bwd, groupnorm, layernorm, elementwise, kd_loss.
 * pea_prof_report synchronises the device.                                                       */
void pea_prof_enable(int on);
void pea_prof_reset(void);
const char* pea_prof_family_name(int fam);
/* CSV of every recorded launch: family, ms, flops, bytes, shape tags (GEMM: M,N,K,epilogue flags) */
int pea_prof_dump(const char* path);
int pea_prof_report(int fam, double* ms, double* flops, double* bytes, long long* launches);
/* Sustained MFMA ceiling of this device: v_mfma_f32_16x16x32_bf16 (shape32 = 1: 32x32x16) issued back to back from registers on
 * random operands, two waves per SIMD on every CU, launches back to back for `seconds` (<= 30; default 2); *tflops = the last
 * launch's FLOP/s from HIP events, *clock_mhz = its in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over
 * the workgroups).  Synchronises the stream.  bench.py reports it as roofline.sustained_peak beside the 2.5 PFLOP/s spec peak. */
int pea_probe_mfma_peak(double seconds, int shape32, double* tflops, double* clock_mhz, void* stream);

/* debugging aid for the parity tests: 1 = ds_read_b64_tr_b16 transpose reads (default), 0 = scalar gathers */
void pea_debug_set_attn_tr(int v);
/* A/B aid: 1 = dQ and dK/dV workgroups of an attention backward in ONE launch (default), 0 = two launches */
void pea_debug_set_attn_fused_bwd(int v);
/* 0: cross-attention backward (<= 128 keys, head_dim 64) on the general kernels instead of the one-pass kernel (A/B) */
void pea_debug_set_attn_xattn(int v);
/* which one-pass cross-attention backward kernel (A/B, parity tests): 0 = the round-3 kernel for every key count; 2 = the
 * specialised-wave kernel (33..96 keys, 7 products); 1 or 3 = the newest that applies (default): the five-product kernel for
 * 33..80 keys, the specialised-wave kernel for 81..96 */
void pea_debug_set_xattn_bwd_v2(int v);
/* A/B aid: 1 = GEGLU backward inside the FF output projection's dgrad GEMM (default), 0 = its own kernel */
void pea_debug_set_geglu_bwd_fused(int v);
/* benchmark aid: force GEMM tile variant (>= 0) or restore the shape-based choice (-1) */
void pea_debug_set_gemm_variant(int v);
/* timing-only probes of the loader/consumer GEMM (results are wrong while set): 1 no DMA, 2 no barriers, 4 no ds_reads */
void pea_debug_set_gemm_debug(int v);
/* experiment aid (scripts/chain_probe.py): arms the NEXT GEMM / conv launch with a prefetch target -- its DMA waves touch
 * [p, p + bytes) behind their last K-step (the product's tapes set the next launch's weight matrix themselves: GemmP::pf_ptr) */
void pea_debug_set_gemm_prefetch(const void* p, long long bytes);

#ifdef __cplusplus
}
#endif
#endif /* PEA_HIP_H */
