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
int pea_op_conv_in(const float* x_nchw, const float* w, const float* bias, void* y_nhwc, int B, int Cin, int H,
                   int W, int Cout, void* stream);
int pea_op_conv_out(const void* x_nhwc, const float* w_packed, const float* bias, float* y_nchw, int B, int Cin,
                    int H, int W, int Cout, void* stream);
int pea_op_conv_out_dgrad(const float* dy_nchw, const float* w_packed, void* dx_nhwc, int B, int Cin, int H, int W,
                          int Cout, void* stream);
int pea_op_pack_conv_out(const float* w, float* out, int Co, int Ci, void* stream);

/* GroupNorm (+ optional SiLU) over x[B][HW][C] bf16; stats fp32 [B][groups][2]; scratch >= 16*B*groups bytes */
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

/* softmax(scale Q K^T) V, head_dim 64 (diffusers AttnProcessor2_0 -> SDPA).  Q/K/V/O bf16 with
 * row strides ld* (elements); head h occupies columns [64h, 64h+64).  lse fp32 [B][H][Sq].        */
int pea_op_attention_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                         float* lse, int B, int H, int Sq, int Skv, float scale, void* stream);
/* dQ/dK/dV (any may be NULL... dK and dV together); delta: fp32 scratch [B][H][Sq] */
int pea_op_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                         int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                         void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                         int accum_dq, int accum_dkv, void* stream);

int pea_op_geglu_fwd(const void* hg, void* y, long long rows, int inner, void* stream);
int pea_op_geglu_bwd(const void* hg, const void* dy, void* dhg, long long rows, int inner, void* stream);
int pea_op_sumpool2(const void* x, void* y, int B, int H, int W, int C, int accum, void* stream);
int pea_op_timestep_embed(const float* t, void* y, int n, int dim, void* stream);
/* DDPM add_noise (train_sdxl_zh.py:322); ac = alphas_cumprod fp32[1000] on device */
int pea_op_add_noise(const float* x0, const float* eps, const long long* t, const float* ac, float* xt, int B,
                     long long per, void* stream);
int pea_op_cast_f32_bf16(const float* x, void* y, long long n, void* stream);
int pea_op_cast_bf16_f32(const void* x, float* y, long long n, void* stream);

/* Fused KD loss of train_sdxl_zh.py:399-441 (SD1.5: train_sd_zh.py:217-276, nan_guard=1).
 * taps_s/taps_t/dtaps: HOST arrays of ntaps device pointers (bf16, elementwise-paired layouts);
 * per: HOST array of per-sample element counts.  eps_*: fp32 [B][per_eps].  zh: int64 [B] device.
 * losses: fp32[4] device = (loss, train_loss, train_loss_logits, train_loss_features).
 * dtaps[k] / deps_s receive dL/d(student tap) / dL/d(eps_s) (may be NULL).
 * workspace: >= 256 bytes of device scratch.                                                       */
int pea_op_kd_loss(int ntaps, const void* const* taps_s, const void* const* taps_t, void* const* dtaps,
                   const long long* per, const float* eps_s, const float* eps, const float* eps_t, float* deps_s,
                   long long per_eps, const long long* zh, int B, float feat_weight, int nan_guard,
                   float grad_scale, float* losses, void* workspace, void* stream);

/* fused AdamW over a flat fp32 buffer (DeepSpeed FusedAdam(adam_w_mode=True), utils/model_utils.py:64-67) */
int pea_op_adamw(float* w, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, void* stream);

/* debugging aid for the parity tests: 1 = ds_read_b64_tr_b16 transpose reads (default), 0 = scalar gathers */
void pea_debug_set_attn_tr(int v);

#ifdef __cplusplus
}
#endif
#endif /* PEA_HIP_H */
