// Host-side launch interface of every HIP kernel family (one translation unit each).
// All launchers enqueue on `stream`, never synchronise, and return PEA_OK / PEA_E_*.
#pragma once
#include "pea_common.h"

// ---------------------------------------------------------------- gemm.hip
// C[M][N] = epi( alpha * A[M][K] . W[N][K]^T ),  bf16 operands, fp32 accumulate (MFMA 32x32x16).
// mode 0: A is a plain row-major matrix (lda).  mode 1: A is the im2col view of an NHWC
// tensor for a 3x3 convolution (implicit GEMM), K = 9*Cin ordered (ky,kx,ci).
struct GemmP {
  int mode;
  const bf16* A; int lda;
  const bf16* W; int ldw;          // W[N][ldw]
  void* C; int ldc; int out_f32;   // bf16 (default) or fp32 output; accum_f32: C(fp32) += result
  int accum_f32;
  int M, N, K;
  float alpha;
  const float* bias;               // [N] fp32 or null
  const bf16* rowvec; int ldrv; int rows_per_batch;   // + rowvec[m / rows_per_batch][n]
  int act;                         // 0 none, 1 GELU(erf), 2 SiLU, 3 quick-GELU x*sigmoid(1.702x)  (after bias/rowvec)
  bf16* preact; int ldpre;         // optional: store the pre-activation value (bf16)
  const bf16* res; int ldres;      // + res[m][n] after activation (may alias C: accumulate)
  // conv (mode 1): source NHWC [B][Hs][Ws][Cin]; output pixels [B][Ho][Wo]; M = B*Ho*Wo
  int Hs, Ws, Cin, Ho, Wo;
  int stride;                      // output -> virtual-input coordinate multiplier (1 or 2)
  int pad_off;                     // 0: padding 1 on every side; 1: the VAE downsampler's (0,1,0,1) padding
  int shift;                       // virtual input = source upsampled by 2^shift (nearest) / zero-stuffed
  int parity;                      // 1: only even virtual coordinates are real (transposed stride-2 conv)
  int kside;                       // tap window: 0 / 3 = 3 x 3;  2 = 2 x 2 (K = 4 Cin);  4 = 16 taps over a depth-to-space source (gemm.hip: conv_tap)
  int pad_dx;                      // extra x offset of the tap window (pad_off moves both axes)
  int pix;                         // elements per source pixel (0 = Cin)
  const bf16* zeros;               // >= 16 bytes of zeros (out-of-bounds taps)
  int debug;                       // timing experiments only (scripts/gemm_loop_probe.py)
  // fused GEGLU: the weight rows are interleaved (h_i, gate_i) so a lane's 4 consecutive columns are two pairs;
  // geglu_y[m][n/2] = h * gelu(gate).  C may be null then (no pre-activation stash: teacher / inference).
  bf16* geglu_y; int ldy;
  int geglu_tanh;                  // GEGLU gate activation: 0 = GELU(erf) (diffusers GEGLU), 1 = gelu_new / tanh form (T5 v1.1 gated-gelu)
  int stash_rows;                  // GEGLU with a stash: rows >= stash_rows (> 0) skip the C store (merged passes: teacher rows)
  int stash_grad;                  // 1: the stash holds what the BACKWARD multiplies by, (gelu(gate), h * gelu'(gate)), instead of (h, gate):
                                   // nothing else ever reads it (no weight gradients), the forward has erf and exp(-g^2/2) at hand
                                   // anyway, and the backward epilogue becomes two multiplies per pair instead of an erf evaluation
  int ksplit; long long split_stride;   // split-K: fp32 partial s is written at C + s*split_stride (then launch_splitk_reduce)
  int epi_fast;                    // set by launch_gemm: the batched-load epilogue (gemm_epilogue16_fast) applies
  // folded LayerNorm on the A operand (A = the UN-normalised rows, W = W' = W.gamma, bias = t):
  //   value = rstd[m] * (acc - mean[m] * ln_s[n]) + bias[n];  ln_stats = [M][2] (mean, rstd).  Needs epi_fast.
  const float* ln_stats; const float* ln_s;
  // GEGLU backward fused into this (dgrad) GEMM: the result is d y[m][n] of y = h * gelu(gate); gbwd_pre[m][2n], [2n+1] are
  // the stashed (h, gate); C becomes [M][2N] (ldc >= 2N): C[m][2n] = dy * gelu(gate), C[m][2n+1] = dy * h * gelu'(gate).
  // Needs epi_fast (bf16 output, no activation / residual / row vector).
  const bf16* gbwd_pre; int ldgp;
  // Column-range output scale: columns n < qscale_cols are multiplied by qscale after alpha, bias and the row vector (n-tile
  // granular: qscale_cols % 16 == 0).  The fused Q|K|V (or to_q) projection hands the attention kernels Q already multiplied by
  // softmax_scale * log2(e) -- ONE rounding from the fp32 accumulator -- so their scores leave the MFMA in the log2 domain.
  int qscale_cols; float qscale;
  int gbwd_form;                   // 0: gbwd_pre = (h, gate);  1: gbwd_pre = (gelu(gate), h * gelu'(gate))  (a stash_grad forward)
  // Next-op weight prefetch: once a workgroup's DMA waves have issued their last K-step they touch their share of
  // [pf_ptr, pf_ptr + pf_bytes) -- the weight matrix of the NEXT GEMM / conv on the stream -- one 4-byte load per 128-byte line,
  // so that it sits in the 256 MiB Infinity Cache when that launch's cold prologue asks for it (the 2.57 B frozen weights are read
  // once per pass: every launch otherwise streams its panel from HBM in 128-byte row pieces).  null / 0 = none.
  const void* pf_ptr; long long pf_bytes;
};
int launch_splitk_reduce(const float* part, int nsplit, long long stride, bf16* out, int ldo, int M, int N, int accum,
                         hipStream_t s);
int launch_gemm(const GemmP& p, hipStream_t stream);

// direct (non-MFMA) 3x3 convs for the 4-channel ends of the UNet
// conv_in:  x NCHW fp32 [B][Cin][H][W] -> y NHWC bf16 [B][H][W][Cout];  w [Cout][Cin][3][3] fp32
int launch_conv_in(const float* x, const float* w, const float* bias, bf16* y, int B, int Cin, int H, int W,
                   int Cout, hipStream_t s, int ldy = 0, int silu = 0);   // ldy: output row stride (0 = Cout)
// conv_out: x NHWC bf16 [B][H][W][Cin] -> y NCHW fp32 [B][Cout][H][W];  w [Cout][3][3][Cin] fp32
int launch_conv_out(const bf16* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                    int Cout, hipStream_t s);
// dgrad of conv_out: dy NCHW fp32 [B][Cout][H][W] -> dx NHWC bf16 [B][H][W][Cin]
int launch_conv_out_dgrad(const float* dy, const float* w, bf16* dx, int B, int Cin, int H, int W, int Cout,
                          hipStream_t s);

// ---------------------------------------------------------------- norm.hip
// GroupNorm over NHWC bf16 [B][HW][C], `groups` groups; stats fp32 [B][groups][2] = (mean, rstd)
int launch_groupnorm_fwd(const bf16* x, const float* gamma, const float* beta, bf16* y, float* stats,
                         double* scratch, int B, int HW, int C, int groups, float eps, int silu, hipStream_t s);
// dx (+= if accum) for y = [silu](GN(x)); gamma/beta frozen
int launch_groupnorm_bwd(const bf16* x, const bf16* dy, const float* gamma, const float* beta, const float* stats,
                         bf16* dx, double* scratch, int B, int HW, int C, int groups, int silu, const bf16* add,
                         hipStream_t s);   // add: optional addend (may be dx itself)
// LayerNorm over rows [R][C] (C % 8 == 0, C <= 4096); stats fp32 [R][2] = (mean, rstd)
int launch_layernorm_fwd(const bf16* x, const float* gamma, const float* beta, bf16* y, float* stats, int R, int C,
                         float eps, hipStream_t s);
// (mean, rstd) per row only; and the LayerNorm -> Linear fold (norm.hip: ln_fold_kernel)
int launch_rmsnorm_fwd(const bf16* x, const float* gamma, bf16* y, int R, int C, float eps, hipStream_t s);
int launch_layernorm_stats(const bf16* x, float* stats, int R, int C, float eps, hipStream_t s);
int launch_ln_fold(const bf16* W, int ldw, const float* gamma, const float* beta, const float* bias, bf16* Wf, float* svec,
                   float* tvec, int N, int K, hipStream_t s);
int launch_layernorm_bwd(const bf16* x, const bf16* dy, const float* gamma, const float* stats, bf16* dx,
                         float* dgamma, float* dbeta, int R, int C, const bf16* add, hipStream_t s);   // add: optional addend (may be dx)

// ---------------------------------------------------------------- attention.hip
// O[b][q][h*D+d] = softmax(scale * Q K^T) V per (b, head); D = 64*nd (nd = 1, 2, 3); heads whose true width is
// not a multiple of 64 are stored zero-padded (the scale stays that of the true width).
// Q rows stride ldq elements, batch stride = Sq*ldq (same for K,V with Skv, ldk/ldv).
struct AttnP {
  const bf16 *Q, *K, *V; int ldq, ldk, ldv;
  bf16* O; int ldo;
  float* lse;                      // [B][H][Sq]  (natural-log-sum-exp of scaled scores), may be null
  int B, H, Sq, Skv;
  float scale;
  // backward
  const bf16* dO; int lddo;
  bf16 *dQ, *dK, *dV; int lddq, lddk, lddv;
  float* delta;                    // [2][B][H][Sq] scratch: -rowsum(dO * O) and -lse * log2(e)
  int accum_dq, accum_dkv;         // += into existing gradients
  float* dkv_part; int nsplit;     // optional fp32 scratch (attention_bwd_scratch_bytes) enabling the query split
  int nd;                          // padded head_dim / 64 (0 is read as 1)
  int xcd_remap;                   // set by the launchers: XCD-aware workgroup order
  int causal;                      // forward only: key index <= query index (text encoders)
  const int* kv_len;               // per-sample number of valid keys (key padding mask; forward and backward), may be null
  int q_prescaled;                 // Q already holds q * scale * log2(e) (GemmP::qscale in the producing projection): scores leave the
                                   // MFMA in the log2 domain with no further rounding; dQ is still the gradient w.r.t. the UNSCALED q
  const float* bias;               // forward, masked instance only: additive score bias [H][Sq][Skv] in the LOG2 domain (T5 relative positions), may be null
  // Deferred split reduce (cross-attention backward, xattn_bwd2_kernel).  defer_reduce = 1: this launch leaves its fp32 dK / dV
  // partials in dkv_part and does NOT launch attn_dkv_reduce_kernel; the caller hands them to a LATER launch (red_* below,
  // filled by attention_set_deferred) whose workgroups add them up while their own first tiles are in flight, or flushes them
  // with launch_attention_dkv_reduce.  The splits are added in the same fixed order either way.
  int defer_reduce;
  const float* red_part; bf16 *red_dK, *red_dV; int red_lddk, red_lddv, red_nsplit, red_B, red_H, red_Skv, red_accum;
};
int attention_bwd_nsplit(int B, int H, int Sq, int Skv);
int attention_bwd_defers(const AttnP& p);                                   // 1: launch_attention_bwd honours p.defer_reduce for this problem
void attention_set_deferred(AttnP& cur, const AttnP& pending);              // cur's launch reduces pending's partials
int launch_attention_dkv_reduce(const AttnP& pending, hipStream_t s);       // flush: the stand-alone reduce of a deferred launch
size_t attention_bwd_scratch_bytes(int B, int H, int Sq, int Skv, int nd = 1);
int launch_attention_fwd(const AttnP& p, hipStream_t s);
int launch_attention_bwd(const AttnP& p, hipStream_t s);

// ---------------------------------------------------------------- elementwise.hip
int launch_geglu_fwd(const bf16* hg, bf16* y, long long rows, int inner, hipStream_t s);   // hg [rows][2*inner]
int launch_geglu_bwd(const bf16* hg, const bf16* dy, bf16* dhg, long long rows, int inner, hipStream_t s);
int launch_add(const bf16* a, const bf16* b, bf16* y, long long n, hipStream_t s);          // y = a + b
int launch_silu_fwd(const bf16* x, bf16* y, long long n, hipStream_t s);
int launch_silu_bwd(const bf16* x, const bf16* dy, bf16* dx, long long n, int accum, hipStream_t s);
int launch_gelu_fwd(const bf16* x, bf16* y, long long n, hipStream_t s);
int launch_gelu_bwd(const bf16* x, const bf16* dy, bf16* dx, long long n, int accum, hipStream_t s);
// concat along channels: y[r][0:C1] = a[r], y[r][C1:C1+C2] = b[r]
// aH, aW != 0: the FIRST operand is stored depth-to-space at full resolution aH x aW (elementwise.hip: d2s_row)
int launch_concat2(const bf16* a, int C1, const bf16* b, int C2, bf16* y, long long rows, hipStream_t s, int aH = 0, int aW = 0);
// split-add (backward of concat): da[r] (+)= dy[r][0:C1]; db[r] (+)= dy[r][C1:]
int launch_split2(const bf16* dy, int C1, int C2, bf16* da, int accum_a, bf16* db, int accum_b, long long rows,
                  hipStream_t s, int aH = 0, int aW = 0);
// 2x2 sum pooling NHWC (backward of nearest 2x upsample): x [B][2H][2W][C] -> y [B][H][W][C]
int launch_sumpool2(const bf16* x, bf16* y, int B, int H, int W, int C, int accum, hipStream_t s);
int launch_cast_f32_bf16(const float* x, bf16* y, long long n, hipStream_t s);
int launch_cast_bf16_f32(const bf16* x, float* y, long long n, hipStream_t s);
int launch_transpose_bf16(const bf16* x, bf16* y, int R, int C, int ldy, hipStream_t s);   // y[c][r] = x[r][c]
int launch_transpose_f32_bf16(const float* x, bf16* y, int R, int C, int ldy, hipStream_t s);
int launch_nhwc_to_nchw_f32(const bf16* x, float* y, int B, int HW, int C, hipStream_t s, int dH = 0, int dW = 0);   // dH, dW: depth-to-space source
int launch_nchw_f32_to_nhwc(const float* x, bf16* y, int B, int HW, int C, hipStream_t s, int dH = 0, int dW = 0);
// conv weight repacks (fp32 torch layout [Co][Ci][3][3]) -> bf16
int launch_pack_conv_fwd(const float* w, bf16* y, int Co, int Ci, hipStream_t s, int Cip = 0);   // Cip: stored (zero-padded) Ci          // y[co][(ky,kx,ci)]
int launch_pack_conv_dgrad(const float* w, bf16* y, int Co, int Ci, hipStream_t s);        // y[ci][(2-ky,2-kx,co)]
int launch_pack_conv_out(const float* w, float* y, int Co, int Ci, hipStream_t s);         // y[co][ky][kx][ci] fp32
// sub-pixel form of conv3x3(nearest_2x(x)): forward y[4][Co][4 Ci], dgrad y[Ci][16 Co]  (elementwise.hip)
int launch_pack_conv_subpix(const float* w, bf16* y, int Co, int Ci, int dgrad, hipStream_t s);
// sinusoidal timestep embedding (cos|sin), out bf16 [n][dim]; t given as fp32 values
int launch_timestep_embed(const float* t, bf16* y, int n, int dim, hipStream_t s);
// x_t = sqrt(ac[t]) x0 + sqrt(1-ac[t]) eps   (fp32 NCHW), ac table fp32[1000]
int launch_add_noise(const float* x0, const float* eps, const long long* t, const float* ac, float* xt, int B,
                     long long per, hipStream_t s);
// y[b] = mask[b] ? u[b] : c[b]  rows of `per` bf16
// (ystride: elements between consecutive samples of y, 0 = per; > per leaves each sample's tail untouched)
int launch_select_rows(const bf16* c, const bf16* u, const unsigned char* mask, bf16* y, int B, long long per,
                       hipStream_t s, long long ystride = 0);
// backward of select: dc[b] = mask? 0 : dy[b]; du[b] = mask ? dy[b] : 0
int launch_select_rows_bwd(const bf16* dy, const unsigned char* mask, bf16* dc, bf16* du, int B, long long per,
                           hipStream_t s, long long dystride = 0);
// mean over tokens: y[b][c] = mean_l x[b][l][c];  bwd: dx[b][l][c] (+)= dy[b][c]/L
int launch_mean_tokens(const bf16* x, bf16* y, int B, int L, int C, hipStream_t s);
int launch_mean_tokens_bwd(const bf16* dy, bf16* dx, int B, int L, int C, int accum, hipStream_t s);
// column sums of a bf16 matrix into fp32 (bias gradient): db[c] (+)= sum_r x[r][c]
int launch_colsum(const bf16* x, float* db, int R, int C, int accum, hipStream_t s);
int launch_fill_random_bf16(bf16* p, long long n, unsigned long long seed, float scale, hipStream_t s);
int launch_fill_random_f32(float* p, long long n, unsigned long long seed, float scale, float offset, hipStream_t s);
// fused AdamW over a flat fp32 buffer
int launch_adamw(float* w, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                 float wd, int step, float gscale, hipStream_t s);

// ---------------------------------------------------------------- kdloss.hip
// Fused KD loss (train_sdxl_zh.py:399-441): all feature taps + the two noise terms in ONE launch.
#define PEA_MAX_TAPS 12
struct KdLossP {
  int ntaps;
  const bf16* fs[PEA_MAX_TAPS];    // student taps (any layout, elementwise-paired with ft)
  const bf16* ft[PEA_MAX_TAPS];
  bf16* dfs[PEA_MAX_TAPS];         // gradient seeds dL/dF_S (overwritten), may be null
  long long per[PEA_MAX_TAPS];     // elements per sample
  const float *eps_s, *eps, *eps_t;  // fp32 [B][per_eps]
  float* deps_s;                   // dL/d eps_s fp32, may be null
  long long per_eps;
  const long long* zh;             // [B] int64 (1 = native caption)
  int B;
  float feat_weight;               // 0.1
  float* partial;                  // [3 + ntaps] fp32 sums (zeroed by the launcher)
  float* losses;                   // [4]: total, noise, logits, features (written by a finishing kernel)
  const int* tap_finite_flags;     // unused (SD1.5 NaN guard handled through `nan_guard`)
  int nan_guard;
  float grad_scale;                // multiplies every seed (1/world for DP averaging, usually 1)
  const int* tmap;                 // optional device int[B]: teacher sample index of student sample b inside ft[] / eps_t (compacted
                                   // teacher rows, Trainer::live_teacher_mask); entries of samples with zh_or_not != 0 are not read
  int kd_samples_hint;             // profiling only: samples with zh_or_not == 0 (their taps are READ; the others' seeds are only
                                   // written as zeros) when the host knows it, else -1 = count every sample as read
};
int launch_kd_loss(const KdLossP& p, hipStream_t s);

// late additions (elementwise.hip)
int launch_colsum_batched(const bf16* x, float* out, int B, int HW, int C, int ldo, float* scratch, hipStream_t s);
size_t colsum_batched_scratch_bytes(int B, int HW, int C);
size_t groupnorm_scratch_bytes(int B, int HW, int C, int groups);
size_t kd_loss_workspace_bytes(int ntaps, const long long* per, long long per_eps, int B);
int launch_accum(const bf16* x, bf16* y, long long n, int accum, hipStream_t s);
int launch_copy2d(const bf16* x, int ldx, bf16* y, int ldy, long long rows, int C, int accum, hipStream_t s);
int launch_cast_i64_f32(const long long* x, float* y, long long n, hipStream_t s);

// ---------------------------------------------------------------- prof.hip
#define PEA_PROF_FAMILIES 8
extern int g_prof_on;
extern int g_prof_tag[4];
void prof_begin_impl(int fam, double flops, double bytes, hipStream_t s);
void prof_end_impl(hipStream_t s);
#define PROF_BEGIN(fam, flops, bytes, s) do { if (g_prof_on) prof_begin_impl(fam, flops, bytes, s); } while (0)
#define PROF_END(s) do { if (g_prof_on) prof_end_impl(s); } while (0)
int launch_prefetch(const void* p, long long bytes, int* sink, hipStream_t s);
int launch_pad_gather(const float* src, int N_t, int K_t, int mode, int d, int dp, bf16* w, int ldw, bf16* wt, int ldwt,
                      int st_n, int st_k, hipStream_t s);
int launch_geglu_bwd_il(const bf16* hg, const bf16* dy, bf16* dhg, long long rows, int inner, hipStream_t s, int form = 0);
int launch_geglu_fwd_il(const bf16* hg, bf16* y, long long rows, int inner, hipStream_t s);
int launch_permute_geglu_vec(const float* src, float* dst, int inner, hipStream_t s);
// sampler.hip: inference denoise-loop glue
size_t cfg_combine_workspace_bytes(int B);
int launch_cfg_combine(const float* eps2, float* out, int B, long long per, float g, float rescale, void* ws, hipStream_t s);
int launch_dpm_update(float* sample, const float* eps, float* x0_prev, long long n, float alpha_s, float sigma_s,
                      float c_s, float c_0, float c_1, hipStream_t s);
int launch_residual_import(const void* src, int dtype, bf16* dst, int B, int C, long long HW, float scale, hipStream_t s);
int launch_softmax_rows(bf16* s, long long rows, int cols, int ld, float scale, hipStream_t st);   // in place
int launch_vae_posterior(const float* h, const float* wq, const float* bq, const float* noise, float* moments,
                         float* latents, int B, int C2, long long HW, float scaling, hipStream_t s);
int launch_vae_post_quant(const float* z, const float* w, const float* b, float* out, int B, int C, long long HW,
                          float inv_scaling, hipStream_t s);
int launch_embed_tokens(const long long* ids, const bf16* tok, const bf16* pos, const bf16* type0, bf16* out, int B, int L,
                        int width, int vocab, hipStream_t s);
int launch_t5_rel_bias(const float* rel, const int* dist_bucket, float* bias, int H, int L, int pitch, int half_buckets, hipStream_t s);
int launch_gather_eos(const long long* ids, const bf16* x, bf16* out, int B, int L, int width, long long eos_id, hipStream_t s);
int launch_kv_len(const long long* ids, int* len, int B, int L, long long pad_id, hipStream_t s);
