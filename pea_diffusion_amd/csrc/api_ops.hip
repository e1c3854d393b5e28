// C-ABI, operator level: thin argument marshalling onto the kernel launchers (include/pea_hip.h).
#include <stdarg.h>
#include <string.h>

#include "../../include/pea_hip.h"
#include "pea_kernels.h"

static thread_local char g_err[512] = "";
void pea_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static bf16* g_zero_page = nullptr;
int pea_zero_page(const bf16** out) {
  if (!g_zero_page) {
    HIPCHK(hipMalloc((void**)&g_zero_page, 256));
    HIPCHK(hipMemset(g_zero_page, 0, 256));
  }
  *out = g_zero_page;
  return PEA_OK;
}

// bodies of the attention operators with the Q convention as a PARAMETER (the entry points below pass 0 or 1): nothing
// process-wide is written around a call, so concurrent callers (ctypes releases the GIL) cannot see each other's mode
static int attention_fwd_impl(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                              float* lse, int B, int H, int Sq, int Skv, float scale, int nd, int q_prescaled, void* stream) {
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.O = (bf16*)O; p.ldo = ldo; p.lse = lse; p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.scale = scale; p.nd = nd;
  p.q_prescaled = q_prescaled;
  return launch_attention_fwd(p, (hipStream_t)stream);
}
static int attention_bwd_impl(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                              int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                              void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                              int accum_dq, int accum_dkv, int nd, void* scratch, int q_prescaled, void* stream) {
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.O = (bf16*)O; p.ldo = ldo; p.lse = (float*)lse; p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.scale = scale;
  p.dO = (const bf16*)dO; p.lddo = lddo; p.delta = delta;
  p.dQ = (bf16*)dQ; p.lddq = lddq; p.dK = (bf16*)dK; p.lddk = lddk; p.dV = (bf16*)dV; p.lddv = lddv;
  p.accum_dq = accum_dq; p.accum_dkv = accum_dkv; p.dkv_part = (float*)scratch; p.nd = nd;
  p.q_prescaled = q_prescaled;
  return launch_attention_bwd(p, (hipStream_t)stream);
}

extern "C" {

const char* pea_last_error(void) { return g_err; }
int pea_version(void) { return 100; }
int pea_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pea_op_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, float alpha,
                const float* bias, const void* rowvec, int ldrv, int rows_per_batch, int act, void* preact,
                int ldpre, const void* res, int ldres, int out_f32, int accum_f32, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.mode = 0;
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.bias = bias;
  p.rowvec = (const bf16*)rowvec; p.ldrv = ldrv; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
  p.act = act; p.preact = (bf16*)preact; p.ldpre = ldpre; p.res = (const bf16*)res; p.ldres = ldres;
  p.out_f32 = out_f32; p.accum_f32 = accum_f32;
  return launch_gemm(p, (hipStream_t)stream);
}

int pea_op_gemm_qscale(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* bias,
                       int qscale_cols, float qscale, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.bias = bias; p.rows_per_batch = 1;
  p.qscale_cols = qscale_cols; p.qscale = qscale;
  return launch_gemm(p, (hipStream_t)stream);
}

int pea_op_gemm_geglu_bwd(const void* A, int lda, const void* W, int ldw, const void* pre, int ldpre, void* C, int ldc,
                          int M, int N, int K, int form, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.rows_per_batch = 1;
  p.gbwd_pre = (const bf16*)pre; p.ldgp = ldpre; p.gbwd_form = form;
  return launch_gemm(p, (hipStream_t)stream);
}

int pea_op_gemm_geglu(const void* A, int lda, const void* W, int ldw, const float* bias, void* y, void* stash, int M, int N,
                      int K, int stash_grad, int stash_rows, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.rows_per_batch = 1;
  p.geglu_y = (bf16*)y; p.ldy = N / 2; p.C = stash; p.ldc = N; p.stash_grad = stash_grad; p.stash_rows = stash_rows;
  return launch_gemm(p, (hipStream_t)stream);
}

int pea_op_ln_linear(const void* x, const float* gamma, const float* beta, const void* W, const float* bias, void* y,
                     void* geglu_y, int M, int N, int K, float eps, void* Wf, float* svec, float* tvec, float* stats,
                     void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_layernorm_stats((const bf16*)x, stats, M, K, eps, st);
  if (rc) return rc;
  rc = launch_ln_fold((const bf16*)W, K, gamma, beta, bias, (bf16*)Wf, svec, tvec, N, K, st);
  if (rc) return rc;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)x; p.lda = K; p.W = (const bf16*)Wf; p.ldw = K; p.M = M; p.N = N; p.K = K; p.alpha = 1.f;
  p.bias = tvec; p.rows_per_batch = 1; p.ln_stats = stats; p.ln_s = svec;
  if (geglu_y) { p.geglu_y = (bf16*)geglu_y; p.ldy = N / 2; p.C = y; p.ldc = N; }
  else { p.C = y; p.ldc = N; }
  return launch_gemm(p, st);
}

int pea_op_conv3x3(const void* x, const void* w, void* y, int B, int Hs, int Ws, int Cin, int Cout, int stride,
                   int upsample2x, int transposed2, const float* bias, const void* rowvec, int ldrv,
                   const void* res, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.mode = 1;
  p.A = (const bf16*)x; p.W = (const bf16*)w; p.ldw = 9 * Cin; p.C = y; p.ldc = Cout;
  p.Hs = Hs; p.Ws = Ws; p.Cin = Cin;
  p.shift = (upsample2x || transposed2) ? 1 : 0;
  p.parity = transposed2 ? 1 : 0;
  p.stride = stride;
  const int Hv = Hs << p.shift, Wv = Ws << p.shift;
  p.Ho = stride == 2 ? (Hv + 1) / 2 : Hv;
  p.Wo = stride == 2 ? (Wv + 1) / 2 : Wv;
  p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin; p.alpha = 1.f; p.bias = bias;
  p.rowvec = (const bf16*)rowvec; p.ldrv = ldrv; p.rows_per_batch = p.Ho * p.Wo;
  p.res = (const bf16*)res; p.ldres = Cout;
  int rc = pea_zero_page(&p.zeros);
  if (rc) return rc;
  return launch_gemm(p, (hipStream_t)stream);
}

// conv3x3(nearest_2x(x)) in its sub-pixel form (elementwise.hip: pack_conv_subpix_kernel; the tape's OP_CONV3 p1 == 2):
// x [B][Hs][Ws][Cin] -> y depth-to-space [B][Hs][Ws][4][Cout];  w = pea_op_pack_conv_subpixel(dgrad = 0)
int pea_op_upconv_subpixel(const void* x, const void* w, void* y, int B, int Hs, int Ws, int Cin, int Cout,
                           const float* bias, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.mode = 1; p.A = (const bf16*)x; p.ldw = 4 * Cin; p.ldc = 4 * Cout;
  p.Hs = Hs; p.Ws = Ws; p.Cin = Cin; p.Ho = Hs; p.Wo = Ws; p.stride = 1; p.kside = 2;
  p.M = B * Hs * Ws; p.N = Cout; p.K = 4 * Cin; p.alpha = 1.f; p.bias = bias; p.rows_per_batch = Hs * Ws;
  int rc = pea_zero_page(&p.zeros);
  if (rc) return rc;
  for (int pl = 0; pl < 4; ++pl) {
    p.W = (const bf16*)w + (size_t)pl * Cout * 4 * Cin;
    p.C = (bf16*)y + (size_t)pl * Cout;
    p.pad_off = pl >> 1; p.pad_dx = (pl & 1) - (pl >> 1);
    rc = launch_gemm(p, (hipStream_t)stream);
    if (rc) return rc;
  }
  return PEA_OK;
}
// its data gradient: dy depth-to-space [B][Hs][Ws][4][Cout] -> dx [B][Hs][Ws][Cin] (+ res);  wt = pea_op_pack_conv_subpixel(dgrad = 1)
int pea_op_upconv_subpixel_dgrad(const void* dy, const void* wt, void* dx, int B, int Hs, int Ws, int Cin, int Cout,
                                 const void* res, void* stream) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.mode = 1; p.A = (const bf16*)dy; p.W = (const bf16*)wt; p.ldw = 16 * Cout; p.C = dx; p.ldc = Cin;
  p.Hs = Hs; p.Ws = Ws; p.Cin = Cout; p.pix = 4 * Cout; p.kside = 4; p.Ho = Hs; p.Wo = Ws; p.stride = 1; p.pad_off = 1;
  p.M = B * Hs * Ws; p.N = Cin; p.K = 16 * Cout; p.alpha = 1.f; p.rows_per_batch = Hs * Ws;
  p.res = (const bf16*)res; p.ldres = Cin;
  int rc = pea_zero_page(&p.zeros);
  if (rc) return rc;
  return launch_gemm(p, (hipStream_t)stream);
}
// channel concat / its backward; aH, aW != 0: the FIRST operand is stored depth-to-space at full resolution aH x aW
int pea_op_concat2(const void* a, int C1, const void* b, int C2, void* y, long long rows, int aH, int aW, void* stream) {
  return launch_concat2((const bf16*)a, C1, (const bf16*)b, C2, (bf16*)y, rows, (hipStream_t)stream, aH, aW);
}
int pea_op_split2(const void* dy, int C1, int C2, void* da, int accum_a, void* db, int accum_b, long long rows, int aH,
                  int aW, void* stream) {
  return launch_split2((const bf16*)dy, C1, C2, (bf16*)da, accum_a, (bf16*)db, accum_b, rows, (hipStream_t)stream, aH, aW);
}
int pea_op_pack_conv_subpixel(const float* w, void* out, int Co, int Ci, int dgrad, void* stream) {
  return launch_pack_conv_subpix(w, (bf16*)out, Co, Ci, dgrad, (hipStream_t)stream);
}

int pea_op_pack_conv(const float* w, void* out, int Co, int Ci, int dgrad, void* stream) {
  return dgrad ? launch_pack_conv_dgrad(w, (bf16*)out, Co, Ci, (hipStream_t)stream)
               : launch_pack_conv_fwd(w, (bf16*)out, Co, Ci, (hipStream_t)stream);
}
int pea_op_pack_conv_out(const float* w, float* out, int Co, int Ci, void* stream) {
  return launch_pack_conv_out(w, out, Co, Ci, (hipStream_t)stream);
}
int pea_op_conv_in(const float* x, const float* w, const float* bias, void* y, int B, int Cin, int H, int W,
                   int Cout, void* stream) {
  return launch_conv_in(x, w, bias, (bf16*)y, B, Cin, H, W, Cout, (hipStream_t)stream);
}
int pea_op_conv_out(const void* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                    int Cout, void* stream) {
  return launch_conv_out((const bf16*)x, w, bias, y, B, Cin, H, W, Cout, (hipStream_t)stream);
}
int pea_op_conv_out_dgrad(const float* dy, const float* w, void* dx, int B, int Cin, int H, int W, int Cout,
                          void* stream) {
  return launch_conv_out_dgrad(dy, w, (bf16*)dx, B, Cin, H, W, Cout, (hipStream_t)stream);
}

int pea_op_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, void* scratch,
                         int B, int HW, int C, int groups, float eps, int silu, void* stream) {
  return launch_groupnorm_fwd((const bf16*)x, gamma, beta, (bf16*)y, stats, (double*)scratch, B, HW, C, groups, eps,
                              silu, (hipStream_t)stream);
}
int pea_op_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats,
                         void* dx, void* scratch, int B, int HW, int C, int groups, int silu, int accum,
                         void* stream) {
  return launch_groupnorm_bwd((const bf16*)x, (const bf16*)dy, gamma, beta, stats, (bf16*)dx, (double*)scratch, B, HW,
                              C, groups, silu, accum ? (const bf16*)dx : nullptr, (hipStream_t)stream);
}
int pea_op_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int R, int C,
                         float eps, void* stream) {
  return launch_layernorm_fwd((const bf16*)x, gamma, beta, (bf16*)y, stats, R, C, eps, (hipStream_t)stream);
}
int pea_op_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* stats, void* dx,
                         float* dgamma, float* dbeta, int R, int C, int accum, void* stream) {
  return launch_layernorm_bwd((const bf16*)x, (const bf16*)dy, gamma, stats, (bf16*)dx, dgamma, dbeta, R, C,
                              accum ? (const bf16*)dx : nullptr, (hipStream_t)stream);
}

int pea_op_attention_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                         float* lse, int B, int H, int Sq, int Skv, float scale, int nd, void* stream) {
  return attention_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, lse, B, H, Sq, Skv, scale, nd, 0, stream);
}
/* the same two operators on a Q that already carries scale * log2(e) (what the model's Q|K|V / to_q projections hand over:
 * pea_op_gemm_qscale); `scale` is still the softmax scale: dQ comes back as the gradient w.r.t. the UNSCALED q */
int pea_op_attention_fwd_prescaled(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                                   float* lse, int B, int H, int Sq, int Skv, float scale, int nd, void* stream) {
  return attention_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, lse, B, H, Sq, Skv, scale, nd, 1, stream);
}
int pea_op_attention_fwd_masked(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                                float* lse, int B, int H, int Sq, int Skv, float scale, int causal, const int* kv_len,
                                void* stream) {
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.O = (bf16*)O; p.ldo = ldo; p.lse = lse; p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.scale = scale; p.nd = 1;
  p.causal = causal; p.kv_len = kv_len;
  return launch_attention_fwd(p, (hipStream_t)stream);
}
int pea_op_attention_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                         int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                         void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                         int accum_dq, int accum_dkv, int nd, void* scratch, void* stream) {
  return attention_bwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, lse, delta, dQ, lddq, dK, lddk, dV, lddv, B, H, Sq, Skv,
                            scale, accum_dq, accum_dkv, nd, scratch, 0, stream);
}
int pea_op_attention_bwd_prescaled(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                                   int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                                   void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                                   int accum_dq, int accum_dkv, int nd, void* scratch, void* stream) {
  return attention_bwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, lse, delta, dQ, lddq, dK, lddk, dV, lddv, B, H, Sq, Skv,
                            scale, accum_dq, accum_dkv, nd, scratch, 1, stream);
}

int pea_op_geglu_fwd(const void* hg, void* y, long long rows, int inner, void* stream) {
  return launch_geglu_fwd((const bf16*)hg, (bf16*)y, rows, inner, (hipStream_t)stream);
}
int pea_op_geglu_bwd(const void* hg, const void* dy, void* dhg, long long rows, int inner, void* stream) {
  return launch_geglu_bwd((const bf16*)hg, (const bf16*)dy, (bf16*)dhg, rows, inner, (hipStream_t)stream);
}
int pea_op_sumpool2(const void* x, void* y, int B, int H, int W, int C, int accum, void* stream) {
  return launch_sumpool2((const bf16*)x, (bf16*)y, B, H, W, C, accum, (hipStream_t)stream);
}
int pea_op_timestep_embed(const float* t, void* y, int n, int dim, void* stream) {
  return launch_timestep_embed(t, (bf16*)y, n, dim, (hipStream_t)stream);
}
int pea_op_add_noise(const float* x0, const float* eps, const long long* t, const float* ac, float* xt, int B,
                     long long per, void* stream) {
  return launch_add_noise(x0, eps, t, ac, xt, B, per, (hipStream_t)stream);
}
int pea_op_cast_f32_bf16(const float* x, void* y, long long n, void* stream) {
  return launch_cast_f32_bf16(x, (bf16*)y, n, (hipStream_t)stream);
}
int pea_op_cast_bf16_f32(const void* x, float* y, long long n, void* stream) {
  return launch_cast_bf16_f32((const bf16*)x, y, n, (hipStream_t)stream);
}

int pea_op_kd_loss(int ntaps, const void* const* taps_s, const void* const* taps_t, void* const* dtaps,
                   const long long* per, const float* eps_s, const float* eps, const float* eps_t, float* deps_s,
                   long long per_eps, const long long* zh, int B, float feat_weight, int nan_guard,
                   float grad_scale, float* losses, void* workspace, void* stream) {
  if (ntaps < 0 || ntaps > PEA_MAX_TAPS) {
    pea_set_error("kd_loss: ntaps=%d out of range", ntaps);
    return PEA_E_SHAPE;
  }
  KdLossP p;
  memset(&p, 0, sizeof(p));
  p.kd_samples_hint = -1;
  p.ntaps = ntaps;
  for (int k = 0; k < ntaps; ++k) {
    p.fs[k] = (const bf16*)taps_s[k];
    p.ft[k] = (const bf16*)taps_t[k];
    p.dfs[k] = dtaps ? (bf16*)dtaps[k] : nullptr;
    p.per[k] = per[k];
  }
  p.eps_s = eps_s; p.eps = eps; p.eps_t = eps_t; p.deps_s = deps_s; p.per_eps = per_eps; p.zh = zh; p.B = B;
  p.feat_weight = feat_weight; p.nan_guard = nan_guard; p.grad_scale = grad_scale; p.losses = losses;
  p.partial = (float*)workspace;
  return launch_kd_loss(p, (hipStream_t)stream);
}

long long pea_op_attention_bwd_scratch_bytes(int B, int H, int Sq, int Skv, int nd) {
  return (long long)attention_bwd_scratch_bytes(B, H, Sq, Skv, nd);
}
long long pea_op_groupnorm_scratch_bytes(int B, int HW, int C, int groups) {
  return (long long)groupnorm_scratch_bytes(B, HW, C, groups);
}
long long pea_op_kd_loss_workspace_bytes(int ntaps, const long long* per, long long per_eps, int B) {
  return (long long)kd_loss_workspace_bytes(ntaps, per, per_eps, B);
}

int pea_op_prefetch(const void* p, long long bytes, void* stream) {
  return launch_prefetch(p, bytes, nullptr, (hipStream_t)stream);
}

int pea_op_adamw(float* w, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, void* stream) {
  return launch_adamw(w, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)stream);
}

// ---- inference denoise-loop glue (sampler.hip)
long long pea_op_cfg_combine_workspace_bytes(int B) { return (long long)cfg_combine_workspace_bytes(B); }
int pea_op_cfg_combine(const float* eps2, float* out, int B, long long per, float guidance_scale, float guidance_rescale,
                       void* workspace, void* stream) {
  if (!eps2 || !out) { pea_set_error("pea_op_cfg_combine: null pointer"); return PEA_E_INVALID; }
  return launch_cfg_combine(eps2, out, B, per, guidance_scale, guidance_rescale, workspace, (hipStream_t)stream);
}
int pea_op_dpm_update(float* sample, const float* eps, float* x0_prev, long long n, float alpha_s, float sigma_s,
                      float c_s, float c_0, float c_1, void* stream) {
  if (!sample || !eps || !x0_prev) { pea_set_error("pea_op_dpm_update: null pointer"); return PEA_E_INVALID; }
  return launch_dpm_update(sample, eps, x0_prev, n, alpha_s, sigma_s, c_s, c_0, c_1, (hipStream_t)stream);
}

}  // extern "C"
