// HBM-bound glue kernels: GEGLU, SiLU/GELU, residual add, concat/split, 2x2 sum-pool,
// casts/transposes/weight repacks, timestep embedding, add_noise, CFG-dropout select,
// token mean, column sums, RNG fill, fused AdamW.  16 bytes per lane everywhere.
//
// Reference call sites: GEGLU / residual adds / concat / upsample inside diffusers' UNet
// (train_sdxl_zh.py:397,415); add_noise train_sdxl_zh.py:322; `torch.where` CFG dropout
// train_sdxl_zh.py:395,413; `torch.mean(x,1)` train_sdxl_zh.py:66; FusedAdam
// utils/model_utils.py:64-67.
#include "pea_kernels.h"

#define EW_GRID(n8) ((int)(cdivl((n8), 256) < 8192 ? cdivl((n8), 256) : 8192))
#define EW_LOOP(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ---- GEGLU: hg [rows][2*inner] = (h | gate);  y = h * gelu(gate)
__global__ void geglu_fwd_kernel(const bf16* __restrict__ hg, bf16* __restrict__ y, long long rows, int inner) {
  const int ck = inner / 8;
  const long long total = rows * ck;
  EW_LOOP(i, total) {
    const long long r = i / ck;
    const int c = (int)(i - r * ck) * 8;
    const bf16x8 h = *(const bf16x8*)(hg + r * 2 * inner + c);
    const bf16x8 g = *(const bf16x8*)(hg + r * 2 * inner + inner + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)h[j] * gelu_erf((float)g[j]));
    *(bf16x8*)(y + r * inner + c) = o;
  }
}
__global__ void geglu_bwd_kernel(const bf16* __restrict__ hg, const bf16* __restrict__ dy, bf16* __restrict__ dhg,
                                 long long rows, int inner) {
  const int ck = inner / 8;
  const long long total = rows * ck;
  EW_LOOP(i, total) {
    const long long r = i / ck;
    const int c = (int)(i - r * ck) * 8;
    const bf16x8 h = *(const bf16x8*)(hg + r * 2 * inner + c);
    const bf16x8 g = *(const bf16x8*)(hg + r * 2 * inner + inner + c);
    const bf16x8 d = *(const bf16x8*)(dy + r * inner + c);
    bf16x8 dh, dg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gf = (float)g[j], df = (float)d[j];
      dh[j] = (bf16)(df * gelu_erf(gf));
      dg[j] = (bf16)(df * (float)h[j] * gelu_erf_grad(gf));
    }
    *(bf16x8*)(dhg + r * 2 * inner + c) = dh;
    *(bf16x8*)(dhg + r * 2 * inner + inner + c) = dg;
  }
}
int launch_geglu_fwd(const bf16* hg, bf16* y, long long rows, int inner, hipStream_t s) {
  SHAPECHK(inner % 8 == 0, "geglu: inner %% 8");
  PROF_BEGIN(6, 0.0, 2.0 * 3.0 * rows * inner, s);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(EW_GRID(rows * (inner / 8))), dim3(256), 0, s, hg, y, rows, inner);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_geglu_bwd(const bf16* hg, const bf16* dy, bf16* dhg, long long rows, int inner, hipStream_t s) {
  SHAPECHK(inner % 8 == 0, "geglu: inner %% 8");
  PROF_BEGIN(6, 0.0, 2.0 * 5.0 * rows * inner, s);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(EW_GRID(rows * (inner / 8))), dim3(256), 0, s, hg, dy, dhg, rows, inner);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- unary / binary maps on flat bf16 arrays (n % 8 == 0)
template <int OP>   // 0 add, 1 silu, 2 silu_bwd, 3 gelu, 4 gelu_bwd
__global__ void map_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y,
                           long long n8, int accum) {
  EW_LOOP(i, n8) {
    const bf16x8 av = *(const bf16x8*)(a + i * 8);
    bf16x8 bv, o;
    if (OP == 0 || OP == 2 || OP == 4) bv = *(const bf16x8*)(b + i * 8);
    if (accum) o = *(const bf16x8*)(y + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)av[j];
      float r;
      if (OP == 0) r = x + (float)bv[j];
      else if (OP == 1) r = siluf_(x);
      else if (OP == 2) r = (float)bv[j] * silu_grad(x);
      else if (OP == 3) r = gelu_erf(x);
      else r = (float)bv[j] * gelu_erf_grad(x);
      if (accum) r += (float)o[j];
      o[j] = (bf16)r;
    }
    *(bf16x8*)(y + i * 8) = o;
  }
}
#define MAP_LAUNCH(OP, a, b, y, n, acc)                                                           \
  SHAPECHK((n) % 8 == 0, "elementwise: n %% 8");                                                  \
  hipLaunchKernelGGL(map_kernel<OP>, dim3(EW_GRID((n) / 8)), dim3(256), 0, s, a, b, y, (n) / 8, acc); \
  HIPCHK(hipGetLastError());                                                                      \
  return PEA_OK;
int launch_add(const bf16* a, const bf16* b, bf16* y, long long n, hipStream_t s) { MAP_LAUNCH(0, a, b, y, n, 0) }
int launch_silu_fwd(const bf16* x, bf16* y, long long n, hipStream_t s) { MAP_LAUNCH(1, x, nullptr, y, n, 0) }
int launch_silu_bwd(const bf16* x, const bf16* dy, bf16* dx, long long n, int accum, hipStream_t s) {
  MAP_LAUNCH(2, x, dy, dx, n, accum)
}
int launch_gelu_fwd(const bf16* x, bf16* y, long long n, hipStream_t s) { MAP_LAUNCH(3, x, nullptr, y, n, 0) }
int launch_gelu_bwd(const bf16* x, const bf16* dy, bf16* dx, long long n, int accum, hipStream_t s) {
  MAP_LAUNCH(4, x, dy, dx, n, accum)
}

// ---- concat / split along the channel (last) dimension
// Depth-to-space storage of an upsampler output (the sub-pixel form of the upsample-folded conv, model.hip OP_CONV3 p1 == 2):
// [B][H/2][W/2][4 = (y & 1) * 2 + (x & 1)][C] instead of [B][H][W][C].  Seen as rows of C elements, full-resolution pixel row r
// lives at row d2s_row(r).  The first operand of concat2 / split2 may be stored that way (H, W = the full resolution; 0 = plain).
__device__ __forceinline__ long long d2s_row(long long r, int H, int W) {
  const long long t = r / W;
  const int x = (int)(r - t * W);
  const long long b = t / H;
  const int y = (int)(t - b * H);
  return (((b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) << 2) + ((y & 1) << 1) + (x & 1);
}
__global__ void concat2_kernel(const bf16* __restrict__ a, int C1, const bf16* __restrict__ b, int C2,
                               bf16* __restrict__ y, long long rows, int aH, int aW) {
  const int ck = (C1 + C2) / 8, k1 = C1 / 8;
  EW_LOOP(i, rows * ck) {
    const long long r = i / ck;
    const int c = (int)(i - r * ck);
    const long long ra = aW ? d2s_row(r, aH, aW) : r;
    const bf16x8 v = c < k1 ? *(const bf16x8*)(a + ra * C1 + c * 8) : *(const bf16x8*)(b + r * C2 + (c - k1) * 8);
    *(bf16x8*)(y + r * (C1 + C2) + c * 8) = v;
  }
}
__global__ void split2_kernel(const bf16* __restrict__ dy, int C1, int C2, bf16* __restrict__ da, int accum_a,
                              bf16* __restrict__ db, int accum_b, long long rows, int aH, int aW) {
  const int ck = (C1 + C2) / 8, k1 = C1 / 8;
  EW_LOOP(i, rows * ck) {
    const long long r = i / ck;
    const int c = (int)(i - r * ck);
    bf16x8 v = *(const bf16x8*)(dy + r * (C1 + C2) + c * 8);
    bf16* dst;
    int acc;
    if (c < k1) { dst = da ? da + (aW ? d2s_row(r, aH, aW) : r) * C1 + c * 8 : nullptr; acc = accum_a; }
    else { dst = db ? db + r * C2 + (c - k1) * 8 : nullptr; acc = accum_b; }
    if (!dst) continue;
    if (acc) {
      const bf16x8 o = *(const bf16x8*)dst;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)o[j]);
    }
    *(bf16x8*)dst = v;
  }
}
int launch_concat2(const bf16* a, int C1, const bf16* b, int C2, bf16* y, long long rows, hipStream_t s, int aH, int aW) {
  SHAPECHK(C1 % 8 == 0 && C2 % 8 == 0, "concat: C %% 8");
  SHAPECHK(!aW || (aH % 2 == 0 && aW % 2 == 0 && rows % ((long long)aH * aW) == 0), "concat: depth-to-space operand %d x %d", aH, aW);
  PROF_BEGIN(6, 0.0, 4.0 * rows * (C1 + C2), s);
  hipLaunchKernelGGL(concat2_kernel, dim3(EW_GRID(rows * ((C1 + C2) / 8))), dim3(256), 0, s, a, C1, b, C2, y, rows, aH, aW);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_split2(const bf16* dy, int C1, int C2, bf16* da, int accum_a, bf16* db, int accum_b, long long rows,
                  hipStream_t s, int aH, int aW) {
  SHAPECHK(C1 % 8 == 0 && C2 % 8 == 0, "split: C %% 8");
  SHAPECHK(!aW || (aH % 2 == 0 && aW % 2 == 0 && rows % ((long long)aH * aW) == 0), "split: depth-to-space operand %d x %d", aH, aW);
  PROF_BEGIN(6, 0.0, 4.0 * rows * (C1 + C2), s);
  hipLaunchKernelGGL(split2_kernel, dim3(EW_GRID(rows * ((C1 + C2) / 8))), dim3(256), 0, s, dy, C1, C2, da, accum_a,
                     db, accum_b, rows, aH, aW);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- 2x2 sum pooling (backward of nearest-2x upsample)
__global__ void sumpool2_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int H, int W, int C,
                                int accum) {
  const int ck = C / 8;
  const long long total = (long long)B * H * W * ck;
  EW_LOOP(i, total) {
    const int c = (int)(i % ck) * 8;
    const long long pix = i / ck;
    const int xw = (int)(pix % W), yh = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const bf16x8 v = *(const bf16x8*)(x + (((long long)b * 2 * H + 2 * yh + dy) * 2 * W + 2 * xw + dx) * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
    bf16x8 o;
    if (accum) o = *(const bf16x8*)(y + pix * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] + (accum ? (float)o[j] : 0.f));
    *(bf16x8*)(y + pix * C + c) = o;
  }
}
int launch_sumpool2(const bf16* x, bf16* y, int B, int H, int W, int C, int accum, hipStream_t s) {
  SHAPECHK(C % 8 == 0, "sumpool: C %% 8");
  PROF_BEGIN(6, 0.0, 2.0 * 5.0 * B * H * W * C, s);
  hipLaunchKernelGGL(sumpool2_kernel, dim3(EW_GRID((long long)B * H * W * (C / 8))), dim3(256), 0, s, x, y, B, H, W,
                     C, accum);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- casts, transposes, repacks (load-time / boundary; not on the per-step hot path)
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long long n) {
  EW_LOOP(i, n) y[i] = (bf16)x[i];
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, long long n) {
  EW_LOOP(i, n) y[i] = (float)x[i];
}
int launch_cast_f32_bf16(const float* x, bf16* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(EW_GRID(n)), dim3(256), 0, s, x, y, n);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_cast_bf16_f32(const bf16* x, float* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(EW_GRID(n)), dim3(256), 0, s, x, y, n);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

template <typename TI>
__global__ void transpose_kernel(const TI* __restrict__ x, bf16* __restrict__ y, int R, int C, int ldy) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = by + i, c = bx + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? (float)x[(long long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = bx + i, r = by + threadIdx.x;
    if (c < C && r < R) y[(long long)c * ldy + r] = (bf16)tile[threadIdx.x][i];
  }
}
int launch_transpose_bf16(const bf16* x, bf16* y, int R, int C, int ldy, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel<bf16>, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, s, x, y, R, C, ldy);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_transpose_f32_bf16(const float* x, bf16* y, int R, int C, int ldy, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel<float>, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, s, x, y, R, C, ldy);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

__global__ void nhwc_to_nchw_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, int B, int HW, int C, int dH, int dW) {
  const long long total = (long long)B * HW * C;
  EW_LOOP(i, total) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long long)HW * C));
    const long long r = (long long)b * HW + p;
    y[i] = (float)x[(dW ? d2s_row(r, dH, dW) : r) * C + c];
  }
}
// dH, dW != 0: x is stored depth-to-space (see d2s_row) at full resolution dH x dW (HW = dH * dW)
int launch_nhwc_to_nchw_f32(const bf16* x, float* y, int B, int HW, int C, hipStream_t s, int dH, int dW) {
  SHAPECHK(!dW || (long long)dH * dW == HW, "nhwc_to_nchw: depth-to-space %d x %d vs HW=%d", dH, dW, HW);
  hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(EW_GRID((long long)B * HW * C)), dim3(256), 0, s, x, y, B, HW, C, dH, dW);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
// the inverse, fp32 NCHW -> the bf16 storage layout (gradient seeds of a feature tap)
__global__ void nchw_f32_to_nhwc_kernel(const float* __restrict__ x, bf16* __restrict__ y, int B, int HW, int C, int dH, int dW) {
  const long long total = (long long)B * HW * C;
  EW_LOOP(i, total) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long long)HW * C));
    const long long r = (long long)b * HW + p;
    y[(dW ? d2s_row(r, dH, dW) : r) * C + c] = (bf16)x[i];
  }
}
int launch_nchw_f32_to_nhwc(const float* x, bf16* y, int B, int HW, int C, hipStream_t s, int dH, int dW) {
  SHAPECHK(!dW || (long long)dH * dW == HW, "nchw_to_nhwc: depth-to-space %d x %d vs HW=%d", dH, dW, HW);
  hipLaunchKernelGGL(nchw_f32_to_nhwc_kernel, dim3(EW_GRID((long long)B * HW * C)), dim3(256), 0, s, x, y, B, HW, C, dH, dW);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// w [Co][Ci][3][3] fp32 -> mode 0: y[co][(ky*3+kx)*Ci + ci]; mode 1: y[ci][((2-ky)*3+(2-kx))*Co + co] (bf16)
// mode 2: y[co][ky][kx][ci] fp32
__global__ void pack_conv_kernel(const float* __restrict__ w, bf16* __restrict__ yb, float* __restrict__ yf, int Co,
                                 int Ci, int mode, int Cip) {
  const long long total = (long long)Co * Ci * 9;
  EW_LOOP(i, total) {
    const int kx = (int)(i % 3), ky = (int)((i / 3) % 3);
    const int ci = (int)((i / 9) % Ci), co = (int)(i / (9LL * Ci));
    const float v = w[i];
    if (mode == 0) yb[(long long)co * 9 * Cip + (ky * 3 + kx) * Cip + ci] = (bf16)v;
    else if (mode == 1) yb[(long long)ci * 9 * Co + ((2 - ky) * 3 + (2 - kx)) * Co + co] = (bf16)v;
    else yf[(long long)co * 9 * Ci + (ky * 3 + kx) * Ci + ci] = v;
  }
}
int launch_pack_conv_fwd(const float* w, bf16* y, int Co, int Ci, hipStream_t s, int Cip) {
  if (Cip <= 0) Cip = Ci;
  if (Cip != Ci) HIPCHK(hipMemsetAsync(y, 0, (size_t)Co * 9 * Cip * 2, s));      // zero-padded input channels
  hipLaunchKernelGGL(pack_conv_kernel, dim3(EW_GRID(9LL * Co * Ci)), dim3(256), 0, s, w, y, nullptr, Co, Ci, 0, Cip);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_pack_conv_dgrad(const float* w, bf16* y, int Co, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(EW_GRID(9LL * Co * Ci)), dim3(256), 0, s, w, y, nullptr, Co, Ci, 1, Ci);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_pack_conv_out(const float* w, float* y, int Co, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(EW_GRID(9LL * Co * Ci)), dim3(256), 0, s, w, nullptr, y, Co, Ci, 2, Ci);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// Sub-pixel form of conv3x3(nearest_2x(x)) (diffusers Upsample2D: interpolate, then conv).  Output pixel (2s + py) reads the
// upsampled rows 2s + py - 1 .. 2s + py + 1, i.e. the SOURCE rows  py = 0: {s-1: w0, s: w1 + w2},  py = 1: {s: w0 + w1, s+1: w2}
// (and the same along x): per output parity pl = py * 2 + px a 2 x 2 kernel of summed taps over the source -- 16 tap products per
// source pixel and output channel where the folded 3 x 3 form executes 36.  The sums are taken in fp32 and rounded once.
//   mode 0 (forward):  y[pl][co][(dy * 2 + dx) * Ci + ci]            window origin of parity pl: source (s - 1 + py, s - 1 + px)
//   mode 1 (dgrad):    y[ci][(pl * 4 + ty * 2 + tx) * Co + co] = forward[pl][co][(1 - ty, 1 - tx)][ci]   (GemmP::kside == 4)
__global__ void pack_conv_subpix_kernel(const float* __restrict__ w, bf16* __restrict__ y, int Co, int Ci, int mode) {
  const long long total = 16LL * Co * Ci;
  EW_LOOP(i, total) {
    const int ci = (int)(i % Ci);
    const int co = (int)((i / Ci) % Co);
    const int t = (int)(i / ((long long)Ci * Co));       // pl * 4 + dy * 2 + dx
    const int pl = t >> 2, dy = (t >> 1) & 1, dx = t & 1, py = pl >> 1, px = pl & 1;
    // taps of the 3-wide kernel that fall on source offset d of parity q:  q=0: d=0 {0}, d=1 {1,2};  q=1: d=0 {0,1}, d=1 {2}
    const int y0 = py == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), y1 = py == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
    const int x0 = px == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), x1 = px == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
    const float* src = w + ((long long)co * Ci + ci) * 9;
    float acc = 0.f;
    for (int ky = y0; ky <= y1; ++ky)
      for (int kx = x0; kx <= x1; ++kx) acc += src[ky * 3 + kx];
    if (mode == 0) y[((long long)pl * Co + co) * 4 * Ci + (dy * 2 + dx) * Ci + ci] = (bf16)acc;
    else y[(long long)ci * 16 * Co + (pl * 4 + (1 - dy) * 2 + (1 - dx)) * Co + co] = (bf16)acc;
  }
}
int launch_pack_conv_subpix(const float* w, bf16* y, int Co, int Ci, int dgrad, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_subpix_kernel, dim3(EW_GRID(16LL * Co * Ci)), dim3(256), 0, s, w, y, Co, Ci, dgrad ? 1 : 0);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- sinusoidal embedding, diffusers get_timestep_embedding(flip_sin_to_cos=True, shift 0): (cos | sin)
__global__ void timestep_embed_kernel(const float* __restrict__ t, bf16* __restrict__ y, int n, int dim) {
  const int half = dim / 2;
  EW_LOOP(i, (long long)n * half) {
    const int k = (int)(i % half), r = (int)(i / half);
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);   // ln(10000)
    const float a = t[r] * freq;
    y[(long long)r * dim + k] = (bf16)cosf(a);
    y[(long long)r * dim + half + k] = (bf16)sinf(a);
  }
}
int launch_timestep_embed(const float* t, bf16* y, int n, int dim, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(EW_GRID((long long)n * dim / 2)), dim3(256), 0, s, t, y, n, dim);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                 const long long* __restrict__ t, const float* __restrict__ ac,
                                 float* __restrict__ xt, int B, long long per) {
  EW_LOOP(i, (long long)B * per) {
    const int b = (int)(i / per);
    const float a = ac[t[b]];
    xt[i] = sqrtf(a) * x0[i] + sqrtf(1.0f - a) * eps[i];
  }
}
int launch_add_noise(const float* x0, const float* eps, const long long* t, const float* ac, float* xt, int B,
                     long long per, hipStream_t s) {
  hipLaunchKernelGGL(add_noise_kernel, dim3(EW_GRID((long long)B * per)), dim3(256), 0, s, x0, eps, t, ac, xt, B, per);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- CFG-dropout row select and its backward (train_sdxl_zh.py:395)
// ystride8 / dystride8: per-sample stride (in 8-element groups) of y / dy; >= per8 when the consumer's tensor holds more
// tokens per sample than the adapter produced (merged passes with a shorter student context: the tail stays zero)
__global__ void select_rows_kernel(const bf16* __restrict__ c, const bf16* __restrict__ u,
                                   const unsigned char* __restrict__ mask, bf16* __restrict__ y, int B,
                                   long long per8, long long ystride8) {
  EW_LOOP(i, (long long)B * per8) {
    const int b = (int)(i / per8);
    const long long o = (long long)b * ystride8 + (i - (long long)b * per8);
    *(bf16x8*)(y + o * 8) = mask[b] ? *(const bf16x8*)(u + i * 8) : *(const bf16x8*)(c + i * 8);
  }
}
__global__ void select_rows_bwd_kernel(const bf16* __restrict__ dy, const unsigned char* __restrict__ mask,
                                       bf16* __restrict__ dc, bf16* __restrict__ du, int B, long long per8,
                                       long long dystride8) {
  EW_LOOP(i, (long long)B * per8) {
    const int b = (int)(i / per8);
    const long long o = (long long)b * dystride8 + (i - (long long)b * per8);
    const bf16x8 v = *(const bf16x8*)(dy + o * 8);
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
    *(bf16x8*)(dc + i * 8) = mask[b] ? z : v;
    *(bf16x8*)(du + i * 8) = mask[b] ? v : z;
  }
}
int launch_select_rows(const bf16* c, const bf16* u, const unsigned char* mask, bf16* y, int B, long long per,
                       hipStream_t s, long long ystride) {
  if (ystride <= 0) ystride = per;
  SHAPECHK(per % 8 == 0 && ystride % 8 == 0 && ystride >= per, "select: per %% 8");
  hipLaunchKernelGGL(select_rows_kernel, dim3(EW_GRID(B * per / 8)), dim3(256), 0, s, c, u, mask, y, B, per / 8, ystride / 8);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_select_rows_bwd(const bf16* dy, const unsigned char* mask, bf16* dc, bf16* du, int B, long long per,
                           hipStream_t s, long long dystride) {
  if (dystride <= 0) dystride = per;
  SHAPECHK(per % 8 == 0 && dystride % 8 == 0 && dystride >= per, "select: per %% 8");
  hipLaunchKernelGGL(select_rows_bwd_kernel, dim3(EW_GRID(B * per / 8)), dim3(256), 0, s, dy, mask, dc, du, B,
                     per / 8, dystride / 8);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- mean over tokens (adapter pooled branch) and backward
__global__ void mean_tokens_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int L, int C) {
  EW_LOOP(i, (long long)B * C) {
    const int c = (int)(i % C), b = (int)(i / C);
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += (float)x[((long long)b * L + l) * C + c];
    y[i] = (bf16)(s / (float)L);
  }
}
__global__ void mean_tokens_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int B, int L, int C,
                                       int accum) {
  EW_LOOP(i, (long long)B * L * C) {
    const int c = (int)(i % C);
    const int b = (int)(i / ((long long)L * C));
    float v = (float)dy[(long long)b * C + c] / (float)L;
    if (accum) v += (float)dx[i];
    dx[i] = (bf16)v;
  }
}
int launch_mean_tokens(const bf16* x, bf16* y, int B, int L, int C, hipStream_t s) {
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(EW_GRID((long long)B * C)), dim3(256), 0, s, x, y, B, L, C);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_mean_tokens_bwd(const bf16* dy, bf16* dx, int B, int L, int C, int accum, hipStream_t s) {
  hipLaunchKernelGGL(mean_tokens_bwd_kernel, dim3(EW_GRID((long long)B * L * C)), dim3(256), 0, s, dy, dx, B, L, C,
                     accum);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// db[c] (+)= sum_r x[r][c].  A block owns 64 columns; its 16 waves take the rows round-robin and their partial sums are added in
// wave order (deterministic) -- one thread per column walked the adapter's 616 rows serially in 144 us.
__global__ __launch_bounds__(1024) void colsum_kernel(const bf16* __restrict__ x, float* __restrict__ db, int R, int C, int accum) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = w; r < R; r += 16) s += (float)x[(long long)r * C + c];
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][lane];
    db[c] = accum ? db[c] + t : t;
  }
}
int launch_colsum(const bf16* x, float* db, int R, int C, int accum, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, s, x, db, R, C, accum);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- counter-based RNG fill (splitmix64 -> uniform(-1,1) * scale): random-init weights of a given
// architecture for bench.py (no checkpoints in the image), deterministic in (seed, index).
__device__ __forceinline__ float rng_uniform(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)((z >> 40) & 0xFFFFFF) * (2.0f / 16777216.0f) - 1.0f;
}
__global__ void fill_random_bf16_kernel(bf16* p, long long n, unsigned long long seed, float scale) {
  EW_LOOP(i, n) p[i] = (bf16)(rng_uniform(seed, (unsigned long long)i) * scale);
}
__global__ void fill_random_f32_kernel(float* p, long long n, unsigned long long seed, float scale, float offset) {
  EW_LOOP(i, n) p[i] = rng_uniform(seed, (unsigned long long)i) * scale + offset;
}
int launch_fill_random_bf16(bf16* p, long long n, unsigned long long seed, float scale, hipStream_t s) {
  hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(EW_GRID(n)), dim3(256), 0, s, p, n, seed, scale);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_fill_random_f32(float* p, long long n, unsigned long long seed, float scale, float offset,
                           hipStream_t s) {
  hipLaunchKernelGGL(fill_random_f32_kernel, dim3(EW_GRID(n)), dim3(256), 0, s, p, n, seed, scale, offset);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- fused AdamW (adam_w_mode=True as FusedAdam in utils/model_utils.py:64-67), fp32 master weights
__global__ void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2, float gscale) {
  EW_LOOP(i, n) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi / bc2) + eps;
    w[i] = w[i] * (1.f - lr * wd) - lr * (mi / bc1) / denom;
  }
}
int launch_adamw(float* w, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                 float wd, int step, float gscale, hipStream_t s) {
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(EW_GRID(n)), dim3(256), 0, s, w, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2,
                     gscale);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- per-batch column sums, deterministic two-stage: out[b][c] = sum_p x[b][p][c]
// (gradient of the per-sample time-embedding row vector added in the ResBlock conv1 epilogue)
__global__ void colsum_batched_kernel(const bf16* __restrict__ x, float* __restrict__ partial, int HW, int C,
                                      int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) char csm[];
  float* tsum = (float*)csm;                       // [ppb][C]
  const int nchunk = C / 8;
  const int ppb = blockDim.x / nchunk;
  const int ck = threadIdx.x % nchunk, pl = threadIdx.x / nchunk;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16* xb = x + (long long)b * HW * C + ck * 8;
  int p = p0 + pl;
  for (; p + 3 * ppb < p1; p += 4 * ppb) {          // four independent 16-byte loads in flight per thread
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(xb + (long long)(p + u * ppb) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[u][j];
  }
  for (; p < p1; p += ppb) {
    const bf16x8 v = *(const bf16x8*)(xb + (long long)p * C);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) tsum[(long long)pl * C + ck * 8 + j] = s[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int l = 0; l < ppb; ++l) a += tsum[(long long)l * C + c];
    partial[((long long)b * gridDim.x + blockIdx.x) * C + c] = a;
  }
}
// second stage: a block owns 64 columns of one sample; its 16 waves take the block partials round-robin and their sums are
// added in wave order (deterministic).  One thread per column walked the up to 171 partials serially (25 us per launch).
__global__ __launch_bounds__(1024) void colsum_batched_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                     int nblk, int C, int ldo) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int k = w; k < nblk; k += 16) a += partial[((long long)b * nblk + k) * C + c];
  }
  part[w][lane] = a;
  __syncthreads();
  if (w == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][lane];
    out[(long long)b * ldo + c] = t;
  }
}
static void colsum_geometry(int HW, int C, int* threads, int* per, int* nblk) {
  const int nchunk = C / 8;
  int ppb = 256 / nchunk;
  if (ppb < 1) ppb = 1;
  *threads = nchunk * ppb;
  int p = ppb * 16;                 // 16 pixels per thread: >= 1000 workgroups at the UNet's sizes (128 per thread left
                                    // 88 workgroups of serial loads on 256 CUs: 40 us for every size)
  if (p > HW) p = HW;
  *per = p;
  *nblk = cdiv(HW, p);
}
size_t colsum_batched_scratch_bytes(int B, int HW, int C) {
  int threads, per, nblk;
  colsum_geometry(HW, C, &threads, &per, &nblk);
  return (size_t)B * nblk * C * sizeof(float) + 256;
}
int launch_colsum_batched(const bf16* x, float* out, int B, int HW, int C, int ldo, float* scratch, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C / 8 <= 1024, "colsum_batched: C=%d", C);
  int threads, per, nblk;
  colsum_geometry(HW, C, &threads, &per, &nblk);
  const int ppb = threads / (C / 8);
  PROF_BEGIN(6, 0.0, 2.0 * B * (double)HW * C, s);
  hipLaunchKernelGGL(colsum_batched_kernel, dim3(nblk, B), dim3(threads), (size_t)ppb * C * 4, s, x, scratch, HW, C,
                     per);
  hipLaunchKernelGGL(colsum_batched_reduce_kernel, dim3(cdiv(C, 64), B), dim3(1024), 0, s, scratch, out, nblk, C, ldo);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// y (+)= x   (bf16), and bf16 <- fp32 with optional accumulate
__global__ void accum_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long long n8, int accum) {
  EW_LOOP(i, n8) {
    bf16x8 v = *(const bf16x8*)(x + i * 8);
    if (accum) {
      const bf16x8 o = *(const bf16x8*)(y + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)o[j]);
    }
    *(bf16x8*)(y + i * 8) = v;
  }
}
int launch_accum(const bf16* x, bf16* y, long long n, int accum, hipStream_t s) {
  SHAPECHK(n % 8 == 0, "accum: n %% 8");
  PROF_BEGIN(6, 0.0, 2.0 * 3.0 * n, s);
  hipLaunchKernelGGL(accum_kernel, dim3(EW_GRID(n / 8)), dim3(256), 0, s, x, y, n / 8, accum);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
// strided 2-D copy/accumulate of bf16 (column slices of fused buffers): y[r][0:C] (+)= x[r][0:C]
__global__ void copy2d_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ y, int ldy, long long rows,
                              int C, int accum) {
  const int ck = C / 8;
  EW_LOOP(i, rows * ck) {
    const long long r = i / ck;
    const int c = (int)(i - r * ck) * 8;
    bf16x8 v = *(const bf16x8*)(x + r * ldx + c);
    if (accum) {
      const bf16x8 o = *(const bf16x8*)(y + r * ldy + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)o[j]);
    }
    *(bf16x8*)(y + r * ldy + c) = v;
  }
}
int launch_copy2d(const bf16* x, int ldx, bf16* y, int ldy, long long rows, int C, int accum, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "copy2d: alignment");
  hipLaunchKernelGGL(copy2d_kernel, dim3(EW_GRID(rows * (C / 8))), dim3(256), 0, s, x, ldx, y, ldy, rows, C, accum);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

__global__ void cast_i64_f32_kernel(const long long* __restrict__ x, float* __restrict__ y, long long n) {
  EW_LOOP(i, n) y[i] = (float)x[i];
}
int launch_cast_i64_f32(const long long* x, float* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(cast_i64_f32_kernel, dim3(EW_GRID(n)), dim3(256), 0, s, x, y, n);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- software prefetch of a read-only buffer into the 256 MiB Infinity Cache / L2 (side stream):
// the frozen UNet weights are streamed once per pass, so a GEMM's first touch of its weight tile is an
// HBM miss; touching the NEXT op's weights while the current op computes turns those misses into
// on-die hits.  Reads 16 B per lane, keeps the value alive without storing it.
__global__ void prefetch_kernel(const uint4* __restrict__ p, long long n16, int* sink) {
  unsigned acc = 0;
  EW_LOOP(i, n16) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9E3779B9u && sink) *sink = 1;      // practically never; defeats dead-code elimination
}
int launch_prefetch(const void* p, long long bytes, int* sink, hipStream_t s) {
  const long long n16 = bytes / 16;
  if (n16 <= 0) return PEA_OK;
  int grid = (int)(cdivl(n16, 256 * 4) < 512 ? cdivl(n16, 256 * 4) : 512);
  hipLaunchKernelGGL(prefetch_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)p, n16, sink);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- Linear weight [N_t][K_t] fp32 (torch) -> bf16 forward layout w[st_n][ldw] and dgrad layout wt[st_k][ldwt] with the
// per-head dimension padded from d to dp: mode 1 pads ROWS (to_q/to_k/to_v: row = head*d + j), mode 2 pads COLUMNS
// (to_out: input feature = head*d + j).  Padding entries are zeros; every stored element is written exactly once.
__global__ void pad_gather_kernel(const float* __restrict__ src, int N_t, int K_t, int mode, int d, int dp,
                                  bf16* __restrict__ w, int ldw, bf16* __restrict__ wt, int ldwt, int st_n, int st_k) {
  EW_LOOP(i, (long long)st_n * st_k) {
    const int rp = (int)(i / st_k), kp = (int)(i % st_k);
    int r = rp, k = kp;
    bool pad = false;
    if (mode == 1) { const int h = rp / dp, j = rp - h * dp; pad = j >= d; r = h * d + j; }
    else if (mode == 2) { const int h = kp / dp, j = kp - h * dp; pad = j >= d; k = h * d + j; }
    else if (mode == 3) { r = (rp & 1) ? (st_n >> 1) + (rp >> 1) : (rp >> 1); }   // GEGLU proj: rows interleaved (h_i, gate_i)
    const float v = pad ? 0.f : src[(long long)r * K_t + k];
    w[(long long)rp * ldw + kp] = (bf16)v;
    if (wt) wt[(long long)kp * ldwt + rp] = (bf16)v;
  }
}
int launch_pad_gather(const float* src, int N_t, int K_t, int mode, int d, int dp, bf16* w, int ldw, bf16* wt, int ldwt,
                      int st_n, int st_k, hipStream_t s) {
  hipLaunchKernelGGL(pad_gather_kernel, dim3(EW_GRID((long long)st_n * st_k)), dim3(256), 0, s, src, N_t, K_t, mode, d,
                     dp, w, ldw, wt, ldwt, st_n, st_k);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- GEGLU on the INTERLEAVED pre-activation layout hg[r][2i] = h_i, hg[r][2i+1] = gate_i (fused FF-proj epilogue)
__global__ void geglu_il_kernel(const bf16* __restrict__ hg, const bf16* __restrict__ dy, bf16* __restrict__ out,
                                long long rows, int inner, int bwd, int form) {
  const int ck = inner / 4;                       // 4 (h, gate) pairs = one 16-byte chunk of hg
  EW_LOOP(i, rows * ck) {
    const long long r = i / ck;
    const int c = (int)(i - r * ck) * 4;
    const bf16x8 v = *(const bf16x8*)(hg + r * 2 * inner + 2 * c);
    if (!bwd) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)v[2 * j] * gelu_erf((float)v[2 * j + 1]));
      *(bf16x4*)(out + r * inner + c) = o;
    } else {
      const bf16x4 d = *(const bf16x4*)(dy + r * inner + c);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float hf = (float)v[2 * j], gf = (float)v[2 * j + 1], df = (float)d[j];
        if (form) {                  // hg holds (gelu(gate), h * gelu'(gate)) (GemmP::stash_grad)
          o[2 * j] = (bf16)(df * hf);
          o[2 * j + 1] = (bf16)(df * gf);
        } else {
          o[2 * j] = (bf16)(df * gelu_erf(gf));
          o[2 * j + 1] = (bf16)(df * hf * gelu_erf_grad(gf));
        }
      }
      *(bf16x8*)(out + r * 2 * inner + 2 * c) = o;
    }
  }
}
int launch_geglu_fwd_il(const bf16* hg, bf16* y, long long rows, int inner, hipStream_t s) {
  SHAPECHK(inner % 4 == 0, "geglu: inner %% 4");
  PROF_BEGIN(6, 0.0, 2.0 * 3.0 * rows * inner, s);
  hipLaunchKernelGGL(geglu_il_kernel, dim3(EW_GRID(rows * (inner / 4))), dim3(256), 0, s, hg, nullptr, y, rows, inner, 0, 0);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
int launch_geglu_bwd_il(const bf16* hg, const bf16* dy, bf16* dhg, long long rows, int inner, hipStream_t s, int form) {
  SHAPECHK(inner % 4 == 0, "geglu: inner %% 4");
  PROF_BEGIN(6, 0.0, 2.0 * 5.0 * rows * inner, s);
  hipLaunchKernelGGL(geglu_il_kernel, dim3(EW_GRID(rows * (inner / 4))), dim3(256), 0, s, hg, dy, dhg, rows, inner, 1, form);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
__global__ void permute_geglu_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int inner) {
  EW_LOOP(i, 2LL * inner) dst[i] = (i & 1) ? src[inner + (i >> 1)] : src[i >> 1];
}
int launch_permute_geglu_vec(const float* src, float* dst, int inner, hipStream_t s) {
  hipLaunchKernelGGL(permute_geglu_vec_kernel, dim3(EW_GRID(2LL * inner)), dim3(256), 0, s, src, dst, inner);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- text encoders: embeddings (token + position [+ token type 0]) and the EOS-token row (CLIP pooled output)
__global__ void embed_tokens_kernel(const long long* __restrict__ ids, const bf16* __restrict__ tok,
                                    const bf16* __restrict__ pos, const bf16* __restrict__ type0,
                                    bf16* __restrict__ out, int B, int L, int width, int vocab) {
  const int ck = width / 8;
  EW_LOOP(i, (long long)B * L * ck) {
    const int c = (int)(i % ck);
    const long long r = i / ck;
    const int l = (int)(r % L);
    long long id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const bf16x8 t = *(const bf16x8*)(tok + id * width + c * 8);
    if (!pos) {                                  // T5: token embedding only (positions enter as an attention bias)
      *(bf16x8*)(out + r * width + c * 8) = t;
      continue;
    }
    const bf16x8 p = *(const bf16x8*)(pos + (long long)l * width + c * 8);
    bf16x8 o;
    if (type0) {
      const bf16x8 ty = *(const bf16x8*)(type0 + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)t[j] + (float)p[j] + (float)ty[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)t[j] + (float)p[j]);
    }
    *(bf16x8*)(out + r * width + c * 8) = o;
  }
}
int launch_embed_tokens(const long long* ids, const bf16* tok, const bf16* pos, const bf16* type0, bf16* out, int B, int L,
                        int width, int vocab, hipStream_t s) {
  SHAPECHK(width % 8 == 0, "embed: width %% 8");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(EW_GRID((long long)B * L * width / 8)), dim3(256), 0, s, ids, tok, pos,
                     type0, out, B, L, width, vocab);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
// T5 relative position bias (HF T5Attention.compute_bias / _relative_position_bucket, bidirectional): bias[h][q][k] =
// rel[bucket(k - q)][h] * log2(e)  (the attention kernel works in the log2 domain); rel: fp32 [num_buckets][H];
// bucket(d) = (d > 0 ? num_buckets/2 : 0) + dist_bucket[|d|], the |d| -> sub-bucket table is computed on the host
// (Tape::alloc) so that the log-spaced bucket edges round exactly as the CPU reference's fp32 arithmetic does
__global__ void t5_rel_bias_kernel(const float* __restrict__ rel, const int* __restrict__ dist_bucket,
                                   float* __restrict__ bias, int H, int L, int pitch, int half_buckets) {
  const long long total = (long long)H * L * L;
  EW_LOOP(i, total) {
    const int k = (int)(i % L), q = (int)((i / L) % L), h = (int)(i / ((long long)L * L));
    const int rp = k - q;
    const int bucket = (rp > 0 ? half_buckets : 0) + dist_bucket[rp < 0 ? -rp : rp];
    bias[((long long)h * L + q) * pitch + k] = rel[(long long)bucket * H + h] * 1.4426950408889634f;
  }
}
int launch_t5_rel_bias(const float* rel, const int* dist_bucket, float* bias, int H, int L, int pitch, int half_buckets,
                       hipStream_t s) {
  hipLaunchKernelGGL(t5_rel_bias_kernel, dim3(EW_GRID((long long)H * L * L)), dim3(256), 0, s, rel, dist_bucket, bias, H, L,
                     pitch, half_buckets);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
// out[b] = x[b][first l with ids[b][l] == eos_id]  (eos_id < 0: argmax of the ids, the CLIP tokenizer's EOS is the largest id)
__global__ void gather_eos_kernel(const long long* __restrict__ ids, const bf16* __restrict__ x, bf16* __restrict__ out,
                                  int L, int width, long long eos_id) {
  const int b = blockIdx.x;
  int pos = 0;
  if (eos_id >= 0) {
    pos = L - 1;
    for (int l = 0; l < L; ++l)
      if (ids[(long long)b * L + l] == eos_id) { pos = l; break; }
  } else {
    long long best = ids[(long long)b * L];
    for (int l = 1; l < L; ++l)
      if (ids[(long long)b * L + l] > best) { best = ids[(long long)b * L + l]; pos = l; }
  }
  for (int c = threadIdx.x; c < width; c += blockDim.x) out[(long long)b * width + c] = x[((long long)b * L + pos) * width + c];
}
int launch_gather_eos(const long long* ids, const bf16* x, bf16* out, int B, int L, int width, long long eos_id, hipStream_t s) {
  hipLaunchKernelGGL(gather_eos_kernel, dim3(B), dim3(256), 0, s, ids, x, out, L, width, eos_id);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
// len[b] = number of leading tokens != pad_id  (BERT-style right padding -> key count for the attention mask)
__global__ void kv_len_kernel(const long long* __restrict__ ids, int* __restrict__ len, int B, int L, long long pad_id) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0;
  for (int l = 0; l < L; ++l)
    if (ids[(long long)b * L + l] != pad_id) n = l + 1;
  len[b] = n > 0 ? n : 1;
}
int launch_kv_len(const long long* ids, int* len, int B, int L, long long pad_id, hipStream_t s) {
  hipLaunchKernelGGL(kv_len_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, ids, len, B, L, pad_id);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
