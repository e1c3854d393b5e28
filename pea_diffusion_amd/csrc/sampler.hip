// Inference denoise-loop glue of tests/test_sdxl_zh.py:376-406: classifier-free-guidance combine (+ the std rescale of
// rescale_noise_cfg, :44-56) and the DPM-Solver++ (2M) latent update.  fp32 NCHW latents; HBM-bound elementwise and
// fixed-order reductions (bit-reproducible).  The schedule's scalar coefficients are computed on the host.
#include "pea_kernels.h"

#define SM_LOOP(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static inline int sm_grid(long long n) { long long g = (n + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

// eps2 = [uncond (B samples) | text (B samples)]; out = u + g * (t - u)
__global__ void cfg_combine_kernel(const float* __restrict__ eps2, float* __restrict__ out, long long n, float g) {
  SM_LOOP(i, n) {
    const float u = eps2[i], t = eps2[n + i];
    out[i] = u + g * (t - u);
  }
}

#define CFG_NBLK 64
// per (sample, block): sums of t, t^2, c, c^2 in double, fixed order inside the block
__global__ __launch_bounds__(256) void cfg_stats_kernel(const float* __restrict__ eps2, long long n, long long per,
                                                        float g, double* __restrict__ part) {
  const int b = blockIdx.y;
  const float* u = eps2 + (long long)b * per;
  const float* t = eps2 + n + (long long)b * per;
  double s[4] = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)CFG_NBLK * 256) {
    const float uv = u[i], tv = t[i];
    const float c = uv + g * (tv - uv);
    s[0] += tv; s[1] += (double)tv * tv; s[2] += c; s[3] += (double)c * c;
  }
  __shared__ double red[4][256];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) part[((long long)b * CFG_NBLK + blockIdx.x) * 4 + threadIdx.x] = red[threadIdx.x][0];
}
// factor[b] = phi * std_text / std_cfg + (1 - phi)   (torch.std: unbiased)
__global__ void cfg_factor_kernel(const double* __restrict__ part, float* __restrict__ factor, int B, long long per,
                                  float phi) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s[4] = {0, 0, 0, 0};
  for (int k = 0; k < CFG_NBLK; ++k)
    for (int j = 0; j < 4; ++j) s[j] += part[((long long)b * CFG_NBLK + k) * 4 + j];
  const double nn = (double)per;
  const double vt = (s[1] - s[0] * s[0] / nn) / (nn - 1.0), vc = (s[3] - s[2] * s[2] / nn) / (nn - 1.0);
  factor[b] = (float)((double)phi * sqrt(vt / vc) + (1.0 - (double)phi));
}
__global__ void cfg_apply_kernel(const float* __restrict__ eps2, float* __restrict__ out, long long n, long long per,
                                 float g, const float* __restrict__ factor) {
  SM_LOOP(i, n) {
    const float u = eps2[i], t = eps2[n + i];
    out[i] = (u + g * (t - u)) * factor[i / per];
  }
}

size_t cfg_combine_workspace_bytes(int B) { return (size_t)B * CFG_NBLK * 4 * sizeof(double) + (size_t)B * sizeof(float); }

int launch_cfg_combine(const float* eps2, float* out, int B, long long per, float g, float rescale, void* ws,
                       hipStream_t s) {
  SHAPECHK(B > 0 && per > 1, "cfg_combine: B=%d per=%lld", B, per);
  const long long n = (long long)B * per;
  if (rescale <= 0.f) {
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(sm_grid(n)), dim3(256), 0, s, eps2, out, n, g);
  } else {
    SHAPECHK(ws != nullptr, "cfg_combine: guidance_rescale needs the workspace");
    double* part = (double*)ws;
    float* factor = (float*)(part + (size_t)B * CFG_NBLK * 4);
    hipLaunchKernelGGL(cfg_stats_kernel, dim3(CFG_NBLK, B), dim3(256), 0, s, eps2, n, per, g, part);
    hipLaunchKernelGGL(cfg_factor_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, part, factor, B, per, rescale);
    hipLaunchKernelGGL(cfg_apply_kernel, dim3(sm_grid(n)), dim3(256), 0, s, eps2, out, n, per, g, factor);
  }
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// x0 = (sample - sigma_s * eps) / alpha_s;  sample <- c_s * sample + c_0 * x0 + c_1 * x0_prev;  x0_prev <- x0
__global__ void dpm_update_kernel(float* __restrict__ sample, const float* __restrict__ eps, float* __restrict__ x0_prev,
                                  long long n, float inv_alpha, float sigma, float cs, float c0, float c1) {
  SM_LOOP(i, n) {
    const float x = sample[i];
    const float x0 = (x - sigma * eps[i]) * inv_alpha;
    float y = cs * x + c0 * x0;
    if (c1 != 0.f) y += c1 * x0_prev[i];
    sample[i] = y;
    x0_prev[i] = x0;
  }
}
int launch_dpm_update(float* sample, const float* eps, float* x0_prev, long long n, float alpha_s, float sigma_s,
                      float c_s, float c_0, float c_1, hipStream_t s) {
  SHAPECHK(n > 0 && alpha_s > 0.f, "dpm_update: n=%lld alpha=%g", n, alpha_s);
  hipLaunchKernelGGL(dpm_update_kernel, dim3(sm_grid(n)), dim3(256), 0, s, sample, eps, x0_prev, n, 1.0f / alpha_s,
                     sigma_s, c_s, c_0, c_1);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ControlNet residual hand-over: [B][C][HW] (fp32 or bf16) -> [B][HW][C] bf16, scaled
template <typename T>
__global__ void nchw_to_nhwc_scaled_kernel(const T* __restrict__ x, bf16* __restrict__ y, int C, long long HW,
                                           long long n, float scale) {
  SM_LOOP(i, n) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const long long p = r % HW, b = r / HW;
    y[i] = (bf16)((float)x[(b * C + c) * HW + p] * scale);
  }
}
__global__ void scale_copy_bf16_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long long n, float scale) {
  SM_LOOP(i, n) y[i] = (bf16)((float)x[i] * scale);
}
int launch_residual_import(const void* src, int dtype, bf16* dst, int B, int C, long long HW, float scale,
                           hipStream_t s) {
  const long long n = (long long)B * C * HW;
  if (dtype == 0) hipLaunchKernelGGL(nchw_to_nhwc_scaled_kernel<float>, dim3(sm_grid(n)), dim3(256), 0, s, (const float*)src, dst, C, HW, n, scale);
  else if (dtype == 1) hipLaunchKernelGGL(nchw_to_nhwc_scaled_kernel<bf16>, dim3(sm_grid(n)), dim3(256), 0, s, (const bf16*)src, dst, C, HW, n, scale);
  else if (dtype == 2) hipLaunchKernelGGL(scale_copy_bf16_kernel, dim3(sm_grid(n)), dim3(256), 0, s, (const bf16*)src, dst, n, scale);
  else SHAPECHK(false, "residual import: dtype %d (0 fp32 NCHW, 1 bf16 NCHW, 2 bf16 NHWC)", dtype);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- VAE encoder tail (train_sdxl_zh.py:306-309): moments = quant_conv(h) (1x1, C2 = 2*latent channels <= 8);
// mean, logvar = chunk(moments); logvar clamped to [-30, 20]; latents = (mean + exp(0.5 logvar) * noise) * scaling
// (diffusers 0.23 DiagonalGaussianDistribution [ext]).  NCHW fp32; one thread per pixel.
__global__ void vae_posterior_kernel(const float* __restrict__ h, const float* __restrict__ wq,
                                     const float* __restrict__ bq, const float* __restrict__ noise,
                                     float* __restrict__ moments, float* __restrict__ latents, int B, int C2,
                                     long long HW, float scaling) {
  SM_LOOP(i, (long long)B * HW) {
    const long long b = i / HW, p = i - b * HW;
    float in[8], m[8];
    for (int c = 0; c < C2; ++c) in[c] = h[(b * C2 + c) * HW + p];
    for (int o = 0; o < C2; ++o) {
      float a = bq[o];
      for (int c = 0; c < C2; ++c) a += wq[o * C2 + c] * in[c];
      m[o] = a;
      if (moments) moments[(b * C2 + o) * HW + p] = a;
    }
    const int L = C2 / 2;
    if (latents)
      for (int c = 0; c < L; ++c) {
        const float lv = fminf(fmaxf(m[L + c], -30.f), 20.f);
        const float nz = noise ? noise[(b * L + c) * HW + p] : 0.f;
        latents[(b * L + c) * HW + p] = (m[c] + __expf(0.5f * lv) * nz) * scaling;
      }
  }
}
int launch_vae_posterior(const float* h, const float* wq, const float* bq, const float* noise, float* moments,
                         float* latents, int B, int C2, long long HW, float scaling, hipStream_t s) {
  SHAPECHK(C2 >= 2 && C2 <= 8 && C2 % 2 == 0, "vae posterior: %d moment channels", C2);
  hipLaunchKernelGGL(vae_posterior_kernel, dim3(sm_grid((long long)B * HW)), dim3(256), 0, s, h, wq, bq, noise, moments,
                     latents, B, C2, HW, scaling);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- row softmax of a materialised score matrix (VAE mid-block attention: ONE head of width 512 over H*W tokens,
// outside the flash kernel's head widths): s[r][:] = softmax(scale * s[r][:]) in place, bf16 storage, fp32 math.
// One workgroup per row; the row is read twice (max+sum pass with online rescale, then the write pass).
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16* __restrict__ s, int cols, int ld, float scale) {
  bf16* row = s + (long long)blockIdx.x * ld;
  const float k = scale * 1.4426950408889634f;
  float m = -INFINITY, l = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    const bf16x8 v = *(const bf16x8*)(row + c);
    float mx = m;
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[j] * k);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += __builtin_amdgcn_exp2f((float)v[j] * k - mx);
    l = l * __builtin_amdgcn_exp2f(m - mx) + acc;
    m = mx;
  }
  __shared__ float sm[4], sl[4];
  const float wm = wave_max(m);
  l *= (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - wm);
  const float wl = wave_sum(l);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = wm; sl[threadIdx.x >> 6] = wl; }
  __syncthreads();
  const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float Lsum = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) Lsum += sl[w] * __builtin_amdgcn_exp2f(sm[w] - M);
  const float inv = 1.0f / Lsum;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    bf16x8 v = *(const bf16x8*)(row + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)(__builtin_amdgcn_exp2f((float)v[j] * k - M) * inv);
    *(bf16x8*)(row + c) = v;
  }
}
int launch_softmax_rows(bf16* s, long long rows, int cols, int ld, float scale, hipStream_t st) {
  SHAPECHK(cols % 8 == 0 && ld % 8 == 0 && rows > 0, "softmax_rows: cols=%d ld=%d", cols, ld);
  PROF_BEGIN(6, 0.0, 6.0 * rows * (double)cols, st);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, cols, ld, scale);
  PROF_END(st);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ---- VAE decode head: z' = post_quant_conv(latents * inv_scaling)  (1x1 conv over <= 8 channels), NCHW fp32
__global__ void vae_post_quant_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                      const float* __restrict__ b, float* __restrict__ out, int B, int C, long long HW,
                                      float inv_scaling) {
  SM_LOOP(i, (long long)B * HW) {
    const long long bb = i / HW, p = i - bb * HW;
    float in[8];
    for (int c = 0; c < C; ++c) in[c] = z[(bb * C + c) * HW + p] * inv_scaling;
    for (int o = 0; o < C; ++o) {
      float a = b[o];
      for (int c = 0; c < C; ++c) a += w[o * C + c] * in[c];
      out[(bb * C + o) * HW + p] = a;
    }
  }
}
int launch_vae_post_quant(const float* z, const float* w, const float* b, float* out, int B, int C, long long HW,
                          float inv_scaling, hipStream_t s) {
  SHAPECHK(C >= 1 && C <= 8, "vae post_quant: %d latent channels", C);
  hipLaunchKernelGGL(vae_post_quant_kernel, dim3(sm_grid((long long)B * HW)), dim3(256), 0, s, z, w, b, out, B, C, HW,
                     inv_scaling);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
