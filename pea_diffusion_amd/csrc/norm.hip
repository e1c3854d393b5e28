// GroupNorm(+SiLU) and LayerNorm, forward and data-gradient backward, NHWC / token-major bf16.
//
// Replaces torch's native_group_norm / native_layer_norm as triggered inside diffusers'
// ResnetBlock2D / Transformer2DModel / BasicTransformerBlock (called from
// train_sdxl_zh.py:397,415) and the adapter's nn.LayerNorm (train_sdxl_zh.py:47,60).
// HBM-bound: every pass moves 16 bytes per lane, statistics are reduced per thread (fp32),
// per block through LDS, and across blocks with fp64 atomics (one per group per block).
#include "pea_kernels.h"
#ifndef PEA_LN_WT
#define PEA_LN_WT 1      // LayerNorm outputs leave as write-through (sc1) stores: a streaming kernel's last ~20 MB otherwise sit dirty in
                         // the L2s until the end-of-kernel release writes them back with nothing else running (whole step, alternating
                         // processes on one box: 101.24 / 101.14 ms -> 101.02 / 100.89; the same in the one-tile GEMM epilogue, where all
                         // stores are one burst at the end, costs 1.8 ms: PEA_EPI_WT stays 0)
#endif

#define GN_MAX_GROUPS 64

// thread t -> channel chunk (t % nchunk) [8 channels], pixel lane (t / nchunk)
// grid: (blocks over pixels, B).  Deterministic (no atomics): per-thread sums -> LDS -> fixed-order sum over
// pixel lanes -> per-channel block sums -> fixed-order sum over each group's channels -> partial[b][blk][g][2];
// gn_finalize_kernel then adds the blocks in order (fp64).
template <int BWD>
__global__ void gn_stats_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ stats, float* __restrict__ partial, int HW, int C,
                                int groups, int pix_per_block, int silu) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  const int nchunk = C / 8;
  const int ppb = blockDim.x / nchunk;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  float* tsum = (float*)gsm;                 // [ppb][nchunk][16]
  float* csum = tsum + ppb * nchunk * 16;    // [2][C]
  const int ck = tid % nchunk, pl = tid / nchunk;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (pl < ppb) {
    const int c0 = ck * 8;
    float gm[8], bt[8], mean[8], rstd[8];
    if (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cpg;
        gm[j] = gamma[c0 + j];
        bt[j] = beta[c0 + j];
        mean[j] = stats[((long long)b * groups + g) * 2];
        rstd[j] = stats[((long long)b * groups + g) * 2 + 1];
      }
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, HW);
  #pragma unroll 4
  for (int p = p0 + pl; p < p1; p += ppb) {      // 4 pixels in flight per thread (latency-bound otherwise)
      const long long off = ((long long)b * HW + p) * C + c0;
      const bf16x8 xv = *(const bf16x8*)(x + off);
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (float)xv[j];
          s1[j] += v;
          s2[j] += v * v;
        }
      } else {
        const bf16x8 dv = *(const bf16x8*)(dy + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)xv[j] - mean[j]) * rstd[j];
          float d = (float)dv[j];
          if (silu) d *= silu_grad(xh * gm[j] + bt[j]);
          d *= gm[j];
          s1[j] += d;
          s2[j] += d * xh;
        }
      }
    }
    float* dst = tsum + ((long long)pl * nchunk + ck) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dst[j] = s1[j];
      dst[8 + j] = s2[j];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {           // per-channel sums over the pixel lanes, fixed order
    float a1 = 0.f, a2 = 0.f;
    const int k = c >> 3, j = c & 7;
    for (int l = 0; l < ppb; ++l) {
      a1 += tsum[((long long)l * nchunk + k) * 16 + j];
      a2 += tsum[((long long)l * nchunk + k) * 16 + 8 + j];
    }
    csum[c] = a1;
    csum[C + c] = a2;
  }
  __syncthreads();
  if (tid < groups) {
    float a1 = 0.f, a2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      a1 += csum[c];
      a2 += csum[C + c];
    }
    float* o = partial + (((long long)b * gridDim.x + blockIdx.x) * groups + tid) * 2;
    o[0] = a1;
    o[1] = a2;
  }
}

// one wave per (b, group): lane-strided sums over the block partials + shuffle tree (fixed order, fp64)
// FWD: writes (mean, rstd); BWD: writes (S1, S2) means
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                         int n_bg, int groups, int nblk, double count, float eps,
                                                         int bwd) {
  const int i = blockIdx.x;
  const int b = i / groups, g = i - b * groups;
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) {
    const float* p = partial + (((long long)b * nblk + k) * groups + g) * 2;
    s1 += (double)p[0];
    s2 += (double)p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (threadIdx.x != 0) return;
  if (bwd) {
    out[2 * i] = (float)(s1 / count);
    out[2 * i + 1] = (float)(s2 / count);
  } else {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0) var = 0;
    out[2 * i] = (float)m;
    out[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// apply passes: same (pixel-block, batch) geometry as the statistics pass; a thread keeps ONE channel chunk,
// so the per-channel affine constants are computed once and the pixel loop is 1 load + 8 FMAs + 1 store
__global__ void gn_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ stats,
                                bf16* __restrict__ y, int HW, int C, int groups, int silu, int pix_per_block) {
  const int nchunk = C / 8;
  const int ppb = blockDim.x / nchunk;
  const int ck = threadIdx.x % nchunk, pl = threadIdx.x / nchunk;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  const int c0 = ck * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int g = (c0 + j) / cpg;
    const float mean = stats[((long long)b * groups + g) * 2], rstd = stats[((long long)b * groups + g) * 2 + 1];
    sc[j] = rstd * gamma[c0 + j];
    sh[j] = beta[c0 + j] - mean * sc[j];
  }
  const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
#pragma unroll 4
  for (int p = p0 + pl; p < p1; p += ppb) {      // 4 pixels in flight per thread (latency-bound otherwise)
    const long long off = ((long long)b * HW + p) * C + c0;
    const bf16x8 xv = *(const bf16x8*)(x + off);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = (float)xv[j] * sc[j] + sh[j];
      if (silu) v = siluf_(v);
      o[j] = (bf16)v;
    }
    *(bf16x8*)(y + off) = o;
  }
}

__global__ void gn_bwd_apply_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ stats, const float* __restrict__ sums,
                                    bf16* dx, int HW, int C, int groups, int silu, const bf16* add,
                                    int pix_per_block) {
  const int nchunk = C / 8;
  const int ppb = blockDim.x / nchunk;
  const int ck = threadIdx.x % nchunk, pl = threadIdx.x / nchunk;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  const int c0 = ck * 8;
  float mean[8], rstd[8], gm[8], bt[8], S1[8], S2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long long sg = (long long)b * groups + (c0 + j) / cpg;
    mean[j] = stats[sg * 2]; rstd[j] = stats[sg * 2 + 1];
    S1[j] = sums[sg * 2]; S2[j] = sums[sg * 2 + 1];
    gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j];
  }
  const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
#pragma unroll 4
  for (int p = p0 + pl; p < p1; p += ppb) {      // 4 pixels in flight per thread (latency-bound otherwise)
    const long long off = ((long long)b * HW + p) * C + c0;
    const bf16x8 xv = *(const bf16x8*)(x + off);
    const bf16x8 dv = *(const bf16x8*)(dy + off);
    bf16x8 o;
    if (add) o = *(const bf16x8*)(add + off);          // may be dx itself (in-place accumulate)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)xv[j] - mean[j]) * rstd[j];
      float d = (float)dv[j];
      if (silu) d *= silu_grad(xh * gm[j] + bt[j]);
      d *= gm[j];
      float r = rstd[j] * (d - S1[j] - xh * S2[j]);
      if (add) r += (float)o[j];
      o[j] = (bf16)r;
    }
    *(bf16x8*)(dx + off) = o;
  }
}

// ---- one-kernel GroupNorm for slabs that fit on chip.  A workgroup owns one (sample, group) slab -- or GW = 2 adjacent
// groups when a group's channel run is not a whole number of VEC-element vectors (10 or 30 channels per group) -- and
// keeps it in registers: NPIX vectors per thread (thread -> vector vk of the slab's channel run, pixel lane pl; pixels
// pl, pl + ppb, ...).  Forward: mean, then the centred second moment (two-pass: the data is resident), normalise +
// affine (+ SiLU), store: one read + one write instead of the three launches / two reads of the general path.
// Backward: the two sums of the input gradient, then the apply, both from the resident (x, dy).  Reductions are
// fixed-order (wave butterfly, then the waves in index order in fp64).  Workgroups are numbered so that an XCD's L2
// sees whole samples (neighbouring groups share cache lines: a group's run is 20-160 bytes of each pixel row).
template <int K>
__device__ __forceinline__ void gn_block_sum(float* v, double (*red)[4], double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave][k] = (double)v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double a = 0.0;
    for (int w = 0; w < nw; ++w) a += red[w][k];
    out[k] = a;
  }
  __syncthreads();
}

// The slab is held as packed bf16 pairs (32-bit registers) and widened at each use; an empty asm between the phases keeps
// hipcc from carrying the widened fp32 copies of the whole slab across them (that spilled every instantiation).
__device__ __forceinline__ float gn_lo(uint32_t r) { return __builtin_bit_cast(float, r << 16); }
__device__ __forceinline__ float gn_hi(uint32_t r) { return __builtin_bit_cast(float, r & 0xffff0000u); }
__device__ __forceinline__ uint32_t gn_pack(float a, float b) {
  bf16x2 v;
  v[0] = (bf16)a; v[1] = (bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
template <int W> struct GnRaw;
template <> struct GnRaw<4> { typedef uint4 T; };
template <> struct GnRaw<2> { typedef uint2 T; };
template <int W>
__device__ __forceinline__ void gn_load(uint32_t (&r)[W], const bf16* p) {
  const typename GnRaw<W>::T v = *(const typename GnRaw<W>::T*)p;
  r[0] = v.x; r[1] = v.y;
  if constexpr (W == 4) { r[2] = v.z; r[3] = v.w; }
}
template <int W>
__device__ __forceinline__ void gn_store(bf16* p, const uint32_t (&r)[W]) {
  typename GnRaw<W>::T v;
  v.x = r[0]; v.y = r[1];
  if constexpr (W == 4) { v.z = r[2]; v.w = r[3]; }
  *(typename GnRaw<W>::T*)p = v;
}
template <int N, int W>
__device__ __forceinline__ void gn_pin(uint32_t (&r)[N][W]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int w = 0; w < W; ++w) asm volatile("" : "+v"(r[i][w]));
}

// orders the unrolled pixel iterations of the backward phases: the next pixel's registers become "new" values only once
// `dep` (a result of the current pixel) exists, so hipcc cannot widen / evaluate all pixels at once (register pressure)
template <int W>
__device__ __forceinline__ void gn_chain(float& dep, uint32_t (&a)[W], uint32_t (&b)[W]) {
  if constexpr (W == 4) asm volatile("" : "+v"(dep), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else asm volatile("" : "+v"(dep), "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}

template <int VEC, int NPIX, int BWD, int GW, int SILU>
__global__ __launch_bounds__((NPIX * VEC * (1 + BWD) > 128) ? 640 : 1024) void gn_fused_kernel(
    const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ stats, bf16* out, const bf16* add, int HW, int C, int groups,
    int ppb, float eps) {
  constexpr int W = VEC / 2;
  __shared__ double red[16][4];
  const int cpg = C / groups, nset = groups / GW, span = GW * cpg, nvec = span / VEC;
  const int total = gridDim.x, bid = blockIdx.x;
  const int idx = (total % 8 == 0) ? (bid % 8) * (total / 8) + bid / 8 : bid;       // XCD x owns a contiguous range of slabs
  const int b = idx / nset, gs = idx - b * nset;
  const int tid = threadIdx.x, vk = tid % nvec, pl0 = tid / nvec;
  const bool act = pl0 < ppb;                 // threads of the rounded-up last wave: read pixel lane 0, contribute nothing, store nothing
  const int pl = act ? pl0 : 0;
  const int c0 = gs * span + vk * VEC;                                              // first channel of this thread's vector
  // addresses = wave-uniform 64-bit base of (sample, pixel step i) + a per-thread 32-bit element offset: the per-pixel
  // addresses live in scalar registers (as 64-bit per-thread values they were CSE'd across the phases and spilled)
  const long long sbase = (long long)b * HW * C;
  const long long istep = (long long)ppb * C;
  const unsigned toff = (unsigned)pl * (unsigned)C + (unsigned)c0;
  bool hi[VEC];                                                                     // element belongs to the second group of the pair
#pragma unroll
  for (int j = 0; j < VEC; ++j) hi[j] = GW == 2 && (vk * VEC + j) >= cpg;
  uint32_t xr[NPIX][W];
  uint32_t dr[BWD ? NPIX : 1][W];
#pragma unroll
  for (int i = 0; i < NPIX; ++i) {
    gn_load<W>(xr[i], x + sbase + i * istep + toff);
    if (BWD) gn_load<W>(dr[i], dy + sbase + i * istep + toff);
  }
#define GN_X(i, j) (((j) & 1) ? gn_hi(xr[i][(j) >> 1]) : gn_lo(xr[i][(j) >> 1]))
#define GN_D(i, j) (((j) & 1) ? gn_hi(dr[i][(j) >> 1]) : gn_lo(dr[i][(j) >> 1]))
#define GN_SEL(j, a0, a1) ((GW == 2 && hi[j]) ? (a1) : (a0))
  const double count = (double)HW * cpg;
  const long long sg = (long long)b * groups + gs * GW;
  if (!BWD) {
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NPIX; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float v = GN_X(i, j);
        if (GW == 2 && hi[j]) s[1] += v; else s[0] += v;
      }
    if (!act) s[0] = s[1] = 0.f;
    double m[2];
    gn_block_sum<GW>(s, red, m);
    gn_pin(xr);
    const float mean0 = (float)(m[0] / count), mean1 = GW == 2 ? (float)(m[1] / count) : 0.f;
    s[0] = s[1] = 0.f;
#pragma unroll
    for (int i = 0; i < NPIX; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float d = GN_X(i, j) - GN_SEL(j, mean0, mean1);
        if (GW == 2 && hi[j]) s[1] += d * d; else s[0] += d * d;
      }
    if (!act) s[0] = s[1] = 0.f;
    double q[2];
    gn_block_sum<GW>(s, red, q);
    gn_pin(xr);
    const float rstd0 = (float)(1.0 / sqrt(q[0] / count + (double)eps));
    const float rstd1 = GW == 2 ? (float)(1.0 / sqrt(q[1] / count + (double)eps)) : 0.f;
    if (tid == 0) {
      stats[sg * 2] = mean0; stats[sg * 2 + 1] = rstd0;
      if (GW == 2) { stats[sg * 2 + 2] = mean1; stats[sg * 2 + 3] = rstd1; }
    }
    if (!act) return;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sc[j] = GN_SEL(j, rstd0, rstd1) * gamma[c0 + j];
      sh[j] = beta[c0 + j] - GN_SEL(j, mean0, mean1) * sc[j];
    }
#pragma unroll
    for (int i = 0; i < NPIX; ++i) {
      uint32_t o[W];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        float v0 = GN_X(i, 2 * w) * sc[2 * w] + sh[2 * w], v1 = GN_X(i, 2 * w + 1) * sc[2 * w + 1] + sh[2 * w + 1];
        if (SILU) { v0 = siluf_(v0); v1 = siluf_(v1); }
        o[w] = gn_pack(v0, v1);
      }
      gn_store<W>(out + sbase + i * istep + toff, o);
    }
  } else {
    const float mean0 = stats[sg * 2], rstd0 = stats[sg * 2 + 1];
    const float mean1 = GW == 2 ? stats[sg * 2 + 2] : 0.f, rstd1 = GW == 2 ? stats[sg * 2 + 3] : 0.f;
    float gm[VEC], bt[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j]; }
    float s[2 * GW];
#pragma unroll
    for (int k = 0; k < 2 * GW; ++k) s[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NPIX; ++i) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (GN_X(i, j) - GN_SEL(j, mean0, mean1)) * GN_SEL(j, rstd0, rstd1);
        float d = GN_D(i, j);
        if (SILU) d *= silu_grad(xh * gm[j] + bt[j]);
        d *= gm[j];
        if (GW == 2 && hi[j]) { s[1] += d; s[3] += d * xh; } else { s[0] += d; s[GW] += d * xh; }
      }
      if (i + 1 < NPIX) gn_chain<W>(s[GW], xr[i + 1], dr[i + 1]);
    }
    if (!act) {
#pragma unroll
      for (int k = 0; k < 2 * GW; ++k) s[k] = 0.f;
    }
    double r[2 * GW];
    gn_block_sum<2 * GW>(s, red, r);
    gn_pin(xr);
    gn_pin(dr);
    if (!act) return;
    const float S10 = (float)(r[0] / count), S20 = (float)(r[GW] / count);
    const float S11 = GW == 2 ? (float)(r[1] / count) : 0.f, S21 = GW == 2 ? (float)(r[2 * GW - 1] / count) : 0.f;
#pragma unroll
    for (int i = 0; i < NPIX; ++i) {
      uint32_t o[W];
      if (add) gn_load<W>(o, add + sbase + i * istep + toff);     // may be out itself (in-place accumulate)
#pragma unroll
      for (int w = 0; w < W; ++w) {
        float rr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * w + h;
          const float rs = GN_SEL(j, rstd0, rstd1);
          const float xh = (GN_X(i, j) - GN_SEL(j, mean0, mean1)) * rs;
          float d = GN_D(i, j);
          if (SILU) d *= silu_grad(xh * gm[j] + bt[j]);
          d *= gm[j];
          rr[h] = rs * (d - GN_SEL(j, S10, S11) - xh * GN_SEL(j, S20, S21));
          if (add) rr[h] += h ? gn_hi(o[w]) : gn_lo(o[w]);
        }
        o[w] = gn_pack(rr[0], rr[1]);
      }
      if (i + 1 < NPIX) {
        float dep = __builtin_bit_cast(float, o[W - 1]);
        gn_chain<W>(dep, xr[i + 1], dr[i + 1]);
        o[W - 1] = __builtin_bit_cast(uint32_t, dep);
      }
      gn_store<W>(out + sbase + i * istep + toff, o);
    }
  }
#undef GN_X
#undef GN_D
#undef GN_SEL
}

// which one-kernel form (if any) holds a [HW][C] sample's group slab in one workgroup's registers
struct GnFused { int vec, npix, gw, threads, ppb; };
static bool gn_fused_geometry(int HW, int C, int groups, int bwd, GnFused* f) {
  static const bool off = getenv("PEA_GN_UNFUSED") != nullptr;      // A/B switch
  if (off) return false;
  const int cpg = C / groups;
  if (cpg % 8 == 0) { f->vec = 8; f->gw = 1; }
  else if (cpg % 4 == 0) { f->vec = 4; f->gw = 1; }
  else if (cpg % 2 == 0 && groups % 2 == 0) { f->vec = 4; f->gw = 2; }
  else return false;
  const int nvec = f->gw * cpg / f->vec;
  for (int npix = 8; npix <= 32; npix *= 2) {
    if (HW % npix) continue;
    const int regs = npix * f->vec / 2 * (1 + bwd);                 // data registers per thread
    if (regs > 128) return false;
    const int maxt = regs > 64 ? 640 : 1024;
    const int thr = nvec * (HW / npix);
    if (thr > maxt) continue;
    f->npix = npix; f->ppb = HW / npix; f->threads = (thr + 63) / 64 * 64;
    return true;
  }
  return false;
}

template <int BWD>
static void gn_fused_launch(const GnFused& f, int blocks, hipStream_t s, const bf16* x, const bf16* dy, const float* gamma,
                            const float* beta, float* stats, bf16* out, const bf16* add, int HW, int C, int groups, float eps,
                            int silu) {
#define GN_S(V, N, G, S) hipLaunchKernelGGL((gn_fused_kernel<V, N, BWD, G, S>), dim3(blocks), dim3(f.threads), 0, s, x, dy, gamma, beta, \
                                            stats, out, add, HW, C, groups, f.ppb, eps)
#define GN_G(V, N, G) do { if (silu) GN_S(V, N, G, 1); else GN_S(V, N, G, 0); } while (0)
#define GN_F(V, N) do { if (f.gw == 2) GN_G(V, N, 2); else GN_G(V, N, 1); } while (0)
  if (f.vec == 8) {
    switch (f.npix) { case 8: GN_F(8, 8); break; case 16: GN_F(8, 16); break; default: GN_F(8, 32); break; }
  } else {
    switch (f.npix) { case 8: GN_F(4, 8); break; case 16: GN_F(4, 16); break; default: GN_F(4, 32); break; }
  }
#undef GN_S
#undef GN_G
#undef GN_F
}

static int gn_geometry(int HW, int C, int* threads, int* ppblk, int* nblk, size_t* lds) {
  const int nchunk = C / 8;
  if (nchunk > 1024) return -1;
  int ppb = 256 / nchunk;
  if (ppb < 1) ppb = 1;
  *threads = nchunk * ppb;
  int per = ppb * 16;
  if (per > HW) per = HW;
  *ppblk = per;
  *nblk = cdiv(HW, per);
  *lds = (size_t)(*threads) * 16 * 4 + (size_t)2 * C * 4;
  return 0;
}

// scratch layout (floats): partial [B][nblk][groups][2] | sums [B][groups][2]
size_t groupnorm_scratch_bytes(int B, int HW, int C, int groups) {
  int threads, ppblk, nblk;
  size_t lds;
  if (gn_geometry(HW, C, &threads, &ppblk, &nblk, &lds)) return 0;
  return ((size_t)B * nblk * groups * 2 + (size_t)B * groups * 2) * sizeof(float) + 256;
}

int launch_groupnorm_fwd(const bf16* x, const float* gamma, const float* beta, bf16* y, float* stats,
                         double* scratch, int B, int HW, int C, int groups, float eps, int silu, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C % groups == 0 && groups <= GN_MAX_GROUPS, "groupnorm: C=%d groups=%d", C, groups);
  int threads, ppblk, nblk;
  size_t lds;
  SHAPECHK(gn_geometry(HW, C, &threads, &ppblk, &nblk, &lds) == 0, "groupnorm: C=%d too wide", C);
  SHAPECHK(threads >= groups, "groupnorm: C=%d too narrow for %d groups", C, groups);
  float* partial = (float*)scratch;
  PROF_BEGIN(4, 0.0, 2.0 * 2.0 * B * (double)HW * C, s);           // algorithmic: read x, write y
  GnFused f;
  if (gn_fused_geometry(HW, C, groups, 0, &f)) {
    gn_fused_launch<0>(f, B * (groups / f.gw), s, x, nullptr, gamma, beta, stats, y, nullptr, HW, C, groups, eps, silu);
    PROF_END(s);
    HIPCHK(hipGetLastError());
    return PEA_OK;
  }
  hipLaunchKernelGGL(gn_stats_kernel<0>, dim3(nblk, B), dim3(threads), lds, s, x, nullptr, nullptr, nullptr, nullptr,
                     partial, HW, C, groups, ppblk, 0);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, s, partial, stats, B * groups, groups, nblk,
                     (double)HW * (C / groups), eps, 0);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, B), dim3(threads), 0, s, x, gamma, beta, stats, y, HW, C, groups,
                     silu, ppblk);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

int launch_groupnorm_bwd(const bf16* x, const bf16* dy, const float* gamma, const float* beta, const float* stats,
                         bf16* dx, double* scratch, int B, int HW, int C, int groups, int silu, const bf16* add,
                         hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C % groups == 0 && groups <= GN_MAX_GROUPS, "groupnorm: C=%d groups=%d", C, groups);
  int threads, ppblk, nblk;
  size_t lds;
  SHAPECHK(gn_geometry(HW, C, &threads, &ppblk, &nblk, &lds) == 0, "groupnorm: C=%d too wide", C);
  SHAPECHK(threads >= groups, "groupnorm: C=%d too narrow for %d groups", C, groups);
  float* partial = (float*)scratch;
  float* sums = partial + (size_t)B * nblk * groups * 2;
  PROF_BEGIN(4, 0.0, 2.0 * (add ? 4.0 : 3.0) * B * (double)HW * C, s);   // algorithmic: read x, dy (, the addend), write dx
  GnFused f;
  if (gn_fused_geometry(HW, C, groups, 1, &f)) {
    gn_fused_launch<1>(f, B * (groups / f.gw), s, x, dy, gamma, beta, const_cast<float*>(stats), dx, add, HW, C, groups, 0.f, silu);
    PROF_END(s);
    HIPCHK(hipGetLastError());
    return PEA_OK;
  }
  hipLaunchKernelGGL(gn_stats_kernel<1>, dim3(nblk, B), dim3(threads), lds, s, x, dy, gamma, beta, stats, partial, HW,
                     C, groups, ppblk, silu);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, s, partial, sums, B * groups, groups, nblk,
                     (double)HW * (C / groups), 0.f, 1);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nblk, B), dim3(threads), 0, s, x, dy, gamma, beta, stats, sums, dx, HW,
                     C, groups, silu, add, ppblk);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ------------------------------------------------------------------------------ LayerNorm
// one wave per row, the whole row lives in registers (C <= 4096 -> <= 8 chunks of 8 per lane)
#define LN_MAXCH 8

// A wave owns LN_NR consecutive rows, all of them loaded before the first reduction: twice the bytes in flight per
// wave (the kernels are latency-bound: 16 waves x 3 KB per CU did not cover the HBM latency), gamma / beta read
// once per wave as 16-byte vectors.
#define LN_NR 2
// STATS_ONLY: only (mean, rstd) per row -- the LayerNorms whose affine part is folded into the consuming GEMM
// (Tape::fold_ln: the normalised tensor is never written)
template <int NCH, bool STATS_ONLY = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y,
                                                     float* __restrict__ stats, int R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_NR;
  if (row0 >= R) return;
  const int nchunk = C / 8;
  bf16x8 v[LN_NR][NCH];
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) {
    const int row = min(row0 + r, R - 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ck = min(lane + 64 * i, nchunk - 1);          // (clamped: every request of the wave leaves in one batch, no branches)
      v[r][i] = *(const bf16x8*)(x + (long long)row * C + ck * 8);
    }
  }
  // gamma / beta requested with the rows (read behind the reductions they added a second memory round trip to the wave's
  // critical path)
  f32x4 g0[STATS_ONLY ? 1 : NCH], g1[STATS_ONLY ? 1 : NCH], b0[STATS_ONLY ? 1 : NCH], b1[STATS_ONLY ? 1 : NCH];
  if constexpr (!STATS_ONLY) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ck = min(lane + 64 * i, nchunk - 1);
      g0[i] = *(const f32x4*)(gamma + ck * 8); g1[i] = *(const f32x4*)(gamma + ck * 8 + 4);
      b0[i] = *(const f32x4*)(beta + ck * 8); b1[i] = *(const f32x4*)(beta + ck * 8 + 4);
    }
  }
  float mean[LN_NR], rstd[LN_NR];
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < nchunk)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[r][i][j];
    mean[r] = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < nchunk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)v[r][i][j] - mean[r];
          q += d * d;
        }
    rstd[r] = rsqrtf(wave_sum(q) / (float)C + eps);
    if (stats && lane == 0 && row0 + r < R) {
      stats[2 * (long long)(row0 + r)] = mean[r];
      stats[2 * (long long)(row0 + r) + 1] = rstd[r];
    }
  }
  if constexpr (STATS_ONLY) return;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nchunk) {
#pragma unroll
      for (int r = 0; r < LN_NR; ++r) {
        if (row0 + r >= R) break;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (bf16)(((float)v[r][i][j] - mean[r]) * rstd[r] * g0[STATS_ONLY ? 0 : i][j] + b0[STATS_ONLY ? 0 : i][j]);
          o[4 + j] = (bf16)(((float)v[r][i][4 + j] - mean[r]) * rstd[r] * g1[STATS_ONLY ? 0 : i][j] + b1[STATS_ONLY ? 0 : i][j]);
        }
        store16<PEA_LN_WT != 0>(y + (long long)(row0 + r) * C + ck * 8, o);
      }
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ stats, bf16* dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int R,
                                                     int C, const bf16* add) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_NR;
  if (row0 >= R) return;
  const int nchunk = C / 8;
  bf16x8 xv[LN_NR][NCH], dv[LN_NR][NCH], av[LN_NR][NCH];
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) {
    const int row = min(row0 + r, R - 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ck = min(lane + 64 * i, nchunk - 1);          // (clamped: all requests of the wave leave in one batch, no branches)
      xv[r][i] = *(const bf16x8*)(x + (long long)row * C + ck * 8);
      dv[r][i] = *(const bf16x8*)(dy + (long long)row * C + ck * 8);
      if (add) av[r][i] = *(const bf16x8*)(add + (long long)row * C + ck * 8);   // may be dx itself (in-place accumulate)
    }
  }
  // gamma and BOTH rows' statistics with the rows: requested behind the first row's arithmetic / store they put two more memory
  // round trips on the wave's critical path
  f32x4 g0[NCH], g1[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ck = min(lane + 64 * i, nchunk - 1);
    g0[i] = *(const f32x4*)(gamma + ck * 8); g1[i] = *(const f32x4*)(gamma + ck * 8 + 4);
  }
  float2 st[LN_NR];
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) st[r] = *(const float2*)(stats + 2 * (long long)min(row0 + r, R - 1));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) {
    const float mean = st[r].x, rstd = st[r].y;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < nchunk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)xv[r][i][j] - mean) * rstd;
          const float d = (float)dv[r][i][j] * (j < 4 ? g0[i][j] : g1[i][j - 4]);
          s1 += d;
          s2 += d * xh;
        }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    if (row0 + r >= R) break;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ck = lane + 64 * i;
      if (ck < nchunk) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)xv[r][i][j] - mean) * rstd;
          float rr = rstd * ((float)dv[r][i][j] * (j < 4 ? g0[i][j] : g1[i][j - 4]) - s1 - xh * s2);
          if (add) rr += (float)av[r][i][j];
          o[j] = (bf16)rr;
        }
        store16<PEA_LN_WT != 0>(dx + (long long)(row0 + r) * C + ck * 8, o);
      }
    }
  }
}

int launch_layernorm_fwd(const bf16* x, const float* gamma, const float* beta, bf16* y, float* stats, int R, int C,
                         float eps, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C <= 64 * 8 * LN_MAXCH, "layernorm: C=%d unsupported", C);
  PROF_BEGIN(5, 0.0, 4.0 * R * (double)C, s);
  const int nch = cdiv(C / 8, 64);          // 16-byte chunks per lane: the row lives in registers
#define LN_FWD(N) hipLaunchKernelGGL(ln_fwd_kernel<N>, dim3(cdiv(R, 4 * LN_NR)), dim3(256), 0, s, x, gamma, beta, y, stats, R, C, eps)
  if (nch <= 1) LN_FWD(1); else if (nch == 2) LN_FWD(2); else if (nch == 3) LN_FWD(3); else if (nch == 4) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// T5LayerNorm (mT5 student text tower, train_sdxl_zh.py:108-112): y = x * rsqrt(mean(x^2) + eps) * w -- no mean
// subtraction, no bias.  One wave per LN_NR rows, the rows live in registers (as ln_fwd_kernel).
template <int NCH>
__global__ __launch_bounds__(256) void rms_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                      bf16* __restrict__ y, int R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_NR;
  if (row0 >= R) return;
  const int nchunk = C / 8;
  bf16x8 v[LN_NR][NCH];
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) {
    const int row = min(row0 + r, R - 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ck = lane + 64 * i;
      if (ck < nchunk) v[r][i] = *(const bf16x8*)(x + (long long)row * C + ck * 8);
    }
  }
  float rinv[LN_NR];
#pragma unroll
  for (int r = 0; r < LN_NR; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < nchunk)
#pragma unroll
        for (int j = 0; j < 8; ++j) q += (float)v[r][i][j] * (float)v[r][i][j];
    rinv[r] = rsqrtf(wave_sum(q) / (float)C + eps);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nchunk) {
      const f32x4 g0 = *(const f32x4*)(gamma + ck * 8), g1 = *(const f32x4*)(gamma + ck * 8 + 4);
#pragma unroll
      for (int r = 0; r < LN_NR; ++r) {
        if (row0 + r >= R) break;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (bf16)((float)v[r][i][j] * rinv[r] * g0[j]);
          o[4 + j] = (bf16)((float)v[r][i][4 + j] * rinv[r] * g1[j]);
        }
        *(bf16x8*)(y + (long long)(row0 + r) * C + ck * 8) = o;
      }
    }
  }
}
int launch_rmsnorm_fwd(const bf16* x, const float* gamma, bf16* y, int R, int C, float eps, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C <= 64 * 8 * LN_MAXCH, "rmsnorm: C=%d unsupported", C);
  PROF_BEGIN(5, 0.0, 4.0 * R * (double)C, s);
  const int nch = cdiv(C / 8, 64);
#define RMS_FWD(N) hipLaunchKernelGGL(rms_fwd_kernel<N>, dim3(cdiv(R, 4 * LN_NR)), dim3(256), 0, s, x, gamma, y, R, C, eps)
  if (nch <= 1) RMS_FWD(1); else if (nch == 2) RMS_FWD(2); else if (nch == 3) RMS_FWD(3); else if (nch == 4) RMS_FWD(4); else RMS_FWD(8);
#undef RMS_FWD
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

int launch_layernorm_stats(const bf16* x, float* stats, int R, int C, float eps, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C <= 64 * 8 * LN_MAXCH, "layernorm: C=%d unsupported", C);
  PROF_BEGIN(5, 0.0, 2.0 * R * (double)C, s);
  const int nch = cdiv(C / 8, 64);
#define LN_ST(N) hipLaunchKernelGGL((ln_fwd_kernel<N, true>), dim3(cdiv(R, 4 * LN_NR)), dim3(256), 0, s, x, nullptr, nullptr, nullptr, stats, R, C, eps)
  if (nch <= 1) LN_ST(1); else if (nch == 2) LN_ST(2); else if (nch == 3) LN_ST(3); else if (nch == 4) LN_ST(4); else LN_ST(8);
#undef LN_ST
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// LayerNorm folded into the Linear that consumes it:  LN(x) . W^T + b  =  rstd (x . W'^T - mean s) + t  with
//   W'[n][k] = W[n][k] gamma[k] (bf16),   s[n] = sum_k W'[n][k],   t[n] = sum_k beta[k] W[n][k] + b[n].
// One wave per output row n; s is summed over the ROUNDED W' (what the MFMA multiplies).
__global__ __launch_bounds__(256) void ln_fold_kernel(const bf16* __restrict__ W, int ldw, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      bf16* __restrict__ Wf, float* __restrict__ svec,
                                                      float* __restrict__ tvec, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float sa = 0.f, ta = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const bf16x8 w = *(const bf16x8*)(W + (long long)n * ldw + k);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float wv = (float)w[j];
      o[j] = (bf16)(wv * gamma[k + j]);
      sa += (float)o[j];
      ta += beta[k + j] * wv;
    }
    *(bf16x8*)(Wf + (long long)n * K + k) = o;
  }
  sa = wave_sum(sa);
  ta = wave_sum(ta);
  if (lane == 0) {
    svec[n] = sa;
    tvec[n] = ta + (bias ? bias[n] : 0.f);
  }
}
int launch_ln_fold(const bf16* W, int ldw, const float* gamma, const float* beta, const float* bias, bf16* Wf, float* svec,
                   float* tvec, int N, int K, hipStream_t s) {
  SHAPECHK(K % 8 == 0 && ldw % 8 == 0, "ln fold: K=%d ldw=%d", K, ldw);
  hipLaunchKernelGGL(ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, ldw, gamma, beta, bias, Wf, svec, tvec, N, K);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// dgamma[c] += sum_r dy[r][c] * xhat[r][c]; dbeta[c] += sum_r dy[r][c].  A block owns 64 columns; its 16 waves take the rows
// round-robin (128-byte row segments per load, four rows in flight per wave) and the 16 partial sums are added in wave
// order: deterministic.  Only the adapter's LayerNorm has trainable affine parameters (R = 2*B*L rows); the one-thread-
// per-column form this replaces walked the 616 rows serially in 245 us.
__global__ __launch_bounds__(1024) void ln_param_grad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                             const float* __restrict__ stats, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int R, int C) {
  __shared__ float sg[16][64], sb[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float g = 0.f, b = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = w; r < R; r += 16) {
      const float d = (float)dy[(long long)r * C + c];
      g += d * ((float)x[(long long)r * C + c] - stats[2 * r]) * stats[2 * r + 1];
      b += d;
    }
  }
  sg[w][lane] = g;
  sb[w][lane] = b;
  __syncthreads();
  if (w == 0 && c < C) {
    float gs = 0.f, bs = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { gs += sg[k][lane]; bs += sb[k][lane]; }
    dgamma[c] += gs;
    dbeta[c] += bs;
  }
}

int launch_layernorm_bwd(const bf16* x, const bf16* dy, const float* gamma, const float* stats, bf16* dx,
                         float* dgamma, float* dbeta, int R, int C, const bf16* add, hipStream_t s) {
  SHAPECHK(C % 8 == 0 && C <= 64 * 8 * LN_MAXCH, "layernorm: C=%d unsupported", C);
  PROF_BEGIN(5, 0.0, 6.0 * R * (double)C, s);
  const int nch = cdiv(C / 8, 64);
#define LN_BWD(N) hipLaunchKernelGGL(ln_bwd_kernel<N>, dim3(cdiv(R, 4 * LN_NR)), dim3(256), 0, s, x, dy, gamma, stats, dx, dgamma, dbeta, R, C, add)
  if (dx) { if (nch <= 1) LN_BWD(1); else if (nch == 2) LN_BWD(2); else if (nch == 3) LN_BWD(3); else if (nch == 4) LN_BWD(4); else LN_BWD(8); }
#undef LN_BWD
  if (dgamma) hipLaunchKernelGGL(ln_param_grad_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, s, x, dy, stats, dgamma, dbeta, R, C);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
