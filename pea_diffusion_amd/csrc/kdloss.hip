// Fused knowledge-distillation loss: the 7 (SDXL) / 9 (SD1.5) masked feature-MSE terms, the
// masked noise-MSE and the masked teacher-logit MSE of train_sdxl_zh.py:399-441
// (train_sd_zh.py:217-276) in ONE launch, which also writes every gradient seed
// (dL/dF_S^k, dL/d eps_S) so the backward pass starts without another sweep.
//
// HBM-bound: per image 2 x 25.56 M tap elements read + 25.56 M written (bf16) for SDXL.
// 16 bytes per lane, fp32 per-thread accumulation, wave shuffle reduction, one partial per block summed
// in fixed order by the finishing kernel (no atomics: bit-reproducible).  Samples whose mask weight is 0 are not read at all (their seeds are zeros).
#include "pea_kernels.h"

#define KD_CHUNKS_PER_BLOCK 4096   // 16-byte chunks per block (64 KiB of each input)

struct KdSeg {
  long long blk0;        // first block of this segment
  long long nchunks;     // chunks in the segment (B * per / 8 for taps, B * per / 4 for eps)
};
struct KdSegs {
  KdSeg s[PEA_MAX_TAPS + 1];
};

// deterministic block reduction: wave shuffle tree + fixed-order sum of the 4 wave results
__device__ __forceinline__ float block_sum(float v) {
  __shared__ float wsum[4];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// partial: fp32 [nblocks][2] (one slot pair per block; no atomics -> bit-reproducible)
__global__ __launch_bounds__(256) void kd_loss_kernel(const KdLossP p, const KdSegs segs, float* partial) {
  int k = 0;
  while (k < p.ntaps && (long long)blockIdx.x >= segs.s[k + 1].blk0) ++k;
  const long long c0 = ((long long)blockIdx.x - segs.s[k].blk0) * KD_CHUNKS_PER_BLOCK;
  const long long c1 = min(c0 + KD_CHUNKS_PER_BLOCK, segs.s[k].nchunks);
  float r0 = 0.f, r1 = 0.f;
  if (k < p.ntaps) {
    const bf16* fs = p.fs[k];
    const bf16* ft = p.ft[k];
    bf16* dfs = p.dfs[k];
    const long long per8 = p.per[k] / 8;
    const float gsc = p.grad_scale * p.feat_weight * 2.0f / ((float)p.per[k] * (float)p.B);
    float acc = 0.f;
    for (long long c = c0 + threadIdx.x; c < c1; c += 256) {
      const int b = (int)(c / per8);
      const bool on = p.zh[b] == 0;          // weight (1 - zh_or_not)
      bf16x8 g;
      // a sample that carries KD weight but whose teacher row was not computed (tmap[b] < 0: the caller's live_teacher_mask
      // contradicts zh_or_not) has no teacher to compare with: its terms come out as NaN -- loss, seeds and with them the
      // adapter gradient -- instead of silently reading another sample's row
      const int trow = (on && p.tmap) ? p.tmap[b] : 0;
      if (on && trow >= 0) {
        const long long ct = p.tmap ? (long long)trow * per8 + (c - (long long)b * per8) : c;
        const bf16x8 a = *(const bf16x8*)(fs + c * 8);
        const bf16x8 t = *(const bf16x8*)(ft + ct * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)a[j] - (float)t[j];
          acc += d * d;
          g[j] = (bf16)(d * gsc);
        }
      } else if (on) {
        acc = __builtin_nanf("");
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (bf16)__builtin_nanf("");
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (bf16)0.f;
      }
      if (dfs) *(bf16x8*)(dfs + c * 8) = g;
    }
    r0 = block_sum(acc);
  } else {
    const long long per4 = p.per_eps / 4;
    const float gsc = p.grad_scale * 2.0f / ((float)p.per_eps * (float)p.B);
    float a0 = 0.f, a1 = 0.f;
    for (long long c = c0 + threadIdx.x; c < c1; c += 256) {
      const int b = (int)(c / per4);
      const bool zh = p.zh[b] != 0;
      const f32x4 es = *(const f32x4*)(p.eps_s + c * 4);
      const int trow = (!zh && p.tmap) ? p.tmap[b] : 0;
      const long long ct = (!zh && p.tmap && trow >= 0) ? (long long)trow * per4 + (c - (long long)b * per4) : c;
      const f32x4 other = zh ? *(const f32x4*)(p.eps + c * 4) : *(const f32x4*)(p.eps_t + ct * 4);
      f32x4 g;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = trow >= 0 ? es[j] - other[j] : __builtin_nanf("");     // (no teacher row for a sample with KD weight: see above)
        acc += d * d;
        g[j] = d * gsc;
      }
      if (zh) a0 += acc; else a1 += acc;
      if (p.deps_s) *(f32x4*)(p.deps_s + c * 4) = g;
    }
    r0 = block_sum(a0);
    r1 = block_sum(a1);
  }
  if (threadIdx.x == 0) {
    partial[2 * (long long)blockIdx.x] = r0;
    partial[2 * (long long)blockIdx.x + 1] = r1;
  }
}

// one block: sums each segment's block partials in fixed order (fp64), then composes the losses.
// losses[0..3] = total, train_loss, train_loss_logits, train_loss_features; skip[k] = 1 when the SD1.5
// NaN/Inf guard (train_sd_zh.py:246-268) drops tap k.
__global__ __launch_bounds__(256) void kd_finish_kernel(const KdLossP p, const KdSegs segs, const float* partial,
                                                        float* losses, int* skip) {
  __shared__ double red[256];
  __shared__ double seg_sum[PEA_MAX_TAPS + 2];
  for (int k = 0; k <= p.ntaps + 1; ++k) {
    const int seg = k <= p.ntaps ? k : p.ntaps;          // last two entries: eps slot 0 / slot 1
    const int slot = k == p.ntaps + 1 ? 1 : 0;
    const long long b0 = segs.s[seg].blk0;
    const long long b1 = seg < p.ntaps ? segs.s[seg + 1].blk0 : b0 + (segs.s[seg].nchunks + KD_CHUNKS_PER_BLOCK - 1) / KD_CHUNKS_PER_BLOCK;
    double a = 0.0;
    for (long long i = b0 + threadIdx.x; i < b1; i += 256) a += (double)partial[2 * i + slot];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) seg_sum[k] = red[0];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double nb = (double)p.per_eps * (double)p.B;
  const double l0 = seg_sum[p.ntaps] / nb, l1 = seg_sum[p.ntaps + 1] / nb;
  double lf = 0.0;
  for (int k = 0; k < p.ntaps; ++k) {
    const double t = seg_sum[k] / ((double)p.per[k] * (double)p.B);
    const bool bad = p.nan_guard && !isfinite(t);
    if (skip) skip[k] = bad ? 1 : 0;
    if (!bad) lf += t;
  }
  losses[1] = (float)l0;
  losses[2] = (float)l1;
  losses[3] = (float)lf;
  losses[0] = (float)(l0 + l1 + (double)p.feat_weight * lf);
}

__global__ void kd_zero_skipped_kernel(const KdLossP p, const int* skip) {
  const int k = blockIdx.y;
  if (!skip[k] || !p.dfs[k]) return;
  const long long n8 = (long long)p.B * p.per[k] / 8;
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x)
    *(bf16x8*)(p.dfs[k] + i * 8) = z;
}

static long long kd_total_blocks(int ntaps, const long long* per, long long per_eps, int B) {
  long long blk = 0;
  for (int k = 0; k < ntaps; ++k) blk += cdivl((long long)B * per[k] / 8, KD_CHUNKS_PER_BLOCK);
  blk += cdivl((long long)B * per_eps / 4, KD_CHUNKS_PER_BLOCK);
  return blk;
}
size_t kd_loss_workspace_bytes(int ntaps, const long long* per, long long per_eps, int B) {
  return (size_t)kd_total_blocks(ntaps, per, per_eps, B) * 2 * sizeof(float) + sizeof(int) * PEA_MAX_TAPS + 256;
}

int launch_kd_loss(const KdLossP& p, hipStream_t s) {
  SHAPECHK(p.ntaps >= 0 && p.ntaps <= PEA_MAX_TAPS, "kd_loss: ntaps=%d", p.ntaps);
  SHAPECHK(p.per_eps % 4 == 0, "kd_loss: per_eps %% 4");
  KdSegs segs;
  long long blk = 0;
  for (int k = 0; k < p.ntaps; ++k) {
    SHAPECHK(p.per[k] % 8 == 0, "kd_loss: tap %d size %% 8", k);
    segs.s[k].blk0 = blk;
    segs.s[k].nchunks = (long long)p.B * p.per[k] / 8;
    blk += cdivl(segs.s[k].nchunks, KD_CHUNKS_PER_BLOCK);
  }
  segs.s[p.ntaps].blk0 = blk;
  segs.s[p.ntaps].nchunks = (long long)p.B * p.per_eps / 4;
  blk += cdivl(segs.s[p.ntaps].nchunks, KD_CHUNKS_PER_BLOCK);
  // workspace: int skip[PEA_MAX_TAPS] | float partial[nblocks][2]   (kd_loss_workspace_bytes)
  int* skip = (int*)p.partial;
  float* partial = (float*)(skip + PEA_MAX_TAPS);
  {
    // bytes actually moved: a sample with zh_or_not == 0 reads both taps and writes its seed (3 x 2 B per element); a masked one
    // is not read at all, only its zero seed is written (2 B).  The mask lives on the device: the count is a caller's hint
    // (pea_trainer_set_option "kd_samples_hint", bench.py); without it every sample is counted as read (an upper bound).
    const int on = (p.kd_samples_hint >= 0 && p.kd_samples_hint <= p.B) ? p.kd_samples_hint : p.B;
    double by = 0;
    for (int k = 0; k < p.ntaps; ++k) by += 2.0 * (3.0 * on + 1.0 * (p.B - on)) * (double)p.per[k] * (p.dfs[k] ? 1.0 : 0.0)
                                            + (p.dfs[k] ? 0.0 : 2.0 * 2.0 * on * (double)p.per[k]);
    by += 4.0 * (p.deps_s ? 3.0 : 2.0) * p.B * (double)p.per_eps;                       // eps: two fp32 reads (+ the seed write)
    PROF_BEGIN(7, 0.0, by, s);
  }
  hipLaunchKernelGGL(kd_loss_kernel, dim3((unsigned)blk), dim3(256), 0, s, p, segs, partial);
  hipLaunchKernelGGL(kd_finish_kernel, dim3(1), dim3(256), 0, s, p, segs, partial, p.losses, skip);
  if (p.nan_guard && p.ntaps > 0)
    hipLaunchKernelGGL(kd_zero_skipped_kernel, dim3(256, p.ntaps), dim3(256), 0, s, p, skip);
  PROF_END(s);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
