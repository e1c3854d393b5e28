// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
// Replaces the cuBLAS/cuDNN calls that diffusers' UNet2DConditionModel triggers from
// train_sdxl_zh.py:397,415 (Linear layers, ResBlock conv3x3, up/down-sample convs) and
// the adapter Linears of train_sdxl_zh.py:48-55.
//
// Structure (round 1): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as
// 2x2 v_mfma_f32_32x32x16_bf16; both operands K-contiguous ("NT"), staged global->LDS with
// global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction) into a 2-stage ring;
// the LDS image is linear, the XOR swizzle (chunk ^ ((row>>1)&7)) is applied to the SOURCE
// address and to the ds_read_b128 address, which makes the 32x32x16 fragment reads of
// 128-byte rows bank-conflict free.  The MFMA is issued with weights as the A operand
// and activations as the B operand, so each lane ends with 4 consecutive output columns
// of one output row -> 8-byte packed bf16 stores and vector bias/residual loads.
// Workgroup ids are remapped XCD-aware (blocks b, b+8 share an L2) and grouped 8 M-tiles
// per N-tile so the 64 tiles resident on one XCD share operand panels.
#include "pea_kernels.h"

#define BK 64

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt literal");
}

// ---- epilogue shared by both kernels.  acc[ni][mi][4g+j] = D[n = 8g + 4h + j][m = lane&31]
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[NI][MI], int m_base, int n_base, int frow,
                                              int fh) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_base + mi * 32 + frow;
    if (m >= p.M) continue;
    const int bidx = p.rowvec ? m / p.rows_per_batch : 0;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n_base + ni * 32 + 8 * g + 4 * fh;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j] * p.alpha;
        if (p.bias) {
          const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += b[j];
        }
        if (p.rowvec) {
          const bf16x4 rv = *(const bf16x4*)(p.rowvec + (long long)bidx * p.ldrv + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
        }
        if (p.preact) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
          *(bf16x4*)(p.preact + (long long)m * p.ldpre + n) = o;
        }
        if (p.act == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        } else if (p.act == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = siluf_(v[j]);
        }
        if (p.res) {
          const bf16x4 rr = *(const bf16x4*)(p.res + (long long)m * p.ldres + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)rr[j];
        }
        if (p.out_f32) {
          float* cp = (float*)p.C + (long long)m * p.ldc + n;
          f32x4 o;
          if (p.accum_f32) {
            o = *(const f32x4*)cp;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v[j];
          }
          *(f32x4*)cp = o;
        } else {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
          *(bf16x4*)((bf16*)p.C + (long long)m * p.ldc + n) = o;
        }
      }
    }
  }
}

// Block tile BM x BN x 64, WM x WN waves (each (BM/WM) x (BN/WN), built from 32x32x16 MFMAs), S-stage LDS
// ring filled by LDS-DMA with a counted vmcnt: tile t+S-1 is issued while tile t is consumed, ONE raw
// s_barrier per K-step (it orders "tile t landed for every wave" and "everyone finished tile t-1").
template <int MODE, int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WM * WN;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int A_BYTES = BM * 128;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;     // 1-KiB LDS-DMA pieces per wave per tile
  constexpr int P = PA + PB;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile / wave split");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;

  // ---- XCD-aware, grouped tile mapping (bijective for any grid size)
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int GROUP = 8;
  const int per_group = GROUP * nbn;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP;
  const int gsize = min(nbm - first_m, GROUP);
  const int bm = first_m + (bid % per_group) % gsize;
  const int bn = (bid % per_group) / gsize;

  // ---- per-thread staging descriptors: PA A rows + PB W rows, one 16-byte chunk each
  const int lrow = lane >> 3;              // row within an 8-row LDS-DMA piece
  const int cpos = lane & 7;               // chunk position inside the LDS row
  const bf16* a_src[PA];
  int a_iy0[PA], a_ix0[PA];                // conv: virtual-input origin of the 3x3 window
  const bf16* w_src[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int r = (wave * PA + j) * 8 + lrow;
    const int chunk = cpos ^ ((r >> 1) & 7);            // source chunk that lands at position cpos
    int gm = bm * BM + r;
    gm = gm < p.M ? gm : p.M - 1;
    if (MODE == 0) {
      a_src[j] = p.A + (long long)gm * p.lda + chunk * 8;
      a_iy0[j] = a_ix0[j] = 0;
    } else {
      const int hw = p.Ho * p.Wo;
      const int b = gm / hw;
      const int rem = gm - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[j] = oy * p.stride - 1;
      a_ix0[j] = ox * p.stride - 1;
      a_src[j] = p.A + (long long)b * p.Hs * p.Ws * p.Cin + chunk * 8;
    }
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int r = (wave * PB + j) * 8 + lrow;
    const int chunk = cpos ^ ((r >> 1) & 7);
    int gn = bn * BN + r;
    gn = gn < p.N ? gn : p.N - 1;
    w_src[j] = p.W + (long long)gn * p.ldw + chunk * 8;
  }
  const int Hv = p.Hs << p.shift, Wv = p.Ws << p.shift;

  auto stage = [&](int st, int k0) {
    char* base = smem + st * STAGE;
    int ky = 0, kx = 0, c0 = 0;
    if (MODE == 1) {
      const int tap = k0 / p.Cin;
      c0 = k0 - tap * p.Cin;
      ky = tap / 3;
      kx = tap - ky * 3;
    }
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const bf16* src;
      if (MODE == 0) {
        src = a_src[j] + k0;
      } else {
        const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
        bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        if (p.parity) ok = ok && (((iy | ix) & 1) == 0);
        const int sy = iy >> p.shift, sx = ix >> p.shift;
        src = ok ? a_src[j] + ((long long)sy * p.Ws + sx) * p.Cin + c0 : p.zeros;
      }
      __builtin_amdgcn_global_load_lds(PEA_GLB(src), PEA_LDS(base + (wave * PA + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j)
      __builtin_amdgcn_global_load_lds(PEA_GLB(w_src[j] + k0), PEA_LDS(base + A_BYTES + (wave * PB + j) * 1024), 16,
                                       0, 0);
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = p.K / BK;
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (i < nt) stage(i, i * BK);

  const int frow = lane & 31, fh = lane >> 5;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    // tile t must have landed: tiles t+1 .. t+S-2 (P loads each) may stay in flight
    if (t + S - 2 < nt) wait_vmcnt<(S - 2) * P>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + S - 1 < nt) {
      int nb = cur + S - 1;
      nb = nb >= S ? nb - S : nb;
      stage(nb, (t + S - 1) * BK);
    }
    const char* As = smem + cur * STAGE;
    const char* Ws = As + A_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8*)(As + swz_off(wr * (BM / WM) + mi * 32 + frow, 2 * s + fh));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8*)(Ws + swz_off(wc * (BN / WN) + ni * 32 + frow, 2 * s + fh));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    cur = cur + 1 == S ? 0 : cur + 1;
  }

  gemm_epilogue<MI, NI>(p, acc, bm * BM + wr * (BM / WM), bn * BN + wc * (BN / WN), frow, fh);
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined kernel.  Same tiling / LDS image as above, but inside a K-step (4 k16 sub-steps):
//   * fragments are double-buffered in registers: sub-step s issues the ds_reads of sub-step s+1
//     (or of the next tile's sub-step 0) before its own MFMAs, so LDS latency hides under MFMA time;
//   * the LDS-DMA pieces of tile t+S-1 are spread over sub-steps 0..2 instead of being issued in one
//     burst, so their issue slots sit behind already-queued MFMAs of the same wave;
//   * the single barrier of the K-step sits before sub-step 3's prefetch of the NEXT tile: "tile t+1
//     landed for every wave" and "every wave has issued all its reads of tile t-1 / t".
// BN may be 160 (5 x 32): every SDXL width is a multiple of 160, so 128x160 / 256x160 tiles cover the
// M = 4096 / 8192 GEMMs of the 32^2 level with exactly 256 workgroups (one per CU).
template <int MODE, int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(WM * WN * 64) void gemm_pipe_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WM * WN;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int A_BYTES = BM * 128;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;
  constexpr int PP = PA + PB;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile / wave split");
  static_assert((BM / WM) % 32 == 0 && (BN / WN) % 32 == 0, "wave tile");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int GROUP = 8;
  const int per_group = GROUP * nbn;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP;
  const int gsize = min(nbm - first_m, GROUP);
  const int bm = first_m + (bid % per_group) % gsize;
  const int bn = (bid % per_group) / gsize;

  const int lrow = lane >> 3, cpos = lane & 7;
  const bf16* a_src[PA];
  int a_iy0[PA], a_ix0[PA];
  const bf16* w_src[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int r = (wave * PA + j) * 8 + lrow;
    const int chunk = cpos ^ ((r >> 1) & 7);
    int gm = bm * BM + r;
    gm = gm < p.M ? gm : p.M - 1;
    if (MODE == 0) {
      a_src[j] = p.A + (long long)gm * p.lda + chunk * 8;
      a_iy0[j] = a_ix0[j] = 0;
    } else {
      const int hw = p.Ho * p.Wo;
      const int b = gm / hw;
      const int rem = gm - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[j] = oy * p.stride - 1;
      a_ix0[j] = ox * p.stride - 1;
      a_src[j] = p.A + (long long)b * p.Hs * p.Ws * p.Cin + chunk * 8;
    }
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int r = (wave * PB + j) * 8 + lrow;
    const int chunk = cpos ^ ((r >> 1) & 7);
    int gn = bn * BN + r;
    gn = gn < p.N ? gn : p.N - 1;
    w_src[j] = p.W + (long long)gn * p.ldw + chunk * 8;
  }
  const int Hv = p.Hs << p.shift, Wv = p.Ws << p.shift;

  // issue the LDS-DMA pieces [j0, j1) of the tile whose K offset is k0 into ring slot `st`
  auto issue = [&](int st, int k0, int j0, int j1) {
    char* base = smem + st * STAGE;
    int ky = 0, kx = 0, c0 = 0;
    if (MODE == 1) {
      const int tap = k0 / p.Cin;
      c0 = k0 - tap * p.Cin;
      ky = tap / 3;
      kx = tap - ky * 3;
    }
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      if (j < j0 || j >= j1) continue;
      if (j < PA) {
        const bf16* src;
        if (MODE == 0) {
          src = a_src[j] + k0;
        } else {
          const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
          bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
          if (p.parity) ok = ok && (((iy | ix) & 1) == 0);
          const int sy = iy >> p.shift, sx = ix >> p.shift;
          src = ok ? a_src[j] + ((long long)sy * p.Ws + sx) * p.Cin + c0 : p.zeros;
        }
        __builtin_amdgcn_global_load_lds(PEA_GLB(src), PEA_LDS(base + (wave * PA + j) * 1024), 16, 0, 0);
      } else {
        const int jb = j - PA;
        __builtin_amdgcn_global_load_lds(PEA_GLB(w_src[jb] + k0), PEA_LDS(base + A_BYTES + (wave * PB + jb) * 1024),
                                         16, 0, 0);
      }
    }
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = p.K / BK;
  const int frow = lane & 31, fh = lane >> 5;
  const int a_row0 = wr * (BM / WM) + frow, w_row0 = wc * (BN / WN) + frow;
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (i < nt) issue(i, i * BK, 0, PP);
  if (nt >= S - 1) wait_vmcnt<(S - 2) * PP>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][MI], wf[2][NI];
  auto load_frags = [&](int which, const char* tile, int s) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[which][mi] = *(const bf16x8*)(tile + swz_off(a_row0 + mi * 32, 2 * s + fh));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      wf[which][ni] = *(const bf16x8*)(tile + A_BYTES + swz_off(w_row0 + ni * 32, 2 * s + fh));
  };
  load_frags(0, smem, 0);

  constexpr int J1 = (PP + 2) / 3, J2 = (2 * PP + 2) / 3;     // piece ranges of the three issue slots
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* tile = smem + cur * STAGE;
    int nxt = cur + 1 == S ? 0 : cur + 1;
    int refill = cur == 0 ? S - 1 : cur - 1;                   // ring slot of tile t-1 == slot of tile t+S-1
    const bool more = t + S - 1 < nt;
    const int kf = (t + S - 1) * BK;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s == 3) {
        if (t + 1 < nt) {
          if (more) wait_vmcnt<(S - 2) * PP>();
          else wait_vmcnt<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          load_frags((s + 1) & 1, smem + nxt * STAGE, 0);
        }
      } else {
        load_frags((s + 1) & 1, tile, s + 1);
      }
      if (more) {
        if (s == 0) issue(refill, kf, 0, J1);
        else if (s == 1) issue(refill, kf, J1, J2);
        else if (s == 2) issue(refill, kf, J2, PP);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s & 1][ni], af[s & 1][mi], acc[ni][mi], 0, 0, 0);
    }
    cur = nxt;
  }
  gemm_epilogue<MI, NI>(p, acc, bm * BM + wr * (BM / WM), bn * BN + wc * (BN / WN), frow, fh);
}

template <int MODE, int BM, int BN, int WM, int WN, int S>
static int launch_pipe(const GemmP& p, hipStream_t stream) {
  constexpr int lds = S * (BM + BN) * 128;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)gemm_pipe_kernel<MODE, BM, BN, WM, WN, S>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
  hipLaunchKernelGGL((gemm_pipe_kernel<MODE, BM, BN, WM, WN, S>), dim3(grid), dim3(WM * WN * 64), lds, stream, p);
  return PEA_OK;
}

// ---- variant table (tile shape x wave grid x ring depth); the launcher picks one per problem shape
int g_gemm_variant = -1;   // >= 0: forced (benchmark / debug)
extern "C" void pea_debug_set_gemm_variant(int v) { g_gemm_variant = v; }

template <int MODE, int BM, int BN, int WM, int WN, int S>
static int launch_variant(const GemmP& p, hipStream_t stream) {
  constexpr int lds = S * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)gemm_bf16_kernel<MODE, BM, BN, WM, WN, S>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
  hipLaunchKernelGGL((gemm_bf16_kernel<MODE, BM, BN, WM, WN, S>), dim3(grid), dim3(WM * WN * 64), lds, stream, p);
  return PEA_OK;
}

#define GEMM_VARIANTS(MODE)                                              \
  switch (v) {                                                           \
    case 0: rc = launch_variant<MODE, 128, 128, 2, 2, 2>(p, stream); break; \
    case 1: rc = launch_variant<MODE, 128, 128, 2, 2, 3>(p, stream); break; \
    case 2: rc = launch_variant<MODE, 128, 128, 2, 2, 4>(p, stream); break; \
    case 3: rc = launch_variant<MODE, 256, 128, 4, 2, 2>(p, stream); break; \
    case 4: rc = launch_variant<MODE, 256, 128, 4, 2, 3>(p, stream); break; \
    case 5: rc = launch_variant<MODE, 256, 256, 2, 4, 2>(p, stream); break; \
    case 6: rc = launch_variant<MODE, 128, 256, 2, 4, 3>(p, stream); break; \
    case 7: rc = launch_variant<MODE, 256, 128, 2, 2, 3>(p, stream); break; \
    case 8: rc = launch_pipe<MODE, 128, 160, 4, 1, 4>(p, stream); break; \
    case 9: rc = launch_pipe<MODE, 256, 160, 4, 1, 3>(p, stream); break; \
    case 10: rc = launch_pipe<MODE, 256, 128, 4, 2, 3>(p, stream); break; \
    case 11: rc = launch_pipe<MODE, 128, 128, 2, 2, 3>(p, stream); break; \
    case 12: rc = launch_pipe<MODE, 256, 256, 2, 4, 2>(p, stream); break; \
    case 13: rc = launch_pipe<MODE, 128, 160, 4, 1, 3>(p, stream); break; \
    default: rc = launch_variant<MODE, 128, 128, 2, 2, 2>(p, stream); break; \
  }

static int pick_variant(const GemmP& p) {
  if (g_gemm_variant >= 0) return g_gemm_variant;
  // measured on the step's shapes with scripts/gemm_bench.py (profiles/r01_gemm_variants.log):
  //   13 = pipelined 128x160, 4 waves, 3 stages     9 = pipelined 256x160, 4 waves, 3 stages
  //   10 = pipelined 256x128, 8 waves, 3 stages    12 = pipelined 256x256, 8 waves, 2 stages
  //   11 = pipelined 128x128, 4 waves, 3 stages     5 = 256x256, 8 waves, 2 stages (plain loop)
  if (p.mode == 1) {
    if (p.N <= 384) return p.K >= 5760 ? 9 : 10;   // 128^2-level convs (N = 320)
    if (p.M >= 16384) return 9;                     // 64^2-level convs (N = 640)
    return 13;                                      // 32^2-level convs (M = 4096, N = 1280)
  }
  if (p.M < 1024) return 11;                        // cross-attention K|V projections, embeddings, adapter
  if (p.N <= 1280) return 13;                       // every N in {640, 1280}: 160-wide tiles fill the chip exactly
  if (p.M >= 8192) return 12;
  return p.N >= 8192 ? 10 : 5;
}

int launch_gemm(const GemmP& p, hipStream_t stream) {
  SHAPECHK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  SHAPECHK(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
  SHAPECHK(p.N % 4 == 0, "gemm: N=%d must be a multiple of 4", p.N);
  SHAPECHK(p.ldw % 8 == 0 && p.ldc % 4 == 0, "gemm: ldw=%d ldc=%d alignment", p.ldw, p.ldc);
  if (p.mode == 0) {
    SHAPECHK(p.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", p.lda);
  } else {
    SHAPECHK(p.Cin % BK == 0 && p.K == 9 * p.Cin, "conv: Cin=%d must be a multiple of %d and K=9*Cin", p.Cin, BK);
    SHAPECHK(p.zeros != nullptr, "conv: zero page missing");
    SHAPECHK(p.M % (p.Ho * p.Wo) == 0, "conv: M=%d not a multiple of Ho*Wo", p.M);
  }
  {
    // algorithmic work: 2*M*N*K; a transposed (zero-stuffed) conv only has 1/4 of its taps real
    double fl = 2.0 * p.M * (double)p.N * p.K * (p.parity ? 0.25 : 1.0);
    double by = 2.0 * ((double)p.M * p.N + (double)p.N * p.K + (p.mode ? (double)p.M * p.K / 9.0 : (double)p.M * p.K));
    PROF_BEGIN(p.mode ? 1 : 0, fl, by, stream);
  }
  const int v = pick_variant(p);
  int rc = PEA_OK;
  if (p.mode == 0) { GEMM_VARIANTS(0) } else { GEMM_VARIANTS(1) }
  PROF_END(stream);
  if (rc != PEA_OK) return rc;
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// ------------------------------------------------------------------------------------------
// 4-channel ends of the UNet (conv_in: K = 36, conv_out: N = 4): negligible FLOPs, direct form.
// conv_in (Cin = 4, K = 36): weights transposed into LDS [36][Cout] fp32; a thread owns 8 output channels and
// walks pixels; the 36 input taps are read once per pixel (L1-shared by the threads of that pixel).
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, bf16* __restrict__ y, int B,
                                                      int Cin, int H, int W, int Cout, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) char cism[];
  float* wl = (float*)cism;                       // [Cin*9][Cout]
  const int K = Cin * 9;
  for (int i = threadIdx.x; i < K * Cout; i += blockDim.x) {
    const int co = i / K, k = i - co * K;        // w[co][ci][ky][kx] -> k = ci*9 + ky*3 + kx
    wl[k * Cout + co] = w[i];
  }
  __syncthreads();
  const int nchunk = Cout / 8;
  const int ppb = blockDim.x / nchunk;
  const int ck = threadIdx.x % nchunk, pl = threadIdx.x / nchunk;
  if (pl >= ppb) return;
  const int c0 = ck * 8;
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = bias[c0 + j];
  const long long npix = (long long)B * H * W;
  const long long p0 = (long long)blockIdx.x * pix_per_block;
  const long long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  for (long long pix = p0 + pl; pix < p1; pix += ppb) {
    const int xw = (int)(pix % W), yh = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bs[j];
    for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = yh + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = xw + kx - 1;
          const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
          const float v = ok ? x[(((long long)b * Cin + ci) * H + iy) * W + ix] : 0.f;
          const float* wr = wl + (ci * 9 + ky * 3 + kx) * Cout + c0;
          const f32x4 w0 = *(const f32x4*)wr, w1 = *(const f32x4*)(wr + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j] += v * w0[j];
            acc[4 + j] += v * w1[j];
          }
        }
      }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
    *(bf16x8*)(y + pix * Cout + c0) = o;
  }
}

int launch_conv_in(const float* x, const float* w, const float* bias, bf16* y, int B, int Cin, int H, int W,
                   int Cout, hipStream_t s) {
  SHAPECHK(Cout % 8 == 0 && Cout / 8 <= 256 && Cin * 9 * Cout * 4 <= 64 * 1024, "conv_in: Cout=%d Cin=%d", Cout, Cin);
  const int nchunk = Cout / 8;
  const int ppb = 256 / nchunk;
  const int per = ppb * 8;
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)cdivl(npix, per)), dim3(256), (size_t)Cin * 9 * Cout * 4, s, x, w,
                     bias, y, B, Cin, H, W, Cout, per);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// conv_out: one wave per output pixel; lanes split the (tap, ci) reduction, 16-byte loads.
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       int B, int Cin, int H, int W, int Cout) {
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long long)B * H * W) return;
  const int xw = (int)(pix % W), yh = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // Cout <= 4
  const int cchunks = Cin / 8;
  for (int i = lane; i < 9 * cchunks; i += 64) {
    const int tap = i / cchunks, c8 = (i - tap * cchunks) * 8;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = yh + ky - 1, ix = xw + kx - 1;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
    const bf16x8 v = *(const bf16x8*)(x + (((long long)b * H + iy) * W + ix) * Cin + c8);
    for (int co = 0; co < Cout; ++co) {
      const float* wp = w + ((long long)co * 9 + tap) * Cin + c8;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a += (float)v[j] * wp[j];
      acc[co] += a;
    }
  }
  for (int co = 0; co < Cout; ++co) {
    const float r = wave_sum(acc[co]);
    if (lane == 0) y[(((long long)b * Cout + co) * H + yh) * W + xw] = r + bias[co];
  }
}

int launch_conv_out(const bf16* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                    int Cout, hipStream_t s) {
  SHAPECHK(Cout <= 4 && Cin % 8 == 0, "conv_out: Cout<=4, Cin%%8");
  const long long pix = (long long)B * H * W;
  hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)cdivl(pix, 4)), dim3(256), 0, s, x, w, bias, y, B, Cin, H, W,
                     Cout);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

// dgrad of conv_out: dx[b][y][x][ci] = sum_{co,ky,kx} dy[b][co][y+1-ky][x+1-kx] * w[co][ky][kx][ci]
__global__ void conv_out_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                      bf16* __restrict__ dx, int B, int Cin, int H, int W, int Cout) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = Cin / 8;
  const long long total = (long long)B * H * W * cq;
  if (idx >= total) return;
  const int c8 = (int)(idx % cq) * 8;
  const long long pix = idx / cq;
  const int xw = (int)(pix % W), yh = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int co = 0; co < Cout; ++co)
    for (int ky = 0; ky < 3; ++ky) {
      const int oy = yh + 1 - ky;
      if ((unsigned)oy >= (unsigned)H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ox = xw + 1 - kx;
        if ((unsigned)ox >= (unsigned)W) continue;
        const float g = dy[(((long long)b * Cout + co) * H + oy) * W + ox];
        const float* wp = w + ((long long)co * 9 + ky * 3 + kx) * Cin + c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += g * wp[j];
      }
    }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
  *(bf16x8*)(dx + pix * Cin + c8) = o;
}

int launch_conv_out_dgrad(const float* dy, const float* w, bf16* dx, int B, int Cin, int H, int W, int Cout,
                          hipStream_t s) {
  const long long total = (long long)B * H * W * (Cin / 8);
  hipLaunchKernelGGL(conv_out_dgrad_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, s, dy, w, dx, B, Cin,
                     H, W, Cout);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
