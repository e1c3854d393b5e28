// Flash-style fused attention for head_dim 64 (SDXL: 10 heads x 4096 tokens, 20 heads x 1024
// tokens, cross-attention with <= 77 keys), forward + data-gradient backward.
//
// Replaces diffusers' AttnProcessor2_0 -> F.scaled_dot_product_attention inside the
// UNet calls of train_sdxl_zh.py:397,415 and its autograd backward.
//
// Forward / dQ kernels: a workgroup = 4 waves = 128 queries, each wave owns 32 queries.  The
// scores are computed TRANSPOSED, S^T = K . Q^T (v_mfma_f32_32x32x16_bf16 with K rows as the A
// operand), so a lane holds one query column and 32 keys in registers: the softmax row
// reduction is in-lane plus one cross-half shuffle, and the bf16-packed P^T accumulator is
// directly the B operand of O^T = V^T . P^T (no LDS round trip for P).  V^T fragments come from
// the row-major V tile in LDS through ds_read_b64_tr_b16 (hardware transpose read).
// dK/dV kernel: a wave owns 32 keys (key on the lane), loops over query tiles; P and dS
// accumulators feed dV^T = dO^T . P and dK^T = Q^T . dS directly; Q / dO tiles are read
// row-wise for S / dP and through the transpose read for the two gradient products.
// K/V (or Q/dO) tiles are 64 x 64 bf16 (128-byte rows) staged by LDS-DMA into a 2-deep ring
// with the same source-side XOR swizzle as gemm.hip.
// Cross-attention (<= 128 keys: the 77-token text context) has its own kernels with K / V resident in LDS for a
// workgroup's whole life: xattn_fwd_kernel; backward xattn_bwd3_kernel (<= 80 keys: S / dP / P / dS evaluated once by
// key-on-the-lane waves, dS handed to a dQ wave through an LDS image -- five products), xattn_bwd2_kernel (<= 96 keys:
// both orientations on specialised waves), xattn_bwd_kernel (round 3: any key count up to 128), each with an ordered
// fp32 split reduce over query ranges (attn_dkv_reduce_kernel).  Every output tile (O, dQ, dK, dV) leaves through a
// per-wave LDS image as whole 128-byte rows.
#include <stdlib.h>

#include "pea_kernels.h"

#define TILE_BYTES 8192   // 64 rows x 128 bytes
#ifndef PEA_ATTN_WT
#define PEA_ATTN_WT 0     // experiment: O / dQ rows leave as write-through (sc1) stores (see norm.hip: PEA_LN_WT)
#endif
#ifndef PEA_ATTN_DQ_READS_DELTA
#define PEA_ATTN_DQ_READS_DELTA 1   // fused backward: the dQ role reads (-delta, -lse log2 e) from attn_delta_kernel's output instead of recomputing delta from O
#endif
#ifndef PEA_ATTN_DKV_LDS_EPI
#define PEA_ATTN_DKV_LDS_EPI 1      // dK / dV leave through a per-wave LDS image as whole 128-byte rows (16 bytes per lane) instead of 8-byte pieces over 32 lines
#endif
#ifndef PEA_ATTN_BWD_PREFETCH
#define PEA_ATTN_BWD_PREFETCH 0   // 1: backward roles request each batch of LDS fragments one phase ahead of its MFMAs (A/B: 1 % slower, +40 registers)
#endif
// raw v_exp_f32: exp2f() expands to a denormal-safe 5-instruction sequence; every argument here is <= 0 (scores
// minus a running max / the log-sum-exp), so a flushed denormal result is an exact zero after bf16 rounding anyway
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#define LOG2E 1.4426950408889634f
// max over the two 32-lane halves of the wave (lane l and l ^ 32), in every lane: one v_permlane32_swap + one v_max.
// __shfl_xor(x, 32) goes through ds_bpermute_b32 -- six address instructions and an LDS round trip in the middle of the
// softmax's dependency chain (scores -> row max -> exp arguments), every key tile.  (The clang builtin folds
// max(swap(x, x)) away, hence the assembly; the s_nops are the data hazards of the swap.)
__device__ __forceinline__ float xhalf_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// The softmax scale rides in the RESIDENT operand: the fragments a workgroup keeps in registers for its whole life (the
// queries of the forward / dQ role, the keys of the dK/dV role) are multiplied once by scale * log2(e), so every score
// leaves the MFMA already in the log2 domain and the row constant (running max, log-sum-exp) is the MFMA's C operand:
// p = exp2(acc) with no per-score FMA.  These loops are bound by vector issue (profiles/EXPERIMENTS.md, attention): the
// forward goes from fma + exp + add + 3/4 max + 1/2 cvt per score to exp + add + 1/2 max + 1/2 cvt.
// AttnP::q_prescaled (the product path: the Q|K|V / to_q projection's epilogue applies the factor to its fp32 accumulator,
// GemmP::qscale): Q arrives scaled, nothing is rounded twice, and the dK/dV role -- which streams Q -- needs no scaled K
// (dK = ln2 * dS^T Q').  Otherwise (operator-level calls with a plain Q) the resident fragments are scaled here, which
// rounds them to bf16 once more: relative 2^-9 per element, i.e. a log2-score error of about 1e-3 for unit-variance scores
// and proportionally more for large logits.
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 f, float c) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)f[j] * c);
  return o;
}
// Running-max threshold of the forward (log2 units): the accumulated offset m_run of a query only moves when a tile's
// scores exceed it by more than this, so p <= 2^8 per score (fp32 sums and bf16's 8-bit mantissa keep their RELATIVE
// precision at any magnitude; nothing overflows) and the rescale branch -- subtract, rescale O and l, rebuild the C operand
// -- runs in the first tile and then almost never.  The result is exact for any threshold: O / l and lse do not depend on
// the offset used.  tests/test_ops_gpu.py forces the branch in a late tile (spiked key).
#define ATTN_MOVE_THR 8.0f

// XOR term of a tile row: the bits of (row >> 1) & 7 rotated so that rows r and r + 2 differ in bit 2.  Any bijection of
// (row >> 1) & 7 keeps the 16-byte row reads (ds_read_b128) conflict free; with this one the transposed reads
// (ds_read_b64_tr_b16: 32 lanes = 4 rows x 4 chunks x 8-byte halves) are conflict free too -- with the plain term rows r and
// r + 2 of such a read hit the same four chunks (2-way conflict: 25-32 % of the LDS cycles of these kernels were conflicts).
__device__ __forceinline__ int swz_x(int row) {
  const int x = (row >> 1) & 7;
  return ((x & 1) << 2) | (x >> 1);
}
__device__ __forceinline__ int swz_rc(int row, int col) {   // byte offset of element (row, col) in a tile
  return row * 128 + ((((col >> 3)) ^ swz_x(row)) << 4) + (col & 7) * 2;
}

// LDS-DMA issued through inline assembly.  With the builtin, hipcc's waitcnt pass knows that an LDS write is pending and puts
// `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint -- every ds_read_b64_tr_b16 (an intrinsic
// without a memory operand) -- i.e. in the middle of the very iteration whose compute is meant to hide the prefetch of the
// next tile (forward: before the P.V reads; dQ role: before the first read).  The asm form is invisible to that pass; the
// loops order DMA and reads themselves (`s_waitcnt vmcnt(0)` + barrier at the end of every iteration).
// vmcnt(0) as the BUILTIN (expcnt / lgkmcnt left at their maxima): the waitcnt pass must see it, or it keeps believing
// that the prologue's global loads are outstanding and re-waits for them with counted vmcnt(N) inside the loop -- which, with
// DMA pieces it does not know about in flight, waits for those pieces instead.
#define WAIT_VM0()                           \
  do {                                       \
    __builtin_amdgcn_s_waitcnt(0x0F70);      \
    asm volatile("" ::: "memory");           \
  } while (0)
__device__ __forceinline__ unsigned lds_addr_u32(const void* p) {
  return (unsigned)(unsigned long long)p;      // a flat pointer into LDS is (shared aperture base << 32) | LDS byte offset
}
__device__ __forceinline__ void lds_dma16(const void* gsrc, const char* lds_dst) {     // 64 lanes x 16 bytes -> lds_dst[0..1023]
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(a), "v"(gsrc) : "m0");
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, const char* lds_dst) {      // 64 lanes x 4 bytes -> lds_dst[0..255]
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(a), "v"(gsrc) : "m0");
}

// A 64x64-tile source for the LDS-DMA in BUFFER form (buffer_load_dwordx4 ... offen lds): a buffer resource over the
// matrix (rows beyond `rows` read as zeros -- they are masked keys / ragged query rows), and this lane's two byte offsets
// inside a tile; the tile's first row goes into the scalar offset.  The 64-bit address form cost ~10 integer vector
// instructions per 1 KiB piece (row clamp, swizzle, 64-bit multiply-add), every key tile, in loops whose bound is the
// vector issue port that the MFMAs share.
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct TileSrc {
  i32x4 rsrc;
  int voff[2];
  int row_bytes;
};
__device__ __forceinline__ TileSrc tile_src(const bf16* base, int ld, int rows, int wave, int lane) {
  TileSrc t;
  const unsigned long long a = (unsigned long long)base;
  t.rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  t.rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  t.rsrc[2] = __builtin_amdgcn_readfirstlane((rows - 1) * ld * 2 + 128);      // num_records (bytes): last valid row's 128 bytes
  t.rsrc[3] = 0x00020000;
  t.row_bytes = ld * 2;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + (lane >> 3);
    t.voff[j] = r * ld * 2 + (((lane & 7) ^ swz_x(r)) << 4);
  }
  return t;
}
// stage rows r0 .. r0+63 -> dst (LDS image of 64 rows x 128 bytes, source-side XOR swizzle); 4 waves x 2 pieces
__device__ __forceinline__ void stage_tile(const TileSrc& t, int r0, char* dst, int wave) {
  const int soff = __builtin_amdgcn_readfirstlane(r0 * t.row_bytes);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr_u32(dst + (wave * 2 + j) * 1024));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(a), "v"(t.voff[j]), "s"(t.rsrc), "s"(soff) : "m0");
  }
}

// MFMA A-operand fragment of T^T for the k-permuted accumulator-as-operand product:
// elements j=0..3 <- rows kbase+4h+j, j=4..7 <- rows kbase+8+4h+(j-4) of the LDS tile, column c0 + (lane&31)
template <bool USE_TR>
__device__ __forceinline__ bf16x8 read_transposed_frag(const char* tile, int kbase, int c0, int lane) {
  bf16x8 out;
  const int h = lane >> 5;
  if (USE_TR) {
    const int i = lane & 15;
    const int col = c0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    const int row = kbase + 4 * h + (i >> 2);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)PEA_LDS(tile + swz_rc(row, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)PEA_LDS(tile + swz_rc(row + 8, col)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      short a = lo[j], b = hi[j];
      out[j] = *(bf16*)&a;
      out[4 + j] = *(bf16*)&b;
    }
  } else {
    const int col = c0 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = kbase + 8 * (j >> 2) + 4 * h + (j & 3);
      out[j] = *(const bf16*)(tile + swz_rc(row, col));
    }
  }
  return out;
}

// the same fragment from precomputed per-lane byte offsets of its two 8-row halves (see attn_q_body)
__device__ __forceinline__ bf16x8 read_transposed_frag_at(const char* lo_addr, const char* hi_addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)PEA_LDS(lo_addr));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)PEA_LDS(hi_addr));
  bf16x8 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    short a = lo[j], b = hi[j];
    out[j] = *(bf16*)&a;
    out[4 + j] = *(bf16*)&b;
  }
  return out;
}

__device__ __forceinline__ bf16x8 read_row_frag(const char* tile, int row, int s, int h) {
  return *(const bf16x8*)(tile + row * 128 + ((((2 * s + h)) ^ swz_x(row)) << 4));
}

// XCD-aware workgroup order.  Workgroups are handed to the 8 XCDs round-robin in dispatch order (x fastest), so the
// query blocks of one head -- which all stream that head's K and V (or Q and dO) -- would land on 8 different L2s and
// every XCD would see every head: a working set of all heads per 4-MB L2.  Remapped, XCD x runs a contiguous range of
// (batch, head, block) triples, i.e. whole heads, and a head's operands are read into one L2 once.  `env
// PEA_ATTN_NO_XCD=1` restores the dispatch order (A/B).
__device__ __forceinline__ void attn_block_coords(int remap, int& bx, int& by, int& bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if (remap) {
    const unsigned total = gx * gy * gz, q = total >> 3, r = total & 7, xcd = lin & 7, j = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  bx = (int)(lin % gx);
  const unsigned t = lin / gx;
  by = (int)(t % gy);
  bz = (int)(t / gy);
}

// ============================================================================= forward / dQ
// MODE 0: forward (writes O, lse).  MODE 1: dQ (reads dO, lse, delta; writes dQ).
// ND = head_dim / 64 (heads are stored padded to ND*64 columns; the padding columns of Q/K/V are zero).  Every
// K/V (Q/dO) tile is kept as ND sub-tiles of 64 x 64 so all LDS images stay 128-byte-row images.
// MODE 0 produces all ND output chunks in one pass; MODE 1 produces ONE 64-wide chunk of dQ per workgroup
// (blockIdx.x enumerates query blocks x ND chunks) so its register budget does not grow with ND.
// WRITE_DELTA: the dQ pass also stores delta[q] for the dK/dV kernel launched behind it (two-launch form); the fused
// backward launch gets delta from attn_delta_kernel instead (both roles run concurrently there)
template <int MODE, bool USE_TR, int ND, bool TXT, bool WRITE_DELTA>
__device__ __forceinline__ void attn_q_body(const AttnP& p, char* smem, int blk_x, int head, int b) {
  constexpr int STG = 2 * ND * TILE_BYTES;                       // smem: [2 stages][K sub-tiles ND | V sub-tiles ND]
  constexpr int NO = MODE == 0 ? ND : 1;                         // output chunks held by this workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qblk = MODE == 0 ? blk_x : blk_x / ND;
  const int chunk = MODE == 0 ? 0 : blk_x % ND;                  // dQ output chunk
  const int q0 = qblk * 128 + wave * 32;
  const int frow = lane & 31, fh = lane >> 5;
  const float c = p.scale * LOG2E;

  const bf16* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * 64 * ND;
  const bf16* Kb = p.K + (long long)b * p.Skv * p.ldk + head * 64 * ND;
  const bf16* Vb = p.V + (long long)b * p.Skv * p.ldv + head * 64 * ND;

  int qrow = q0 + frow;
  const bool qvalid = qrow < p.Sq;
  qrow = qvalid ? qrow : p.Sq - 1;
  bf16x8 qf[ND][4], dof[MODE == 1 ? ND : 1][4];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[nd][s] = *(const bf16x8*)(Qb + (long long)qrow * p.ldq + nd * 64 + 16 * s + 8 * fh);
      if (!p.q_prescaled) qf[nd][s] = scale_frag(qf[nd][s], c);
    }
  float lse2 = 0.f, dlt = 0.f;
  if (MODE == 1) {
    const bf16* dOb = p.dO + (long long)b * p.Sq * p.lddo + head * 64 * ND;
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        dof[nd][s] = *(const bf16x8*)(dOb + (long long)qrow * p.lddo + nd * 64 + 16 * s + 8 * fh);
    const long long li = ((long long)b * p.H + head) * p.Sq + qrow;
    if (!WRITE_DELTA && PEA_ATTN_DQ_READS_DELTA) {
      // fused launch: attn_delta_kernel ran in front of this grid and left -delta[q] and -lse[q] * log2(e) for the dK/dV role;
      // the dQ role takes the same two floats instead of re-reading its O row (4 x 16 bytes per lane over 32 lines per
      // instruction) and redoing the 64 multiply-adds -- bit-identical inputs for both roles
      dlt = -p.delta[li];
      lse2 = -p.delta[(long long)p.B * p.H * p.Sq + li];
    } else {
      lse2 = p.lse[li] * LOG2E;
      // delta[q] = sum_d dO[q][d] * O[q][d], computed here from the dO fragments already in registers (one extra read of
      // the O row) instead of a separate kernel; the dK/dV kernel, launched after this one, reads it from p.delta
      const bf16* Ob = p.O + (long long)b * p.Sq * p.ldo + head * 64 * ND;
      float dsum = 0.f;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 o = *(const bf16x8*)(Ob + (long long)qrow * p.ldo + nd * 64 + 16 * s + 8 * fh);
#pragma unroll
          for (int j = 0; j < 8; ++j) dsum += (float)dof[nd][s][j] * (float)o[j];
        }
      { float da = dsum, db = dsum; swap_halves32(da, db); dsum = da + db; }
      dlt = dsum;
      if (WRITE_DELTA && qvalid && fh == 0 && chunk == 0) {       // row constants of the dK/dV kernel (see attn_delta_kernel)
        p.delta[li] = -dsum;
        p.delta[(long long)p.B * p.H * p.Sq + li] = -lse2;
      }
    }
  }
  // dP^T accumulates ON TOP OF -delta[q] (the MFMA's C operand: this lane's query, the same value in all 16 registers, for
  // every key tile), so dS^T = P^T (dP^T - delta) costs one multiply per score; the softmax scale is applied once, to the
  // finished dQ^T.  The backward loops are VALU-issue-bound (a probe with a 4-cycle stand-in for the 8-cycle v_exp_f32 ran
  // 11 % faster): per score fma + exp + mul + half a conversion instead of fma + exp + sub + mul + mul + conversion.
  // rowc: the C operand of the S^T products -- forward: -m_run (the accumulated offset of this lane's query, 0 before the
  // first tile); dQ role: -lse2 (so that P^T = exp2(S^T) directly)
  f32x16 negd, rowc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { negd[r] = -dlt; rowc[r] = MODE == 1 ? -lse2 : 0.f; }

  f32x16 oacc[2 * NO];
#pragma unroll
  for (int i = 0; i < 2 * NO; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = 0.f, l_run = 0.f;            // m_run: log2-domain offset the P values of this query are relative to
  [[maybe_unused]] bool seen = false;        // TXT instance: this lane's query has met a key that is not masked

  TileSrc ksrc[ND], vsrc[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) {
    ksrc[nd] = tile_src(Kb + nd * 64, p.ldk, p.Skv, wave, lane);
    vsrc[nd] = tile_src(Vb + nd * 64, p.ldv, p.Skv, wave, lane);
  }
  auto stage_kv = [&](char* dst, int r0) {
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) {
      stage_tile(ksrc[nd], r0, dst + nd * TILE_BYTES, wave);
      stage_tile(vsrc[nd], r0, dst + (ND + nd) * TILE_BYTES, wave);
    }
  };
  const int nt = (p.Skv + 63) / 64;
  // read ONCE, before the loop: a global load inside it makes the compiler wait for vmcnt(0) there, i.e. for the DMA
  // prefetch of the next tile as well
  const int skv_all = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.Skv);
  stage_kv(smem, 0);
  WAIT_VM0();
  __syncthreads();

  // Per-lane LDS byte offsets, computed once.  The swizzle term of a row is unchanged by +16 / +32 rows, so every fragment
  // address in the loop is (stage base) + (one of these eight per-lane constants) + (a compile-time offset): eight integer
  // adds per key tile instead of one address computation per ds_read (45 of the ~200 VALU instructions of a tile of the
  // VALU-bound forward loop were address arithmetic).
  int rf_off[4], tr_off[2][2];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) rf_off[s4] = frow * 128 + (((2 * s4 + fh) ^ swz_x(frow)) << 4);
  {
    const int i16 = lane & 15, rr = 4 * fh + (i16 >> 2), cc = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) tr_off[db][hl] = swz_rc(rr + 8 * hl, db * 32 + cc);
  }

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage_kv(smem + (cur ^ 1) * STG, (t + 1) * 64);
    const char* Ks = smem + cur * STG;
    const char* Vs = Ks + ND * TILE_BYTES;
    int rfc[4], trc[2][2];
    {
      const int sb = cur * STG;
      const int tb = sb + (MODE == 0 ? ND * TILE_BYTES : chunk * TILE_BYTES);   // fwd: V sub-tiles; dQ: this chunk's K sub-tile
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) rfc[s4] = rf_off[s4] + sb;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) trc[db][hl] = tr_off[db][hl] + tb;
    }

    // S^T[key][q] = K . Q^T  (sum over the ND sub-tiles of the head dimension).  All K fragments of a sub-tile are requested
    // before the first MFMA: read-wait-MFMA per fragment (what the compiler emits for the fused loop under a 128-register
    // budget) exposes one LDS latency per MFMA, i.e. the matrix pipe runs at a quarter of its rate in this phase.
    f32x16 sacc[2];
    // dQ role: the V fragments of the dP product are requested BEFORE the S MFMAs and the transposed K fragments of the dQ
    // product before the exponentials, so each batch of LDS reads lands under the work in front of it (two waves per SIMD
    // do not cover an LDS round trip per batch; the forward's three do, and it has no registers to spare)
    bf16x8 vfr0[2][4];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) {
      bf16x8 kfr[2][4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) kfr[kb][s] = *(const bf16x8*)(smem + rfc[s] + (nd * TILE_BYTES + kb * 4096));
      if (MODE == 1 && nd == ND - 1 && PEA_ATTN_BWD_PREFETCH) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int s = 0; s < 4; ++s) vfr0[kb][s] = *(const bf16x8*)(smem + rfc[s] + (ND * TILE_BYTES + kb * 4096));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][s], qf[nd][s], (nd == 0 && s == 0) ? rowc : sacc[kb], 0, 0, 0);
    }
    const int kv0 = t * 64;
    bf16x8 pf[4];   // P^T (fwd) or dS^T (dQ) as B-operand fragments, k-permuted
    bf16x8 tfr1[MODE == 1 ? 4 : 1][2];   // dQ role: transposed K fragments, requested early
    if (MODE == 0) {
      // keys beyond Skv are masked only in the tile that contains them.
      // TXT (text encoders, inference): causal mask and / or a per-sample key count (padding) -- a separate instance
      // so the UNet's kernel keeps its register budget
      int skv_b = p.Skv;
      if constexpr (TXT) skv_b = skv_all;
      if constexpr (TXT) {
        if (p.bias) {
          // additive score bias (T5 relative positions), given in the log2 domain with a row pitch of 64 * ceil(Skv / 64)
          const int pitch = ((p.Skv + 63) >> 6) << 6;
          const float* brow = p.bias + ((long long)head * p.Sq + qrow) * pitch + kv0 + 4 * fh;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 bv = *(const f32x4*)(brow + kb * 32 + 8 * g);
#pragma unroll
              for (int j = 0; j < 4; ++j) sacc[kb][4 * g + j] += bv[j];
            }
        }
      }
      const bool boundary = (TXT && p.causal) || kv0 + 64 > skv_b;   // wave-uniform
      if (boundary) {
        int kmax = skv_b;
        if constexpr (TXT) kmax = p.causal ? min(skv_b, qrow + 1) : skv_b;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            sacc[kb][r] = key < kmax ? sacc[kb][r] : -INFINITY;
          }
      }
      // online softmax in the log2 domain.  sacc holds (scaled score - m_run): the offset came in through the MFMA's C
      // operand, so p = exp2(sacc) as it stands.  Only when some query's tile maximum exceeds its offset by more than
      // ATTN_MOVE_THR (always in the first tile) does the wave take the rescale branch (exact for any threshold).
      float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);     // one v_max3 per pair of scores
      mx = xhalf_max(mx);
      // "first" = this query has not seen a valid key yet.  The UNet instance always has one in tile 0 (Skv >= 1, checked on the
      // host).  TXT: a per-sample key count of 0 (or a padded context shorter than a tile boundary) leaves whole tiles at -inf;
      // the offset then stays 0 -- with m_run = -1e30 the next tile's scores would be absorbed by the C operand and every key
      // would come out with weight 1 -- and is set by the first tile with a finite maximum (a row without any key ends as 0 / 0).
      bool first = t == 0;
      if constexpr (TXT) first = !seen;
      const bool moved = (first || mx > ATTN_MOVE_THR) && (!TXT || mx > -INFINITY);
      if constexpr (TXT) seen = seen || mx > -INFINITY;
      if (__any(moved)) {
        const float mrel = moved ? mx : 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[kb][r] -= mrel;
        const float alpha = first ? 0.f : fast_exp2(-mrel);      // O and l are still zero before the first valid key
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 2 * NO; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        m_run += mrel;
#pragma unroll
        for (int r = 0; r < 16; ++r) rowc[r] = -m_run;
      }
      float ls = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = fast_exp2(sacc[kb][r]);
          sacc[kb][r] = e;
          ls += e;
        }
      l_run += ls;
    } else {
      // P^T = exp2(S^T) (scale in Q, -lse2[q] in the C operand);  dP^T = V . dO^T;  dS^T = P^T (dP^T - delta[q]) scale
      const int skv_b = skv_all;                                 // per-sample valid keys (padded contexts)
      f32x16 dpacc[2];
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        bf16x8 vfr[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (nd == 0 && PEA_ATTN_BWD_PREFETCH) vfr[kb][s] = vfr0[kb][s];            // requested before the S products
            else vfr[kb][s] = *(const bf16x8*)(smem + rfc[s] + ((ND + nd) * TILE_BYTES + kb * 4096));
          }
        if (nd == ND - 1 && PEA_ATTN_BWD_PREFETCH) {           // the dQ product's transposed K fragments: under dP and the exponentials
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              if constexpr (USE_TR)
                tfr1[ks][db] = read_transposed_frag_at(smem + trc[db][0] + ks * 2048, smem + trc[db][1] + ks * 2048);
              else tfr1[ks][db] = read_transposed_frag<false>(Ks + chunk * TILE_BYTES, ks * 16, db * 32, lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
            dpacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kb][s], dof[MODE == 1 ? nd : 0][s],
                                                                (nd == 0 && s == 0) ? negd : dpacc[kb], 0, 0, 0);
      }
      if (kv0 + 64 > skv_b) {                                    // uniform: only the tile that holds masked keys
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            sacc[kb][r] = key < skv_b ? sacc[kb][r] : -INFINITY;   // P = exp2(-inf) = 0
          }
        __builtin_amdgcn_sched_barrier(0);                         // (keeps the compiler from if-converting the block into 64 selects per tile)
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = fast_exp2(sacc[kb][r]);                 // S^T came out of the MFMA as c s - lse2 (C operand)
          sacc[kb][r] = pr * dpacc[kb][r];                         // = P (dP - delta); x scale in the epilogue
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[ks][j] = (bf16)sacc[ks >> 1][8 * (ks & 1) + j];

    // fwd: O^T[d][q] += V^T[d][key] P^T[key][q]  (all chunks);   dQ: dQ^T[d][q] += K^T[d][key] dS^T[key][q]  (one chunk)
#pragma unroll
    for (int no = 0; no < NO; ++no) {
      const char* Ts = MODE == 0 ? Vs + no * TILE_BYTES : Ks + chunk * TILE_BYTES;
      bf16x8 tfr[4][2];                        // all eight transposed fragments first (the score registers are free by now)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          if constexpr (MODE == 1 && PEA_ATTN_BWD_PREFETCH) { tfr[ks][db] = tfr1[ks][db]; continue; }
          if constexpr (USE_TR)
            tfr[ks][db] = read_transposed_frag_at(smem + trc[db][0] + (no * TILE_BYTES + ks * 2048),
                                                  smem + trc[db][1] + (no * TILE_BYTES + ks * 2048));
          else tfr[ks][db] = read_transposed_frag<false>(Ts, ks * 16, db * 32, lane);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[2 * no + db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfr[ks][db], pf[ks], oacc[2 * no + db], 0, 0, 0);
    }
    WAIT_VM0();
    __syncthreads();
  }

  // epilogue: oacc[2*no+db][4g+j] = X^T[d = no*64 + db*32 + 8g + 4h + j][q = lane&31]
  float inv = MODE == 1 ? p.scale : 1.f;
  if (MODE == 0) {
    float la = l_run, lb = l_run;
    swap_halves32(la, lb);                                     // (l_run + its partner half, without an LDS round trip)
    const float l_tot = la + lb;
    inv = 1.f / l_tot;
    if (p.lse && qvalid && fh == 0)
      p.lse[((long long)b * p.H + head) * p.Sq + qrow] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
  }
  // The accumulator layout has a query row per lane: written straight out that is eight 8-byte stores per lane, each
  // touching 32 different 128-byte lines.  The wave's 32 x 64 block goes through an LDS image instead (its own 4.5 KiB of
  // the K / V ring, which every wave has left behind the loop's last barrier; 144-byte pitch), from which every store
  // instruction writes eight whole rows, 16 bytes per lane.
  const bool accum = MODE == 1 && p.accum_dq;
  char* const ost = smem + wave * (32 * 144);
  const int r8 = lane >> 3, ch = lane & 7;
#pragma unroll
  for (int no = 0; no < NO; ++no) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(oacc[2 * no + db][4 * g + j] * inv);
        *(bf16x4*)(ost + frow * 144 + (db * 32 + 8 * g + 4 * fh) * 2) = o;
      }
    bf16* Ob = MODE == 0 ? p.O + (long long)b * p.Sq * p.ldo + head * 64 * ND + no * 64 + ch * 8
                         : p.dQ + (long long)b * p.Sq * p.lddq + head * 64 * ND + chunk * 64 + ch * 8;
    const int ldo_ = MODE == 0 ? p.ldo : p.lddq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + r8;
      if (q0 + row < p.Sq) {
        bf16x8 v = *(const bf16x8*)(ost + row * 144 + ch * 16);
        bf16* dst = Ob + (long long)(q0 + row) * ldo_;
        if (accum) {
          const bf16x8 old = *(const bf16x8*)dst;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)old[j]);
        }
        store16<PEA_ATTN_WT != 0>(dst, v);
      }
    }
  }
}

template <int MODE, bool USE_TR, int ND, bool TXT = false>
__global__ __launch_bounds__(256, (ND == 1 ? (MODE == 0 ? 3 : 2) : 1)) void attn_q_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int blk_x, head, b;
  attn_block_coords(p.xcd_remap, blk_x, head, b);
  attn_q_body<MODE, USE_TR, ND, TXT, true>(p, smem, blk_x, head, b);
}

// per-row constants of a 64-query tile (lse, delta) -> LDS, 4 bytes per lane, issued by wave 0 only
__device__ __forceinline__ void stage_rowconst(const float* lse, const float* dlt, int r0, int rmax, char* dst,
                                               int wave, int lane) {
  if (wave != 0) return;
  int r = r0 + lane;
  r = r < rmax ? r : rmax - 1;
  lds_dma4(lse + r, dst);
  lds_dma4(dlt + r, dst + 256);
}

// ============================================================================= dK / dV
// workgroup = 4 waves = 128 keys (wave owns 32, key on the lane); loops over 64-query tiles.  ONE 64-wide chunk of
// dK/dV per workgroup: blockIdx.x enumerates (key block, query split, chunk).
// With p.nsplit > 1 (cross-attention: few keys, many queries) blockIdx.x also enumerates query ranges and the
// block writes fp32 partial dK/dV to p.dkv_part[split][b][h][key][2][64*ND]; attn_dkv_reduce_kernel adds the
// splits in order (deterministic, no atomics).
template <bool USE_TR, int ND>
__device__ __forceinline__ void attn_dkv_body(const AttnP& p, char* smem, int blk_x, int head, int b) {
  constexpr int STG = 2 * ND * TILE_BYTES + 512;                 // smem: [2 stages][Q sub-tiles | dO sub-tiles | lse,delta]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
  const int chunk = blk_x % ND;
  const int bx = blk_x / ND;
  const int kblk = bx / nsplit, split = bx - kblk * nsplit;
  const int k0 = kblk * 128 + wave * 32;
  const int frow = lane & 31, fh = lane >> 5;
  const float c = p.scale * LOG2E;

  const bf16* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * 64 * ND;
  const bf16* dOb = p.dO + (long long)b * p.Sq * p.lddo + head * 64 * ND;
  const bf16* Kb = p.K + (long long)b * p.Skv * p.ldk + head * 64 * ND;
  const bf16* Vb = p.V + (long long)b * p.Skv * p.ldv + head * 64 * ND;
  // row constants written by attn_delta_kernel (or the dQ pass): -delta[q] and -lse[q] * log2(e)
  const float* dltb = p.delta + ((long long)b * p.H + head) * p.Sq;
  const float* lseb = dltb + (long long)p.B * p.H * p.Sq;

  int krow = k0 + frow;
  const bool kstored = krow < p.Skv;          // the row exists in K / V / dK / dV
  const int skv_b = p.kv_len ? p.kv_len[b] : p.Skv;   // keys >= skv_b are padding: P = 0, so their dK = dV = 0
  const bool kvalid = krow < skv_b;
  krow = kstored ? krow : p.Skv - 1;
  const bool wave_active = k0 < p.Skv;       // wave-uniform
  bf16x8 kf[ND][4], vf[ND][4];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // K is only the operand of S here (dK^T = Q^T dS uses the staged Q tiles): it carries scale * log2(e)
      kf[nd][s] = *(const bf16x8*)(Kb + (long long)krow * p.ldk + nd * 64 + 16 * s + 8 * fh);
      if (!p.q_prescaled) kf[nd][s] = scale_frag(kf[nd][s], c);        // (a prescaled Q tile already carries the factor)
      vf[nd][s] = *(const bf16x8*)(Vb + (long long)krow * p.ldv + nd * 64 + 16 * s + 8 * fh);
    }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = 0.f;
  const bool all_valid = __all(kvalid);        // wave-uniform: no padding key among this wave's 32

  TileSrc qsrc[ND], dosrc[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) {
    qsrc[nd] = tile_src(Qb + nd * 64, p.ldq, p.Sq, wave, lane);
    dosrc[nd] = tile_src(dOb + nd * 64, p.lddo, p.Sq, wave, lane);
  }
  auto stage_q = [&](char* dst, int r0) {
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) {
      stage_tile(qsrc[nd], r0, dst + nd * TILE_BYTES, wave);
      stage_tile(dosrc[nd], r0, dst + (ND + nd) * TILE_BYTES, wave);
    }
    stage_rowconst(lseb, dltb, r0, p.Sq, dst + 2 * ND * TILE_BYTES, wave, lane);
  };
  const int nt_all = (p.Sq + 63) / 64;
  const int tps = (nt_all + nsplit - 1) / nsplit;            // query tiles per split
  const int t_begin = split * tps;
  const int nt = min(nt_all, t_begin + tps);
  if (t_begin < nt) stage_q(smem, t_begin * 64);
  WAIT_VM0();
  __syncthreads();

  for (int t = t_begin; t < nt; ++t) {
    const int cur = (t - t_begin) & 1;
    if (t + 1 < nt) stage_q(smem + (cur ^ 1) * STG, (t + 1) * 64);
    const char* Qs = smem + cur * STG;
    const char* dOs = Qs + ND * TILE_BYTES;
    const float* rc = (const float*)(Qs + 2 * ND * TILE_BYTES);
    if (t * 64 + 64 > p.Sq) {
      // ragged last tile (uniform, once per workgroup): the staged rows >= Sq repeat row Sq-1; give them -lse*log2e = -inf
      // (P = exp2(-inf) = 0) so the loop below needs no per-row predicate
      if (wave == 0 && t * 64 + lane >= p.Sq) {
        float* rw = (float*)(Qs + 2 * ND * TILE_BYTES);
        rw[lane] = -INFINITY;
        rw[64 + lane] = 0.f;
      }
      __syncthreads();
    }
    if (wave_active) {
      // S[q][key] = Q . K^T ; dP[q][key] = dO . V^T   (rows q in registers, key on the lane).  Both accumulations start
      // from a C operand instead of zero -- the staged row constants (row q = register index: the four 16-byte LDS loads ARE
      // the MFMA's C operand): S from -lse[q] * log2(e) (-inf on the lane of a padding key: P = 0 without a select per
      // score), dP from -delta[q].  With scale * log2(e) in the K fragments, P = exp2(S) as the MFMA leaves it.
      // LDS reads run one step ahead of the MFMAs that consume them (registers: two buffers of row fragments, two of
      // transposed fragments): as read -> wait -> MFMA pairs (what the fused loop compiled to) every pair of MFMAs stood
      // behind an LDS round trip, and with two waves per SIMD nothing covers it -- a key tile took 2.6 k cycles of its SIMD
      // for 1 k cycles of matrix work.
      f32x16 sacc[2], dpacc[2];
      bf16x8 rq[2][ND * 4], rd[2][ND * 4];
      auto load_rows = [&](int buf, int qb) {
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            rq[buf][nd * 4 + s] = read_row_frag(Qs + nd * TILE_BYTES, qb * 32 + frow, s, fh);
            rd[buf][nd * 4 + s] = read_row_frag(dOs + nd * TILE_BYTES, qb * 32 + frow, s, fh);
          }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 d4 = *(const f32x4*)(rc + 64 + qb * 32 + 8 * g + 4 * fh);
          const f32x4 l4 = *(const f32x4*)(rc + qb * 32 + 8 * g + 4 * fh);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dpacc[qb][4 * g + j] = d4[j];
            sacc[qb][4 * g + j] = l4[j];
          }
        }
      };
      bf16x8 dot[2][2][2], qt[2][2][2];          // [k-slice pair][kk][db]
      auto load_tr = [&](int bt) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dot[bt][kk][db] = read_transposed_frag<USE_TR>(dOs + chunk * TILE_BYTES, (2 * bt + kk) * 16, db * 32, lane);
            qt[bt][kk][db] = read_transposed_frag<USE_TR>(Qs + chunk * TILE_BYTES, (2 * bt + kk) * 16, db * 32, lane);
          }
      };
      load_rows(0, 0);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (PEA_ATTN_BWD_PREFETCH) {
          if (qb == 0) load_rows(1, 1);
          else load_tr(0);                        // lands under the second block's MFMAs and the exponentials
        } else if (qb == 1) {
          load_rows(1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!all_valid) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[qb][r] = kvalid ? sacc[qb][r] : -INFINITY;
        }
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            sacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rq[qb][nd * 4 + s], kf[nd][s], sacc[qb], 0, 0, 0);
            dpacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd[qb][nd * 4 + s], vf[nd][s], dpacc[qb], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x8 pfr[4], dsfr[4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const float pr = fast_exp2(sacc[qb][r]);
            const float ds = pr * dpacc[qb][r];                // P (dP - delta); x scale on the finished dK
            const int ks = qb * 2 + (g >> 1), e = (g & 1) * 4 + j;
            pfr[ks][e] = (bf16)pr;
            dsfr[ks][e] = (bf16)ds;
          }
        }
      // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]   (d in this block's chunk)
      // two k-slices (8 transposed fragments = 32 registers) per batch of 8 MFMAs; the second batch's fragments are read
      // while the first batch's MFMAs run
      __builtin_amdgcn_sched_barrier(0);
      if (!PEA_ATTN_BWD_PREFETCH) load_tr(0);
      load_tr(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[bt][kk][db], pfr[2 * bt + kk], dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[bt][kk][db], dsfr[2 * bt + kk], dk[db], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WAIT_VM0();
    __syncthreads();
  }
  // (whole waves store rows there: ragged lanes stay; an accumulating store keeps the direct form -- fp32 sum, ONE rounding)
  const bool lds_epi = PEA_ATTN_DKV_LDS_EPI && p.nsplit <= 1 && !p.accum_dkv;
  if (lds_epi ? !wave_active : !kstored) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[i][r] *= p.q_prescaled ? 0.6931471805599453f : p.scale;   // dS^T Q' = scale log2(e) dS^T Q
  const int D = 64 * ND;
  if (p.nsplit > 1) {
    float* pr = p.dkv_part + ((((long long)split * p.B + b) * p.H + head) * p.Skv + krow) * 2 * D + chunk * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * fh;
        f32x4 a, cc;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = dk[db][4 * g + j]; cc[j] = dv[db][4 * g + j]; }
        *(f32x4*)(pr + d) = a;
        *(f32x4*)(pr + D + d) = cc;
      }
    return;
  }
  if (lds_epi) {
    // The accumulator layout has a key row per lane: written straight out that is sixteen 8-byte stores per lane, each
    // touching 32 different 128-byte lines (store-issue-bound: ~9 k cycles per workgroup, paid once per 16 query tiles at
    // 1024 tokens).  Each wave's 32 x 64 blocks of dK, then dV, go through its own 4.5 KiB LDS image (the ring is free
    // behind the loop's last barrier; 144-byte pitch) and leave as whole rows, 16 bytes per lane, eight rows per instruction.
    char* const ost = smem + wave * (32 * 144);
    const int r8 = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      f32x16* acc = which ? dv : dk;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)acc[db][4 * g + j];
          *(bf16x4*)(ost + frow * 144 + (db * 32 + 8 * g + 4 * fh) * 2) = o;
        }
      bf16* base = (which ? p.dV : p.dK) + (long long)b * p.Skv * (which ? p.lddv : p.lddk) + head * D + chunk * 64 + ch * 8;
      const int ld_ = which ? p.lddv : p.lddk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + r8;
        if (k0 + row < p.Skv) {
          bf16x8 v = *(const bf16x8*)(ost + row * 144 + ch * 16);
          bf16* dst = base + (long long)(k0 + row) * ld_;
          if (p.accum_dkv) {
            const bf16x8 old = *(const bf16x8*)dst;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)old[j]);
          }
          *(bf16x8*)dst = v;
        }
      }
    }
    return;
  }
  bf16* dKr = p.dK + ((long long)b * p.Skv + krow) * p.lddk + head * D + chunk * 64;
  bf16* dVr = p.dV + ((long long)b * p.Skv + krow) * p.lddv + head * D + chunk * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = db * 32 + 8 * g + 4 * fh;
      bf16x4 ok, ov;
      if (p.accum_dkv) {
        ok = *(const bf16x4*)(dKr + d);
        ov = *(const bf16x4*)(dVr + d);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ok[j] = (bf16)(dk[db][4 * g + j] + (p.accum_dkv ? (float)ok[j] : 0.f));
        ov[j] = (bf16)(dv[db][4 * g + j] + (p.accum_dkv ? (float)ov[j] : 0.f));
      }
      *(bf16x4*)(dKr + d) = ok;
      *(bf16x4*)(dVr + d) = ov;
    }
}

template <bool USE_TR, int ND>
__global__ __launch_bounds__(256, (ND == 1 ? 2 : 1)) void attn_dkv_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int blk_x, head, b;
  attn_block_coords(p.xcd_remap, blk_x, head, b);
  attn_dkv_body<USE_TR, ND>(p, smem, blk_x, head, b);
}

// ============================================================================= fused backward launch
// dQ workgroups and dK/dV workgroups of ONE attention in ONE grid.  As two launches each pass ends on a partly filled
// last round of workgroups (self-attention over 1024 tokens, B*H = 80: 640 workgroups on 512 slots = 1.25 rounds, i.e.
// 62 % of the slots busy; 4096 tokens: 2.5 rounds, 83 %); together they are 2.5 / 5 rounds.  Both roles keep their own
// code (the role is uniform per workgroup).  Task order: all workgroups of one (batch, head) are adjacent -- dQ blocks,
// then dK/dV blocks -- and XCD x owns a contiguous range of heads, so a head's Q / K / V / dO are read into one L2 once
// for both roles.  delta comes from attn_delta_kernel (the roles run concurrently, so the dQ role cannot hand it over).
template <bool USE_TR, int ND>
__global__ __launch_bounds__(256, (ND == 1 ? 2 : 1)) void attn_bwd_fused_kernel(const AttnP p, int n_dq, int n_dkv, int heavy_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned total = gridDim.x;
  unsigned lin = blockIdx.x;
  const int per_head = n_dq + n_dkv;
  if (p.xcd_remap) {
    const unsigned q = total >> 3, r = total & 7, xcd = lin & 7, j = lin >> 3;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1u : 0u);
    if (heavy_first && r == 0 && cnt % per_head == 0) {
      // Dispatch order inside an XCD's range of heads: ALL dK/dV workgroups (4 products per tile) of those heads first, then
      // all dQ workgroups (3 products).  The grid is 2.5 rounds of the chip's slots (1024 tokens, B * H = 80); the hardware
      // hands a free slot the next workgroup in order, so what is left for the last, partly filled round should be the
      // lighter role, and the heavy ones must not start last.  A head's operands then meet one L2 twice instead of once:
      // the loops do not care (profiles/EXPERIMENTS.md: cache-resident operands are worth 2 % to self-attention).
      const unsigned nh = cnt / per_head, first_bh = start / per_head;
      unsigned bh_l, rem;
      if (j < nh * (unsigned)n_dkv) { bh_l = j / n_dkv; rem = j - bh_l * n_dkv; }
      else { const unsigned j2 = j - nh * n_dkv; bh_l = j2 / n_dq; rem = n_dkv + (j2 - bh_l * n_dq); }
      const int bh = (int)(first_bh + bh_l);
      const int b = bh / p.H, head = bh - b * p.H;
      if ((int)rem < n_dkv) attn_dkv_body<USE_TR, ND>(p, smem, (int)rem, head, b);
      else attn_q_body<1, USE_TR, ND, false, false>(p, smem, (int)rem - n_dkv, head, b);
      return;
    }
    lin = start + j;
  }
  const int bh = (int)(lin / per_head), rem = (int)(lin - (unsigned)bh * per_head);
  const int b = bh / p.H, head = bh - b * p.H;
  if (rem < n_dkv) attn_dkv_body<USE_TR, ND>(p, smem, rem, head, b);          // heavier role (4 products) first
  else attn_q_body<1, USE_TR, ND, false, false>(p, smem, rem - n_dkv, head, b);
}

// ============================================================================= cross-attention forward, K / V resident
// <= 128 keys (the 77-token text context: KB = 3 blocks of 32), head_dim 64.  The general forward spends a workgroup per 128
// queries on two 64-key tiles (25 % of them padding), a K / V staging round trip and two barriers for 2 x 8 MFMAs per wave.
// Here a workgroup stages K and V ONCE and its four waves then walk 128-query units independently -- no barrier in the loop:
// Q fragments straight from HBM (this lane's query row: four 16-byte loads), S^T over KB key blocks, the softmax in one
// shot (all scores of a query are in registers: no running max, no rescale), O^T = V^T P^T, store.  Per-sample key counts
// (merged passes with a shorter student context) mask through the same compare as the padding keys.
__device__ __forceinline__ float xhalf_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
template <int KB>
__global__ __launch_bounds__(256, 3) void xattn_fwd_kernel(const AttnP p, int upw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KT = (KB + 1) / 2;                               // 64-key tiles
  char* const Ksm = smem;
  char* const Vsm = smem + KT * TILE_BYTES;
  char* const Osm = smem + 2 * KT * TILE_BYTES;                  // 4 waves x 32 rows x 144 bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  int split, head, b;
  attn_block_coords(p.xcd_remap, split, head, b);
  const int nu = (p.Sq + 127) >> 7;
  const int u_begin = split * upw, u_end = min(nu, u_begin + upw);
  const float c = p.q_prescaled ? 1.f : p.scale * LOG2E;         // prescaled Q: the scores are in the log2 domain already
  const bf16* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * 64;
  const bf16* Kb = p.K + (long long)b * p.Skv * p.ldk + head * 64;
  const bf16* Vb = p.V + (long long)b * p.Skv * p.ldv + head * 64;
  const int skv_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.Skv);
  {
    const TileSrc ksrc = tile_src(Kb, p.ldk, p.Skv, wave, lane), vsrc = tile_src(Vb, p.ldv, p.Skv, wave, lane);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      stage_tile(ksrc, t * 64, Ksm + t * TILE_BYTES, wave);
      stage_tile(vsrc, t * 64, Vsm + t * TILE_BYTES, wave);
    }
  }
  int rf_off[4], tr_off[2][2];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) rf_off[s4] = frow * 128 + (((2 * s4 + fh) ^ swz_x(frow)) << 4);
  {
    const int i16 = lane & 15, rr = 4 * fh + (i16 >> 2), cc = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) tr_off[db][hl] = swz_rc(rr + 8 * hl, db * 32 + cc);
  }
  int qrow = u_begin * 128 + wave * 32 + frow;
  bf16x8 qf[4];
  // Q: this wave's 32 rows as a 4 KiB swizzled image in its own LDS region, filled by LDS-DMA (whole 128-byte rows per
  // 8 lanes) one unit ahead; direct 16-byte loads of a row per lane touched 32 lines per instruction
  char* const qst = Osm + 4 * 32 * 144 + wave * 4096;
  const TileSrc qsrc = tile_src(Qb, p.ldq, p.Sq, 0, lane);      // pieces 0 / 1 of a tile: rows 0..15; + 16 rows via the scalar offset
  auto stage_q = [&](int row0) {
    stage_tile(qsrc, row0, qst, 0);
    stage_tile(qsrc, row0 + 16, qst + 2048, 0);
  };
  if (u_begin < u_end) stage_q(u_begin * 128 + wave * 32);
  WAIT_VM0();
  __syncthreads();
  for (int u = u_begin; u < u_end; ++u, qrow += 128) {
    const bool qvalid = qrow < p.Sq;
    WAIT_VM0();                                                  // this wave's Q image (and its stores of the last unit)
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qst + rf_off[s]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (u + 1 < u_end) stage_q((u + 1) * 128 + wave * 32);      // next unit's rows fly under this unit's work
    // S^T[key][q] = K . Q^T over the KB key blocks
    f32x16 sacc[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      bf16x8 kfr[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kfr[s] = *(const bf16x8*)(Ksm + rf_off[s] + ((kb >> 1) * TILE_BYTES + (kb & 1) * 4096));
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < 4; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], sacc[kb], 0, 0, 0);
    }
    // padding keys / keys past this sample's count: -inf (only the blocks that hold any)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
      if (kb * 32 + 32 > skv_b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          sacc[kb][r] = key < skv_b ? sacc[kb][r] : -INFINITY;
        }
      }
    float mx = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = (kb == 0 ? 1 : 0); r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
    mx = xhalf_max(mx);
    const float mc = mx * c;
    float ls = 0.f;
    bf16x8 pf[2 * KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = fast_exp2(fmaf(sacc[kb][r], c, -mc));
        ls += e;
        pf[2 * kb + (r >> 3)][r & 7] = (bf16)e;
      }
    const float l_tot = xhalf_sum(ls);
    // O^T[d][q] = V^T[d][key] P^T[key][q]
    f32x16 oacc[2];
#pragma unroll
    for (int ks = 0; ks < 2 * KB; ++ks) {
      const int vb = (ks >> 2) * TILE_BYTES + (ks & 3) * 2048;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x8 tf = read_transposed_frag_at(Vsm + tr_off[db][0] + vb, Vsm + tr_off[db][1] + vb);
        if (ks == 0) {
          const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf[0], zero16, 0, 0, 0);
        } else {
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf[ks], oacc[db], 0, 0, 0);
        }
      }
    }
    const float inv = 1.f / l_tot;
    if (qvalid && p.lse && fh == 0)
      p.lse[((long long)b * p.H + head) * p.Sq + qrow] = mx * (p.q_prescaled ? 0.6931471805599453f : p.scale) + log2f(l_tot) * 0.6931471805599453f;
    // O leaves through a per-wave LDS image (32 rows x 128 bytes, 144-byte pitch): the accumulator layout has a query row
    // per lane, i.e. eight 8-byte stores per lane that each touch 32 different 128-byte lines; from the image every store
    // instruction writes eight whole rows (16 bytes per lane).  With 77 keys the output is most of this kernel's traffic.
    char* const ost = Osm + wave * (32 * 144);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(oacc[db][4 * g + j] * inv);
        *(bf16x4*)(ost + frow * 144 + (db * 32 + 8 * g + 4 * fh) * 2) = o;
      }
    {
      const int r8 = lane >> 3, ch = lane & 7;
      const int q0w = u * 128 + wave * 32;
      bf16* Ob = p.O + (long long)b * p.Sq * p.ldo + head * 64 + ch * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + r8;
        const bf16x8 v = *(const bf16x8*)(ost + row * 144 + ch * 16);
        if (q0w + row < p.Sq) store16<PEA_ATTN_WT != 0>(Ob + (long long)(q0w + row) * p.ldo, v);
      }
    }
  }
}

// ============================================================================= cross-attention backward, ONE pass
// Few keys (the text context: 77 tokens -> KB = 3 blocks of 32), many queries, head_dim 64.  The general path costs three
// launches (delta, dQ + dK/dV roles, split reduce) that each stream Q / dO again and are latency-bound on their short
// key loops.  Here a workgroup owns a run of 128-query units of one (batch, head): K and V stay in LDS for its whole
// life, a unit's Q / dO tiles (+ lse) arrive by LDS-DMA one unit ahead, delta = rowsum(dO * O) is formed in-kernel
// (dO from the staged tile, O prefetched from HBM one unit ahead), and both orientations run off the same tiles:
//   query on the lane (wave w = queries 32w..32w+31):  S^T, dP^T -> dS^T -> dQ^T = K^T dS^T           (written per unit)
//   key on the lane   (wave w = keys 32w..32w+31, w < KB):  S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS
// dK / dV partial sums of the workgroup go to p.dkv_part[split] (fixed-order reduce, attn_dkv_reduce_kernel) or, with
// one split, straight to dK / dV.
template <int KB>
__global__ __launch_bounds__(256, 2) void xattn_bwd_kernel(const AttnP p, int upw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: K rows 0..127 | V rows 0..127 (two 64-row tiles each; 32 KB) | ONE stage: Q tiles 0,1 | dO tiles 0,1 | lse | delta
  // = 65 KB, two workgroups per CU: the other workgroup's compute covers this one's DMA wait (the stage is not double
  // buffered: with one wave per SIMD and 98 KB the single resident workgroup ran its dependency chains back to back,
  // 10 us per unit)
  char* const Ksm = smem;
  char* const Vsm = smem + 2 * TILE_BYTES;
  char* const S0 = smem + 4 * TILE_BYTES;
  float* const rc = (float*)(S0 + 4 * TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  int split, head, b;
  attn_block_coords(p.xcd_remap, split, head, b);
  const int nu = (p.Sq + 127) >> 7;
  const int u_begin = split * upw, u_end = min(nu, u_begin + upw);
  const float c = p.q_prescaled ? 1.f : p.scale * LOG2E;         // prescaled Q: the scores are in the log2 domain already
  const bf16* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * 64;
  const bf16* dOb = p.dO + (long long)b * p.Sq * p.lddo + head * 64;
  const bf16* Ob = p.O + (long long)b * p.Sq * p.ldo + head * 64;
  const bf16* Kb = p.K + (long long)b * p.Skv * p.ldk + head * 64;
  const bf16* Vb = p.V + (long long)b * p.Skv * p.ldv + head * 64;
  const float* lseb = p.lse + ((long long)b * p.H + head) * p.Sq;
  const int skv_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.Skv);

  const TileSrc qsrc = tile_src(Qb, p.ldq, p.Sq, wave, lane), dosrc = tile_src(dOb, p.lddo, p.Sq, wave, lane);
  auto stage_unit = [&](int u) {
    const int r0 = u * 128;
    stage_tile(qsrc, r0, S0, wave);
    stage_tile(qsrc, r0 + 64, S0 + TILE_BYTES, wave);
    stage_tile(dosrc, r0, S0 + 2 * TILE_BYTES, wave);
    stage_tile(dosrc, r0 + 64, S0 + 3 * TILE_BYTES, wave);
    if (wave < 2) {
      int r = r0 + wave * 64 + lane;
      r = r < p.Sq ? r : p.Sq - 1;
      lds_dma4(lseb + r, S0 + 4 * TILE_BYTES + wave * 256);
    }
  };
  // delta: thread = (query of the unit, half of the head dimension); its O values are fetched together with the stage
  const int dq_l = tid >> 1, dhalf = tid & 1;
  bf16x8 o_pf[4];
  auto fetch_o = [&](int u) {
    int r = u * 128 + dq_l;
    r = r < p.Sq ? r : p.Sq - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) o_pf[i] = *(const bf16x8*)(Ob + (long long)r * p.ldo + dhalf * 32 + 8 * i);
  };

  {
    const TileSrc ksrc = tile_src(Kb, p.ldk, p.Skv, wave, lane), vsrc = tile_src(Vb, p.ldv, p.Skv, wave, lane);
    stage_tile(ksrc, 0, Ksm, wave);
    stage_tile(vsrc, 0, Vsm, wave);
    if (KB > 2) {
      stage_tile(ksrc, 64, Ksm + TILE_BYTES, wave);
      stage_tile(vsrc, 64, Vsm + TILE_BYTES, wave);
    }
  }
  if (u_begin < u_end) {
    stage_unit(u_begin);
    fetch_o(u_begin);
  }
  // key-on-the-lane role: this wave's 32 keys as B-operand fragments, for the whole kernel
  int krow = wave * 32 + frow;
  const bool kstored = krow < p.Skv, kvalid = krow < skv_b;
  krow = kstored ? krow : p.Skv - 1;
  const bool wave_keys = wave < KB && wave * 32 < p.Skv;       // wave-uniform
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = 0.f;
  // this wave's keys as B-operand fragments come from the K / V images in LDS, per query block (kept in registers for the
  // whole kernel they cost 32 VGPRs and the 256-register budget of two workgroups per CU spills)
  const bool wave_pad = wave * 32 + 32 > skv_b;                 // wave-uniform: this wave's key block holds padding keys
  const char* const Kmine = Ksm + (wave >> 1) * TILE_BYTES;
  const char* const Vmine = Vsm + (wave >> 1) * TILE_BYTES;
  const int kmine_row = (wave & 1) * 32 + frow;

  for (int u = u_begin; u < u_end; ++u) {
    WAIT_VM0();
    __syncthreads();                                             // the unit's tiles, lse and this thread's O values are here
    {
      const char* dt = S0 + (2 + (dq_l >> 6)) * TILE_BYTES;
      const int row = dq_l & 63;
      float dsum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 d = *(const bf16x8*)(dt + row * 128 + (((dhalf * 4 + i) ^ swz_x(row)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum += (float)d[j] * (float)o_pf[i][j];
      }
      dsum += dpp_move<0xB1>(dsum);                          // lane ^ 1, a DPP move
      if (dhalf == 0) {                       // the unit's row constants, negated: C operand of dP, addend of the exp2 argument
        const bool rv = u * 128 + dq_l < p.Sq;             // rows past Sq (ragged last unit): P = exp2(-inf) = 0
        rc[128 + dq_l] = rv ? -dsum : 0.f;
        rc[dq_l] = rv ? -rc[dq_l] * LOG2E : -INFINITY;
      }
    }
    __syncthreads();

    // ---- query on the lane: dQ of this wave's 32 queries
    {
      const char* Qt = S0 + (wave >> 1) * TILE_BYTES;
      const char* dOt = S0 + (2 + (wave >> 1)) * TILE_BYTES;
      const int rb0 = (wave & 1) * 32;
      const int qg = u * 128 + wave * 32 + frow;
      bf16x8 qf[4], dof[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[s] = read_row_frag(Qt, rb0 + frow, s, fh);
        dof[s] = read_row_frag(dOt, rb0 + frow, s, fh);
      }
      const float nlse2 = rc[wave * 32 + frow], ndlt = rc[128 + wave * 32 + frow];
      f32x16 negd;
#pragma unroll
      for (int r = 0; r < 16; ++r) negd[r] = ndlt;
      f32x16 oacc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
#pragma unroll 1
      for (int kb = 0; kb < KB; ++kb) {
        const char* Kt = Ksm + (kb >> 1) * TILE_BYTES;
        const char* Vt = Vsm + (kb >> 1) * TILE_BYTES;
        const int kr0 = (kb & 1) * 32;
        f32x16 sacc, dpacc;
        {
          bf16x8 kfr[4], vfr[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            kfr[s] = read_row_frag(Kt, kr0 + frow, s, fh);
            vfr[s] = read_row_frag(Vt, kr0 + frow, s, fh);
          }
          const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qf[0], zero16, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0], dof[0], negd, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 4; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], dof[s], dpacc, 0, 0, 0);
          }
        }
        bf16x8 tfr[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int db = 0; db < 2; ++db) tfr[k2][db] = read_transposed_frag<true>(Kt, kr0 + k2 * 16, db * 32, lane);
        bf16x8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pr = fast_exp2(fmaf(sacc[r], c, nlse2));
          if (kb * 32 + 32 > skv_b)                               // uniform: only the key block that holds padding keys
            pr = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh < skv_b ? pr : 0.f;
          pf[r >> 3][r & 7] = (bf16)(pr * dpacc[r]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int db = 0; db < 2; ++db)
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfr[k2][db], pf[k2], oacc[db], 0, 0, 0);
      }
      if (qg < p.Sq) {
        bf16* dst0 = p.dQ + ((long long)b * p.Sq + qg) * p.lddq + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16* dst = dst0 + db * 32 + 8 * g + 4 * fh;
            bf16x4 o;
            if (p.accum_dq) o = *(const bf16x4*)dst;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)(oacc[db][4 * g + j] * p.scale + (p.accum_dq ? (float)o[j] : 0.f));
            *(bf16x4*)dst = o;
          }
      }
    }
    // ---- key on the lane: dK / dV of this wave's 32 keys, the unit's four 32-query blocks
    if (wave_keys) {
#pragma unroll 1
      for (int qb = 0; qb < 4; ++qb) {
        const char* Qt = S0 + (qb >> 1) * TILE_BYTES;
        const char* dOt = S0 + (2 + (qb >> 1)) * TILE_BYTES;
        const int rb0 = (qb & 1) * 32;
        f32x16 sacc, dpacc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                            // dP starts from -delta[q] (row = register index)
          const f32x4 d4 = *(const f32x4*)(rc + 128 + qb * 32 + 8 * g + 4 * fh);
#pragma unroll
          for (int j = 0; j < 4; ++j) dpacc[4 * g + j] = d4[j];
        }
        {
          bf16x8 qfr[4], kf[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            qfr[s] = read_row_frag(Qt, rb0 + frow, s, fh);
            kf[s] = read_row_frag(Kmine, kmine_row, s, fh);
          }
          const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[0], kf[0], zero16, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], sacc, 0, 0, 0);
        }
        {
          bf16x8 dfr[4], vf[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            dfr[s] = read_row_frag(dOt, rb0 + frow, s, fh);
            vf[s] = read_row_frag(Vmine, kmine_row, s, fh);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[s], vf[s], dpacc, 0, 0, 0);
        }
        bf16x8 dot[2][2], qt[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dot[k2][db] = read_transposed_frag<true>(dOt, rb0 + k2 * 16, db * 32, lane);
            qt[k2][db] = read_transposed_frag<true>(Qt, rb0 + k2 * 16, db * 32, lane);
          }
        bf16x8 pfr[2], dsfr[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *(const f32x4*)(rc + qb * 32 + 8 * g + 4 * fh);     // -lse * log2(e) of 4 consecutive query rows
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            float pr = fast_exp2(fmaf(sacc[r], c, l4[j]));
            if (wave_pad) pr = kvalid ? pr : 0.f;                // only the wave whose key block holds padding keys
            const float ds = pr * dpacc[r];
            pfr[g >> 1][(g & 1) * 4 + j] = (bf16)pr;
            dsfr[g >> 1][(g & 1) * 4 + j] = (bf16)ds;
          }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[k2][db], pfr[k2], dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[k2][db], dsfr[k2], dk[db], 0, 0, 0);
          }
      }
    }
    __syncthreads();                                             // every wave is done with the stage
    if (u + 1 < u_end) {
      stage_unit(u + 1);
      fetch_o(u + 1);
    }
  }
  if (!wave_keys || !kstored) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[i][r] *= p.q_prescaled ? 0.6931471805599453f : p.scale;
  if (p.nsplit > 1) {
    float* pr = p.dkv_part + ((((long long)split * p.B + b) * p.H + head) * p.Skv + krow) * 128;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * fh;
        f32x4 a, cc;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = dk[db][4 * g + j]; cc[j] = dv[db][4 * g + j]; }
        *(f32x4*)(pr + d) = a;
        *(f32x4*)(pr + 64 + d) = cc;
      }
    return;
  }
  bf16* dKr = p.dK + ((long long)b * p.Skv + krow) * p.lddk + head * 64;
  bf16* dVr = p.dV + ((long long)b * p.Skv + krow) * p.lddv + head * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = db * 32 + 8 * g + 4 * fh;
      bf16x4 ok, ov;
      if (p.accum_dkv) {
        ok = *(const bf16x4*)(dKr + d);
        ov = *(const bf16x4*)(dVr + d);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ok[j] = (bf16)(dk[db][4 * g + j] + (p.accum_dkv ? (float)ok[j] : 0.f));
        ov[j] = (bf16)(dv[db][4 * g + j] + (p.accum_dkv ? (float)ov[j] : 0.f));
      }
      *(bf16x4*)(dKr + d) = ok;
      *(bf16x4*)(dVr + d) = ov;
    }
}

// one 16-byte element of the split reduce (head_dim 64): out[b][key][h*64+d] (+)= sum_split part[split][b][h][key][{dK,dV}][d],
// splits added in order.  Shared by attn_dkv_reduce_kernel's nd == 1 case in spirit; used by xattn_bwd2_kernel's prologue.
__device__ __forceinline__ void dkv_reduce_elem64(const float* __restrict__ part, bf16* dK, bf16* dV, int lddk, int lddv, int ns,
                                                  int B, int H, int Skv, int accum, long long idx) {
  const int d4 = (int)(idx & 15) * 4;
  long long r = idx >> 4;
  const int which = (int)(r & 1); r >>= 1;
  const int key = (int)(r % Skv); r /= Skv;
  const int head = (int)(r % H);
  const int b = (int)(r / H);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < ns; ++s) {
    const f32x4 v = *(const f32x4*)(part + ((((long long)s * B + b) * H + head) * Skv + key) * 128 + which * 64 + d4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
  bf16* dst = which ? dV + ((long long)b * Skv + key) * lddv + head * 64 + d4 : dK + ((long long)b * Skv + key) * lddk + head * 64 + d4;
  bf16x4 o;
  if (accum) o = *(const bf16x4*)dst;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[j] + (accum ? (float)o[j] : 0.f));
  *(bf16x4*)dst = o;
}

// ============================================================================= cross-attention backward, v2 (round 6)
// The one-pass kernel above runs BOTH orientations on every wave, one after the other, off a single-buffered 128-query
// stage: per unit a wave walks LDS read -> MFMA -> exp -> MFMA chains of 100 MFMAs with nothing beside it on its SIMD but
// the other workgroup's wave, and the DMA of the next unit starts only when the unit is done (30 us for 42 MB of traffic).
// Here the unit is 64 queries, the stage is double buffered (the next unit's Q / dO / lse fly under this unit's work) and
// the waves SPECIALISE, so the two orientations run side by side on different SIMDs:
//   waves 0, 1  query on the lane: S^T, dP^T -> dS^T -> dQ^T = K^T dS^T of 32 queries each; dQ leaves through a per-wave
//               LDS image as whole 128-byte rows (the row-per-lane accumulator layout is sixteen 8-byte pieces over 32 lines)
//   waves 2, 3  key on the lane: S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS.  KB = 3 key blocks x 2 query blocks
//               = 6 tasks of 16 MFMAs: wave 2 takes (kb 0, both query blocks) + (kb 2, queries 0..31), wave 3 takes
//               (kb 1, both) + (kb 2, queries 32..63) -- 48 MFMAs each against 36 on the dQ waves; the two partial sums
//               of key block 2 meet through LDS behind the loop, (wave 2) + (wave 3), a fixed order.
// LDS: K | V images (2 x 16 KB) + 2 stages x (Q tile | dO tile | lse | delta = 16.5 KB) + 2 x 4.5 KB dQ images = 74 KB:
// two workgroups per CU.  KB = 2 (<= 64 keys: the 52-token student context un-merged) and KB = 3 (77 keys); other key
// counts keep the kernel above.  PRE (Q prescaled by its projection, the product path): the scores leave the MFMA in the
// log2 domain with -lse log2(e) as the C operand, p = exp2(acc) with no per-score FMA.
template <int KB, bool PRE>
__global__ __launch_bounds__(256, 2) void xattn_bwd2_kernel(const AttnP p, int upw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STG = 2 * TILE_BYTES + 512;
  char* const Ksm = smem;
  char* const Vsm = smem + 2 * TILE_BYTES;
  char* const St = smem + 4 * TILE_BYTES;            // [2 stages][Q tile | dO tile | lse[64] | delta[64]]
  char* const Ost = St + 2 * STG;                    // [2 waves][32 rows x 144 bytes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  int split, head, b;
  attn_block_coords(p.xcd_remap, split, head, b);
  const int nu = (p.Sq + 63) >> 6;
  const int u_begin = split * upw, u_end = min(nu, u_begin + upw);
  const float c = PRE ? 1.f : p.scale * LOG2E;
  const bf16* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * 64;
  const bf16* dOb = p.dO + (long long)b * p.Sq * p.lddo + head * 64;
  const bf16* Ob = p.O + (long long)b * p.Sq * p.ldo + head * 64;
  const bf16* Kb = p.K + (long long)b * p.Skv * p.ldk + head * 64;
  const bf16* Vb = p.V + (long long)b * p.Skv * p.ldv + head * 64;
  const float* lseb = p.lse + ((long long)b * p.H + head) * p.Sq;
  const int skv_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.Skv);

  const TileSrc qsrc = tile_src(Qb, p.ldq, p.Sq, wave, lane), dosrc = tile_src(dOb, p.lddo, p.Sq, wave, lane);
  auto stage_unit = [&](int u, int st) {
    char* const S0 = St + st * STG;
    const int r0 = u * 64;
    stage_tile(qsrc, r0, S0, wave);
    stage_tile(dosrc, r0, S0 + TILE_BYTES, wave);
    if (wave == 0) {
      int r = r0 + lane;
      r = r < p.Sq ? r : p.Sq - 1;
      lds_dma4(lseb + r, S0 + 2 * TILE_BYTES);
    }
  };
  // delta = rowsum(dO * O): thread = (query of the unit, quarter of the head dimension); O fetched one unit ahead
  const int dq_l = tid >> 2, dqt = tid & 3;
  bf16x8 o_pf[2];
  auto fetch_o = [&](int u) {
    int r = u * 64 + dq_l;
    r = r < p.Sq ? r : p.Sq - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) o_pf[i] = *(const bf16x8*)(Ob + (long long)r * p.ldo + dqt * 16 + 8 * i);
  };
  {
    const TileSrc ksrc = tile_src(Kb, p.ldk, p.Skv, wave, lane), vsrc = tile_src(Vb, p.ldv, p.Skv, wave, lane);
    stage_tile(ksrc, 0, Ksm, wave);
    stage_tile(vsrc, 0, Vsm, wave);
    if (KB > 2) {
      stage_tile(ksrc, 64, Ksm + TILE_BYTES, wave);
      stage_tile(vsrc, 64, Vsm + TILE_BYTES, wave);
    }
  }
  if (u_begin < u_end) {
    stage_unit(u_begin, 0);
    fetch_o(u_begin);
  }
  // Deferred split reduce of ANOTHER launch (the previous cross-attention layer of the backward pass: AttnP::red_*): every
  // workgroup adds up its share of those partials while its own K / V / first unit are in flight -- 70 launches of a 5.8 us
  // kernel per step otherwise.  Nothing here depends on this workgroup's own data.
  if (p.red_part) {
    const long long total = (long long)p.red_B * p.red_H * p.red_Skv * 32;
    const long long nthr = (long long)gridDim.x * gridDim.y * gridDim.z * 256;
    const long long first = ((long long)blockIdx.x + gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z)) * 256 + tid;
    for (long long idx = first; idx < total; idx += nthr)
      dkv_reduce_elem64(p.red_part, p.red_dK, p.red_dV, p.red_lddk, p.red_lddv, p.red_nsplit, p.red_B, p.red_H, p.red_Skv, p.red_accum, idx);
  }
  const bool key_wave = wave >= 2;
  // the shared head of a unit, executed by every wave: wait for the unit's tiles, start the next unit's, form delta.
  // Both role loops below call it once per unit, so every wave passes the same sequence of workgroup barriers.
  auto unit_head = [&](int u) -> char* {
    const int cur = (u - u_begin) & 1;
    char* const S0 = St + cur * STG;
    float* const rc = (float*)(S0 + 2 * TILE_BYTES);           // [0..63] lse -> -lse log2(e); [64..127] -delta
    WAIT_VM0();
    __syncthreads();                                           // unit u landed; every wave has left unit u - 1
    if (u + 1 < u_end) stage_unit(u + 1, cur ^ 1);             // flies under this unit's work
    {
      const char* dt = S0 + TILE_BYTES;
      float dsum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 d = *(const bf16x8*)(dt + dq_l * 128 + (((dqt * 2 + i) ^ swz_x(dq_l)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum += (float)d[j] * (float)o_pf[i][j];
      }
      dsum += dpp_move<0xB1>(dsum);                            // lanes ^ 1, ^ 2: the four quarters of a query
      dsum += dpp_move<0x4E>(dsum);
      if (dqt == 0) {
        const bool rv = u * 64 + dq_l < p.Sq;                  // rows past Sq (ragged last unit): P = exp2(-inf) = 0
        rc[64 + dq_l] = rv ? -dsum : 0.f;
        rc[dq_l] = rv ? -rc[dq_l] * LOG2E : -INFINITY;
      }
    }
    if (u + 1 < u_end) fetch_o(u + 1);
    __syncthreads();
    return S0;
  };

  if (!key_wave) {
    // ================= waves 0, 1: query on the lane -- dQ of queries 32 wave .. 32 wave + 31 of every unit
    for (int u = u_begin; u < u_end; ++u) {
      const char* const S0 = unit_head(u);
      const char* const Qt = S0;
      const char* const dOt = S0 + TILE_BYTES;
      const float* const rc = (const float*)(S0 + 2 * TILE_BYTES);
      const int rb0 = wave * 32;
      const int qg = u * 64 + rb0;
      bf16x8 qf[4], dof[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[s] = read_row_frag(Qt, rb0 + frow, s, fh);
        dof[s] = read_row_frag(dOt, rb0 + frow, s, fh);
      }
      const float nlse2 = rc[rb0 + frow], ndlt = rc[64 + rb0 + frow];
      f32x16 negd, rowc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { negd[r] = ndlt; rowc[r] = PRE ? nlse2 : 0.f; }
      f32x16 oacc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
#pragma unroll 1
      for (int kb = 0; kb < KB; ++kb) {
        const char* Kt = Ksm + (kb >> 1) * TILE_BYTES;
        const char* Vt = Vsm + (kb >> 1) * TILE_BYTES;
        const int kr0 = (kb & 1) * 32;
        f32x16 sacc, dpacc;
        {
          bf16x8 kfr[4], vfr[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            kfr[s] = read_row_frag(Kt, kr0 + frow, s, fh);
            vfr[s] = read_row_frag(Vt, kr0 + frow, s, fh);
          }
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qf[0], rowc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0], dof[0], negd, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 4; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], dof[s], dpacc, 0, 0, 0);
          }
        }
        bf16x8 tfr[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int db = 0; db < 2; ++db) tfr[k2][db] = read_transposed_frag<true>(Kt, kr0 + k2 * 16, db * 32, lane);
        bf16x8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pr = PRE ? fast_exp2(sacc[r]) : fast_exp2(fmaf(sacc[r], c, nlse2));
          if (kb * 32 + 32 > skv_b)                               // uniform: only the key block that holds padding keys
            pr = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh < skv_b ? pr : 0.f;
          pf[r >> 3][r & 7] = (bf16)(pr * dpacc[r]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int db = 0; db < 2; ++db)
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfr[k2][db], pf[k2], oacc[db], 0, 0, 0);
      }
      // dQ^T[d][q] -> LDS image [q][d] -> whole rows, 16 bytes per lane
      char* const ost = Ost + wave * (32 * 144);
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)(oacc[db][4 * g + j] * p.scale);
          *(bf16x4*)(ost + frow * 144 + (db * 32 + 8 * g + 4 * fh) * 2) = o;
        }
      const int r8 = lane >> 3, ch = lane & 7;
      bf16* dQb = p.dQ + (long long)b * p.Sq * p.lddq + head * 64 + ch * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + r8;
        if (qg + row < p.Sq) {
          bf16x8 v = *(const bf16x8*)(ost + row * 144 + ch * 16);
          bf16* dst = dQb + (long long)(qg + row) * p.lddq;
          if (p.accum_dq) {
            const bf16x8 old = *(const bf16x8*)dst;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)old[j]);
          }
          *(bf16x8*)dst = v;
        }
      }
    }
    if (KB > 2) {                                                // the two barriers of the key waves' hand-over below
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  // ================= waves 2, 3: key on the lane.  Accumulator slots: 0 = the wave's own key block (wave 2: kb 0, wave 3:
  // kb 1), 1 = its half of kb 2.  (The role loops are separate code paths so that these 128 registers are not live across
  // the dQ role's code: as one loop with a branch per unit the kernel needed 312 registers and spilled 110.)
  constexpr int NS = KB > 2 ? 2 : 1;
  f32x16 dk[NS][2], dv[NS][2];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[sl][i][r] = dv[sl][i][r] = 0.f;
  const int kw = wave - 2;                                        // 0 / 1
  for (int u = u_begin; u < u_end; ++u) {
    const char* const S0 = unit_head(u);
    const char* const Qt = S0;
    const char* const dOt = S0 + TILE_BYTES;
    const float* const rc = (const float*)(S0 + 2 * TILE_BYTES);
    // three (key block, query block) tasks per wave (two when KB == 2); phases fenced (sched_barrier) so that hipcc does not
    // hoist the next phase's fragment reads over the MFMAs beside the 128 accumulator registers
    auto key_task = [&](const int kb, const int qb, f32x16 (&dkt)[2], f32x16 (&dvt)[2]) {
      if (kb * 32 >= p.Skv) return;                             // uniform
      const char* Kmine = Ksm + (kb >> 1) * TILE_BYTES;
      const char* Vmine = Vsm + (kb >> 1) * TILE_BYTES;
      const int kmine_row = (kb & 1) * 32 + frow;
      const bool kvalid = kb * 32 + frow < skv_b;
      const bool blk_pad = kb * 32 + 32 > skv_b;                // uniform: this key block holds padding keys
      const int rb0 = qb * 32;
      f32x16 sacc, dpacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                            // rows = queries: the staged row constants ARE the C operands
        const f32x4 d4 = *(const f32x4*)(rc + 64 + rb0 + 8 * g + 4 * fh);
        const f32x4 l4 = *(const f32x4*)(rc + rb0 + 8 * g + 4 * fh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dpacc[4 * g + j] = d4[j];
          sacc[4 * g + j] = PRE ? l4[j] : 0.f;
        }
      }
      {
        bf16x8 qfr[4], kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qfr[s] = read_row_frag(Qt, rb0 + frow, s, fh);
          kf[s] = read_row_frag(Kmine, kmine_row, s, fh);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], sacc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        bf16x8 dfr[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          dfr[s] = read_row_frag(dOt, rb0 + frow, s, fh);
          vf[s] = read_row_frag(Vmine, kmine_row, s, fh);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[s], vf[s], dpacc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 pfr[2], dsfr[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
        if (!PRE) l4 = *(const f32x4*)(rc + rb0 + 8 * g + 4 * fh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          float pr = PRE ? fast_exp2(sacc[r]) : fast_exp2(fmaf(sacc[r], c, l4[j]));
          if (blk_pad) pr = kvalid ? pr : 0.f;
          const float ds = pr * dpacc[r];
          pfr[g >> 1][(g & 1) * 4 + j] = (bf16)pr;
          dsfr[g >> 1][(g & 1) * 4 + j] = (bf16)ds;
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 dot[2], qt[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dot[db] = read_transposed_frag<true>(dOt, rb0 + k2 * 16, db * 32, lane);
          qt[db] = read_transposed_frag<true>(Qt, rb0 + k2 * 16, db * 32, lane);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[db], pfr[k2], dvt[db], 0, 0, 0);
          dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[db], dsfr[k2], dkt[db], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    key_task(kw, 0, dk[0], dv[0]);
    key_task(kw, 1, dk[0], dv[0]);
    if (KB > 2) key_task(2, kw, dk[NS - 1], dv[NS - 1]);
  }
  // ---- dK / dV out.  KB == 3: wave 3 hands its half of key block 2 to wave 2 through LDS (lane-private 16-byte slots:
  // both waves hold the same (key, d) elements on the same lanes); (wave 2) + (wave 3), always in that order.
  if (KB > 2) {
    __syncthreads();                                             // every wave is done with the stages
    float* const X = (float*)St;                                 // 16 slots x 64 lanes x 16 bytes = 16 KB
    if (wave == 3) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 a, cc;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = dk[NS - 1][db][4 * g + j]; cc[j] = dv[NS - 1][db][4 * g + j]; }
          *(f32x4*)(X + ((db * 4 + g) * 2 + 0) * 256 + lane * 4) = a;
          *(f32x4*)(X + ((db * 4 + g) * 2 + 1) * 256 + lane * 4) = cc;
        }
    }
    __syncthreads();
    if (wave == 2) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = *(const f32x4*)(X + ((db * 4 + g) * 2 + 0) * 256 + lane * 4);
          const f32x4 cc = *(const f32x4*)(X + ((db * 4 + g) * 2 + 1) * 256 + lane * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { dk[NS - 1][db][4 * g + j] += a[j]; dv[NS - 1][db][4 * g + j] += cc[j]; }
        }
    }
  }
  const float dks = PRE ? 0.6931471805599453f : p.scale;         // dS^T Q' = scale log2(e) dS^T Q
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    if (sl == 1 && wave != 2) break;                              // the shared block leaves from wave 2
    const int kb = sl == 0 ? wave - 2 : 2;
    const int krow = kb * 32 + frow;
    if (krow >= p.Skv) continue;
    if (p.nsplit > 1) {
      float* pr = p.dkv_part + ((((long long)split * p.B + b) * p.H + head) * p.Skv + krow) * 128;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * fh;
          f32x4 a, cc;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = dk[sl][db][4 * g + j] * dks; cc[j] = dv[sl][db][4 * g + j]; }
          *(f32x4*)(pr + d) = a;
          *(f32x4*)(pr + 64 + d) = cc;
        }
    } else {
      bf16* dKr = p.dK + ((long long)b * p.Skv + krow) * p.lddk + head * 64;
      bf16* dVr = p.dV + ((long long)b * p.Skv + krow) * p.lddv + head * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * fh;
          bf16x4 ok, ov;
          if (p.accum_dkv) {
            ok = *(const bf16x4*)(dKr + d);
            ov = *(const bf16x4*)(dVr + d);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ok[j] = (bf16)(dk[sl][db][4 * g + j] * dks + (p.accum_dkv ? (float)ok[j] : 0.f));
            ov[j] = (bf16)(dv[sl][db][4 * g + j] + (p.accum_dkv ? (float)ov[j] : 0.f));
          }
          *(bf16x4*)(dKr + d) = ok;
          *(bf16x4*)(dVr + d) = ov;
        }
    }
  }
}

// ============================================================================= cross-attention backward, v3: FIVE products
// v2 above still computes S and dP twice (once per orientation): 7 matrix products and two exponentials per score for an
// algorithm that has 5 and one.  With <= 80 keys the whole key range of a (batch, head) sits in one workgroup, so dQ needs no
// sum across workgroups and the two orientations can share one evaluation: the KEY waves (key on the lane) compute S, dP ->
// P, dS once, feed dV^T += dO^T P and dK^T += Q^T dS from their accumulators as before, and drop dS (bf16) into an LDS
// image [key][query]; the dQ wave reads it back through the transpose read as the B operand of dQ^T = K^T dS^T (the same
// k-permutation as its A operand, the transposed K fragment), one unit behind the key waves.
//   wave 0 = dQ of all 64 queries of the previous unit (KB = 3: 20 MFMAs -- the sixth 16-key slice, keys 80..95, is padding
//   and skipped), waves 1 .. KB = key blocks 0 .. KB - 1 (32 MFMAs each); KB = 2 (33..64 keys) leaves wave 3 idle.
// 120 MFMAs and 96 exponentials per 64-query unit instead of 168 and 192, no accumulator hand-over behind the loop.
// LDS: K | V images of 96 rows (24 KB) + 2 stages x 16.5 KB + 2 dS images of 80 key rows (20 KB) = 77 KB: two workgroups
// per CU; the dQ wave's output image reuses the dS image it has just read into registers.
template <int KB, bool PRE>
__global__ __launch_bounds__(256, 2) void xattn_bwd3_kernel(const AttnP p, int upw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STG = 2 * TILE_BYTES + 512;
  constexpr int KV_IMG = TILE_BYTES + TILE_BYTES / 2;            // 96 rows
  constexpr int DS_IMG = TILE_BYTES + 2048;                      // 80 key rows x 128 bytes (64 queries)
  char* const Ksm = smem;
  char* const Vsm = smem + KV_IMG;
  char* const St = smem + 2 * KV_IMG;                            // [2 stages][Q tile | dO tile | lse[64] | delta[64]]
  char* const Dsm = St + 2 * STG;                                // [2][80 keys][64 queries] bf16, tile format
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  int split, head, b;
  attn_block_coords(p.xcd_remap, split, head, b);
  const int nu = (p.Sq + 63) >> 6;
  const int u_begin = split * upw, u_end = min(nu, u_begin + upw);
  const float c = PRE ? 1.f : p.scale * LOG2E;
  const bf16* Qb = p.Q + (long long)b * p.Sq * p.ldq + head * 64;
  const bf16* dOb = p.dO + (long long)b * p.Sq * p.lddo + head * 64;
  const bf16* Ob = p.O + (long long)b * p.Sq * p.ldo + head * 64;
  const bf16* Kb = p.K + (long long)b * p.Skv * p.ldk + head * 64;
  const bf16* Vb = p.V + (long long)b * p.Skv * p.ldv + head * 64;
  const float* lseb = p.lse + ((long long)b * p.H + head) * p.Sq;
  const int skv_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.Skv);

  const TileSrc qsrc = tile_src(Qb, p.ldq, p.Sq, wave, lane), dosrc = tile_src(dOb, p.lddo, p.Sq, wave, lane);
  auto stage_unit = [&](int u, int st) {
    char* const S0 = St + st * STG;
    const int r0 = u * 64;
    stage_tile(qsrc, r0, S0, wave);
    stage_tile(dosrc, r0, S0 + TILE_BYTES, wave);
    if (wave == 0) {
      int r = r0 + lane;
      r = r < p.Sq ? r : p.Sq - 1;
      lds_dma4(lseb + r, S0 + 2 * TILE_BYTES);
    }
  };
  const int dq_l = tid >> 2, dqt = tid & 3;                       // delta: thread = (query of the unit, quarter of the head dim)
  bf16x8 o_pf[2];
  auto fetch_o = [&](int u) {
    int r = u * 64 + dq_l;
    r = r < p.Sq ? r : p.Sq - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) o_pf[i] = *(const bf16x8*)(Ob + (long long)r * p.ldo + dqt * 16 + 8 * i);
  };
  {
    const TileSrc ksrc = tile_src(Kb, p.ldk, p.Skv, wave, lane), vsrc = tile_src(Vb, p.ldv, p.Skv, wave, lane);
    stage_tile(ksrc, 0, Ksm, wave);
    stage_tile(vsrc, 0, Vsm, wave);
    if (KB > 2 && wave < 2) {                                     // rows 64..95: the first four 1 KiB pieces of the second tile
      stage_tile(ksrc, 64, Ksm + TILE_BYTES, wave);
      stage_tile(vsrc, 64, Vsm + TILE_BYTES, wave);
    }
  }
  if (u_begin < u_end) {
    stage_unit(u_begin, 0);
    fetch_o(u_begin);
  }
  if (p.red_part) {                                               // deferred split reduce of another launch (see xattn_bwd2_kernel)
    const long long total = (long long)p.red_B * p.red_H * p.red_Skv * 32;
    const long long nthr = (long long)gridDim.x * gridDim.y * gridDim.z * 256;
    const long long first = ((long long)blockIdx.x + gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z)) * 256 + tid;
    for (long long idx = first; idx < total; idx += nthr)
      dkv_reduce_elem64(p.red_part, p.red_dK, p.red_dV, p.red_lddk, p.red_lddv, p.red_nsplit, p.red_B, p.red_H, p.red_Skv, p.red_accum, idx);
  }
  // wave roles
  const bool key_wave = wave >= 1 && wave <= KB;
  const bool dq_wave = wave == 0;
  const int my_kb = wave - 1;                                     // key waves

  auto unit_head = [&](int u) -> char* {                          // (as in xattn_bwd2_kernel)
    const int cur = (u - u_begin) & 1;
    char* const S0 = St + cur * STG;
    float* const rc = (float*)(S0 + 2 * TILE_BYTES);
    WAIT_VM0();
    __syncthreads();                                              // unit u landed; unit u - 1 is complete (its dS image too)
    if (u + 1 < u_end) stage_unit(u + 1, cur ^ 1);
    {
      const char* dt = S0 + TILE_BYTES;
      float dsum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 d = *(const bf16x8*)(dt + dq_l * 128 + (((dqt * 2 + i) ^ swz_x(dq_l)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum += (float)d[j] * (float)o_pf[i][j];
      }
      dsum += dpp_move<0xB1>(dsum);
      dsum += dpp_move<0x4E>(dsum);
      if (dqt == 0) {
        const bool rv = u * 64 + dq_l < p.Sq;
        rc[64 + dq_l] = rv ? -dsum : 0.f;
        rc[dq_l] = rv ? -rc[dq_l] * LOG2E : -INFINITY;
      }
    }
    if (u + 1 < u_end) fetch_o(u + 1);
    __syncthreads();
    return S0;
  };

  if (!key_wave) {
    // ================= dQ wave (and, with KB == 2, an idle wave that only keeps the barriers' count):
    // dQ^T[d][q] = K^T[d][key] dS^T[key][q] of the PREVIOUS unit, from its dS image
    auto dq_unit = [&](int u) {                                   // u: the unit whose dS image is complete
      char* const img = Dsm + ((u - u_begin) & 1) * DS_IMG;
      constexpr int NSL = KB > 2 ? 5 : 4;                         // 16-key slices that hold keys (< 80)
      f32x16 oacc[2][2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[q2][i][r] = 0.f;
      // all dS fragments first: the image is then free and becomes this wave's output image
      bf16x8 dsf[2][NSL];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
          dsf[q2][sl] = read_transposed_frag<true>(img + (sl >> 2) * TILE_BYTES, (sl & 3) * 16, q2 * 32, lane);
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        bf16x8 kt[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) kt[db] = read_transposed_frag<true>(Ksm + (sl >> 2) * TILE_BYTES, (sl & 3) * 16, db * 32, lane);
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
          for (int db = 0; db < 2; ++db)
            oacc[q2][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[db], dsf[q2][sl], oacc[q2][db], 0, 0, 0);
      }
      const int r8 = lane >> 3, ch = lane & 7;
      bf16* dQb = p.dQ + (long long)b * p.Sq * p.lddq + head * 64 + ch * 8;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        char* const o2 = img + q2 * (32 * 144);                   // (2 x 4.5 KiB <= the 10 KiB image)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)(oacc[q2][db][4 * g + j] * p.scale);
            *(bf16x4*)(o2 + frow * 144 + (db * 32 + 8 * g + 4 * fh) * 2) = o;
          }
        const int qg = u * 64 + q2 * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + r8;
          if (qg + row < p.Sq) {
            bf16x8 v = *(const bf16x8*)(o2 + row * 144 + ch * 16);
            bf16* dst = dQb + (long long)(qg + row) * p.lddq;
            if (p.accum_dq) {
              const bf16x8 old = *(const bf16x8*)dst;
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)old[j]);
            }
            *(bf16x8*)dst = v;
          }
        }
      }
    };
    for (int u = u_begin; u < u_end; ++u) {
      (void)unit_head(u);
      if (dq_wave && u > u_begin) dq_unit(u - 1);
    }
    if (u_begin < u_end) {
      __syncthreads();                                            // the key waves have finished the last unit's dS image
      if (dq_wave) dq_unit(u_end - 1);
    }
    return;
  }

  // ================= key waves: one key block each, S / dP / P / dS ONCE per (key block, query block)
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = 0.f;
  const int kb = my_kb;
  const bool blk_live = kb * 32 < p.Skv;                          // uniform
  const char* const Kmine = Ksm + (kb >> 1) * TILE_BYTES;
  const char* const Vmine = Vsm + (kb >> 1) * TILE_BYTES;
  const int kmine_row = (kb & 1) * 32 + frow;
  const int key = kb * 32 + frow;
  const bool kvalid = key < skv_b;
  const bool blk_pad = kb * 32 + 32 > skv_b;                      // uniform: this key block holds padding keys
  for (int u = u_begin; u < u_end; ++u) {
    const char* const S0 = unit_head(u);
    const char* const Qt = S0;
    const char* const dOt = S0 + TILE_BYTES;
    const float* const rc = (const float*)(S0 + 2 * TILE_BYTES);
    char* const img = Dsm + ((u - u_begin) & 1) * DS_IMG;
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
      const int rb0 = qb * 32;
      f32x16 sacc, dpacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 d4 = *(const f32x4*)(rc + 64 + rb0 + 8 * g + 4 * fh);
        const f32x4 l4 = *(const f32x4*)(rc + rb0 + 8 * g + 4 * fh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dpacc[4 * g + j] = d4[j];
          sacc[4 * g + j] = PRE ? l4[j] : 0.f;
        }
      }
      {
        bf16x8 qfr[4], kf[4], dfr[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qfr[s] = read_row_frag(Qt, rb0 + frow, s, fh);
          kf[s] = read_row_frag(Kmine, kmine_row, s, fh);
          dfr[s] = read_row_frag(dOt, rb0 + frow, s, fh);
          vf[s] = read_row_frag(Vmine, kmine_row, s, fh);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {                             // two independent accumulation chains, interleaved
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], sacc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[s], vf[s], dpacc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 pfr[2], dsfr[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
        if (!PRE) l4 = *(const f32x4*)(rc + rb0 + 8 * g + 4 * fh);
        bf16x4 ds4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          float pr = PRE ? fast_exp2(sacc[r]) : fast_exp2(fmaf(sacc[r], c, l4[j]));
          if (blk_pad) pr = kvalid ? pr : 0.f;
          const float ds = pr * dpacc[r];
          pfr[g >> 1][(g & 1) * 4 + j] = (bf16)pr;
          dsfr[g >> 1][(g & 1) * 4 + j] = (bf16)ds;
          ds4[j] = (bf16)ds;
        }
        // dS[key][q = rb0 + 8 g + 4 fh .. + 3] -> the image row of this lane's key (keys >= 80 have no row: they are padding)
        if (key < 80) *(bf16x4*)(img + (kb >> 1) * TILE_BYTES + swz_rc(kmine_row, rb0 + 8 * g + 4 * fh)) = ds4;
      }
      if (blk_live) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 dot[2], qt[2];
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dot[db] = read_transposed_frag<true>(dOt, rb0 + k2 * 16, db * 32, lane);
            qt[db] = read_transposed_frag<true>(Qt, rb0 + k2 * 16, db * 32, lane);
          }
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[db], pfr[k2], dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[db], dsfr[k2], dk[db], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (u_begin < u_end) __syncthreads();                           // (the dQ waves' last hand-over)
  const int krow = kb * 32 + frow;
  if (krow >= p.Skv) return;
  const float dks = PRE ? 0.6931471805599453f : p.scale;
  if (p.nsplit > 1) {
    float* pr = p.dkv_part + ((((long long)split * p.B + b) * p.H + head) * p.Skv + krow) * 128;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * fh;
        f32x4 a, cc;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = dk[db][4 * g + j] * dks; cc[j] = dv[db][4 * g + j]; }
        *(f32x4*)(pr + d) = a;
        *(f32x4*)(pr + 64 + d) = cc;
      }
  } else {
    bf16* dKr = p.dK + ((long long)b * p.Skv + krow) * p.lddk + head * 64;
    bf16* dVr = p.dV + ((long long)b * p.Skv + krow) * p.lddv + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * fh;
        bf16x4 ok, ov;
        if (p.accum_dkv) {
          ok = *(const bf16x4*)(dKr + d);
          ov = *(const bf16x4*)(dVr + d);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ok[j] = (bf16)(dk[db][4 * g + j] * dks + (p.accum_dkv ? (float)ok[j] : 0.f));
          ov[j] = (bf16)(dv[db][4 * g + j] + (p.accum_dkv ? (float)ov[j] : 0.f));
        }
        *(bf16x4*)(dKr + d) = ok;
        *(bf16x4*)(dVr + d) = ov;
      }
  }
}

// row constants of the backward kernels: delta[0][b][h][q] = -sum_d dO[q][h*D+d] * O[q][h*D+d], delta[1][b][h][q] = -lse * log2(e);
// 8 lanes per (row, head), each sums D/8 elements
#ifndef PEA_ATTN_DELTA_ROWS
#define PEA_ATTN_DELTA_ROWS 4      // (row, head) slots per thread, all requested before the first use: a thread with ONE pair of 16-byte loads in
                                   // flight left the kernel latency-bound (9.6 us for 21 MB); 1 restores that form
#endif
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnP p) {
  constexpr int NR = PEA_ATTN_DELTA_ROWS;
  const long long total = (long long)p.B * p.Sq * p.H * 8;            // one lane per (b, q, head, 8-lane slot)
  const long long stride = (long long)gridDim.x * 256;                // slot r of a thread: idx + r * stride (coalesced per r)
  const long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const int D = 64 * p.nd;
  bf16x8 a[NR], o[NR];
  float lse[NR];
  long long li[NR];
  bool ok[NR];
  if (p.nd == 1) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const long long idx = idx0 + r * stride;
      ok[r] = idx < total;
      const int sub = (int)(idx & 7);
      const long long row = (ok[r] ? idx : 0) >> 3;
      const int head = (int)(row % p.H);
      const long long bq = row / p.H;
      const int b = (int)(bq / p.Sq), q = (int)(bq % p.Sq);
      li[r] = ((long long)b * p.H + head) * p.Sq + q;
      a[r] = *(const bf16x8*)(p.dO + bq * p.lddo + head * D + sub * 8);
      o[r] = *(const bf16x8*)(p.O + bq * p.ldo + head * D + sub * 8);
      lse[r] = p.lse[li[r]];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)a[r][j] * (float)o[r][j];
      s += dpp_move<0xB1>(s);                                  // xor 1, xor 2, xor 4 over the 8 lanes of a row: DPP moves, no LDS
      s += dpp_move<0x4E>(s);
      s += dpp_move<0x141>(s);
      if (ok[r] && ((idx0 + r * stride) & 7) == 0) {
        p.delta[li[r]] = -s;                                                     // C operand of the dP products
        p.delta[(long long)p.B * p.H * p.Sq + li[r]] = -lse[r] * LOG2E;          // addend of the exp2 argument
      }
    }
    return;
  }
  for (int r = 0; r < NR; ++r) {                                // padded head widths (text towers, SD1.5): the plain loop
    const long long idx = idx0 + r * stride;
    const int sub = (int)(idx & 7);
    float s = 0.f, l = 0.f;
    long long row = idx >> 3, lin = 0;
    if (idx < total) {
      const int head = (int)(row % p.H);
      const long long bq = row / p.H;
      const int b = (int)(bq / p.Sq), q = (int)(bq % p.Sq);
      lin = ((long long)b * p.H + head) * p.Sq + q;
      if (sub == 0) l = p.lse[lin];
      for (int nd = 0; nd < p.nd; ++nd) {
        const bf16x8 av = *(const bf16x8*)(p.dO + bq * p.lddo + head * D + nd * 64 + sub * 8);
        const bf16x8 ov = *(const bf16x8*)(p.O + bq * p.ldo + head * D + nd * 64 + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)av[j] * (float)ov[j];
      }
    }
    s += dpp_move<0xB1>(s);
    s += dpp_move<0x4E>(s);
    s += dpp_move<0x141>(s);
    if (idx < total && sub == 0) {
      p.delta[lin] = -s;
      p.delta[(long long)p.B * p.H * p.Sq + lin] = -l * LOG2E;
    }
  }
}

// out[b][key][h*D+d] (+)= sum_split part[split][b][h][key][{dK,dV}][d]   (splits added in order)
__global__ void attn_dkv_reduce_kernel(const AttnP p) {
  const int D = 64 * p.nd, D4 = D / 4;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // (b, h, key, which, d/4)
  const long long total = (long long)p.B * p.H * p.Skv * 2 * D4;
  if (idx >= total) return;
  const int d4 = (int)(idx % D4) * 4;
  long long r = idx / D4;
  const int which = (int)(r & 1); r >>= 1;
  const int key = (int)(r % p.Skv); r /= p.Skv;
  const int head = (int)(r % p.H);
  const int b = (int)(r / p.H);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.nsplit; ++s) {
    const f32x4 v = *(const f32x4*)(p.dkv_part + ((((long long)s * p.B + b) * p.H + head) * p.Skv + key) * 2 * D +
                                    which * D + d4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
  bf16* dst = which ? p.dV + ((long long)b * p.Skv + key) * p.lddv + head * D + d4
                    : p.dK + ((long long)b * p.Skv + key) * p.lddk + head * D + d4;
  bf16x4 o;
  if (p.accum_dkv) o = *(const bf16x4*)dst;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[j] + (p.accum_dkv ? (float)o[j] : 0.f));
  *(bf16x4*)dst = o;
}

// query-range splitting of the cross-attention backward (few keys, many queries): a workgroup owns `u` consecutive
// 128-query units of one (batch, head); pick the u that minimises (rounds of 512 workgroups: two per CU) x u, then the
// split count (fp32 partials per split).  1024 tokens, B*H = 80: u = 2 -> 320 workgroups, 4 splits; 4096 tokens, 40:
// u = 3 -> 440 workgroups, 11 splits.
// v2 of the one-pass cross-attention backward (xattn_bwd2_kernel: 64-query units, specialised waves): 33 .. 96 keys
// which one-pass kernel: 0 = round 3 (xattn_bwd_kernel) for every key count; 2 = xattn_bwd2_kernel where it applies (33..96
// keys); 1 / 3 (default) = the newest that applies: xattn_bwd3_kernel for 33..80 keys, xattn_bwd2_kernel for 81..96.
// PEA_XATTN_BWD_V1=1 / PEA_XATTN_BWD_VER=n in the environment, pea_debug_set_xattn_bwd_v2(n) at run time (A/B, parity tests).
static int g_xattn_v2 = getenv("PEA_XATTN_BWD_V1") ? 0 : (getenv("PEA_XATTN_BWD_VER") ? atoi(getenv("PEA_XATTN_BWD_VER")) : 3);
extern "C" void pea_debug_set_xattn_bwd_v2(int v) { g_xattn_v2 = v; }
static bool xattn_v2_keys(int Skv) { return g_xattn_v2 && Skv > 32 && Skv <= 96; }     // v2 OR v3: 64-query units, deferrable reduce
static bool xattn_v3_keys(int Skv) { return g_xattn_v2 != 0 && g_xattn_v2 != 2 && Skv > 32 && Skv <= 80; }
int attention_bwd_nsplit(int B, int H, int Sq, int Skv) {
  if (Skv > 128 || Sq < 512) return 1;
  const bool v2 = xattn_v2_keys(Skv);
  const int uq = v2 ? 64 : 128;                                          // queries per unit
  // v2: a workgroup's fixed part (K / V staging, the first unit's flight, the hand-over and the partial stores) is worth
  // about two of its 64-query units; without it the rule cut B * H = 160 (per-GPU batch 8) into 2560 one-unit workgroups:
  // 57.8 us against the 41.2 us of the round-3 kernel.  (v1 keeps the rule it was tuned with.)
  const int fixed = v2 ? 2 : 0;
  const int nu = (Sq + uq - 1) / uq;
  const long long bh = (long long)B * H;
  long long best = -1;
  int best_ns = 1;
  for (int u = 1; u <= nu; ++u) {
    const int ns = (nu + u - 1) / u;
    const long long rounds = (bh * ns + 511) / 512;
    const long long cost = rounds * (u + fixed) * (uq / 2) + ns;
    if (best < 0 || cost < best) { best = cost; best_ns = ns; }
  }
  return best_ns;
}
size_t attention_bwd_scratch_bytes(int B, int H, int Sq, int Skv, int nd) {
  const int ns = attention_bwd_nsplit(B, H, Sq, Skv);
  return ns > 1 ? (size_t)ns * B * H * Skv * 128 * nd * sizeof(float) : 0;
}

static int g_attn_use_tr = 1;
static int g_attn_xattn = getenv("PEA_XATTN_OFF") ? 0 : 1;               // PEA_XATTN_OFF=1: cross-attention on the general kernels (A/B)
extern "C" void pea_debug_set_attn_xattn(int v) { g_attn_xattn = v; }
static int g_attn_xcd = getenv("PEA_ATTN_NO_XCD") ? 0 : 1;
extern "C" void pea_debug_set_attn_tr(int v) { g_attn_use_tr = v; }

static int attn_check(const AttnP& p) {
  SHAPECHK(p.B > 0 && p.H > 0 && p.Sq > 0 && p.Skv > 0, "attention: empty problem");
  SHAPECHK(p.nd >= 1 && p.nd <= 3, "attention: padded head_dim must be 64, 128 or 192 (nd=%d)", p.nd);
  SHAPECHK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0, "attention: leading dims must be multiples of 8");
  return PEA_OK;
}

#define ATTN_DISPATCH(KERNEL_TR, KERNEL_NOTR, grid, lds)                                        \
  do {                                                                                          \
    if (g_attn_use_tr) hipLaunchKernelGGL(KERNEL_TR, grid, dim3(256), lds, s, p);               \
    else hipLaunchKernelGGL(KERNEL_NOTR, grid, dim3(256), lds, s, p);                           \
  } while (0)

template <int ND>
static int attn_set_lds_attr() {
  static bool done = false;
  if (done) return PEA_OK;
  const int lq = 2 * 2 * ND * TILE_BYTES, lk = 2 * (2 * ND * TILE_BYTES + 512);
  HIPCHK(hipFuncSetAttribute((const void*)attn_q_kernel<0, true, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lq));
  HIPCHK(hipFuncSetAttribute((const void*)attn_q_kernel<0, false, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lq));
  HIPCHK(hipFuncSetAttribute((const void*)attn_q_kernel<1, true, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lq));
  HIPCHK(hipFuncSetAttribute((const void*)attn_q_kernel<1, false, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lq));
  HIPCHK(hipFuncSetAttribute((const void*)attn_dkv_kernel<true, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lk));
  HIPCHK(hipFuncSetAttribute((const void*)attn_dkv_kernel<false, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lk));
  done = true;
  return PEA_OK;
}

template <int ND>
static int attn_fwd_nd(const AttnP& p, hipStream_t s) {
  int rc = attn_set_lds_attr<ND>();
  if (rc) return rc;
  const dim3 grid(cdiv(p.Sq, 128), p.H, p.B);
  if constexpr (ND == 1) {
    if (g_attn_xattn && g_attn_use_tr && p.Skv <= 128 && !p.causal && !p.bias && p.Sq >= 128) {   // cross-attention: K / V resident
      const int kb = cdiv(p.Skv, 32), kt = (kb + 1) / 2;
      const int lds = 2 * kt * TILE_BYTES + 4 * 32 * 144 + 4 * 4096;
      const int nu = cdiv(p.Sq, 128);
      const long long units = (long long)p.B * p.H * nu;
      int upw = (int)((units + 511) / 512);                       // one round of two workgroups per CU (66 KB of LDS each)
      upw = upw < 1 ? 1 : (upw > nu ? nu : upw);
      const dim3 g2(cdiv(nu, upw), p.H, p.B);
      if (kb == 1) hipLaunchKernelGGL(xattn_fwd_kernel<1>, g2, dim3(256), lds, s, p, upw);
      else if (kb == 2) hipLaunchKernelGGL(xattn_fwd_kernel<2>, g2, dim3(256), lds, s, p, upw);
      else if (kb == 3) hipLaunchKernelGGL(xattn_fwd_kernel<3>, g2, dim3(256), lds, s, p, upw);
      else hipLaunchKernelGGL(xattn_fwd_kernel<4>, g2, dim3(256), lds, s, p, upw);
      return PEA_OK;
    }
  }
  if (p.causal || p.kv_len || p.bias) {       // text-encoder masks / score bias: separate instance (head_dim 64 only)
    SHAPECHK(ND == 1, "attention: causal / key-length masks need head_dim 64");
    if constexpr (ND == 1) {
      static bool attr = false;
      if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)attn_q_kernel<0, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   2 * 2 * TILE_BYTES));
        attr = true;
      }
      hipLaunchKernelGGL((attn_q_kernel<0, true, 1, true>), grid, dim3(256), 2 * 2 * TILE_BYTES, s, p);
    }
    return PEA_OK;
  }
  ATTN_DISPATCH((attn_q_kernel<0, true, ND>), (attn_q_kernel<0, false, ND>), grid, 2 * 2 * ND * TILE_BYTES);
  return PEA_OK;
}
static int g_attn_fused_bwd = getenv("PEA_ATTN_BWD_SPLIT") ? 0 : 1;      // PEA_ATTN_BWD_SPLIT=1: the two-launch form (A/B)
// the one-pass cross-attention backward: head_dim 64, at most 128 keys, all three gradients wanted
static bool attn_use_xattn(const AttnP& p) {
  return g_attn_xattn && g_attn_use_tr && p.nd == 1 && p.Skv <= 128 && p.dQ && p.dK && p.dV && (p.nsplit <= 1 || p.dkv_part);
}
extern "C" void pea_debug_set_attn_fused_bwd(int v) { g_attn_fused_bwd = v; }
// PEA_XATTN_DEFER=1: a cross-attention layer's split reduce rides in the next such launch's prologue instead of its own 5.8 us
// launch.  Built for VERDICT r05 item 1a ("a single batched launch per step instead of 70"), parity-green (per-layer K / V
// gradient checks of the full model), and worth NOTHING in the step: 99.86 / 99.76 ms with it, 99.78 / 99.87 without, alternating
// processes on one box (profiles/r06_ab_defer_reduce.log) -- the prologue's extra loads cost what the launches did.  Off.
static int g_xattn_defer = getenv("PEA_XATTN_DEFER") ? atoi(getenv("PEA_XATTN_DEFER")) : 0;
int attention_bwd_defers(const AttnP& p0) {
  AttnP p = p0;
  if (p.nd == 0) p.nd = 1;
  p.nsplit = p.dkv_part ? attention_bwd_nsplit(p.B, p.H, p.Sq, p.Skv) : 1;
  return (g_xattn_defer && attn_use_xattn(p) && xattn_v2_keys(p.Skv) && p.nsplit > 1) ? 1 : 0;
}
void attention_set_deferred(AttnP& cur, const AttnP& pend) {
  cur.red_part = pend.dkv_part; cur.red_dK = pend.dK; cur.red_dV = pend.dV; cur.red_lddk = pend.lddk; cur.red_lddv = pend.lddv;
  cur.red_nsplit = attention_bwd_nsplit(pend.B, pend.H, pend.Sq, pend.Skv);
  cur.red_B = pend.B; cur.red_H = pend.H; cur.red_Skv = pend.Skv; cur.red_accum = pend.accum_dkv;
}
int launch_attention_dkv_reduce(const AttnP& pend, hipStream_t s) {
  AttnP p = pend;
  if (p.nd == 0) p.nd = 1;
  p.nsplit = attention_bwd_nsplit(p.B, p.H, p.Sq, p.Skv);
  SHAPECHK(p.dkv_part && p.nsplit > 1 && p.dK && p.dV, "attention: nothing to reduce");
  const long long total = (long long)p.B * p.H * p.Skv * 2 * 16 * p.nd;
  hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, s, p);
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
template <int ND>
static int attn_bwd_nd(const AttnP& p, hipStream_t s) {
  int rc = attn_set_lds_attr<ND>();
  if (rc) return rc;
  if constexpr (ND == 1) {
    if (attn_use_xattn(p)) {
      constexpr int lds = 4 * TILE_BYTES + (4 * TILE_BYTES + 1024);
      static bool attr = false;
      if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
      }
      const int ns = p.nsplit > 1 ? p.nsplit : 1;
      const dim3 grid(ns, p.H, p.B);
      const int kb = cdiv(p.Skv, 32);
      if (xattn_v2_keys(p.Skv)) {
        constexpr int lds2 = 4 * TILE_BYTES + 2 * (2 * TILE_BYTES + 512) + 2 * 32 * 144;
        static bool attr2 = false;
        if (!attr2) {
          HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd2_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
          HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd2_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
          HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd2_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
          HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd2_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
          attr2 = true;
        }
        const int upw2 = cdiv(cdiv(p.Sq, 64), ns);
        if (xattn_v3_keys(p.Skv)) {
          constexpr int lds3 = 2 * (TILE_BYTES + TILE_BYTES / 2) + 2 * (2 * TILE_BYTES + 512) + 2 * (TILE_BYTES + 2048);
          static bool attr3 = false;
          if (!attr3) {
            HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd3_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
            HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd3_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
            HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd3_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
            HIPCHK(hipFuncSetAttribute((const void*)xattn_bwd3_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
            attr3 = true;
          }
          if (kb == 2) {
            if (p.q_prescaled) hipLaunchKernelGGL((xattn_bwd3_kernel<2, true>), grid, dim3(256), lds3, s, p, upw2);
            else hipLaunchKernelGGL((xattn_bwd3_kernel<2, false>), grid, dim3(256), lds3, s, p, upw2);
          } else {
            if (p.q_prescaled) hipLaunchKernelGGL((xattn_bwd3_kernel<3, true>), grid, dim3(256), lds3, s, p, upw2);
            else hipLaunchKernelGGL((xattn_bwd3_kernel<3, false>), grid, dim3(256), lds3, s, p, upw2);
          }
        } else if (kb == 2) {
          if (p.q_prescaled) hipLaunchKernelGGL((xattn_bwd2_kernel<2, true>), grid, dim3(256), lds2, s, p, upw2);
          else hipLaunchKernelGGL((xattn_bwd2_kernel<2, false>), grid, dim3(256), lds2, s, p, upw2);
        } else {
          if (p.q_prescaled) hipLaunchKernelGGL((xattn_bwd2_kernel<3, true>), grid, dim3(256), lds2, s, p, upw2);
          else hipLaunchKernelGGL((xattn_bwd2_kernel<3, false>), grid, dim3(256), lds2, s, p, upw2);
        }
      } else {
      const int nu = cdiv(p.Sq, 128), upw = cdiv(nu, ns);
      if (kb == 1) hipLaunchKernelGGL(xattn_bwd_kernel<1>, grid, dim3(256), lds, s, p, upw);
      else if (kb == 2) hipLaunchKernelGGL(xattn_bwd_kernel<2>, grid, dim3(256), lds, s, p, upw);
      else if (kb == 3) hipLaunchKernelGGL(xattn_bwd_kernel<3>, grid, dim3(256), lds, s, p, upw);
      else hipLaunchKernelGGL(xattn_bwd_kernel<4>, grid, dim3(256), lds, s, p, upw);
      }
      if (p.nsplit > 1 && !(p.defer_reduce && xattn_v2_keys(p.Skv))) {
        const long long total = (long long)p.B * p.H * p.Skv * 2 * 16;
        hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, s, p);
      }
      return PEA_OK;
    }
  }
  if (p.dQ && p.dK && p.dV && g_attn_fused_bwd) {
    static bool attr = false;
    constexpr int lds = 2 * (2 * ND * TILE_BYTES + 512);
    if (!attr) {
      HIPCHK(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<true, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      HIPCHK(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<false, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr = true;
    }
    const int n_dq = cdiv(p.Sq, 128) * ND, n_dkv = cdiv(p.Skv, 128) * (p.nsplit > 1 ? p.nsplit : 1) * ND;
    const dim3 grid((unsigned)((n_dq + n_dkv) * p.H * p.B));
    static const int heavy_first = getenv("PEA_ATTN_BWD_HEAVY_FIRST") ? atoi(getenv("PEA_ATTN_BWD_HEAVY_FIRST")) : 0;
    if (g_attn_use_tr) hipLaunchKernelGGL((attn_bwd_fused_kernel<true, ND>), grid, dim3(256), lds, s, p, n_dq, n_dkv, heavy_first);
    else hipLaunchKernelGGL((attn_bwd_fused_kernel<false, ND>), grid, dim3(256), lds, s, p, n_dq, n_dkv, heavy_first);
    if (p.nsplit > 1) {
      const long long total = (long long)p.B * p.H * p.Skv * 2 * 16 * ND;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, s, p);
    }
    return PEA_OK;
  }
  if (p.dQ) {
    const dim3 grid(cdiv(p.Sq, 128) * ND, p.H, p.B);
    ATTN_DISPATCH((attn_q_kernel<1, true, ND>), (attn_q_kernel<1, false, ND>), grid, 2 * 2 * ND * TILE_BYTES);
  }
  if (p.dK && p.dV) {
    const dim3 grid(cdiv(p.Skv, 128) * (p.nsplit > 1 ? p.nsplit : 1) * ND, p.H, p.B);
    ATTN_DISPATCH((attn_dkv_kernel<true, ND>), (attn_dkv_kernel<false, ND>), grid, 2 * (2 * ND * TILE_BYTES + 512));
    if (p.nsplit > 1) {
      const long long total = (long long)p.B * p.H * p.Skv * 2 * 16 * ND;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, s, p);
    }
  }
  return PEA_OK;
}

int launch_attention_fwd(const AttnP& p0, hipStream_t s) {
  AttnP p = p0;
  p.xcd_remap = g_attn_xcd;
  if (p.nd == 0) p.nd = 1;
  int rc = attn_check(p);
  if (rc) return rc;
  SHAPECHK(p.ldo % 4 == 0, "attention: ldo %% 4");
  if (g_prof_on) { g_prof_tag[0] = p.B * p.H; g_prof_tag[1] = p.Sq; g_prof_tag[2] = p.Skv; g_prof_tag[3] = p.nd; }
  PROF_BEGIN(2, 4.0 * p.B * p.H * (double)p.Sq * p.Skv * 64 * p.nd, 2.0 * p.B * p.H * 64 * p.nd * (2.0 * p.Sq + 2.0 * p.Skv), s);
  rc = p.nd == 1 ? attn_fwd_nd<1>(p, s) : p.nd == 2 ? attn_fwd_nd<2>(p, s) : attn_fwd_nd<3>(p, s);
  PROF_END(s);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return PEA_OK;
}

int launch_attention_bwd(const AttnP& p0, hipStream_t s) {
  AttnP p = p0;
  p.xcd_remap = g_attn_xcd;
  if (p.nd == 0) p.nd = 1;
  p.nsplit = p.dkv_part ? attention_bwd_nsplit(p.B, p.H, p.Sq, p.Skv) : 1;
  int rc = attn_check(p);
  if (rc) return rc;
  SHAPECHK(p.lse && p.delta && p.dO && p.O, "attention bwd: lse/delta/dO/O required");
  SHAPECHK(p.Sq % 4 == 0, "attention bwd: Sq=%d must be a multiple of 4", p.Sq);
  const long long total = (long long)p.B * p.Sq * p.H * 8;
  // algorithmic: 5 products (S, dP, dV, dK, dQ) = 10*B*H*Sq*Skv*D flops (the two-kernel form recomputes S and dP)
  if (g_prof_on) { g_prof_tag[0] = p.B * p.H; g_prof_tag[1] = p.Sq; g_prof_tag[2] = p.Skv; g_prof_tag[3] = p.nd; }
  PROF_BEGIN(3, 10.0 * p.B * p.H * (double)p.Sq * p.Skv * 64 * p.nd, 2.0 * p.B * p.H * 64 * p.nd * (4.0 * p.Sq + 4.0 * p.Skv), s);
  // delta: its own (memory-bound) kernel for the fused launch and for dK/dV-only calls; the two-launch form computes it
  // inside the dQ pass
  if (!attn_use_xattn(p) && (!p.dQ || (p.dK && p.dV && g_attn_fused_bwd)))
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdivl(total, 256 * PEA_ATTN_DELTA_ROWS)), dim3(256), 0, s, p);
  rc = p.nd == 1 ? attn_bwd_nd<1>(p, s) : p.nd == 2 ? attn_bwd_nd<2>(p, s) : attn_bwd_nd<3>(p, s);
  PROF_END(s);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return PEA_OK;
}
