// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PEA_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define PEA_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

#define PEA_OK 0
#define PEA_E_INVALID (-1)
#define PEA_E_HIP (-2)
#define PEA_E_SHAPE (-3)
#define PEA_E_STATE (-4)
#define PEA_E_NOTFOUND (-5)
#define PEA_E_TIMEOUT (-6)

// thread-local last error text (C-ABI: pea_last_error)
void pea_set_error(const char* fmt, ...);

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e__ = (x);                                                                      \
    if (e__ != hipSuccess) {                                                                   \
      pea_set_error("%s:%d hip error %d (%s) in %s", __FILE__, __LINE__, (int)e__,            \
                    hipGetErrorString(e__), #x);                                              \
      return PEA_E_HIP;                                                                        \
    }                                                                                          \
  } while (0)

#define SHAPECHK(cond, ...)                                                                    \
  do {                                                                                         \
    if (!(cond)) {                                                                             \
      pea_set_error(__VA_ARGS__);                                                              \
      return PEA_E_SHAPE;                                                                      \
    }                                                                                          \
  } while (0)

// 16-byte store, write-through when WT (sc1: the line leaves the XCD's L2 as it is written instead of waiting there for the
// end-of-kernel release).  Inline assembly: invisible to hipcc's vmcnt bookkeeping -- only for stores nothing in the kernel waits on.
typedef __attribute__((ext_vector_type(4))) unsigned pea_u32x4;
template <bool WT>
__device__ __forceinline__ void store16(void* ptr, bf16x8 v) {
#if defined(__gfx950__) || defined(__gfx942__) || !defined(__HIP_DEVICE_COMPILE__)
  if constexpr (WT) {
    union { bf16x8 h; pea_u32x4 u; } c;
    c.h = v;
    // s_nop 1: a 16-byte store reads its data registers for two more cycles -- the wait states hipcc pads before a following write
    // of those registers when IT emits the store.  Without it the next chunk's values overwrote the data in flight (round 5: the
    // first form of this helper stored garbage in the LayerNorm backward; caught by smoke(), not by the timing A/B).
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(ptr), "v"(c.u) : "memory");
  } else {
    *(bf16x8*)ptr = v;
  }
#else      // `make ARCH=...` for another target: the sc1 cache-policy bit and the wait-state count are gfx94x / gfx950 forms
  *(bf16x8*)ptr = v;
#endif
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-epsilon level; every consumer rounds to bf16):
// 1 v_rcp + 1 v_exp + 7 fma/mul instead of libm's two-branch erff -- the GELU sits in GEMM epilogues where the
// VALU work is not hidden behind HBM
__device__ __forceinline__ float erf_fast(float x, float* exp_mx2 = nullptr) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);   // exp(-x^2)
  if (exp_mx2) *exp_mx2 = e;
  return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
// gelu_new (HF NewGELUActivation, T5 v1.1 / mT5): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))); tanh(u) = 1 - 2 / (exp(2u) + 1)
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * u);          // exp(2u)
  const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  return 0.5f * x * (1.0f + th);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x);  phi(x) = exp(-x^2/2)/sqrt(2 pi) shares the exponential with erf(x/sqrt2)
  float e;
  const float er = erf_fast(x * 0.70710678118654752f, &e);
  return 0.5f * (1.0f + er) + x * 0.39894228040143268f * e;
}
// gelu(g) and gelu'(g) from one erf evaluation (the forward's stash_grad form)
__device__ __forceinline__ void gelu_val_grad(float g, float& val, float& grad) {
  float e;
  const float phi = 0.5f * (1.0f + erf_fast(g * 0.70710678118654752f, &e));
  val = g * phi;
  grad = phi + g * 0.39894228040143268f * e;
}
// backward of y = h * gelu(gate) for one pair, one erf for both factors: dh = d * gelu(gate), dgate = d * h * gelu'(gate)
__device__ __forceinline__ void geglu_pair_bwd(float h, float g, float d, float& dh, float& dg) {
  float e;
  const float phi = 0.5f * (1.0f + erf_fast(g * 0.70710678118654752f, &e));
  dh = d * (g * phi);
  dg = d * h * (phi + g * 0.39894228040143268f * e);
}
// v_rcp_f32 (1 ulp) instead of a correctly rounded division (~10 VALU instructions); results are stored as bf16
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// Butterfly reductions over the 64 lanes, without LDS.  The four steps inside a 16-lane row are DPP moves (one VALU
// instruction each: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after each step every lane of
// the group holds the group's value, so the mirrors act as the xor-4 / xor-8 exchanges); the two cross-row steps are
// v_permlane16_swap (odd rows of one copy <-> even rows of the other: the sum of the two copies is the xor-16 step) and
// v_permlane32_swap (upper half <-> lower half: xor-32).  As ds_bpermute round trips those two steps were ~250 cycles of
// every reduction -- eight of them in a row on the critical path of a LayerNorm wave.  (Inline assembly: with fp32 operands
// bit-cast in and out of the builtins hipcc 7.2 folds the second result onto the first; the s_nops are the VALU hazards.)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_halves32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  float a = v, b = v;
  swap_rows16(a, b);
  v = a + b;
  a = v; b = v;
  swap_halves32(a, b);
  return a + b;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  float a = v, b = v;
  swap_rows16(a, b);
  v = fmaxf(a, b);
  a = v; b = v;
  swap_halves32(a, b);
  return fmaxf(a, b);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
