// Micro-benchmark: do vector instructions placed between a wave's own MFMAs overlap with them, and what do two waves
// of one SIMD do to each other?  Each wave runs ITER trips of 8 x { v_mfma_f32_32x32x16_bf16 ; K vector instructions }
// (two accumulator chains, operands in registers, vector work independent of the MFMAs).  Variants: K = 0 (MFMA only),
// 2 v_exp_f32, 2 v_exp_f32 + 2 v_add_f32, 4 v_fma_f32, 6 v_fma_f32; and the same vector work WITHOUT the MFMAs.
// One workgroup per CU of 256 threads (one wave per SIMD) or 512 (two per SIMD).  Output: cycles per trip per wave
// (s_memtime), i.e. per 8 MFMAs = 256 matrix-pipe cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VARIANT, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = -0.001f * (lane + j);
  float sum = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (WITH_MFMA) {
        if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      }
      if (VARIANT == 1) {
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[m & 7]), "+v"(v[(m + 3) & 7]));
      } else if (VARIANT == 2) {
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %5"
                     : "+v"(v[m & 7]), "+v"(v[(m + 3) & 7]), "+v"(v[(m + 5) & 7]) : "v"(v[(m + 1) & 7]), "v"(sum), "v"(v[(m + 2) & 7]));
      } else if (VARIANT == 3) {
        asm volatile("v_fma_f32 %0, %0, %4, %0\n\tv_fma_f32 %1, %1, %4, %1\n\tv_fma_f32 %2, %2, %4, %2\n\tv_fma_f32 %3, %3, %4, %3"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(v[4]));
      } else if (VARIANT == 4) {
        asm volatile("v_fma_f32 %0, %0, %6, %0\n\tv_fma_f32 %1, %1, %6, %1\n\tv_fma_f32 %2, %2, %6, %2\n\tv_fma_f32 %3, %3, %6, %3\n\t"
                     "v_fma_f32 %4, %4, %6, %4\n\tv_fma_f32 %5, %5, %6, %5"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "v"(v[6]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int j = 0; j < 8; ++j) sum += v[j];
  for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
  if (sum == 123.456f) out[0] = sum;
  if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int VARIANT, bool WITH_MFMA>
int run(const char* name, int threads, float* out, unsigned long long* dcyc) {
  const int iters = 2000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<VARIANT, WITH_MFMA>), dim3(256), dim3(threads), 0, 0, out, dcyc, iters);
  CHK(hipDeviceSynchronize());
  std::vector<unsigned long long> c(256 * 8);
  CHK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> per;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < threads / 64; ++w) per.push_back((double)c[b * 8 + w] / iters);
  std::sort(per.begin(), per.end());
  printf("%-34s %s waves/SIMD %d: %7.1f cycles per trip of 8 gaps (median), %5.1f per gap\n", name, WITH_MFMA ? "with MFMA" : "no MFMA  ", threads / 256,
         per[per.size() / 2], per[per.size() / 2] / 8);
  return 0;
}

int main() {
  float* out;
  unsigned long long* dcyc;
  CHK(hipMalloc(&out, 64));
  CHK(hipMalloc(&dcyc, 256 * 8 * 8));
  for (int threads : {256, 512}) {
    run<0, true>("MFMA only", threads, out, dcyc);
    run<1, true>("2 v_exp per gap", threads, out, dcyc);
    run<1, false>("2 v_exp per gap", threads, out, dcyc);
    run<2, true>("2 v_exp + 2 v_add per gap", threads, out, dcyc);
    run<2, false>("2 v_exp + 2 v_add per gap", threads, out, dcyc);
    run<3, true>("4 v_fma per gap", threads, out, dcyc);
    run<3, false>("4 v_fma per gap", threads, out, dcyc);
    run<4, true>("6 v_fma per gap", threads, out, dcyc);
    run<4, false>("6 v_fma per gap", threads, out, dcyc);
  }
  return 0;
}
