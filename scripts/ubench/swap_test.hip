#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, const float* in) {
  const int lane = threadIdx.x;
  float a = in[lane] * 2.f + 1.f, b = in[64 + lane] * 2.f + 1.f;      // VALU-produced operands
  const auto w = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  out[lane] = __builtin_bit_cast(float, w[0]);
  out[64 + lane] = __builtin_bit_cast(float, w[1]);
  // the loop form used in the epilogue
  float lo[4], hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto v = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a + j), __builtin_bit_cast(unsigned, b + j), false, false);
    lo[j] = __builtin_bit_cast(float, v[0]);
    hi[j] = __builtin_bit_cast(float, v[1]);
  }
  out[128 + lane] = lo[0] + lo[1] + lo[2] + lo[3];
  out[192 + lane] = hi[0] + hi[1] + hi[2] + hi[3];
}
int main() {
  float h[128], o[256], *di, *dd;
  for (int i = 0; i < 64; ++i) { h[i] = i; h[64 + i] = 100 + i; }
  hipMalloc(&di, sizeof(h)); hipMalloc(&dd, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dd, di);
  hipMemcpy(o, dd, sizeof(o), hipMemcpyDeviceToHost);
  // a = 2*lane+1, b = 2*(100+lane)+1
  for (int r = 0; r < 4; ++r) printf("lane row %d: w0=%g w1=%g | lo-sum=%g hi-sum=%g\n", r, o[16 * r], o[64 + 16 * r], o[128 + 16 * r], o[192 + 16 * r]);
  return 0;
}
