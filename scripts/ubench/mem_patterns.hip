// Micro-benchmark: what a GEMM tile epilogue's global stores / residual loads cost per CU, by access shape.
// One 512-thread workgroup per CU (the consumer waves of gemm_lcp_kernel<256,160>) writes / reads 256x160 bf16 tiles of
// a [M][N = 1280] matrix, tile after tile, with the lane -> address maps under test:
//   0  "quad"      16 rows x 64 B per wave-instruction (16 B per lane: today's paired epilogue store)
//   1  "quad8"     16 rows x 32 B per wave-instruction ( 8 B per lane: today's residual load)
//   2  "row160"    whole 160-B wave-tile rows: 10 lanes x 16 B per row, 6.4 rows per instruction
//   3  "row320"    whole 320-B tile rows: 20 lanes x 16 B per row, 3.2 rows per instruction (an LDS-transposed epilogue)
//   4  "line128"   8 rows x 128 B per instruction
//   5  "contig"    1 KiB contiguous per instruction (upper bound; not a tile shape)
// Output: cycles per tile per CU and bytes per clock per CU (s_memtime around the tile loop, median over workgroups),
// wall time, for stores and for loads.  Build: hipcc --offload-arch=gfx950 -O3 -o mem_patterns mem_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int BM = 256, BN = 160, N = 1280;

// byte offset inside the matrix for (tile-local instruction i, lane) under pattern P; returns bytes per lane
template <int P>
__device__ __forceinline__ long long addr(int wave, int i, int lane, int& ok) {
  ok = 1;
  const int wr = wave >> 1, wc = wave & 1;                 // 4 x 2 waves, 64 x 80 each
  if (P == 0) {                                            // 12 instr per wave: mt 0..3 x {pair0, pair1, (single 8B as 16B half)}
    const int mt = i / 3, pr = i % 3;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int row = wr * 64 + mt * 16 + r16;
    if (pr == 2) { ok = q4 < 2; return (long long)row * N * 2 + (wc * 80 + 64 + (q4 & 1) * 8) * 2; }
    return (long long)row * N * 2 + (wc * 80 + pr * 32 + q4 * 8) * 2;
  }
  if (P == 1) {                                            // 20 instr per wave: mt x nt, 8 B per lane
    const int mt = i / 5, nt = i % 5;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int row = wr * 64 + mt * 16 + r16;
    return (long long)row * N * 2 + (wc * 80 + nt * 16 + q4 * 4) * 2;
  }
  if (P == 2) {                                            // wave tile 64 x 160 B = 640 chunks of 16 B: 10 instr
    const int c = i * 64 + lane;
    const int row = wr * 64 + c / 10, ch = c % 10;
    return (long long)row * N * 2 + wc * 160 + ch * 16;
  }
  if (P == 3) {                                            // tile rows of 320 B: the wave's 32 rows x 20 chunks = 640 chunks
    const int c = i * 64 + lane;
    const int row = wave * 32 + c / 20, ch = c % 20;
    return (long long)row * N * 2 + ch * 16;
  }
  if (P == 4) {                                            // 8 rows x 128 B: the wave's 32 rows x 320 B = 2.5 lines per row ->
    const int c = i * 64 + lane;                           // treat as 80 line-pieces of 128 B (row, piece 0..2; piece 2 half)
    const int row = wave * 32 + (c >> 3) / 3, piece = (c >> 3) % 3, ch = c & 7;
    ok = !(piece == 2 && ch >= 4);
    return (long long)row * N * 2 + piece * 128 + ch * 16;
  }
  const int c = i * 64 + lane;                             // P == 5: contiguous
  return (long long)wave * 10240 + (long long)c * 16;
}
template <int P> constexpr int n_instr() { return P == 0 ? 12 : P == 1 ? 20 : P == 4 ? 12 : 10; }

template <int P, bool STORE>
__global__ __launch_bounds__(512) void k(char* buf, int tiles_per_wg, int nbm, unsigned long long* cyc, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4 v = {threadIdx.x, 1u, 2u, 3u};
  float acc = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int bm = tile % nbm, bn = tile / nbm;
    char* base = buf + ((long long)bm * BM * N + bn * BN) * 2;
    if (P == 5) base = buf + (long long)tile * 81920;
#pragma unroll
    for (int i = 0; i < n_instr<P>(); ++i) {
      int ok;
      const long long a = addr<P>(wave, i, lane, ok);
      if (STORE) {
        if (P == 1) { if (ok) *(u32x2*)(base + a) = (u32x2){v[0], v[1]}; }
        else { if (ok) *(u32x4*)(base + a) = v; }
      } else {
        if (P == 1) { if (ok) { u32x2 r = *(const u32x2*)(base + a); acc += (float)r[0] + (float)r[1]; } }
        else { if (ok) { u32x4 r = *(const u32x4*)(base + a); acc += (float)r[0] + (float)r[3]; } }
      }
    }
    v[1] += 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 12345.f) sink[0] = acc;
}

template <int P, bool STORE>
void run(const char* name, char* buf, int M, unsigned long long* dcyc, float* sink, int tiles_per_wg) {
  const int nbm = M / BM;
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<P, STORE>), dim3(256), dim3(512), 0, 0, buf, tiles_per_wg, nbm, dcyc, sink);
  CHK(hipEventRecord(a));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<P, STORE>), dim3(256), dim3(512), 0, 0, buf, tiles_per_wg, nbm, dcyc, sink);
  CHK(hipEventRecord(b));
  CHK(hipDeviceSynchronize());
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> c(256);
  CHK(hipMemcpy(c.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost));
  std::sort(c.begin(), c.end());
  const double per_tile = (double)c[128] / tiles_per_wg;
  const double bytes = 81920.0;
  printf("%-6s %-8s tiles/wg %3d: %8.0f clk/tile/CU  %6.1f B/clk/CU  %7.2f us/launch  %6.2f TB/s chip\n", STORE ? "store" : "load", name,
         tiles_per_wg, per_tile, bytes / per_tile, ms * 1e3 / reps, 256.0 * tiles_per_wg * bytes / (ms * 1e-3 / reps) / 1e12);
}

int main() {
  // [M][1280] bf16 with M = 8192 * 8: 8 tiles per workgroup column-walk; 168 MB (beyond L2, inside the Infinity Cache)
  const int M = 65536;
  char* buf;
  CHK(hipMalloc(&buf, (size_t)M * N * 2 + (1 << 20)));
  CHK(hipMemset(buf, 1, (size_t)M * N * 2));
  unsigned long long* dcyc;
  float* sink;
  CHK(hipMalloc(&dcyc, 256 * 8));
  CHK(hipMalloc(&sink, 64));
  for (int tp : {1, 8}) {
    run<0, true>("quad", buf, M, dcyc, sink, tp);
    run<1, true>("quad8", buf, M, dcyc, sink, tp);
    run<2, true>("row160", buf, M, dcyc, sink, tp);
    run<3, true>("row320", buf, M, dcyc, sink, tp);
    run<4, true>("line128", buf, M, dcyc, sink, tp);
    run<5, true>("contig", buf, M, dcyc, sink, tp);
    run<0, false>("quad", buf, M, dcyc, sink, tp);
    run<1, false>("quad8", buf, M, dcyc, sink, tp);
    run<2, false>("row160", buf, M, dcyc, sink, tp);
    run<3, false>("row320", buf, M, dcyc, sink, tp);
    run<4, false>("line128", buf, M, dcyc, sink, tp);
    run<5, false>("contig", buf, M, dcyc, sink, tp);
  }
  return 0;
}
