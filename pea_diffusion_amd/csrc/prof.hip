// Optional per-launch HIP-event timing by kernel family (used by bench.py's roofline leg).
// Off by default: when off, prof_begin/prof_end are a single predictable branch.
#include <algorithm>
#include <vector>

#include "../../include/pea_hip.h"
#include "pea_kernels.h"

struct ProfRec { hipEvent_t a, b; int fam; double flops, bytes; int tag[4]; };
int g_prof_tag[4] = {0, 0, 0, 0};
int g_prof_on = 0;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;
static const char* kFam[PEA_PROF_FAMILIES] = {"gemm_lc[p]_kernel<plain>", "gemm_lc[p]_kernel<conv3x3>", "attn_fwd",
                                              "attn_bwd", "groupnorm", "layernorm", "elementwise", "kd_loss"};

static hipEvent_t take_event() {
  if (g_pool_next == g_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    g_pool.push_back(e);
  }
  return g_pool[g_pool_next++];
}
void prof_begin_impl(int fam, double flops, double bytes, hipStream_t s) {
  ProfRec r;
  r.a = take_event(); r.b = take_event(); r.fam = fam; r.flops = flops; r.bytes = bytes;
  for (int i = 0; i < 4; ++i) { r.tag[i] = g_prof_tag[i]; g_prof_tag[i] = 0; }
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void prof_end_impl(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }

extern "C" {
void pea_prof_enable(int on) { g_prof_on = on; }
void pea_prof_reset(void) {
  g_recs.clear();
  g_pool_next = 0;
}
const char* pea_prof_family_name(int fam) { return fam >= 0 && fam < PEA_PROF_FAMILIES ? kFam[fam] : "?"; }
int pea_prof_dump(const char* path) {
  if (hipDeviceSynchronize() != hipSuccess) return PEA_E_HIP;
  FILE* f = fopen(path, "w");
  if (!f) return PEA_E_INVALID;
  fprintf(f, "family,ms,flops,bytes,t0,t1,t2,t3\n");
  for (const ProfRec& r : g_recs) {
    float e = 0.f;
    (void)hipEventElapsedTime(&e, r.a, r.b);
    fprintf(f, "%s,%.6f,%.0f,%.0f,%d,%d,%d,%d\n", kFam[r.fam], e, r.flops, r.bytes, r.tag[0], r.tag[1], r.tag[2], r.tag[3]);
  }
  fclose(f);
  return PEA_OK;
}
int pea_prof_report(int fam, double* ms, double* flops, double* bytes, long long* launches) {
  if (hipDeviceSynchronize() != hipSuccess) return PEA_E_HIP;
  double t = 0, f = 0, by = 0;
  long long n = 0;
  for (const ProfRec& r : g_recs) {
    if (r.fam != fam) continue;
    float e = 0.f;
    if (hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) return PEA_E_HIP;
    t += e; f += r.flops; by += r.bytes; ++n;
  }
  if (ms) *ms = t;
  if (flops) *flops = f;
  if (bytes) *bytes = by;
  if (launches) *launches = n;
  return PEA_OK;
}
}

// ------------------------------------------------------------------------------------------------
// Sustained MFMA ceiling of THIS device (bench.py: roofline.sustained_peak).  The 2.5 PFLOP/s dense bf16 figure is the data
// sheet's at 2.4 GHz; under a matrix-dense load on random operands the chip holds a lower clock (MI355X_MICROARCH.md, DVFS
// give-back), so the fraction of the spec peak mixes "how well the kernel feeds the pipes" with "what clock this box holds".
// The probe is the GEMM family's instruction (v_mfma_f32_16x16x32_bf16) issued back to back from registers -- 16 independent
// accumulators, 4 x 4 fragments of random bf16, two waves per SIMD on every CU, no memory traffic inside the loop -- run
// back to back for `seconds`; it reports the FLOP/s of the LAST launch from HIP events and the in-kernel clock of that launch,
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz around the loop, median over the workgroups (stamps go to a buffer of their
// own that nothing reads on the device).  SHAPE32 = 1: the same with v_mfma_f32_32x32x16_bf16 (the attention kernels' shape).
template <int SHAPE32>
__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(int iters, unsigned seed, float* sink, unsigned long long* stamps) {
  const int lane = threadIdx.x & 63;
  unsigned h = seed ^ (blockIdx.x * 512u + threadIdx.x) * 2654435761u;
  auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f; };
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[i][j] = (bf16)rnd(); b[i][j] = (bf16)rnd(); }
  unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
  if constexpr (SHAPE32) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + rep) & 3], acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;                       // keeps the accumulators alive
  } else {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
  }
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  (void)lane;
}

extern "C" int pea_probe_mfma_peak(double seconds, int shape32, double* tflops, double* clock_mhz, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (cus <= 0) { pea_set_error("pea_probe_mfma_peak: no device"); return PEA_E_HIP; }
  if (!(seconds > 0.0) || seconds > 30.0) seconds = 2.0;
  const int iters = 20000;                                   // 16 MFMAs x 16384 FLOP (8 x 32768) per wave and iteration: ~7-10 ms a launch
  const double flop_per_launch = (double)cus * 8.0 * iters * 16.0 * 16384.0;
  float* sink = nullptr;
  unsigned long long* stamps = nullptr;
  HIPCHK(hipMalloc((void**)&sink, 256));
  if (hipMalloc((void**)&stamps, (size_t)cus * 16) != hipSuccess) { (void)hipFree(sink); pea_set_error("pea_probe_mfma_peak: hipMalloc"); return PEA_E_HIP; }
  hipEvent_t e0, e1, ea, eb;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
  (void)hipEventRecord(e0, s);
  float elapsed = 0.f, last = 0.f;
  int launches = 0;
  do {
    (void)hipEventRecord(ea, s);
    if (shape32) hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(cus), dim3(512), 0, s, iters, 1234u + launches, sink, stamps);
    else hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(cus), dim3(512), 0, s, iters, 1234u + launches, sink, stamps);
    (void)hipEventRecord(eb, s);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) break;
    (void)hipEventElapsedTime(&elapsed, e0, e1);
    (void)hipEventElapsedTime(&last, ea, eb);
    ++launches;
  } while (elapsed < seconds * 1e3f && launches < 100000);
  std::vector<unsigned long long> hst((size_t)cus * 2);
  hipError_t ce = hipMemcpy(hst.data(), stamps, (size_t)cus * 16, hipMemcpyDeviceToHost);
  (void)hipFree(sink); (void)hipFree(stamps);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
  if (ce != hipSuccess || last <= 0.f) { pea_set_error("pea_probe_mfma_peak: probe failed"); return PEA_E_HIP; }
  std::vector<double> mhz;
  for (int i = 0; i < cus; ++i)
    if (hst[2 * i + 1] > 0) mhz.push_back((double)hst[2 * i] / (double)hst[2 * i + 1] * 100.0);
  std::sort(mhz.begin(), mhz.end());
  if (tflops) *tflops = flop_per_launch / (last * 1e-3) / 1e12;
  if (clock_mhz) *clock_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
  return PEA_OK;
}
