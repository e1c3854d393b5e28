// Optional per-launch HIP-event timing by kernel family (used by bench.py's roofline leg).
// Off by default: when off, prof_begin/prof_end are a single predictable branch.
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
