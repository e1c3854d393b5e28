// Data-parallel collective of the KD step: ONE RCCL all-reduce of the flat fp32 adapter
// gradient per step on a dedicated HIP stream, followed by the 1/world scaling on that same stream.  Replaces
// DeepSpeed ZeRO-1's gradient all-reduce + parameter all-gather (train_sdxl_zh.sh:22,87; utils/model_utils.py:57-67):
// the optimizer is replicated, so no parameter traffic exists.  The step's compute stream never blocks on the
// collective: pea_allreduce_grads() makes the comm stream wait for what the compute stream has enqueued so far and
// returns; pea_comm_join() makes a stream wait for the result (called right before the optimizer), so the next batch's
// VAE encode (side stream) overlaps the xGMI transfer.
//
// RCCL is bound at run time (dlopen): the library a process already carries (torch ships librccl.so) is reused so that
// one process never holds two copies; libpea_hip.so itself loads without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/pea_hip.h"
#include "pea_common.h"

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.h) return PEA_OK;
  void* h = nullptr;
  // a copy already mapped into this process first (RTLD_NOLOAD), then the search path, then the ROCm install
  for (const char* nm : {"librccl.so", "librccl.so.1"}) {
    h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  if (!h)
    for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
  if (!h) {
    pea_set_error("pea_comm: cannot load librccl.so (%s)", dlerror());
    return PEA_E_STATE;
  }
  Rccl r;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GetErrorString) {
    pea_set_error("pea_comm: librccl.so lacks a required symbol");
    return PEA_E_STATE;
  }
  g_rccl = r;
  return PEA_OK;
}

#define NCCLCHK(x)                                                                                      \
  do {                                                                                                  \
    ncclResult_t r__ = (x);                                                                             \
    if (r__ != ncclSuccess) {                                                                           \
      pea_set_error("%s:%d rccl error %d (%s) in %s", __FILE__, __LINE__, (int)r__,                    \
                    g_rccl.GetErrorString(r__), #x);                                                   \
      return PEA_E_HIP;                                                                                 \
    }                                                                                                   \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;                    // the dedicated comm stream
  hipEvent_t ev_ready = nullptr;                   // compute stream -> comm stream (gradients complete)
  hipEvent_t ev_t0 = nullptr, ev_done = nullptr;   // around the collective (timing enabled) / result ready
  hipEvent_t ev_j0 = nullptr, ev_j1 = nullptr;     // on the joining stream, either side of its wait (exposed time)
  bool pending = false, joined = false;
};

__global__ void scale_kernel(float* __restrict__ p, long long n, float f) {
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    f32x4 v = *(f32x4*)(p + i4);
    v *= f;
    *(f32x4*)(p + i4) = v;
  } else {
    for (long long i = i4; i < n; ++i) p[i] *= f;
  }
}
}  // namespace

extern "C" {

int pea_comm_unique_id(void* out128) {
  if (!out128) { pea_set_error("pea_comm_unique_id: null output"); return PEA_E_INVALID; }
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return PEA_OK;
}

// Bounded rendezvous.  ncclCommInitRank blocks until all `world` ranks have arrived; if one never does (a rank that died
// in its own start-up, a wrong WORLD_SIZE) it blocks forever and RCCL offers no cancel for a BLOCKING communicator.  The
// reference gets its deadline from torch.distributed.run's rendezvous (train_sdxl_zh.sh:108-113).  Here the call runs on a
// helper thread bound to the caller's device and the caller waits for it with a deadline: past it the function returns
// PEA_E_TIMEOUT and the helper is left behind, detached, still inside RCCL's bootstrap -- the process is expected to exit
// with a non-zero status (bench.py does); the state the helper touches is kept alive by a shared_ptr, so a late return of
// ncclCommInitRank releases the communicator it then holds and nothing else.
namespace {
struct InitJob {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false, abandoned = false;
  ncclResult_t res = ncclSuccess;
  ncclComm_t comm = nullptr;
  hipError_t dev_err = hipSuccess;
};
double comm_default_timeout_s() {
  const char* e = getenv("PEA_COMM_TIMEOUT_S");      // <= 0: wait forever (the pre-round-6 behaviour)
  if (e && *e) return atof(e);
  return 600.0;
}
}  // namespace

int pea_comm_init_timeout(int rank, int world, const void* unique_id128, double timeout_s, void** out) {
  if (!out || !unique_id128) { pea_set_error("pea_comm_init: null argument"); return PEA_E_INVALID; }
  SHAPECHK(world >= 1 && rank >= 0 && rank < world, "pea_comm_init: rank %d of %d", rank, world);
  int rc = load_rccl();
  if (rc) return rc;
  Comm* c = new Comm();
  c->rank = rank; c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    pea_set_error("pea_comm_init: no HIP device");
    return PEA_E_HIP;
  }
  ncclUniqueId id;
  memcpy(id.internal, unique_id128, NCCL_UNIQUE_ID_BYTES);
  auto job = std::make_shared<InitJob>();
  const int device = c->device;
  const Rccl api = g_rccl;
  std::thread([job, api, id, world, rank, device]() {
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclSuccess;
    const hipError_t de = hipSetDevice(device);                        // the helper thread starts on device 0
    if (de == hipSuccess) r = api.CommInitRank(&comm, world, id, rank);  // blocking rendezvous of all ranks
    std::unique_lock<std::mutex> lk(job->mu);
    job->dev_err = de;
    job->res = r;
    job->comm = comm;
    job->done = true;
    if (job->abandoned && comm) (void)api.CommDestroy(comm);           // nobody is waiting any more
    lk.unlock();
    job->cv.notify_all();
  }).detach();
  {
    std::unique_lock<std::mutex> lk(job->mu);
    if (timeout_s > 0) {
      const auto dl = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                             std::chrono::duration<double>(timeout_s));
      if (!job->cv.wait_until(lk, dl, [&] { return job->done; })) {
        job->abandoned = true;
        lk.unlock();
        delete c;
        pea_set_error("pea_comm_init: rendezvous of %d ranks not complete after %.1f s (rank %d waited in ncclCommInitRank; "
                      "a rank is missing or never reached it) -- exit this process", world, timeout_s, rank);
        return PEA_E_TIMEOUT;
      }
    } else {
      job->cv.wait(lk, [&] { return job->done; });
    }
  }
  if (job->dev_err != hipSuccess) {
    pea_set_error("pea_comm_init: hipSetDevice(%d) failed on the rendezvous thread: %s", device, hipGetErrorString(job->dev_err));
    delete c;
    return PEA_E_HIP;
  }
  if (job->res != ncclSuccess) {
    pea_set_error("pea_comm_init: ncclCommInitRank failed: %s", g_rccl.GetErrorString(job->res));
    delete c;
    return PEA_E_HIP;
  }
  c->comm = job->comm;
  // any failure past the rendezvous goes through pea_comm_destroy (communicator, stream and events released)
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_t0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_done);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_j0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_j1);
  if (e != hipSuccess) {
    pea_set_error("pea_comm_init: stream / event creation failed: %s", hipGetErrorString(e));
    (void)pea_comm_destroy(c);
    return PEA_E_HIP;
  }
  *out = c;
  return PEA_OK;
}

int pea_comm_init(int rank, int world, const void* unique_id128, void** out) {
  return pea_comm_init_timeout(rank, world, unique_id128, comm_default_timeout_s(), out);
}

int pea_comm_destroy(void* h) {
  if (!h) return PEA_OK;
  Comm* c = (Comm*)h;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_j0) (void)hipEventDestroy(c->ev_j0);
  if (c->ev_j1) (void)hipEventDestroy(c->ev_j1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return PEA_OK;
}

int pea_comm_world(void* h) { return h ? ((Comm*)h)->world : 1; }
int pea_comm_rank(void* h) { return h ? ((Comm*)h)->rank : 0; }

int pea_allreduce_grads(void* h, float* grads, long long n, void* compute_stream) {
  if (!h || !grads) { pea_set_error("pea_allreduce_grads: null argument"); return PEA_E_INVALID; }
  Comm* c = (Comm*)h;
  SHAPECHK(n > 0, "pea_allreduce_grads: n=%lld", n);
  SHAPECHK(((uintptr_t)grads & 15) == 0, "pea_allreduce_grads: grads must be 16-byte aligned (got %p)", (void*)grads);
  HIPCHK(hipEventRecord(c->ev_ready, (hipStream_t)compute_stream));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
  HIPCHK(hipEventRecord(c->ev_t0, c->stream));
  NCCLCHK(g_rccl.AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, c->comm, c->stream));
  if (c->world > 1) {
    const int thr = 256;
    const long long blocks = (n / 4 + thr) / thr;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(thr), 0, c->stream, grads, n, 1.0f / (float)c->world);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(c->ev_done, c->stream));
  c->pending = true;
  c->joined = false;
  return PEA_OK;
}

int pea_comm_join(void* h, void* stream) {
  if (!h) { pea_set_error("pea_comm_join: null communicator"); return PEA_E_INVALID; }
  Comm* c = (Comm*)h;
  if (c->pending && !c->joined) {
    // ev_j0 completes when `stream` has drained what was enqueued before the join, ev_j1 when the wait is satisfied:
    // their distance is the time the joining stream stood still for the collective (pea_comm_last_exposed_ms)
    HIPCHK(hipEventRecord(c->ev_j0, (hipStream_t)stream));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_done, 0));
    HIPCHK(hipEventRecord(c->ev_j1, (hipStream_t)stream));
    c->joined = true;
  } else if (c->pending) {
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_done, 0));
  }
  return PEA_OK;
}

int pea_comm_last_exposed_ms(void* h, float* ms) {
  if (!h || !ms) { pea_set_error("pea_comm_last_exposed_ms: null argument"); return PEA_E_INVALID; }
  Comm* c = (Comm*)h;
  if (!c->pending || !c->joined) { *ms = 0.f; return PEA_OK; }
  HIPCHK(hipEventSynchronize(c->ev_j1));
  HIPCHK(hipEventElapsedTime(ms, c->ev_j0, c->ev_j1));
  return PEA_OK;
}

int pea_comm_last_ms(void* h, float* ms) {
  if (!h || !ms) { pea_set_error("pea_comm_last_ms: null argument"); return PEA_E_INVALID; }
  Comm* c = (Comm*)h;
  if (!c->pending) { *ms = 0.f; return PEA_OK; }
  HIPCHK(hipEventSynchronize(c->ev_done));
  HIPCHK(hipEventElapsedTime(ms, c->ev_t0, c->ev_done));
  return PEA_OK;
}

int pea_comm_broadcast(void* h, float* buf, long long n, int root, void* stream) {
  if (!h || !buf) { pea_set_error("pea_comm_broadcast: null argument"); return PEA_E_INVALID; }
  Comm* c = (Comm*)h;
  SHAPECHK(n > 0 && root >= 0 && root < c->world, "pea_comm_broadcast: n=%lld root=%d", n, root);
  NCCLCHK(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, (hipStream_t)stream));
  return PEA_OK;
}

}  // extern "C"
