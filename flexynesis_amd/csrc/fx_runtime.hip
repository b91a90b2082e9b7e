// Host-side runtime pieces of libfxhip (no kernels): memory LEASES and the library's own hipGraph capture / replay.
//
// Leases.  The engine sub-allocates the three arrays of every wide weight from long-lived pools laid out over the GPU's memory
// partitions (flexynesis_amd.engine.PartitionArena).  A range of such a pool may only be handed to the next
// model when NOTHING refers to it any more -- not the ParamStore that took it, not an nn.Parameter's .data, not a state_dict a
// caller still holds.  The host therefore wraps every range in a DLPack managed tensor whose deleter is fx's: the framework that
// imports it (torch.from_dlpack on the Python host) calls the deleter when the last view of that storage has gone, from whatever
// thread that happens on, possibly while the interpreter is shutting down -- so the deleter is plain C: it pushes the lease id on
// a queue the host drains (fx_lease_drain) before its next allocation.  The queue is host-owned state behind an opaque handle;
// the library keeps no globals for it.
//
// Graphs.  fx_graph_begin / fx_graph_end capture whatever is launched on a stream (and on streams forked from it by event
// waits) between the two calls into a hipGraphExec; fx_graph_launch replays it.  Capture mode RELAXED: the engine owns every
// buffer a captured launch touches (nothing is allocated inside a capture), and other host threads (trials in flight, the
// garbage collector) keep calling the runtime meanwhile.  This replaces torch.cuda.CUDAGraph on the engine's path
// (ops.graph_capture, FX_GRAPH_BACKEND): no private allocator pool, no generator registration, no device synchronisation in the
// destructor -- the pieces of torch's graph layer that the replay crashes of rounds 2-5 were traced to.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "fx_common.h"

namespace {

// DLPack (dmlc/dlpack, include/dlpack/dlpack.h, ABI of the unversioned "dltensor" capsule), restated: plain C structs.
struct FxDLDevice { int32_t device_type; int32_t device_id; };
struct FxDLDataType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct FxDLTensor {
  void* data;
  FxDLDevice device;
  int32_t ndim;
  FxDLDataType dtype;
  int64_t* shape;
  int64_t* strides;
  uint64_t byte_offset;
};
struct FxDLManagedTensor {
  FxDLTensor dl_tensor;
  void* manager_ctx;
  void (*deleter)(FxDLManagedTensor*);
};

struct LeaseQueue {
  std::mutex mu;
  std::vector<long long> ids;
  long long wrapped = 0, released = 0;
};

struct Lease {
  FxDLManagedTensor m;      // first member: the managed tensor's address is the lease's
  int64_t shape[1];
  LeaseQueue* q;
  long long id;
};

void lease_deleter(FxDLManagedTensor* m) {
  Lease* l = reinterpret_cast<Lease*>(m);
  {
    std::lock_guard<std::mutex> g(l->q->mu);
    l->q->ids.push_back(l->id);
    l->q->released++;
  }
  free(l);
}

}  // namespace

extern "C" {

void* fx_lease_queue_create(void) { return new LeaseQueue(); }

// (never destroyed while leases are outstanding: the host keeps one queue per process for life)
void fx_lease_queue_destroy(void* q) { delete static_cast<LeaseQueue*>(q); }

// A DLManagedTensor* describing n fp32 elements at ptr on (device_type, device_id) -- DLPack device types: 1 = host memory
// (tests), 10 = ROCm.  Whoever consumes it (PyCapsule "dltensor" -> torch.from_dlpack) owns it and calls its deleter once.
void* fx_lease_wrap(void* queue, void* ptr, long long n, int device_type, int device_id, long long id) {
  if (queue == nullptr || ptr == nullptr || n <= 0) {
    fx_set_error("fx_lease_wrap: queue %p, ptr %p, n %lld", queue, ptr, n);
    return nullptr;
  }
  Lease* l = static_cast<Lease*>(calloc(1, sizeof(Lease)));
  if (l == nullptr) {
    fx_set_error("fx_lease_wrap: out of host memory");
    return nullptr;
  }
  l->shape[0] = n;
  l->q = static_cast<LeaseQueue*>(queue);
  l->id = id;
  l->m.dl_tensor.data = ptr;
  l->m.dl_tensor.device = FxDLDevice{device_type, device_id};
  l->m.dl_tensor.ndim = 1;
  l->m.dl_tensor.dtype = FxDLDataType{2 /* kDLFloat */, 32, 1};
  l->m.dl_tensor.shape = l->shape;
  l->m.dl_tensor.strides = nullptr;
  l->m.dl_tensor.byte_offset = 0;
  l->m.manager_ctx = l;
  l->m.deleter = lease_deleter;
  {
    std::lock_guard<std::mutex> g(l->q->mu);
    l->q->wrapped++;
  }
  return &l->m;
}

// for a managed tensor that was never consumed (an error between fx_lease_wrap and the import): runs its deleter
void fx_lease_discard(void* managed) {
  if (managed != nullptr) {
    FxDLManagedTensor* m = static_cast<FxDLManagedTensor*>(managed);
    m->deleter(m);
  }
}

// up to max_ids ids of leases whose storage has been released since the last call; returns how many were written
int fx_lease_drain(void* queue, long long* ids, int max_ids) {
  LeaseQueue* q = static_cast<LeaseQueue*>(queue);
  if (q == nullptr || ids == nullptr || max_ids <= 0) return 0;
  std::lock_guard<std::mutex> g(q->mu);
  int n = 0;
  while (n < max_ids && !q->ids.empty()) {
    ids[n++] = q->ids.back();
    q->ids.pop_back();
  }
  return n;
}

long long fx_lease_outstanding(void* queue) {
  LeaseQueue* q = static_cast<LeaseQueue*>(queue);
  if (q == nullptr) return 0;
  std::lock_guard<std::mutex> g(q->mu);
  return q->wrapped - q->released;
}

// ---- hipGraph capture / replay -------------------------------------------------------------------------------------------------
static int graph_fail(const char* what, hipError_t e) {
  fx_set_error("%s: %s", what, hipGetErrorString(e));
  return -(int)e;
}

// mode: 0 global, 1 thread-local, 2 relaxed (what the engine uses)
int fx_graph_begin(hipStream_t stream, int mode) {
  hipStreamCaptureMode m = mode == 0 ? hipStreamCaptureModeGlobal : mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
  hipError_t e = hipStreamBeginCapture(stream, m);
  if (e != hipSuccess) return graph_fail("fx_graph_begin: hipStreamBeginCapture", e);
  return 0;
}

// ends the capture that fx_graph_begin started on this stream; *exec_out = the instantiated graph (NULL on failure), *n_nodes its
// number of nodes (kernel launches + event / dependency nodes; optional)
int fx_graph_end(hipStream_t stream, void** exec_out, int* n_nodes) {
  if (exec_out == nullptr) {
    fx_set_error("fx_graph_end: exec_out is NULL");
    return FX_EINVAL;
  }
  *exec_out = nullptr;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(stream, &g);
  if (e != hipSuccess || g == nullptr) {
    (void)hipGetLastError();
    return graph_fail("fx_graph_end: hipStreamEndCapture", e == hipSuccess ? hipErrorUnknown : e);
  }
  if (n_nodes != nullptr) {
    size_t n = 0;
    *n_nodes = hipGraphGetNodes(g, nullptr, &n) == hipSuccess ? (int)n : -1;
  }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return graph_fail("fx_graph_end: hipGraphInstantiate", e);
  *exec_out = x;
  return 0;
}

// abandon a capture after an error inside it (the stream leaves capture mode; whatever was captured is dropped)
int fx_graph_abort(hipStream_t stream) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(stream, &g);
  if (g != nullptr) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int fx_graph_launch(void* exec, hipStream_t stream) {
  if (exec == nullptr) {
    fx_set_error("fx_graph_launch: exec is NULL");
    return FX_EINVAL;
  }
  hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, stream);
  if (e != hipSuccess) return graph_fail("fx_graph_launch: hipGraphLaunch", e);
  return 0;
}

int fx_graph_destroy(void* exec) {
  if (exec == nullptr) return 0;
  hipError_t e = hipGraphExecDestroy((hipGraphExec_t)exec);
  if (e != hipSuccess) return graph_fail("fx_graph_destroy: hipGraphExecDestroy", e);
  return 0;
}

// 1 while the stream is capturing, 0 otherwise, < 0 on error
int fx_graph_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(stream, &st);
  if (e != hipSuccess) return graph_fail("fx_graph_capturing: hipStreamIsCapturing", e);
  return st == hipStreamCaptureStatusActive ? 1 : 0;
}

}  // extern "C"
