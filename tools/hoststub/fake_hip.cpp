// A HOST-ONLY stand-in for the HIP runtime calls libdmmfods_hip.so makes, for the sanitizer build of the library's host code
// (tools/hoststub/build.sh: every translation unit compiled with `-x hip --offload-host-only -fsanitize=address,undefined`).
// Test infrastructure: nothing here computes anything.  "Device memory" is host memory (so hipMemcpy / hipMemset are bounds-checked
// by AddressSanitizer), streams and events are heap objects with a magic word (a double destroy, a destroy of a handle that was
// never created or a use after destroy is an ASan report or a counted violation), kernels are not run: hipLaunchKernel checks the
// launch geometry and counts.  The stream model is the one the teardown contract is written against:
//   * work enqueued on a stream is "pending" until that stream (or the device) is synchronised;
//   * destroying a stream with pending work, or an event whose last record is still pending, is a VIOLATION (the real runtime
//     defers such destroys; the library's contract is not to rely on that);
//   * every create must be matched by a destroy (fakehip_live_objects() at the end of a run).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace {
constexpr unsigned STREAM_MAGIC = 0x5712ea77u, EVENT_MAGIC = 0xe7e27000u, DEAD = 0xdeadbeefu;
struct FakeStream { unsigned magic; int id; long pending; int priority; unsigned flags; bool capturing; };
struct FakeEvent { unsigned magic; int id; FakeStream* recorded_on; long recorded_seq; bool timing; };
std::mutex g_mu;
std::set<FakeStream*> g_streams;
std::set<FakeEvent*> g_events;
std::set<void*> g_async_allocs;
FakeStream g_null_stream{STREAM_MAGIC, 0, 0, 0, 0, false};
long g_launches = 0, g_violations = 0, g_stream_creates = 0, g_event_creates = 0;
int g_next_id = 1;
std::vector<std::string> g_violation_log;
bool g_trace = getenv("FAKEHIP_TRACE") != nullptr;

void violation(const std::string& what) {
  ++g_violations;
  g_violation_log.push_back(what);
  fprintf(stderr, "[fakehip] VIOLATION: %s\n", what.c_str());
}
FakeStream* S(hipStream_t st) {
  if (st == nullptr) return &g_null_stream;
  FakeStream* s = reinterpret_cast<FakeStream*>(st);
  if (!g_streams.count(s)) { violation("use of a stream handle that is not alive"); return nullptr; }
  if (s->magic != STREAM_MAGIC) { violation("stream handle with a bad magic word"); return nullptr; }
  return s;
}
FakeEvent* E(hipEvent_t ev) {
  FakeEvent* e = reinterpret_cast<FakeEvent*>(ev);
  if (!g_events.count(e)) { violation("use of an event handle that is not alive"); return nullptr; }
  if (e->magic != EVENT_MAGIC) { violation("event handle with a bad magic word"); return nullptr; }
  return e;
}
}  // namespace

// ---- introspection for the driver (tools/hoststub/drive.cpp) ----
extern "C" long fakehip_launches() { return g_launches; }
extern "C" long fakehip_violations() { return g_violations; }
extern "C" long fakehip_live_objects() { std::lock_guard<std::mutex> l(g_mu); return (long)(g_streams.size() + g_events.size() + g_async_allocs.size()); }
extern "C" long fakehip_live_streams() { std::lock_guard<std::mutex> l(g_mu); return (long)g_streams.size(); }
extern "C" long fakehip_live_events() { std::lock_guard<std::mutex> l(g_mu); return (long)g_events.size(); }
extern "C" long fakehip_stream_creates() { return g_stream_creates; }
extern "C" long fakehip_event_creates() { return g_event_creates; }

extern "C" {

// ---- registration stubs emitted by the host side of a HIP translation unit ----
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
static thread_local struct { dim3 g, b; size_t sh; hipStream_t st; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { t_cfg = {g, b, sh, st}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) { *g = t_cfg.g; *b = t_cfg.b; *sh = t_cfg.sh; *st = t_cfg.st; return hipSuccess; }

hipError_t hipLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeStream* s = S(st);
  if (!s) return hipErrorInvalidResourceHandle;
  if (func == nullptr) { violation("hipLaunchKernel: null function"); return hipErrorInvalidDeviceFunction; }
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || block.y == 0 || block.z == 0) { violation("hipLaunchKernel: empty grid or block"); return hipErrorInvalidConfiguration; }
  if ((size_t)block.x * block.y * block.z > 1024) { violation("hipLaunchKernel: block larger than 1024 threads"); return hipErrorInvalidConfiguration; }
  if (shmem > 160 * 1024) { violation("hipLaunchKernel: more than 160 KB of dynamic LDS"); return hipErrorInvalidConfiguration; }
  if ((unsigned long long)grid.x * grid.y * grid.z > 0x7fffffffull) { violation("hipLaunchKernel: grid too large"); return hipErrorInvalidConfiguration; }
  (void)args;
  ++g_launches;
  ++s->pending;
  return hipSuccess;
}

hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  p->sharedMemPerBlock = 160 * 1024;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->warpSize = 64;
  p->maxThreadsPerBlock = 1024;
  strcpy(p->gcnArchName, "gfx950");
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -2; return hipSuccess; }

hipError_t hipDeviceSynchronize(void) {
  std::lock_guard<std::mutex> l(g_mu);
  g_null_stream.pending = 0;
  for (FakeStream* s : g_streams) s->pending = 0;
  return hipSuccess;
}

// ---- memory: device memory is host memory ----
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t st) {
  { std::lock_guard<std::mutex> l(g_mu); FakeStream* s = S(st); if (!s) return hipErrorInvalidResourceHandle; ++s->pending; }
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  { std::lock_guard<std::mutex> l(g_mu); FakeStream* fs = S(st); if (!fs) return hipErrorInvalidResourceHandle; ++fs->pending; }
  memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (!S(st)) return hipErrorInvalidResourceHandle;
  *p = malloc(n ? n : 1);
  g_async_allocs.insert(*p);
  return hipSuccess;
}
hipError_t hipFreeAsync(void* p, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (!S(st)) return hipErrorInvalidResourceHandle;
  if (!g_async_allocs.erase(p)) { violation("hipFreeAsync of a pointer hipMallocAsync did not return"); return hipErrorInvalidValue; }
  free(p);
  return hipSuccess;
}

// ---- streams ----
static hipError_t make_stream(hipStream_t* out, unsigned flags, int prio) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeStream* s = new FakeStream{STREAM_MAGIC, g_next_id++, 0, prio, flags, false};
  g_streams.insert(s);
  ++g_stream_creates;
  if (g_trace) fprintf(stderr, "[fakehip] stream %d created\n", s->id);
  *out = reinterpret_cast<hipStream_t>(s);
  return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* out, unsigned flags, int prio) { return make_stream(out, flags, prio); }
hipError_t hipStreamCreateWithFlags(hipStream_t* out, unsigned flags) { return make_stream(out, flags, 0); }
hipError_t hipStreamCreate(hipStream_t* out) { return make_stream(out, 0, 0); }
hipError_t hipStreamSynchronize(hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeStream* s = S(st);
  if (!s) return hipErrorInvalidResourceHandle;
  s->pending = 0;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (st == nullptr) { violation("hipStreamDestroy(null stream)"); return hipErrorInvalidResourceHandle; }
  FakeStream* s = S(st);
  if (!s) return hipErrorInvalidResourceHandle;
  if (s->pending) violation("hipStreamDestroy of stream " + std::to_string(s->id) + " with " + std::to_string(s->pending) + " unsynchronised operations");
  if (s->capturing) violation("hipStreamDestroy of a capturing stream");
  for (FakeEvent* e : g_events)
    if (e->recorded_on == s) e->recorded_on = nullptr;   // (the stream was synchronised or the violation is already counted)
  if (g_trace) fprintf(stderr, "[fakehip] stream %d destroyed\n", s->id);
  g_streams.erase(s);
  s->magic = DEAD;
  delete s;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t ev, unsigned) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeStream* s = S(st);
  FakeEvent* e = E(ev);
  if (!s || !e) return hipErrorInvalidResourceHandle;
  // waiting makes `st` depend on the recorded stream's work: model it as pending work on `st`
  if (e->recorded_on && e->recorded_on->pending) ++s->pending;
  return hipSuccess;
}
// graph capture is not modelled: the library falls back to eager launches when a capture fails
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { if (g) *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { violation("hipGraphExecDestroy: no graph can exist here"); return hipErrorInvalidValue; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }

// ---- events ----
static hipError_t make_event(hipEvent_t* out, bool timing) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeEvent* e = new FakeEvent{EVENT_MAGIC, g_next_id++, nullptr, 0, timing};
  g_events.insert(e);
  ++g_event_creates;
  *out = reinterpret_cast<hipEvent_t>(e);
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* out) { return make_event(out, true); }
hipError_t hipEventCreateWithFlags(hipEvent_t* out, unsigned flags) { return make_event(out, !(flags & hipEventDisableTiming)); }
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeStream* s = S(st);
  FakeEvent* e = E(ev);
  if (!s || !e) return hipErrorInvalidResourceHandle;
  e->recorded_on = s;
  ++s->pending;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t ev) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeEvent* e = E(ev);
  if (!e) return hipErrorInvalidResourceHandle;
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeEvent *ea = E(a), *eb = E(b);
  if (!ea || !eb) return hipErrorInvalidResourceHandle;
  if (!ea->timing || !eb->timing) { violation("hipEventElapsedTime on an event without timing"); return hipErrorInvalidResourceHandle; }
  *ms = 0.001f;
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t ev) {
  std::lock_guard<std::mutex> l(g_mu);
  FakeEvent* e = E(ev);
  if (!e) return hipErrorInvalidResourceHandle;
  if (e->recorded_on && e->recorded_on->pending)
    violation("hipEventDestroy of event " + std::to_string(e->id) + " whose stream " + std::to_string(e->recorded_on->id) + " still has unsynchronised operations");
  g_events.erase(e);
  e->magic = DEAD;
  delete e;
  return hipSuccess;
}

}  // extern "C"
