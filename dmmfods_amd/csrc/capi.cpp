// extern "C" surface of libdmmfods_hip.so (see include/dmmfods_hip.h) and the launch-list executor.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <chrono>
#include <mutex>
#include <string>
#include <unistd.h>

#include "plan.h"

using namespace dmm;

void plan_build_tables(dmm_plan* p);
void plan_bind(dmm_plan* p, void* ws);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) return fail(DMM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

static bool g_overlap_wgrad = getenv("DMM_NO_OVERLAP") == nullptr;
static int g_graph = getenv("DMM_GRAPH") ? std::max(1, atoi(getenv("DMM_GRAPH"))) : 0;  // off by default: see launch_list (1: both lists, 2: the forward list only)
static unsigned long long g_option_epoch = 1;  // bumped by every dmm_set_option: captured graphs have the options of their time baked in
static int g_bucket_mb = getenv("DMM_GRAD_BUCKET_MB") ? atoi(getenv("DMM_GRAD_BUCKET_MB")) : 25;

// ------------------------------------------------------------------------------------------------ streams and events of the process
// One pool per device, created on first use and NEVER torn down: the three helper streams (side: weight gradients, backward leaves,
// the second encoder; pack: the late layers' weight pack; capture: hipGraph capture) live as long as the process, events go back
// to free lists when a plan is destroyed.  Round 4's teardown destroyed two low-priority streams and ~200 events per plan, ~350 plans
// per test process, without synchronising anything - and the driver's GPU suite died of a segmentation fault raised on a runtime
// thread (no Python thread state: faulthandler marked no thread "Current") while the main thread was inside dmm_plan_destroy.  The
// last user of a priority level's hardware queue going away is a queue destruction inside the runtime; a plan has no business causing
// one.  Nothing here is destroyed at exit either (no static destructor may touch a runtime that is unloading): the pointers leak by design.
namespace {
struct DevicePool {
  std::mutex mu;
  hipStream_t side = nullptr, pack = nullptr, capture = nullptr;
  std::vector<hipEvent_t> free_plain, free_timing;   // hipEventDisableTiming / timing events handed back by destroyed plans
  long events_made = 0;
};
constexpr int MAX_DEVICES = 64;
DevicePool* g_pools[MAX_DEVICES] = {};
std::mutex g_pools_mu;

DevicePool* pool_of(int device) {
  if (device < 0 || device >= MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> l(g_pools_mu);
  if (g_pools[device] == nullptr) g_pools[device] = new DevicePool();
  return g_pools[device];
}
// the helper streams of the CURRENT device's pool (created on first use)
hipError_t pool_streams(DevicePool* dp, bool want_capture) {
  std::lock_guard<std::mutex> l(dp->mu);
  if (dp->side == nullptr) {
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e != hipSuccess) return e;
    // lowest priority: the data-gradient chain on the caller's stream is the longer dependency chain (mid / high measured in round 2:
    // 33.6 / 33.8 / 33.9 ms; confining the stream to a CU mask in round 4: 41-57 ms instead of 28)
    hipStream_t s = nullptr, k = nullptr;
    if ((e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo)) != hipSuccess) return e;
    if ((e = hipStreamCreateWithPriority(&k, hipStreamNonBlocking, lo)) != hipSuccess) return e;   // (s leaks: the process is in trouble anyway)
    dp->side = s; dp->pack = k;
  }
  if (want_capture && dp->capture == nullptr) {
    hipStream_t c = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    dp->capture = c;
  }
  return hipSuccess;
}
hipEvent_t pool_event(DevicePool* dp, bool timing) {
  std::lock_guard<std::mutex> l(dp->mu);
  auto& fl = timing ? dp->free_timing : dp->free_plain;
  if (!fl.empty()) { hipEvent_t e = fl.back(); fl.pop_back(); return e; }
  hipEvent_t e = nullptr;
  const hipError_t rc = timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) return nullptr;
  ++dp->events_made;
  return e;
}
void pool_return(DevicePool* dp, std::vector<void*>& evs, bool timing) {
  std::lock_guard<std::mutex> l(dp->mu);
  auto& fl = timing ? dp->free_timing : dp->free_plain;
  for (void* e : evs) if (e) fl.push_back((hipEvent_t)e);
  evs.clear();
}
// DMM_TRACE_DESTROY=1: one line per teardown step, written straight to fd 2 (no buffering: the line in front of a fault survives it)
void trace_destroy(const char* what) {
  static const bool on = getenv("DMM_TRACE_DESTROY") != nullptr;
  if (!on) return;
  char buf[160];
  const int n = snprintf(buf, sizeof(buf), "[dmm] destroy: %s\n", what);
  if (n > 0) { ssize_t w = write(2, buf, (size_t)std::min<int>(n, (int)sizeof(buf) - 1)); (void)w; }
}
}  // namespace

extern "C" {

const char* dmm_last_error(void) { return g_err.c_str(); }
int dmm_version(void) { return 100; }

int dmm_set_option(const char* name, int value) {
  if (!name) return fail(DMM_ERR_INVALID, "null argument");
  ++g_option_epoch;
  if (std::string(name) == "overlap_wgrad") { g_overlap_wgrad = value != 0; return DMM_OK; }
  if (std::string(name) == "graph") { g_graph = value; return DMM_OK; }
  if (std::string(name) == "thin_logits") { dmm::thin_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "conv3") { dmm::conv3_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "wg3") { dmm::wg3_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "wgp") { dmm::wgp_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "wg5") { dmm::wg5_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "cvp") { dmm::cvp_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "bw1") { dmm::bw1_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "pig") { dmm::pig_set_enabled(value != 0); return DMM_OK; }
  if (std::string(name) == "grad_bucket_mb") {  // applies to plans created afterwards; 0 = one bucket
    if (value < 0) return fail(DMM_ERR_INVALID, "grad_bucket_mb must be >= 0");
    g_bucket_mb = value;
    return DMM_OK;
  }
  return fail(DMM_ERR_INVALID, std::string("unknown option ") + name);
}

int dmm_plan_create(const dmm_model_desc* desc, dmm_plan** out) {
  if (!desc || !out) return fail(DMM_ERR_INVALID, "null argument");
  if (desc->num_blocks < 2 || desc->num_blocks > 8) return fail(DMM_ERR_INVALID, "num_blocks must be in [2, 8]");
  if (desc->dtype != DMM_F32 && desc->dtype != DMM_F16 && desc->dtype != DMM_BF16)
    return fail(DMM_ERR_INVALID, "dtype must be DMM_F32, DMM_F16 or DMM_BF16");
  if (desc->dtype == DMM_BF16 && !desc->use_mfma) return fail(DMM_ERR_INVALID, "DMM_BF16 has no scalar check kernels: use_mfma must be 1");
  if (desc->batch < 1) return fail(DMM_ERR_INVALID, "batch must be >= 1");
  if (!(desc->loss_scale > 0)) return fail(DMM_ERR_INVALID, "loss_scale must be > 0");
  dmm_plan* p = new dmm_plan();
  p->desc = *desc;
  p->sw = PlanSwitches::from_environment();   // the only place a plan's switches are read
  p->bucket_bytes = (size_t)g_bucket_mb << 20;
  try {
    plan_build_tables(p);
  } catch (const std::domain_error& e) {
    delete p;
    return fail(DMM_ERR_SHAPE, e.what());
  } catch (const std::exception& e) {
    delete p;
    return fail(DMM_ERR_INVALID, e.what());
  }
  *out = p;
  return DMM_OK;
}

// Teardown contract (tests/test_host_cpu.py drives it under AddressSanitizer against a fake runtime that counts violations):
//   * every helper stream this plan has launched on is synchronised first - when the call returns nothing the library enqueued
//     outside the caller's own stream still reads the workspace or the arenas, so the caller may free them (what it enqueued on
//     the stream it passed in is the caller's to order, as with any stream-ordered allocator);
//   * no stream is destroyed (they belong to the process pool), events go back to the pool, a graph is destroyed behind its stream's
//     synchronisation; every HIP return code is looked at, the first failure is reported (DMM_ERR_HIP) and the teardown still completes;
//   * destroying NULL is DMM_OK; a handle must not be used after the call, whatever it returned.
int dmm_plan_destroy(dmm_plan* plan) {
  if (!plan) return DMM_OK;
  trace_destroy("begin");
  std::string first_err;
  auto note = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && first_err.empty()) first_err = std::string(what) + ": " + hipGetErrorString(e);
  };
  DevicePool* dp = plan->device >= 0 ? pool_of(plan->device) : nullptr;
  if (dp != nullptr) {
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != plan->device && hipSetDevice(plan->device) == hipSuccess;
    if (plan->used_side) {
      trace_destroy("synchronise the side stream");
      note(hipStreamSynchronize(dp->side), "hipStreamSynchronize(side stream)");
      trace_destroy("synchronise the pack stream");
      note(hipStreamSynchronize(dp->pack), "hipStreamSynchronize(pack stream)");
    }
    if (plan->used_capture && dp->capture) {
      trace_destroy("synchronise the capture stream");
      note(hipStreamSynchronize(dp->capture), "hipStreamSynchronize(capture stream)");
    }
    trace_destroy("graphs");
    for (auto& gc : plan->graphs) {
      for (auto& e : gc.entries) note(hipGraphExecDestroy((hipGraphExec_t)e.exec), "hipGraphExecDestroy");
      gc.entries.clear();
    }
    trace_destroy("events back to the pool");
    pool_return(dp, plan->fork_events, false);
    pool_return(dp, plan->join_events, false);
    pool_return(dp, plan->bucket_events, false);
    for (auto& which : plan->prof_events)
      for (auto& pass : which) pool_return(dp, pass, true);
    if (switched) (void)hipSetDevice(cur);
  }
  trace_destroy("delete");
  delete plan;
  trace_destroy("done");
  return first_err.empty() ? DMM_OK : fail(DMM_ERR_HIP, "dmm_plan_destroy: " + first_err);
}

int dmm_plan_num_tensors(const dmm_plan* plan) { return plan ? (int)plan->tensors.size() : 0; }

int dmm_plan_tensor_info(const dmm_plan* plan, int index, const char** name, int32_t* kind, int32_t* ndim, int64_t shape[4],
                         int64_t* arena_offset) {
  if (!plan || index < 0 || index >= (int)plan->tensors.size()) return fail(DMM_ERR_INVALID, "tensor index out of range");
  const TensorInfo& t = plan->tensors[index];
  if (name) *name = t.name.c_str();
  if (kind) *kind = t.kind;
  if (ndim) *ndim = t.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  if (arena_offset) *arena_offset = t.off;
  return DMM_OK;
}

int64_t dmm_plan_num_params(const dmm_plan* plan) { return plan ? plan->nparams : 0; }
int64_t dmm_plan_num_buffer_elems(const dmm_plan* plan) { return plan ? plan->nbuf : 0; }
size_t dmm_plan_workspace_bytes(const dmm_plan* plan) { return plan ? plan->zero_bytes + plan->zero_bwd_bytes + plan->main_bytes : 0; }
double dmm_plan_forward_flops(const dmm_plan* plan) { return plan ? plan->fwd_flops : 0; }

int dmm_plan_bind(dmm_plan* plan, void* workspace, size_t workspace_bytes, float* params, float* grads, float* buffers) {
  if (!plan || !workspace || !params || !grads || !buffers) return fail(DMM_ERR_INVALID, "null argument");
  if (workspace_bytes < dmm_plan_workspace_bytes(plan)) return fail(DMM_ERR_INVALID, "workspace too small");
  if ((uintptr_t)workspace % 256) return fail(DMM_ERR_INVALID, "workspace must be 256-byte aligned");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DMM_ERR_NO_DEVICE, "no HIP device");
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (plan->device >= 0 && plan->device != dev) return fail(DMM_ERR_STATE, "the plan is bound to another device");
  if (dev >= MAX_DEVICES) return fail(DMM_ERR_INVALID, "device index out of range");
  plan->device = dev;
  plan->params = params;
  plan->grads = grads;
  plan->buffers = buffers;
  plan->graphs[0].epoch = plan->graphs[1].epoch = 0;  // captured graphs hold the old pointers
  plan->bound = false;
  try {
    plan_bind(plan, workspace);
  } catch (const dmm::plan_sizing_error& e) {
    return fail(DMM_ERR_STATE, e.what());
  } catch (const std::exception& e) {
    return fail(DMM_ERR_INVALID, e.what());
  }
  HIPCHK(hipDeviceSynchronize());
  plan->bound = true;
  return DMM_OK;
}

// Launches ops[begin, end) in order (end = 0: to the end of the list).
// capturing: the calls are being recorded into a hipGraph (no profiling, no bucket events: launch_list records those behind the graph).
static int run_ops(dmm_plan* p, std::vector<Op>& ops, hipStream_t st, int prof_which = -1, size_t begin = 0, size_t end = 0, bool capturing = false) {
  if (end == 0 || end > ops.size()) end = ops.size();
  const size_t ev_offset = 0;
  const int dt = p->desc.dtype;
  const bool mfma = p->desc.use_mfma != 0;
  std::vector<void*>* evs = nullptr;
  if (prof_which >= 0 && p->prof_max_passes > 0 && p->prof_pass[prof_which] < p->prof_max_passes) {
    auto& sets = p->prof_events[prof_which];
    const int pass = p->prof_pass[prof_which]++;
    if ((int)sets.size() <= pass) sets.resize(pass + 1);
    evs = &sets[pass];
    const size_t need = 2 * (ops.size() + ev_offset);
    DevicePool* pdp = pool_of(p->device);
    while (evs->size() < need) {
      hipEvent_t e = pdp ? pool_event(pdp, true) : nullptr;
      if (e == nullptr) return fail(DMM_ERR_HIP, "hipEventCreate failed");
      evs->push_back((void*)e);
    }
  }
  const std::string& filt = p->prof_filter;
  auto selected = [&](const Op& o) { return evs != nullptr && (filt.empty() || strncmp(o.label, filt.c_str(), filt.size()) == 0); };
  // Overlap: a weight gradient only needs what the ops before it produced, and nothing but the final unpack reads it, so
  // it goes to a low-priority side stream (fork event before, one join before unpack).  Late layers launch a few hundred
  // workgroups of a few microseconds; two independent chains fill the CUs that one chain leaves idle.  Per-op profiling keeps
  // everything on one stream so that the event pairs bracket each kernel alone; a FILTERED profile (one kernel class) runs
  // as in production.
  const bool overlap = g_overlap_wgrad && !capturing && (evs == nullptr || !filt.empty());
  constexpr int nside = 1;  // 2 and 3 side streams measured: 0.3 / 0.6 ms slower; and the leaf chains need one FIFO
  size_t nfork = 0;
  bool forked = false;
  DevicePool* dp = pool_of(p->device);
  hipStream_t side_st = nullptr, pack_st = nullptr;
  if (overlap) {
    if (dp == nullptr || pool_streams(dp, false) != hipSuccess) return fail(DMM_ERR_HIP, "side stream");
    side_st = dp->side; pack_st = dp->pack;
    while ((int)p->join_events.size() < nside + 1) {  // [nside]: the pack stream's
      hipEvent_t je = pool_event(dp, false);
      if (je == nullptr) return fail(DMM_ERR_HIP, "hipEventCreate failed");
      p->join_events.push_back((void*)je);
    }
    p->used_side = true;
  }
  // The pack stream: the launch that packs the weights of the LATE layers (leaf == 2) runs there beside the first
  // layers of the forward pass; OP_JOIN with epi == 1 in front of the first late layer makes the main stream wait for it.
  bool pack_pending = false;
  auto join_pack = [&]() {
    if (pack_pending) {
      hipStreamWaitEvent(st, (hipEvent_t)p->join_events[nside], 0);
      pack_pending = false;
    }
  };
  auto join_side = [&]() {
    if (forked) {
      for (int k = 0; k < nside; ++k) {
        hipEventRecord((hipEvent_t)p->join_events[k], side_st);
        hipStreamWaitEvent(st, (hipEvent_t)p->join_events[k], 0);
      }
      forked = false;
    }
  };
  auto join = [&]() { join_side(); join_pack(); };
  // DMM_HOST_PROF=1: host time of the enqueue calls by op kind, printed when a list has run 20 times (tools/host_bound.py)
  static const bool host_prof = lab_flag("DMM_HOST_PROF");
  static double hp_launch[32] = {0}, hp_fork[32] = {0};
  static long hp_n[32] = {0}, hp_lists = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (size_t i = begin; i < end; ++i) {
    Op& o = ops[i];
    hipError_t e = hipSuccess;
    const double hp_t0 = host_prof ? now() : 0.0;
    double hp_t1 = hp_t0;
    // side-stream launches: weight gradients and the leaves of the backward graph (stem, raw-input branches)
    hipStream_t lst = st;
    if (overlap && o.leaf == 3 && forked) {
      // a launch that continues a chain on the side stream (the second stream's encoder): everything it reads was produced by the
      // launch in front of it on that stream, or before the chain's first launch (which forked from the main stream).  Only while
      // the side stream is forked in THIS call: a range that starts mid-chain, or a join inside the chain, makes the next launch
      // of the chain fork again like any leaf.
      lst = side_st;
    } else if (overlap && !o.chain && (o.kind == OP_WGRAD || o.leaf)) {
      if (nfork >= p->fork_events.size()) {
        hipEvent_t fe = pool_event(dp, false);
        if (fe == nullptr) { join(); return fail(DMM_ERR_HIP, "hipEventCreate failed"); }
        p->fork_events.push_back((void*)fe);
      }
      lst = o.leaf == 2 ? pack_st : side_st;
      hipEvent_t fe = (hipEvent_t)p->fork_events[nfork++];
      hipEventRecord(fe, st);
      hipStreamWaitEvent(lst, fe, 0);
      if (o.leaf == 2) pack_pending = true; else forked = true;
      if (host_prof) hp_t1 = now();
    }
    const bool sel = selected(o);
    if (sel) hipEventRecord((hipEvent_t)(*evs)[2 * (ev_offset + i)], lst);
    switch (o.kind) {
      case OP_MEMSET: e = hipMemsetAsync(o.ms.p, 0, o.ms.bytes, lst); break;
      case OP_COPY: e = hipMemcpyAsync(o.cp.dst, o.cp.src, o.cp.bytes, hipMemcpyDeviceToDevice, lst); break;
      case OP_CONVERT: e = launch_convert_input(o.cv, dt, lst); break;
      case OP_IGEMM: e = launch_igemm(o.c, dt, o.epi, mfma, lst, o.impl); break;
      case OP_WGRAD: e = launch_wgrad(o.w, dt, mfma, lst, o.impl); break;
      case OP_BW1: e = launch_bw1(o.b1, dt, lst); break;
      case OP_BW1RED: e = launch_bw1_reduce(o.b1, lst); break;
      case OP_RAWFIN: e = launch_wg5_rawfin(o.rf, lst); break;
      case OP_FIN64: e = launch_wg5_fin64(o.f64, lst); break;
      case OP_JOIN: if (o.epi == 1) join_pack(); else join_side(); break;  // the main stream waits for what the side (epi 1: pack) stream has been given so far
      case OP_BNFIN: e = launch_bn_finalize(o.bf, lst); break;
      case OP_BNBWD: e = launch_bn_bwd_finalize(o.bb, lst); break;
      case OP_POOL: e = launch_maxpool_fwd(o.mp, dt, lst); break;
      case OP_POOLBWD: e = launch_maxpool_bwd(o.mpb, dt, lst); break;
      case OP_BCE: e = launch_bce_metrics(o.bce, dt, lst); break;
      case OP_PACK: e = launch_pack(o.pk.descs, o.pk.prefix, o.pk.ndesc, o.pk.total_rows, dt, lst, o.pk.tdescs, o.pk.tiles, o.pk.nt1, o.pk.nt9); break;
      case OP_APPLYCORR: e = launch_apply_corr(o.ac, dt, lst); break;
      case OP_UNPACK: e = launch_unpack(o.pk.descs, o.pk.prefix, o.pk.ndesc, o.pk.total_rows, dt, o.pk.grad_scale, lst, o.pk.tdescs, o.pk.tiles, o.pk.nt1, o.pk.nt9); break;
      default: join(); return fail(DMM_ERR_STATE, "unknown op");
    }
    if (host_prof && o.kind < 32) { const double t2 = now(); hp_fork[o.kind] += hp_t1 - hp_t0; hp_launch[o.kind] += t2 - hp_t1; hp_n[o.kind]++; }
    if (e != hipSuccess) join();  // leave the main stream ordered after whatever the side stream already got
    if (e != hipSuccess) return fail(DMM_ERR_HIP, "op " + std::to_string(i) + " kind " + std::to_string(o.kind) + ": " + hipGetErrorString(e));
    if (sel) hipEventRecord((hipEvent_t)(*evs)[2 * (ev_offset + i) + 1], lst);
    if (pack_pending && o.leaf == 2) hipEventRecord((hipEvent_t)p->join_events[nside], lst);
    if (o.signal >= 0 && !capturing) {  // a gradient bucket is final on this stream from here on
      while ((int)p->bucket_events.size() <= o.signal) {
        hipEvent_t be = dp ? pool_event(dp, false) : nullptr;
        if (be == nullptr) { join(); return fail(DMM_ERR_HIP, "hipEventCreate failed"); }
        p->bucket_events.push_back((void*)be);
      }
      hipEventRecord((hipEvent_t)p->bucket_events[o.signal], lst);
    }
  }
  join();
  if (host_prof && ++hp_lists == 40) {
    double tot = 0;
    for (int k = 0; k < 32; ++k) tot += hp_launch[k] + hp_fork[k];
    fprintf(stderr, "[host] 40 lists: %.2f ms of enqueue calls per list pair\n", tot / 20 / 1e3);
    for (int k = 0; k < 32; ++k)
      if (hp_n[k]) fprintf(stderr, "[host] op kind %2d: %6ld calls per pair, launch %.2f us each, fork events %.2f us each\n", k, hp_n[k] / 20,
                           hp_launch[k] / hp_n[k], hp_fork[k] / hp_n[k]);
  }
  return DMM_OK;
}

// Every bucket event of the list, recorded on `st`: behind a graph replay all buckets are final at once.
static int record_bucket_events(dmm_plan* p, const std::vector<Op>& ops, hipStream_t st) {
  for (const Op& o : ops) {
    if (o.signal < 0) continue;
    while ((int)p->bucket_events.size() <= o.signal) {
      DevicePool* dp = pool_of(p->device);
      hipEvent_t be = dp ? pool_event(dp, false) : nullptr;
      if (be == nullptr) return fail(DMM_ERR_HIP, "hipEventCreate failed");
      p->bucket_events.push_back((void*)be);
    }
    HIPCHK(hipEventRecord((hipEvent_t)p->bucket_events[o.signal], st));
  }
  return DMM_OK;
}

// Runs a whole launch list.  The launches [seg_begin, seg_end) - everything that does not touch a caller pointer - are captured ONCE
// into a hipGraph the second time the list runs (on a plan-owned stream: the caller's may be the legacy default stream, which cannot
// be captured) as ONE chain - the weight gradients are not sent to the side stream - and replayed by one call from then on; the few
// launches in front of and behind the segment (input conversion, the stem convolution up to the join with the weight-packing stream,
// the logits kernel; the loss kernel) take the caller's pointers and stay eager, so the graph never has to be patched or captured
// again.  `which`: 0 training forward, 1 loss + backward.
// Measured on MI355X / ROCm 7.2 (round 3, tools/host_bound.py): a replay enqueues a step in 0.5 ms of host time instead of 14 ms (C1) /
// 25 ms (C2), but the GPU time of a step does not change (C1 18.6 ms, C2 31.7 ms replayed = the same launches eager on one stream):
// the small configurations are bound by the latency of ~700 dependent tiny-grid kernels, not by the host, and the eager
// two-stream schedule is faster than either (C1 16.5 ms, C2 29.5 ms).  A capture of the two-stream schedule (fork / join events
// between the streams) replayed at 37 ms (C1) / 41 ms (C2) and returned different weight gradients than the eager launches, so the
// capture is single-stream.  Hence OFF by default (dmm_set_option("graph", 1) / DMM_GRAPH=1): it frees the host, it does not buy time.
static int launch_list(dmm_plan* p, int which, std::vector<Op>& ops, size_t seg_begin, size_t seg_end, hipStream_t st) {
  dmm_plan::GraphCache& gc = p->graphs[which];
  const bool profiling = p->prof_max_passes > 0 && p->prof_pass[which] < p->prof_max_passes;
  if (gc.epoch != g_option_epoch) {  // an option changed (or the plan was re-bound) since the graph was captured
    for (auto& e : gc.entries) hipGraphExecDestroy((hipGraphExec_t)e.exec);
    gc.entries.clear();
    gc.nseen = 0;
    gc.epoch = g_option_epoch;
  }
  const bool want = g_graph && !(g_graph == 2 && which == 1) && !p->graph_failed && !profiling && !(which == 1 && p->dp_used) && seg_end > seg_begin + 8;
  if (!want) return run_ops(p, ops, st, which);
  if (gc.entries.empty() && gc.nseen > 0) {  // the second run of the list: capture the segment
    DevicePool* dp = pool_of(p->device);
    if (dp == nullptr || pool_streams(dp, true) != hipSuccess) return fail(DMM_ERR_HIP, "capture stream");
    hipStream_t cs = dp->capture;
    p->used_capture = true;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int rc = run_ops(p, ops, cs, -1, seg_begin, seg_end, true);
      const bool ended = hipStreamEndCapture(cs, &graph) == hipSuccess && graph != nullptr;
      ok = rc == DMM_OK && ended;
      if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
    }
    gc.captures++;
    static const bool gtrace = lab_flag("DMM_GRAPH_TRACE");
    if (gtrace) fprintf(stderr, "[dmm] graph capture of list %d, launches [%zu, %zu): %s\n", which, seg_begin, seg_end, ok ? "ok" : "FAILED");
    if (!ok) {
      (void)hipGetLastError();
      p->graph_failed = true;  // stay eager on this plan
    } else {
      dmm_plan::GraphEntry e;
      for (int i = 0; i < 4; ++i) e.key[i] = nullptr;
      e.exec = (void*)exec; e.stamp = 0;
      gc.entries.push_back(e);
    }
  }
  gc.nseen = 1;
  if (gc.entries.empty()) return run_ops(p, ops, st, which);
  int rc = seg_begin > 0 ? run_ops(p, ops, st, -1, 0, seg_begin) : DMM_OK;
  if (rc) return rc;
  HIPCHK(hipGraphLaunch((hipGraphExec_t)gc.entries[0].exec, st));
  p->graph_replays[which]++;
  if (seg_end < ops.size()) { rc = run_ops(p, ops, st, -1, seg_end, ops.size()); if (rc) return rc; }
  return which == 1 ? record_bucket_events(p, ops, st) : DMM_OK;
}

int dmm_plan_forward(dmm_plan* plan, const float* stream_1, const float* stream_2, float* logits_out, int training, void* stream) {
  if (!plan || !plan->bound) return fail(DMM_ERR_STATE, "plan not bound");
  if (!stream_1 || !logits_out) return fail(DMM_ERR_INVALID, "null argument");
  if (plan->desc.stream_2_in_channels > 0 && !stream_2) return fail(DMM_ERR_INVALID, "stream_2 required");
  std::vector<Op>& ops = training ? plan->fwd_train : plan->fwd_eval;
  for (int idx : (training ? plan->convert_ops_train : plan->convert_ops_eval)) {
    Op& o = ops[idx];
    if (o.epi == 0) { o.cv.src1 = stream_1; o.cv.src2 = nullptr; }
    else if (o.epi == 1) { o.cv.src1 = stream_1; o.cv.src2 = stream_2; }
    else { o.cv.src1 = stream_2; o.cv.src2 = nullptr; }
  }
  ops[training ? plan->logits_op_train : plan->logits_op_eval].c.logits = logits_out;
  if (!training) return run_ops(plan, ops, (hipStream_t)stream, -1);
  // the replayed segment: behind the join with the weight-packing stream (the launches in front of it read the caller's inputs),
  // in front of the logits kernel (which writes the caller's tensor)
  size_t seg_begin = 0;
  for (size_t i = 0; i < ops.size(); ++i) if (ops[i].kind == OP_JOIN) seg_begin = i + 1;
  if (seg_begin == 0) for (int idx : plan->convert_ops_train) seg_begin = std::max(seg_begin, (size_t)idx + 1);
  return launch_list(plan, 0, ops, seg_begin, (size_t)plan->logits_op_train, (hipStream_t)stream);
}

static void set_loss_fields(const dmm_plan* plan, BceArgs& a) {
  a.kind = plan->loss_kind;
  for (int i = 0; i < 8; ++i) { a.alpha[i] = plan->loss_alpha[i]; a.gamma[i] = plan->loss_gamma[i]; }
}

int dmm_plan_set_loss(dmm_plan* plan, int kind, const float* alpha, const float* gamma, int nclass) {
  if (!plan) return fail(DMM_ERR_INVALID, "null plan");
  if (kind != DMM_LOSS_BCE && kind != DMM_LOSS_FOCAL) return fail(DMM_ERR_INVALID, "loss kind must be DMM_LOSS_BCE or DMM_LOSS_FOCAL");
  if (kind == DMM_LOSS_FOCAL) {
    if (!alpha || !gamma || nclass != plan->desc.num_classes) return fail(DMM_ERR_INVALID, "focal loss needs num_classes alpha / gamma values");
    for (int i = 0; i < nclass; ++i) { plan->loss_alpha[i] = alpha[i]; plan->loss_gamma[i] = gamma[i]; }
  }
  plan->loss_kind = kind;
  return DMM_OK;
}

int dmm_loss_forward(int kind, int from_prob, const float* alpha, const float* gamma, const float* input, const float* target,
                     float* loss_out, float* dinput_out, int batch, int nclass, int height, int width, void* stream) {
  if (!input || !target || (!loss_out && !dinput_out)) return fail(DMM_ERR_INVALID, "null argument");
  if (kind != DMM_LOSS_BCE && kind != DMM_LOSS_FOCAL) return fail(DMM_ERR_INVALID, "loss kind must be DMM_LOSS_BCE or DMM_LOSS_FOCAL");
  if (nclass < 1 || nclass > 8 || batch < 1 || height < 1 || width < 1) return fail(DMM_ERR_INVALID, "1..8 classes (dim 1) supported");
  if (kind == DMM_LOSS_FOCAL && (!alpha || !gamma)) return fail(DMM_ERR_INVALID, "focal loss needs alpha / gamma");
  BceArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = input; a.target = target; a.loss_out = loss_out; a.dx_out = dinput_out;
  a.B = batch; a.NC = nclass; a.H = height; a.W = width;
  a.thr = 0.f; a.loss_scale = 1.f; a.kind = kind; a.from_prob = from_prob ? 1 : 0;
  for (int i = 0; i < nclass; ++i) { a.alpha[i] = alpha ? alpha[i] : 1.f; a.gamma[i] = gamma ? gamma[i] : 2.f; }
  HIPCHK(launch_bce_metrics(a, DT_F32, (hipStream_t)stream));
  return DMM_OK;
}

int dmm_plan_loss_backward(dmm_plan* plan, const float* logits, const float* target, double* metrics_out, void* stream) {
  if (!plan || !plan->bound) return fail(DMM_ERR_STATE, "plan not bound");
  if (!logits || !target) return fail(DMM_ERR_INVALID, "null argument");
  Op& b = plan->bwd[plan->bce_op];
  b.bce.logits = logits;
  b.bce.target = target;
  set_loss_fields(plan, b.bce);
  // the loss kernel (caller pointers, loss parameters) stays eager; everything behind it is the replayed segment
  int rc = launch_list(plan, 1, plan->bwd, (size_t)plan->bce_op + 1, plan->bwd.size(), (hipStream_t)stream);
  if (rc) return rc;
  if (metrics_out) HIPCHK(hipMemcpyAsync(metrics_out, plan->metrics, plan->metrics_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DMM_OK;
}

int dmm_plan_backward(dmm_plan* plan, const float* dlogits, void* stream) {
  if (!plan || !plan->bound) return fail(DMM_ERR_STATE, "plan not bound");
  if (!dlogits) return fail(DMM_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  const BceArgs& b = plan->bwd[plan->bce_op].bce;
  ConvertArgs cv;
  memset(&cv, 0, sizeof(cv));
  cv.src1 = dlogits; cv.C1 = b.NC; cv.dst = b.dlogits; cv.B = b.B; cv.H = b.H; cv.W = b.W;
  cv.scale = plan->desc.loss_scale;
  // everything the fused path runs in front of the loss kernel (backward accumulators AND the gradient arena: unpack adds
  // into it for merged-tap / shared-master weights), then the external d(loss)/d(logit) instead of the loss kernel
  int rc = run_ops(plan, plan->bwd, st, -1, 0, (size_t)plan->bce_op);
  if (rc) return rc;
  HIPCHK(launch_convert_input(cv, plan->desc.dtype, st));
  return run_ops(plan, plan->bwd, st, -1, (size_t)plan->bce_op + 1, 0);
}

long long dmm_plan_num_graph_replays(const dmm_plan* plan, int which) {
  return (plan && which >= 0 && which <= 1) ? plan->graph_replays[which] : 0;
}

int dmm_plan_num_grad_buckets(const dmm_plan* plan) { return plan ? (int)plan->buckets.size() : 0; }

int dmm_plan_grad_bucket(const dmm_plan* plan, int index, int64_t* offset, int64_t* count) {
  if (!plan || index < 0 || index >= (int)plan->buckets.size()) return fail(DMM_ERR_INVALID, "bucket index out of range");
  if (offset) *offset = plan->buckets[index].off;
  if (count) *count = plan->buckets[index].n;
  return DMM_OK;
}

int dmm_plan_grad_bucket_wait(dmm_plan* plan, int index, void* stream) {
  if (!plan || index < 0 || index >= (int)plan->buckets.size()) return fail(DMM_ERR_INVALID, "bucket index out of range");
  const GradBucket& b = plan->buckets[index];
  plan->dp_used = true;  // from now on the backward runs eagerly: its bucket events must be recorded where the buckets become final
  for (int ev : {b.ev_main, b.ev_side}) {
    if (ev < 0) continue;
    if (ev >= (int)plan->bucket_events.size()) return fail(DMM_ERR_STATE, "no backward pass has been enqueued on this plan yet");
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)plan->bucket_events[ev], 0));
  }
  return DMM_OK;
}

int dmm_plan_profile_begin(dmm_plan* plan, int max_passes) {
  if (!plan) return fail(DMM_ERR_INVALID, "null plan");
  plan->prof_max_passes = max_passes;
  plan->prof_pass[0] = plan->prof_pass[1] = 0;
  return DMM_OK;
}

int dmm_plan_profile_filter(dmm_plan* plan, const char* label_prefix) {
  if (!plan) return fail(DMM_ERR_INVALID, "null plan");
  plan->prof_filter = label_prefix ? label_prefix : "";
  return DMM_OK;
}

int dmm_plan_profile_num_ops(const dmm_plan* plan, int which) {
  if (!plan || which < 0 || which > 1) return 0;
  return (int)(which == 0 ? plan->fwd_train.size() : plan->bwd.size());
}

int dmm_plan_profile_op(const dmm_plan* plan, int which, int index, const char** label, double* flops, double* bytes) {
  if (!plan || which < 0 || which > 1) return fail(DMM_ERR_INVALID, "bad argument");
  const std::vector<Op>& ops = which == 0 ? plan->fwd_train : plan->bwd;
  if (index < 0 || index >= (int)ops.size()) return fail(DMM_ERR_INVALID, "op index out of range");
  if (label) *label = ops[index].label;
  if (flops) *flops = ops[index].flops;
  if (bytes) *bytes = ops[index].bytes;
  return DMM_OK;
}

/* Sum of per-op durations (ms) over the recorded passes; the caller must have synchronised the stream. */
int dmm_plan_profile_collect(dmm_plan* plan, int which, double* ms_sum, int n, int* passes) {
  if (!plan || which < 0 || which > 1 || !ms_sum) return fail(DMM_ERR_INVALID, "bad argument");
  const int nops = dmm_plan_profile_num_ops(plan, which);
  if (n < nops) return fail(DMM_ERR_INVALID, "buffer too small");
  for (int i = 0; i < nops; ++i) ms_sum[i] = 0;
  int np = 0;
  for (auto& evs : plan->prof_events[which]) {
    if ((int)evs.size() < 2 * nops || np >= plan->prof_pass[which]) continue;
    const std::vector<Op>& ops = which == 0 ? plan->fwd_train : plan->bwd;
    const std::string& filt = plan->prof_filter;
    for (int i = 0; i < nops; ++i) {
      if (!filt.empty() && strncmp(ops[i].label, filt.c_str(), filt.size()) != 0) continue;
      float ms = 0;
      if (hipEventElapsedTime(&ms, (hipEvent_t)evs[2 * i], (hipEvent_t)evs[2 * i + 1]) != hipSuccess) return fail(DMM_ERR_HIP, "hipEventElapsedTime failed");
      ms_sum[i] += ms;
    }
    ++np;
  }
  if (passes) *passes = np;
  return DMM_OK;
}

int dmm_plan_loss_metrics(dmm_plan* plan, const float* logits, const float* target, double* metrics_out, void* stream) {
  if (!plan || !plan->bound) return fail(DMM_ERR_STATE, "plan not bound");
  if (!logits || !target || !metrics_out) return fail(DMM_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(plan->metrics, 0, plan->metrics_bytes, st));
  Op o = plan->bce_only;
  o.bce.logits = logits;
  o.bce.target = target;
  set_loss_fields(plan, o.bce);
  HIPCHK(launch_bce_metrics(o.bce, plan->desc.dtype, st));
  HIPCHK(hipMemcpyAsync(metrics_out, plan->metrics, plan->metrics_bytes, hipMemcpyDeviceToDevice, st));
  return DMM_OK;
}

int dmm_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return fail(DMM_ERR_INVALID, "bad Adam argument");
  AdamArgs a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = (size_t)n;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
  a.step_size = (float)(lr / bc1);
  a.bc2_sqrt = (float)std::sqrt(bc2);
  a.grad_scale = grad_scale;
  HIPCHK(launch_adam(a, (hipStream_t)stream));
  return DMM_OK;
}

// ------------------------------------------------------------------------------------------------ single-kernel entry points
namespace {
// Descriptor upload of the single-kernel entry points.  The source is stack / local-vector memory of the call, so the copy must be
// COMPLETE when this returns (an asynchronous copy from pageable memory only happens to be staged at once by the runtime), and what
// the stream still runs may be reading the descriptors the scratch held before.
hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
  const hipError_t e = hipStreamSynchronize(st);
  return e != hipSuccess ? e : hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}
struct OneConv {
  int esz, SLOT, BK;
  int Ho, Wo, Hout, Wout, ostride, istride;
  std::vector<std::vector<Tap>> phase_taps;
  std::vector<std::pair<int, int>> phase_xy;
  int Cst;  // storage channels of the input
};
int rup(int v, int m) { return (v + m - 1) / m * m; }

bool geometry(const dmm_conv_desc* d, OneConv& g) {
  g.esz = (int)dtype_size(d->dtype);
  g.SLOT = 16 / g.esz;
  g.BK = 4 * g.SLOT;
  g.Cst = d->Cin;
  g.ostride = 1;
  g.istride = 1;
  if (d->Cin % 8 || d->Cout % 8 == 7) return false;
  if (d->transposed) {
    g.Ho = d->H; g.Wo = d->W; g.Hout = 2 * d->H; g.Wout = 2 * d->W; g.ostride = 2;
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) { g.phase_taps.push_back(taps_convT_phase(py, px)); g.phase_xy.push_back({py, px}); }
    return true;
  }
  if (d->mode == 2) { g.Ho = d->H / 2; g.Wo = d->W / 2; g.phase_taps.push_back(taps_conv(1, 1, 0)); }
  else if (d->mode == 1) { g.Ho = 2 * d->H; g.Wo = 2 * d->W; g.phase_taps.push_back(taps_conv(d->R, d->S, d->pad)); }
  else {
    g.Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1;
    g.Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    g.istride = d->stride;
    g.phase_taps.push_back(taps_conv(d->R, d->S, d->pad));
  }
  g.Hout = g.Ho; g.Wout = g.Wo;
  g.phase_xy.push_back({0, 0});
  return true;
}

struct Scratch {
  uint8_t* base;
  size_t off = 0;
  void* take(size_t bytes) {
    off = (off + 255) / 256 * 256;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

// builds forward pack descriptors (one per phase); returns bytes used
size_t layout_fwd(const dmm_conv_desc* d, const OneConv& g, uint8_t* scratch, const float* w, float* dw, std::vector<PackDesc>& packs,
                  PackDesc** dev_descs, int** dev_prefix, std::vector<int>& prefix, int& total_rows) {
  Scratch S{scratch};
  *dev_descs = (PackDesc*)S.take(8 * sizeof(PackDesc));
  *dev_prefix = (int*)S.take(8 * sizeof(int));
  total_rows = 0;
  const long long RS = (long long)d->R * d->S;
  for (size_t ph = 0; ph < g.phase_taps.size(); ++ph) {
    PackDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.w = w; pd.gw = dw;
    pd.N = d->Cout; pd.Npad = rup(d->Cout, 32); pd.nseg = 1;
    if (!d->transposed) { pd.sn = d->Cin * RS; pd.sk = RS; } else { pd.sn = RS; pd.sk = (long long)d->Cout * RS; }
    pd.st = 1;
    fill_pack_seg(pd.seg[0], g.phase_taps[ph], d->Cin, g.Cst, 0, g.BK);
    const size_t elems = (size_t)pd.seg[0].nchunks * pd.Npad * g.BK;
    pd.dst = S.take(elems * g.esz);
    pd.dpack = (float*)S.take(elems * sizeof(float));
    prefix.push_back(total_rows);
    total_rows += pd.seg[0].nchunks * pd.Npad;
    packs.push_back(pd);
  }
  return S.off;
}
}  // namespace

size_t dmm_conv_scratch_bytes(const dmm_conv_desc* d) {
  OneConv g;
  if (!d || !geometry(d, g)) return 0;
  std::vector<PackDesc> packs;
  std::vector<int> prefix;
  PackDesc* dd; int* dp; int tr;
  size_t fwd = layout_fwd(d, g, nullptr, nullptr, nullptr, packs, &dd, &dp, prefix, tr);
  // dgrad pack: [chunks][rup(Cin,32)][BK] with K' = R*S*rup(Cout,8)
  const size_t dg = (size_t)((d->R * d->S * rup(d->Cout, 8) + g.BK - 1) / g.BK + 16) * rup(d->Cin, 128) * g.BK * g.esz;
  // + the fp32 dgrad-shaped packed gradient of the transposed-form weight gradient, zero tables, descriptors
  return fwd + dg + 2 * dg + (size_t)rup(d->Cout, 8) * 8 + 8192 + (size_t)W5_SBUF64_FLOATS * sizeof(float);
}

static void fill_one_seg(Seg& s, const dmm_conv_desc* d, const OneConv& g, const void* x, const float* scale, const float* shift,
                         const std::vector<Tap>& taps) {
  memset(&s, 0, sizeof(s));
  s.src = x; s.ld = g.Cst; s.Hs = d->H; s.Ws = d->W; s.C = g.Cst; s.Cpad = g.Cst;
  s.mode = d->transposed ? 0 : d->mode;
  s.istride = g.istride;
  if (d->bn_relu) { s.scale = scale; s.shift = shift; }
  fill_seg_taps(s, taps, g.BK);
}

int dmm_last_impl(void) { return g_last_impl; }
unsigned dmm_impl_mask(int reset) { const unsigned m = g_impl_mask; if (reset) g_impl_mask = 0; return m; }
const char* dmm_impl_name(int impl) {
  static const char* const names[IMPL_COUNT] = {"auto", "generic", "thin", "conv3", "cvp", "halo", "wg3", "wg5", "wgp", "pig", "bw1", "hf", "cf", "wgpw", "cvw"};
  return impl >= 0 && impl < IMPL_COUNT ? names[impl] : "?";
}

int dmm_conv_forward(const dmm_conv_desc* d, const void* x, const float* w, const float* scale, const float* shift, void* y,
                     double* stats, void* scratch, void* stream) {
  OneConv g;
  if (!d || !geometry(d, g)) return fail(DMM_ERR_INVALID, "unsupported conv descriptor");
  hipStream_t st = (hipStream_t)stream;
  std::vector<PackDesc> packs;
  std::vector<int> prefix;
  PackDesc* dd; int* dp; int total_rows;
  layout_fwd(d, g, (uint8_t*)scratch, w, nullptr, packs, &dd, &dp, prefix, total_rows);
  HIPCHK(upload(dd, packs.data(), packs.size() * sizeof(PackDesc), st));
  HIPCHK(upload(dp, prefix.data(), prefix.size() * sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(launch_pack(dd, dp, (int)packs.size(), total_rows, d->dtype, st));
  if (stats) HIPCHK(hipMemsetAsync(stats, 0, 2 * d->Cout * sizeof(double), st));
  for (size_t ph = 0; ph < packs.size(); ++ph) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = 1;
    fill_one_seg(a.seg[0], d, g, x, scale, shift, g.phase_taps[ph]);
    a.B = d->B; a.Ho = g.Ho; a.Wo = g.Wo; a.M = d->B * g.Ho * g.Wo;
    a.wpack = packs[ph].dst; a.N = d->Cout; a.Npad = packs[ph].Npad;
    a.out = y; a.ldo = d->Cout; a.Hout = g.Hout; a.Wout = g.Wout; a.ostride = g.ostride;
    a.py = g.phase_xy[ph].first; a.px = g.phase_xy[ph].second;
    if (stats) { a.stat_sum = stats; a.stat_sq = stats + d->Cout; }
    HIPCHK(launch_igemm(a, d->dtype, EPI_STORE, d->use_mfma != 0, st));
  }
  return DMM_OK;
}

// dy_eff = dy + q[c] + r[c] * yfwd (the deferred BatchNorm-backward correction of the plan's gradient buffers) when q != NULL
static void set_eff_grad(Seg& s, const void* yfwd, int ldy, const float* q, const float* r, const float* zeros) {
  if (q == nullptr) return;
  s.src2 = yfwd; s.ld2 = ldy; s.q = q; s.r = r; s.ql = zeros; s.rl = zeros;
}

// the dgrad-shaped pack of a convolution: [chunks over (tap, output channel)][Npad input channels][BK]
static PackDesc dgrad_pack_desc(const dmm_conv_desc* d, const OneConv& g, const std::vector<Tap>& taps, const float* w, int npad) {
  PackDesc pd;
  memset(&pd, 0, sizeof(pd));
  const long long RS = (long long)d->R * d->S;
  pd.w = w;
  pd.N = d->Cin; pd.Npad = npad; pd.nseg = 1;
  if (!d->transposed) { pd.sn = RS; pd.sk = d->Cin * RS; } else { pd.sn = (long long)d->Cout * RS; pd.sk = RS; }
  pd.st = 1;
  fill_pack_seg(pd.seg[0], taps, d->Cout, rup(d->Cout, 8), 0, g.BK);
  return pd;
}

int dmm_conv_wgrad_ex(const dmm_conv_desc* d, const void* x, const void* dy, const float* scale, const float* shift, const void* yfwd,
                      const float* q, const float* r, int transposed_form, float* dw, void* scratch, void* stream) {
  OneConv g;
  if (!d || !geometry(d, g)) return fail(DMM_ERR_INVALID, "unsupported conv descriptor");
  if ((q != nullptr) != (r != nullptr) || (q != nullptr && yfwd == nullptr)) return fail(DMM_ERR_INVALID, "q, r and yfwd come together");
  hipStream_t st = (hipStream_t)stream;
  const size_t wn = (size_t)d->Cin * d->Cout * d->R * d->S;
  if (transposed_form) {
    // taps on the gradient side (wgrad.hip "transposed form", what the plan builds for thin outputs: plan.cpp finish_conv):
    // the result has the shape of the data-gradient pack and is scattered into the master layout through that descriptor
    if (d->transposed || d->mode != 0 || d->stride != 1 || d->R * d->S <= 1 || !d->bn_relu)
      return fail(DMM_ERR_INVALID, "the transposed form serves unit-stride multi-tap convolutions behind BN+ReLU");
    const std::vector<Tap> taps = taps_conv_dgrad(d->R, d->S, d->pad);
    const int cs = g.Cst;
    const int npad = rup(cs, cs >= 384 || cs % 128 == 0 ? 128 : (cs >= 64 ? 64 : 32));
    Scratch S{(uint8_t*)scratch};
    PackDesc* dd = (PackDesc*)S.take(sizeof(PackDesc));
    int* dp = (int*)S.take(sizeof(int));
    float* zeros = (float*)S.take((size_t)rup(d->Cout, 8) * sizeof(float) + 64);
    PackDesc pd = dgrad_pack_desc(d, g, taps, dw, npad);
    const size_t elems = (size_t)pd.seg[0].nchunks * pd.Npad * g.BK;
    pd.dpack = (float*)S.take(elems * sizeof(float));
    pd.gw = dw;
    if (S.off > dmm_conv_scratch_bytes(d)) return fail(DMM_ERR_INVALID, "scratch too small");
    HIPCHK(hipMemsetAsync(scratch, 0, S.off, st));
    int zero = 0;
    HIPCHK(upload(dd, &pd, sizeof(pd), st));
    HIPCHK(upload(dp, &zero, sizeof(int), st));
    HIPCHK(hipStreamSynchronize(st));
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = 1;
    Seg& sq = a.seg[0];  // Q: the output gradient under the flipped taps
    sq.src = dy; sq.ld = d->Cout; sq.Hs = g.Hout; sq.Ws = g.Wout; sq.C = rup(d->Cout, 8); sq.Cpad = sq.C; sq.mode = G_PLAIN; sq.istride = 1;
    set_eff_grad(sq, yfwd, d->Cout, q, r, zeros);
    fill_seg_taps(sq, taps, g.BK);
    a.B = d->B; a.Ho = g.Ho; a.Wo = g.Wo; a.M = d->B * g.Ho * g.Wo;
    fill_one_seg(a.dy, d, g, x, scale, shift, taps_conv(1, 1, 0));  // P: the activated input, once
    a.dy.istride = 1;
    a.N = g.Cst; a.Npad = pd.Npad;
    a.dpack = (float*)pd.dpack;
    void* part = nullptr;
    if (d->use_mfma && wg3_handles(a, d->dtype)) {  // the plan gives the family its slots; so does this entry point
      HIPCHK(hipMallocAsync(&part, (size_t)W3_MAX_SLOTS * W3_SLOT_FLOATS * sizeof(float), st));
      a.part = (float*)part;
      a.part_slots = W3_MAX_SLOTS;
    }
    const hipError_t e1 = launch_wgrad(a, d->dtype, d->use_mfma != 0, st);
    if (part != nullptr) hipFreeAsync(part, st);
    HIPCHK(e1);
    HIPCHK(hipMemsetAsync(dw, 0, wn * sizeof(float), st));
    HIPCHK(launch_unpack(dd, dp, 1, pd.seg[0].nchunks * pd.Npad, d->dtype, 1.0f, st));
    return DMM_OK;
  }
  std::vector<PackDesc> packs;
  std::vector<int> prefix;
  PackDesc* dd; int* dp; int total_rows;
  const size_t used = layout_fwd(d, g, (uint8_t*)scratch, dw /*unused as w*/, dw, packs, &dd, &dp, prefix, total_rows);
  Scratch S{(uint8_t*)scratch, used};
  float* zeros = (float*)S.take((size_t)rup(d->Cout, 8) * sizeof(float) + 64);
  if (S.off > dmm_conv_scratch_bytes(d)) return fail(DMM_ERR_INVALID, "scratch too small");
  HIPCHK(hipMemsetAsync(scratch, 0, S.off, st));
  HIPCHK(upload(dd, packs.data(), packs.size() * sizeof(PackDesc), st));
  HIPCHK(upload(dp, prefix.data(), prefix.size() * sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));
  for (size_t ph = 0; ph < packs.size(); ++ph) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.nseg = 1;
    fill_one_seg(a.seg[0], d, g, x, scale, shift, g.phase_taps[ph]);
    a.B = d->B; a.Ho = g.Ho; a.Wo = g.Wo; a.M = d->B * g.Ho * g.Wo;
    a.dy.src = dy; a.dy.ld = d->Cout; a.dy.Hs = g.Hout; a.dy.Ws = g.Wout; a.dy.C = rup(d->Cout, 8); a.dy.Cpad = a.dy.C;
    a.dy.mode = G_PLAIN; a.dy.istride = g.ostride; a.dy.ntaps = 1; a.dy.nchunks = 1;
    a.dy.taps[0] = (short)((g.phase_xy[ph].first & 0xff) | ((g.phase_xy[ph].second & 0xff) << 8));
    set_eff_grad(a.dy, yfwd, d->Cout, q, r, zeros);
    a.N = d->Cout; a.Npad = packs[ph].Npad;
    a.dpack = (float*)packs[ph].dpack;
    HIPCHK(launch_wgrad(a, d->dtype, d->use_mfma != 0, st));
  }
  HIPCHK(hipMemsetAsync(dw, 0, wn * sizeof(float), st));
  HIPCHK(launch_unpack(dd, dp, (int)packs.size(), total_rows, d->dtype, 1.0f, st));
  return DMM_OK;
}

int dmm_conv5_wgrad_stats(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale, const float* shift,
                          float* dw, double* red, void* scratch, void* stream) {
  OneConv g;
  if (!d || !geometry(d, g)) return fail(DMM_ERR_INVALID, "unsupported conv descriptor");
  if (d->transposed || d->mode != 0 || d->stride != 1 || d->R != 5 || d->S != 5 || d->pad != 2 || !d->bn_relu || d->Cin != 64 || d->Cout > 4 ||
      d->dtype == DMM_F32 || !d->use_mfma)
    return fail(DMM_ERR_INVALID, "serves the 5x5 convolution of 64 channels onto <= 4 behind BN+ReLU in 16-bit storage");
  hipStream_t st = (hipStream_t)stream;
  const size_t wn = (size_t)d->Cin * d->Cout * 25;
  const std::vector<Tap> taps = taps_conv_dgrad(5, 5, 2);
  Scratch S{(uint8_t*)scratch};
  PackDesc* dd = (PackDesc*)S.take(sizeof(PackDesc));
  int* dp = (int*)S.take(sizeof(int));
  PackDesc pd = dgrad_pack_desc(d, g, taps, dw, 64);
  const size_t elems = (size_t)pd.seg[0].nchunks * pd.Npad * g.BK;
  pd.dpack = (float*)S.take(elems * sizeof(float));
  pd.gw = dw;
  float* sbuf = (float*)S.take((size_t)W5_SBUF64_FLOATS * sizeof(float));
  if (S.off > dmm_conv_scratch_bytes(d)) return fail(DMM_ERR_INVALID, "scratch too small");
  HIPCHK(hipMemsetAsync(scratch, 0, S.off, st));
  HIPCHK(hipMemsetAsync(red, 0, 2 * (size_t)d->Cin * sizeof(double), st));
  int zero = 0;
  HIPCHK(upload(dd, &pd, sizeof(pd), st));
  HIPCHK(upload(dp, &zero, sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.nseg = 1;
  Seg& sq = a.seg[0];  // Q: the output gradient under the flipped taps
  sq.src = dy; sq.ld = d->Cout; sq.Hs = g.Hout; sq.Ws = g.Wout; sq.C = rup(d->Cout, 8); sq.Cpad = sq.C; sq.mode = G_PLAIN; sq.istride = 1;
  fill_seg_taps(sq, taps, g.BK);
  a.B = d->B; a.Ho = g.Ho; a.Wo = g.Wo; a.M = d->B * g.Ho * g.Wo;
  fill_one_seg(a.dy, d, g, x, scale, shift, taps_conv(1, 1, 0));  // P: the input, entered as the two factors of its activation
  a.dy.istride = 1;
  a.N = g.Cst; a.Npad = pd.Npad;
  a.dpack = (float*)pd.dpack;
  a.sbuf = sbuf;
  a.t_mean = shift + d->Cin; a.t_invstd = shift + 2 * d->Cin;
  HIPCHK(launch_wgrad(a, d->dtype, true, st));
  if (g_last_impl != IMPL_WG5) return fail(DMM_ERR_STATE, "the factor form is wg5.hip's");
  Fin64Args f;
  memset(&f, 0, sizeof(f));
  f.sbuf = sbuf; f.dpack = a.dpack; f.Npad = a.Npad;
  f.w = w; f.Kin = d->Cin; f.nreal = d->Cout; f.dtype = d->dtype;
  for (int t = 0; t < 25; ++t) f.tapw[t] = (unsigned char)(pd.seg[0].tapw[t] & 0xff);
  f.scale = scale; f.shift = shift; f.mean = shift + d->Cin; f.invstd = shift + 2 * d->Cin;
  f.red1 = red; f.red2 = red + d->Cin;
  HIPCHK(launch_wg5_fin64(f, st));
  HIPCHK(hipMemsetAsync(dw, 0, wn * sizeof(float), st));
  HIPCHK(launch_unpack(dd, dp, 1, pd.seg[0].nchunks * pd.Npad, d->dtype, 1.0f, st));
  return DMM_OK;
}

int dmm_conv_wgrad(const dmm_conv_desc* d, const void* x, const void* dy, const float* scale, const float* shift, float* dw, void* scratch,
                   void* stream) {
  return dmm_conv_wgrad_ex(d, x, dy, scale, shift, nullptr, nullptr, nullptr, 0, dw, scratch, stream);
}

// Fills the data-gradient launch of one convolution (EPI_BNBWD) and packs its weights; shared by dmm_conv_dgrad_ex and
// dmm_conv1x1_backward_fused.
static int build_dgrad(const dmm_conv_desc* d, const OneConv& g, const void* x, const void* dy, const float* w, const float* scale,
                       const float* shift, const void* yfwd, const float* q, const float* r, void* gx, int accumulate, double* red,
                       Scratch& S, int npad, ConvArgs& a, hipStream_t st) {
  std::vector<Tap> taps;
  int istride = 1, rH = d->H, rW = d->W, pool2 = 0, ostride = 1;
  if (d->transposed) { taps = taps_convT_dgrad(); istride = 2; }
  else if (d->mode == 2) { taps = taps_conv(1, 1, 0); rH = d->H / 2; rW = d->W / 2; pool2 = 1; ostride = 2; }
  else if (d->mode == 1) { taps = taps_up2_merged_dgrad(); istride = 2; }
  else taps = taps_conv_dgrad(d->R, d->S, d->pad);
  PackDesc* dd = (PackDesc*)S.take(sizeof(PackDesc));
  int* dp = (int*)S.take(sizeof(int));
  float* zeros = (float*)S.take((size_t)rup(d->Cout, 8) * sizeof(float) + 64);
  PackDesc pd = dgrad_pack_desc(d, g, taps, w, npad);
  pd.dst = S.take((size_t)pd.seg[0].nchunks * pd.Npad * g.BK * g.esz);
  if (S.off > dmm_conv_scratch_bytes(d)) return fail(DMM_ERR_INVALID, "scratch too small");
  int zero = 0;
  HIPCHK(hipMemsetAsync(zeros, 0, (size_t)rup(d->Cout, 8) * sizeof(float) + 64, st));
  HIPCHK(upload(dd, &pd, sizeof(pd), st));
  HIPCHK(upload(dp, &zero, sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(launch_pack(dd, dp, 1, pd.seg[0].nchunks * pd.Npad, d->dtype, st));
  HIPCHK(hipMemsetAsync(red, 0, 2 * d->Cin * sizeof(double), st));
  memset(&a, 0, sizeof(a));
  a.nseg = 1;
  Seg& s = a.seg[0];
  const int kc = rup(d->Cout, 8);
  s.src = dy; s.ld = d->Cout; s.Hs = g.Hout; s.Ws = g.Wout; s.C = kc; s.Cpad = kc; s.mode = G_PLAIN; s.istride = istride;
  set_eff_grad(s, yfwd, d->Cout, q, r, zeros);
  fill_seg_taps(s, taps, g.BK);
  a.B = d->B; a.Ho = rH; a.Wo = rW; a.M = d->B * rH * rW;
  a.wpack = pd.dst; a.N = d->Cin; a.Npad = pd.Npad;
  a.out = gx; a.ldo = d->Cin; a.Hout = d->H; a.Wout = d->W; a.ostride = ostride;
  a.bx = x; a.ldbx = d->Cin; a.bscale = scale; a.bshift = shift;
  a.bmean = shift + d->Cin; a.binvstd = shift + 2 * d->Cin;  // test entry point: mean / invstd follow `shift`
  a.red1 = red; a.red2 = red + d->Cin;
  a.accumulate = accumulate ? 1 : 0; a.pool2 = pool2;
  return DMM_OK;
}

int dmm_conv_dgrad_ex(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale, const float* shift,
                      const void* yfwd, const float* q, const float* r, void* gx, int accumulate, double* red, void* scratch,
                      void* stream) {
  OneConv g;
  if (!d || !geometry(d, g) || !d->bn_relu) return fail(DMM_ERR_INVALID, "unsupported conv descriptor (dgrad needs bn_relu)");
  if (d->mode == 0 && !d->transposed && d->stride != 1) return fail(DMM_ERR_INVALID, "strided conv dgrad is not on the hot path");
  if ((q != nullptr) != (r != nullptr) || (q != nullptr && yfwd == nullptr)) return fail(DMM_ERR_INVALID, "q, r and yfwd come together");
  hipStream_t st = (hipStream_t)stream;
  Scratch S{(uint8_t*)scratch};
  ConvArgs a;
  // column tile of the plan (plan.cpp add_pack): 128 once the padding waste is <= 25 %, else 64, else 32
  const int cs = g.Cst;
  const int npad = rup(cs, cs >= 384 || cs % 128 == 0 ? 128 : (cs >= 64 ? 64 : 32));
  const int rc = build_dgrad(d, g, x, dy, w, scale, shift, yfwd, q, r, gx, accumulate, red, S, npad, a, st);
  if (rc) return rc;
  HIPCHK(launch_igemm(a, d->dtype, EPI_BNBWD, d->use_mfma != 0, st));
  return DMM_OK;
}

int dmm_conv_dgrad(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale, const float* shift, void* gx,
                   double* red, void* scratch, void* stream) {
  return dmm_conv_dgrad_ex(d, x, dy, w, scale, shift, nullptr, nullptr, nullptr, gx, 0, red, scratch, stream);
}

int dmm_conv1x1_backward_fused(const dmm_conv_desc* d, const void* x, const void* dy, const float* w, const float* scale,
                               const float* shift, const void* yfwd, const float* q, const float* r, void* gx, int accumulate,
                               float* dw, double* red, void* scratch, void* stream) {
  OneConv g;
  if (!d || !geometry(d, g) || !d->bn_relu || d->R != 1 || d->S != 1 || d->stride != 1 || d->transposed || d->mode != 0)
    return fail(DMM_ERR_INVALID, "the fused backward serves 1x1 unit-stride convolutions behind BN+ReLU");
  if ((q != nullptr) != (r != nullptr) || (q != nullptr && yfwd == nullptr)) return fail(DMM_ERR_INVALID, "q, r and yfwd come together");
  hipStream_t st = (hipStream_t)stream;
  // forward-shaped packed gradient first (its descriptor scatters dW into the master layout), then the data-gradient launch
  std::vector<PackDesc> packs;
  std::vector<int> prefix;
  PackDesc* dd; int* dp; int total_rows;
  const size_t used = layout_fwd(d, g, (uint8_t*)scratch, dw, dw, packs, &dd, &dp, prefix, total_rows);
  HIPCHK(hipMemsetAsync(scratch, 0, used, st));
  HIPCHK(upload(dd, packs.data(), packs.size() * sizeof(PackDesc), st));
  HIPCHK(upload(dp, prefix.data(), prefix.size() * sizeof(int), st));
  Scratch S{(uint8_t*)scratch, used};
  Bw1Args b;
  memset(&b, 0, sizeof(b));
  const int cs = g.Cst;
  const int npad = rup(cs, cs >= 384 || cs % 128 == 0 ? 128 : (cs >= 64 ? 64 : 32));
  const int rc = build_dgrad(d, g, x, dy, w, scale, shift, yfwd, q, r, gx, accumulate, red, S, npad, b.c, st);
  if (rc) return rc;
  WgradArgs wa;  // the weight-gradient launch the plan would have emitted: only to test the pair's eligibility
  memset(&wa, 0, sizeof(wa));
  wa.nseg = 1;
  fill_one_seg(wa.seg[0], d, g, x, scale, shift, g.phase_taps[0]);
  wa.B = d->B; wa.Ho = g.Ho; wa.Wo = g.Wo; wa.M = d->B * g.Ho * g.Wo;
  wa.dy = b.c.seg[0];
  wa.N = d->Cout; wa.Npad = packs[0].Npad;
  wa.dpack = (float*)packs[0].dpack;
  if (!bw1_eligible(wa, b.c, d->dtype)) return fail(DMM_ERR_INVALID, "not a pair bw1.hip fuses (16-bit storage, 128 output channels, Cin % 32 == 0)");
  b.dpack = wa.dpack; b.dNpad = wa.Npad; b.wC = wa.seg[0].C;
  {  // the production form: per-workgroup slots + the reduction launch (stream-ordered scratch of this call)
    const Bw1Geom q = bw1_geometry(b.c);
    b.part_slots = q.nsplit * q.nct;
    void* part = nullptr;
    HIPCHK(hipMallocAsync(&part, (size_t)b.part_slots * B1_SLOT_FLOATS * sizeof(float), st));
    b.part = (float*)part;
    const hipError_t e1 = launch_bw1(b, d->dtype, st);
    const hipError_t e2 = e1 == hipSuccess ? launch_bw1_reduce(b, st) : e1;
    hipFreeAsync(part, st);
    HIPCHK(e2);
  }
  const size_t wn = (size_t)d->Cin * d->Cout;
  HIPCHK(hipMemsetAsync(dw, 0, wn * sizeof(float), st));
  HIPCHK(launch_unpack(dd, dp, (int)packs.size(), total_rows, d->dtype, 1.0f, st));
  return DMM_OK;
}

}  // extern "C"
