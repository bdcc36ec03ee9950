// Host-side plan of the Dense_U_Net_lidar training step: topology -> buffers -> kernel launch list.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dmmfods_hip.h"
#include "common.h"
#include "pointwise.h"

namespace dmm {

// plan_bind: the bound pass did not take the workspace bytes the sizing pass reported (dmm_plan_bind returns DMM_ERR_STATE)
struct plan_sizing_error : std::runtime_error { using std::runtime_error::runtime_error; };

struct TensorInfo {
  std::string name;
  int kind, ndim;
  int64_t shape[4];
  int64_t off;  // elements into the param arena / buffer arena, -1 for num_batches_tracked
};

struct Tap {
  int dy, dx;
  unsigned tapw;  // up to four master tap indices (8 bits each, 0xff = none)
};

// tap geometry helpers (shared with the single-kernel test entry points)
std::vector<Tap> taps_conv(int R, int S, int pad);            // forward stride-1/2 conv: (ky-pad, kx-pad)
std::vector<Tap> taps_conv_dgrad(int R, int S, int pad);      // data gradient of a stride-1 conv: (pad-ky, pad-kx)
std::vector<Tap> taps_convT_phase(int py, int px);            // ConvTranspose 3x3 s2 p1, output parity (py,px)
std::vector<Tap> taps_convT_dgrad();                          // stride-2 gather over the output gradient
std::vector<Tap> taps_up2_merged_dgrad();                     // 3x3 conv over a nearest-x2 source: 4x4 s2 merged taps
std::vector<Tap> taps_up2_phase(int e, int f);                // same conv, forward, output parity (e,f): 2x2 merged half-res taps
std::vector<Tap> taps_conv_phase_s2(int e, int f);            // 3x3 p1 conv sampled at output parity (e,f) from a full-res source
void fill_seg_taps(Seg& s, const std::vector<Tap>& taps, int BK);
void fill_pack_seg(PackSeg& p, const std::vector<Tap>& taps, int Creal, int Cpad, int koff, int BK);

enum OpKind { OP_MEMSET = 0, OP_CONVERT, OP_IGEMM, OP_WGRAD, OP_BNFIN, OP_BNBWD, OP_POOL, OP_POOLBWD, OP_BCE, OP_PACK, OP_UNPACK,
              OP_COPY, OP_APPLYCORR, OP_BW1, OP_JOIN, OP_BW1RED, OP_RAWFIN, OP_FIN64 };

struct MemsetArgs { void* p; size_t bytes; };
struct CopyArgs { void* dst; const void* src; size_t bytes; };
struct PackArgs {
  const PackDesc* descs; const int* prefix; int ndesc, total_rows; float grad_scale;
  const PackDesc* tdescs; const PackTile* tiles; int nt1, nt9;  // tile kernels: descriptor array the tiles index, the range's 1x1 / 3x3 tiles
};

struct Op {
  int kind;
  int epi;
  int leaf;        // nothing on the data-gradient chain reads what this launch produces (may run beside it): 1 side stream, forked from
                   // the main stream; 3 side stream, continuing the chain of the launch in front of it there (no fork); 2 pack stream
  int chain;       // 1: a weight-gradient-side launch the data-gradient chain WAITS for (its results feed the next launch of the chain):
                   // it runs on the main stream although its kind says side stream
  int signal;      // gradient bucket event to record behind this launch on the stream it ran on (-1: none)
  int impl;        // kernel family (enum Impl) chosen for this launch when the plan was built
  double flops;    // algorithmic 2*MACs of this launch (reference formulation)
  double bytes;    // algorithmic HBM bytes: every operand read once, every result written once
  char label[56];  // kernel class / layer
  union {
    MemsetArgs ms;
    CopyArgs cp;
    ConvertArgs cv;
    ConvArgs c;
    WgradArgs w;
    BnFinalizeArgs bf;
    BnBwdFinalizeArgs bb;
    MaxpoolArgs mp;
    MaxpoolBwdArgs mpb;
    BceArgs bce;
    PackArgs pk;
    ApplyCorrArgs ac;
    Bw1Args b1;
    RawFinArgs rf;
    Fin64Args f64;
  };
  Op() : kind(0), epi(0), leaf(0), chain(0), signal(-1), impl(IMPL_AUTO), flops(0), bytes(0) { label[0] = 0; }
};

}  // namespace dmm

// A contiguous range of the gradient arena (whole tensors) whose gradients become final at a known point of the backward
// launch list: data-parallel training all-reduces it from there on, beside the rest of backward.
struct GradBucket {
  int64_t off = 0, n = 0;             // elements of the gradient arena
  int ev_main = -1, ev_side = -1;     // indices into dmm_plan::bucket_events (-1: nothing on that stream writes the bucket)
  int ready = -1;                     // index of the last launch that writes into it
};

// The run-time switches of a plan, read from the environment ONCE, by dmm_plan_create, and kept in the plan: the sizing pass
// (create) and the bound pass (bind) cannot disagree about them, and nothing in the library calls getenv while a plan is built or run.
// Each selects between two correct schedules of the same arithmetic and has an A/B test (tests/test_timed_kernels_gpu.py; the host
// harness of tests/test_host_cpu.py drives every one through create -> bind -> run -> destroy under AddressSanitizer):
//   DMM_NO_PACK_TILES=1     weights packed / gradients unpacked by the generic kernels only
//   DMM_NO_HF=1             the head's first convolution on conv3.hip's one-launch path instead of hf.hip
//   DMM_NO_C3_MERGE=1       ... as four launches, one per output parity
//   DMM_NO_CVP_MERGE=1      a decoder ConvTranspose's four parity phases as four launches
//   DMM_NO_WGP_MERGE=1      the head's weight-gradient phases as four launches
//   DMM_NO_TWO_PASS=1       the 5x5 data gradient as one pass + apply_corr instead of two passes
//   DMM_NO_EFF_COMPACT=1    wg3.hip gathers its gradient operand from the block buffers instead of reading conv3.hip's compact copy
//   DMM_NO_S2_INTERLEAVE=1  mid fusion: the second encoder's launch records behind the first's instead of alternating
//   DMM_PACK_CUT=<n>        the forward record in front of which the late layers' weight pack is joined (1 = behind the stem)
//   DMM_DEFER_WGRAD=1       the head's / decoder's multi-tap weight gradients held back until backward reaches the encoder
//   DMM_NO_R1_STATS=1       the BatchNorm-backward sums of the head's norm1 from the reductions-only first pass of the 5x5 data gradient
//                           instead of from the 5x5 weight gradient's factor correlations (wg5.hip, PA = 3)
//   DMM_NO_RAW_STATS=1      the BatchNorm-backward sums of the head's raw-input channels from a data-gradient pass of their own
//                           (round 4) instead of from the weight gradient's factor correlations (wg5.hip, PY = 2)
// Process-wide (capi.cpp, read when the library is loaded; also dmm_set_option): DMM_NO_OVERLAP, DMM_GRAPH, DMM_GRAD_BUCKET_MB;
// diagnostics: DMM_TRACE_DESTROY.  Everything else that used to be an environment switch is a compile-time lab knob (common.h).
struct PlanSwitches {
  bool no_pack_tiles = false, no_hf = false, no_c3_merge = false, no_cvp_merge = false, no_wgp_merge = false, no_two_pass = false,
       no_eff_compact = false, no_s2_interleave = false, defer_wgrad = false, no_raw_stats = false, no_r1_stats = false;
  int pack_cut = 0;  // 0: by weight count
  static PlanSwitches from_environment();
};

struct dmm_plan {
  dmm_model_desc desc;
  PlanSwitches sw;
  std::vector<dmm::TensorInfo> tensors;
  int64_t nparams = 0, nbuf = 0;
  size_t zero_bytes = 0, zero_bwd_bytes = 0, main_bytes = 0;
  double fwd_flops = 0;
  bool bound = false;
  void* ws = nullptr;
  float *params = nullptr, *grads = nullptr, *buffers = nullptr;
  // launch lists (built by bind)
  std::vector<dmm::Op> fwd_train, fwd_eval, bwd;
  // indices of ops whose pointers are patched per call
  std::vector<int> convert_ops_train, convert_ops_eval;
  int logits_op_train = -1, logits_op_eval = -1;
  int bce_op = -1, bce_only_valid = 0;
  dmm::Op bce_only;     // loss + metrics without gradient
  int loss_kind = 0;          // 0 BCE, 1 focal (dmm_plan_set_loss)
  float loss_alpha[8] = {1, 1, 1, 1, 1, 1, 1, 1}, loss_gamma[8] = {2, 2, 2, 2, 2, 2, 2, 2};
  double* metrics = nullptr;  // device, inside the zero region
  size_t metrics_bytes = 0;
  std::vector<dmm::PackDesc> packs;
  std::vector<int> pack_prefix;
  // unpack tables, grouped by gradient bucket
  std::vector<dmm::PackDesc> unpacks;
  std::vector<int> unpack_prefix;
  std::vector<dmm::PackTile> pack_tiles, unpack_tiles;  // tile-kernel work lists (pointwise.h PackTile)
  size_t bucket_bytes = 0;
  std::vector<GradBucket> buckets;       // in the order they become ready
  std::vector<void*> bucket_events;      // hipEvent_t, created on first use
  // profiling: per-op events on the launch stream, one set per recorded pass
  int prof_max_passes = 0;
  int prof_pass[2] = {0, 0};                       // passes recorded for [0] training forward, [1] backward
  std::vector<std::vector<void*>> prof_events[2];  // [which][pass] -> one (start, end) event pair per op
  std::string prof_filter;                         // only ops whose label starts with this are bracketed (empty: all)
  // Streams: the side stream (weight gradients, leaves, the second encoder), the pack stream and the capture stream belong to a
  // PROCESS-LIFETIME pool per device (capi.cpp: DevicePool) - a plan never creates or destroys a stream.  Events (fork / join /
  // bucket / profiling) are taken from that pool's free lists on first use and handed back by dmm_plan_destroy.
  int device = -1;                         // the device the plan was bound on
  bool used_side = false, used_capture = false;
  std::vector<void*> join_events;          // [0] side stream, [1] pack stream
  std::vector<void*> fork_events;
  // hipGraph replay of a launch list (capi.cpp: launch_list): instantiated graphs keyed by the caller's pointers of the call
  struct GraphEntry { const void* key[4]; void* exec; unsigned long long stamp; };
  struct GraphCache {
    std::vector<GraphEntry> entries;
    const void* seen[4][4] = {};   // the caller pointers of the last eager runs (a list is captured when its pointers come back)
    int nseen = 0;
    int captures = 0;
    unsigned long long epoch = 0;  // option epoch the entries were captured under (dmm_set_option invalidates them)
  };
  GraphCache graphs[2];          // [0] training forward, [1] loss + backward
  bool graph_failed = false;     // a capture did not work on this plan: stay eager
  bool dp_used = false;          // a data-parallel reducer waits for bucket events: backward stays eager (events must be real)
  unsigned long long graph_clock = 0;
  long long graph_replays[2] = {0, 0};
};
