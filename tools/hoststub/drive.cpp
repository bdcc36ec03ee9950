// Drives the HOST code of libdmmfods_hip.so (plan.cpp, capi.cpp and the launchers of every kernel file, built for the host with
// AddressSanitizer + UBSan against tools/hoststub/fake_hip.cpp) through the life of a plan:
//   create -> bind (workspace = host memory) -> training forward x2 -> loss + backward -> backward from an external gradient ->
//   eval forward -> loss metrics -> a profiled pass -> bucket waits -> destroy
// and checks what a GPU cannot show:  (a) host heap errors in the sizing and the bound pass (ASan), (b) every device pointer of
// every launch record lies inside the workspace or one of the caller's arenas, (c) the bound pass takes exactly the bytes the sizing
// pass reported, (d) the teardown contract: no stream or event is destroyed with unsynchronised work behind it, nothing is destroyed
// twice, and every stream / event made for a plan is either back in the process pool or destroyed when the plan is gone.
// Test infrastructure (tests/test_host_cpu.py runs it); environment switches (DMM_NO_PACK_TILES, ...) come from the caller's environment.
//
//   drive <arch> <dtype> <batch> <H> <W> [repeat]
//   arch: d121e d121m d169m d201m d121n tiny_mid tiny_early tiny_no g8_mid     dtype: f32 f16 bf16
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../dmmfods_amd/csrc/plan.h"

extern "C" long fakehip_launches();
extern "C" long fakehip_violations();
extern "C" long fakehip_live_objects();
extern "C" long fakehip_live_streams();
extern "C" long fakehip_live_events();
extern "C" long fakehip_stream_creates();
extern "C" long fakehip_event_creates();

using namespace dmm;

namespace {
struct Region { const uint8_t* lo; const uint8_t* hi; const char* name; };
std::vector<Region> g_regions;
long g_bad = 0, g_checked = 0;

void chk(const void* p, const char* what, const char* label) {
  if (p == nullptr) return;
  ++g_checked;
  const uint8_t* q = (const uint8_t*)p;
  for (auto& r : g_regions)
    if (q >= r.lo && q < r.hi) return;
  ++g_bad;
  if (g_bad <= 20) fprintf(stderr, "[drive] pointer %s of launch '%s' = %p lies in none of the caller's regions\n", what, label, p);
}
#define CHK(field) chk((const void*)(field), #field, o.label)

void chk_seg(const Seg& s, const Op& o) {
  CHK(s.src); CHK(s.src2); CHK(s.scale); CHK(s.shift); CHK(s.q); CHK(s.r); CHK(s.ql); CHK(s.rl);
}
void chk_conv(const ConvArgs& a, const Op& o) {
  for (int s = 0; s < a.nseg; ++s) chk_seg(a.seg[s], o);
  CHK(a.wpack); CHK(a.out); CHK(a.stat_sum); CHK(a.stat_sq); CHK(a.logits); CHK(a.bx); CHK(a.bscale); CHK(a.bshift); CHK(a.bmean);
  CHK(a.binvstd); CHK(a.red1); CHK(a.red2); CHK(a.eff_out); CHK(a.eq); CHK(a.er);
  for (int p = 0; p < a.nphase; ++p) CHK(a.ph_wpack[p]);
}
void chk_wgrad(const WgradArgs& a, const Op& o) {
  for (int s = 0; s < a.nseg; ++s) chk_seg(a.seg[s], o);
  chk_seg(a.dy, o);
  CHK(a.dpack); CHK(a.part);
  for (int p = 0; p < a.nphase; ++p) CHK(a.ph_dpack[p]);
}
void chk_ops(const std::vector<Op>& ops) {
  for (const Op& o : ops) {
    switch (o.kind) {
      case OP_MEMSET: CHK(o.ms.p); if (o.ms.bytes) chk((const uint8_t*)o.ms.p + o.ms.bytes - 1, "ms.p + bytes - 1", o.label); break;
      case OP_COPY: CHK(o.cp.dst); CHK(o.cp.src); break;
      case OP_CONVERT: CHK(o.cv.src1); CHK(o.cv.src2); CHK(o.cv.dst); CHK(o.cv.stat_sum); CHK(o.cv.stat_sq); break;
      case OP_IGEMM: chk_conv(o.c, o); break;
      case OP_WGRAD: chk_wgrad(o.w, o); break;
      case OP_BW1: case OP_BW1RED: chk_conv(o.b1.c, o); CHK(o.b1.dpack); CHK(o.b1.part); break;
      case OP_BNFIN: CHK(o.bf.sum); CHK(o.bf.sq); CHK(o.bf.gamma); CHK(o.bf.beta); CHK(o.bf.running_mean); CHK(o.bf.running_var);
                     CHK(o.bf.scale); CHK(o.bf.shift); CHK(o.bf.mean); CHK(o.bf.invstd); break;
      case OP_BNBWD: CHK(o.bb.red1); CHK(o.bb.red2); CHK(o.bb.mean); CHK(o.bb.invstd); CHK(o.bb.scale); CHK(o.bb.dgamma); CHK(o.bb.dbeta);
                     CHK(o.bb.qd); CHK(o.bb.rd); CHK(o.bb.q); CHK(o.bb.r); CHK(o.bb.ql); CHK(o.bb.rl); break;
      case OP_POOL: CHK(o.mp.y0); CHK(o.mp.scale); CHK(o.mp.shift); CHK(o.mp.out); CHK(o.mp.argmax); CHK(o.mp.stat_sum); CHK(o.mp.stat_sq); break;
      case OP_POOLBWD: CHK(o.mpb.y0); CHK(o.mpb.scale); CHK(o.mpb.shift); CHK(o.mpb.gpool); CHK(o.mpb.xpool); CHK(o.mpb.q); CHK(o.mpb.r);
                       CHK(o.mpb.ql); CHK(o.mpb.rl); CHK(o.mpb.mean); CHK(o.mpb.invstd); CHK(o.mpb.argmax); CHK(o.mpb.gy0); CHK(o.mpb.red1);
                       CHK(o.mpb.red2); break;
      case OP_BCE: CHK(o.bce.logits); CHK(o.bce.target); CHK(o.bce.dlogits); CHK(o.bce.out); CHK(o.bce.loss_out); CHK(o.bce.dx_out); break;
      case OP_PACK: case OP_UNPACK: CHK(o.pk.descs); CHK(o.pk.prefix); CHK(o.pk.tdescs); CHK(o.pk.tiles); break;
      case OP_APPLYCORR: CHK(o.ac.g); CHK(o.ac.y); CHK(o.ac.q); CHK(o.ac.r); CHK(o.ac.ql); CHK(o.ac.rl); break;
      case OP_FIN64: CHK(o.f64.sbuf); CHK(o.f64.dpack); CHK(o.f64.w); CHK(o.f64.scale); CHK(o.f64.shift); CHK(o.f64.mean); CHK(o.f64.invstd); CHK(o.f64.red1); CHK(o.f64.red2); break;
      case OP_RAWFIN: CHK(o.rf.sbuf); CHK(o.rf.dpack); CHK(o.rf.w); CHK(o.rf.gamma); CHK(o.rf.beta); CHK(o.rf.red1); CHK(o.rf.red2); break;
      case OP_JOIN: break;
      default: ++g_bad; fprintf(stderr, "[drive] unknown op kind %d\n", o.kind);
    }
  }
}
// the pack / unpack descriptor tables as uploaded ("device" memory is host memory here)
void chk_descs(const dmm_plan* p) {
  Op o; snprintf(o.label, sizeof(o.label), "pack tables");
  for (auto& pd : p->packs) { CHK(pd.w); CHK(pd.dst); CHK(pd.gw); CHK(pd.dpack); }
  for (auto& pd : p->unpacks) { CHK(pd.w); CHK(pd.dst); CHK(pd.gw); CHK(pd.dpack); }
}

bool fill_desc(const std::string& arch, dmm_model_desc& d) {
  memset(&d, 0, sizeof(d));
  d.growth_rate = 32; d.num_init_features = 64; d.bn_size = 4; d.num_classes = 3;
  d.stream_1_in_channels = 3; d.loss_scale = 1.0f; d.bn_momentum = 0.1f; d.bn_eps = 1e-5f; d.iou_threshold = 0.7f; d.use_mfma = 1;
  auto cfg = [&](std::initializer_list<int> bc) { d.num_blocks = 0; for (int v : bc) d.block_config[d.num_blocks++] = v; };
  if (arch == "d121e") { cfg({6, 12, 24, 16}); d.concat_before_block_num = 1; d.stream_2_in_channels = 3; }
  else if (arch == "d121n") { cfg({6, 12, 24, 16}); d.concat_before_block_num = 1; d.stream_2_in_channels = 0; }
  else if (arch == "d121m") { cfg({6, 12, 24, 16}); d.concat_before_block_num = 3; d.stream_2_in_channels = 3; }
  else if (arch == "d121m2") { cfg({6, 12, 24, 16}); d.concat_before_block_num = 2; d.stream_2_in_channels = 3; }
  else if (arch == "d169m") { cfg({6, 12, 32, 32}); d.concat_before_block_num = 3; d.stream_2_in_channels = 3; }
  else if (arch == "d201m") { cfg({6, 12, 48, 32}); d.concat_before_block_num = 3; d.stream_2_in_channels = 3; }
  else if (arch == "d161m") { cfg({6, 12, 36, 24}); d.growth_rate = 48; d.num_init_features = 96; d.concat_before_block_num = 3; d.stream_2_in_channels = 3; }
  else if (arch == "tiny_mid") { cfg({2, 2, 2}); d.concat_before_block_num = 2; d.stream_2_in_channels = 3; }       // the A/B tests' net
  else if (arch == "tiny_early") { cfg({2, 2, 2}); d.concat_before_block_num = 1; d.stream_2_in_channels = 3; }
  else if (arch == "tiny_no") { cfg({2, 2, 2}); d.concat_before_block_num = 1; d.stream_2_in_channels = 0; }
  else if (arch == "g8_mid") { cfg({2, 2, 2, 2}); d.growth_rate = 8; d.num_init_features = 16; d.concat_before_block_num = 3; d.stream_2_in_channels = 3; }  // smoke()'s net
  else return false;
  return true;
}

#define MUST(call)                                                                           \
  do {                                                                                       \
    const int rc__ = (call);                                                                 \
    if (rc__ != 0) { fprintf(stderr, "[drive] %s -> %d: %s\n", #call, rc__, dmm_last_error()); return 2; } \
  } while (0)
}  // namespace

static int one_life(const dmm_model_desc& d, int life) {
  dmm_plan* plan = nullptr;
  MUST(dmm_plan_create(&d, &plan));
  const size_t wsb = dmm_plan_workspace_bytes(plan);
  const int64_t np = dmm_plan_num_params(plan), nb = std::max<int64_t>(dmm_plan_num_buffer_elems(plan), 1);
  // 256-byte aligned workspace with nothing mapped... ASan red zones sit on either side of each allocation
  uint8_t* ws = (uint8_t*)aligned_alloc(256, (wsb + 255) / 256 * 256);
  float* params = (float*)malloc(np * 4); float* grads = (float*)malloc(np * 4); float* buffers = (float*)malloc(nb * 4);
  const size_t px = (size_t)d.batch * d.height * d.width;
  float* in1 = (float*)malloc(px * std::max(1, d.stream_1_in_channels) * 4);
  float* in2 = (float*)malloc(px * std::max(1, d.stream_2_in_channels) * 4);
  float* logits = (float*)malloc(px * d.num_classes * 4);
  float* target = (float*)malloc(px * d.num_classes * 4);
  const size_t nmet = 2 * d.num_classes + (size_t)d.batch * 2 * d.num_classes;
  double* metrics = (double*)malloc(nmet * 8);
  g_regions = {{ws, ws + wsb, "workspace"}, {(uint8_t*)params, (uint8_t*)(params + np), "params"}, {(uint8_t*)grads, (uint8_t*)(grads + np), "grads"},
               {(uint8_t*)buffers, (uint8_t*)(buffers + nb), "buffers"}, {(uint8_t*)in1, (uint8_t*)in1 + px * 3 * 4, "in1"},
               {(uint8_t*)in2, (uint8_t*)in2 + px * 3 * 4, "in2"}, {(uint8_t*)logits, (uint8_t*)logits + px * d.num_classes * 4, "logits"},
               {(uint8_t*)target, (uint8_t*)target + px * d.num_classes * 4, "target"}};
  // (a kernel family switched off between sizing and binding must NOT change what the plan reserves: buffers are reserved by shape)
  if (const char* t = getenv("DRIVE_TOGGLE_BETWEEN_CREATE_AND_BIND")) MUST(dmm_set_option(t, 0));
  // (test of the bind-time check: a plan whose switches were tampered with after it was sized reserves other bytes -> DMM_ERR_STATE)
  if (getenv("DRIVE_FLIP_SWITCH_BETWEEN_CREATE_AND_BIND")) plan->sw.no_eff_compact = !plan->sw.no_eff_compact;
  MUST(dmm_plan_bind(plan, ws, wsb, params, grads, buffers));
  const long l0 = fakehip_launches();
  void* st = nullptr;  // the caller's stream: the null stream, as torch's default
  for (int rep = 0; rep < 2; ++rep) {
    MUST(dmm_plan_forward(plan, in1, d.stream_2_in_channels ? in2 : nullptr, logits, 1, st));
    MUST(dmm_plan_loss_backward(plan, logits, target, metrics, st));
  }
  chk_ops(plan->fwd_train); chk_ops(plan->fwd_eval); chk_ops(plan->bwd); chk_descs(plan);
  MUST(dmm_plan_forward(plan, in1, d.stream_2_in_channels ? in2 : nullptr, logits, 1, st));
  MUST(dmm_plan_backward(plan, target /*stands for d(loss)/d(logit)*/, st));
  MUST(dmm_plan_forward(plan, in1, d.stream_2_in_channels ? in2 : nullptr, logits, 0, st));
  MUST(dmm_plan_loss_metrics(plan, logits, target, metrics, st));
  // the data-parallel hooks: wait for every bucket on the caller's stream
  for (int b = 0; b < dmm_plan_num_grad_buckets(plan); ++b) MUST(dmm_plan_grad_bucket_wait(plan, b, st));
  // a profiled pass (per-op events), then one bracketing a single class as bench.py does in its timed region
  MUST(dmm_plan_profile_begin(plan, 1));
  MUST(dmm_plan_forward(plan, in1, d.stream_2_in_channels ? in2 : nullptr, logits, 1, st));
  MUST(dmm_plan_loss_backward(plan, logits, target, metrics, st));
  {
    std::vector<double> ms(dmm_plan_profile_num_ops(plan, 1));
    int passes = 0;
    MUST(dmm_plan_profile_collect(plan, 1, ms.data(), (int)ms.size(), &passes));
  }
  MUST(dmm_plan_profile_filter(plan, "bw1."));
  MUST(dmm_plan_profile_begin(plan, 1));
  MUST(dmm_plan_forward(plan, in1, d.stream_2_in_channels ? in2 : nullptr, logits, 1, st));
  MUST(dmm_plan_loss_backward(plan, logits, target, metrics, st));
  MUST(dmm_plan_profile_begin(plan, 0));
  // one more step WITHOUT synchronising anything, then destroy: teardown must not rely on the caller having drained the device
  MUST(dmm_plan_forward(plan, in1, d.stream_2_in_channels ? in2 : nullptr, logits, 1, st));
  MUST(dmm_plan_loss_backward(plan, logits, target, metrics, st));
  const long launches = fakehip_launches() - l0;
  const size_t nf = plan->fwd_train.size(), nbw = plan->bwd.size();
  MUST(dmm_plan_destroy(plan));
  free(ws); free(params); free(grads); free(buffers); free(in1); free(in2); free(logits); free(target); free(metrics);
  printf("life %d: workspace %.1f MiB, %zu + %zu launch records, %ld launches, %ld pointers checked, %ld bad, streams alive %ld (created %ld), events alive %ld (created %ld), violations %ld\n",
         life, wsb / 1048576.0, nf, nbw, launches, g_checked, g_bad, fakehip_live_streams(), fakehip_stream_creates(), fakehip_live_events(),
         fakehip_event_creates(), fakehip_violations());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: drive <arch> <dtype> <batch> <H> <W> [lives]\n"); return 64; }
  dmm_model_desc d;
  if (!fill_desc(argv[1], d)) { fprintf(stderr, "unknown arch %s\n", argv[1]); return 64; }
  const std::string dt = argv[2];
  d.dtype = dt == "f32" ? DMM_F32 : (dt == "f16" ? DMM_F16 : DMM_BF16);
  d.batch = atoi(argv[3]); d.height = atoi(argv[4]); d.width = atoi(argv[5]);
  const int lives = argc > 6 ? atoi(argv[6]) : 2;
  long streams_after_first = -1, events_after_first = -1;
  for (int life = 0; life < lives; ++life) {
    const int rc = one_life(d, life);
    if (rc) return rc;
    // the pool: what the first plan made is what every later plan uses - stream and event counts must not grow with the plans
    if (life == 0) { streams_after_first = fakehip_stream_creates(); events_after_first = fakehip_event_creates(); }
  }
  int rc = 0;
  if (g_bad) { fprintf(stderr, "[drive] FAIL: %ld device pointers outside the caller's regions\n", g_bad); rc = 1; }
  if (fakehip_violations()) { fprintf(stderr, "[drive] FAIL: %ld teardown / handle violations\n", fakehip_violations()); rc = 1; }
  if (lives > 1 && fakehip_stream_creates() != streams_after_first) {
    fprintf(stderr, "[drive] FAIL: streams are created per plan (%ld after the first plan, %ld after %d)\n", streams_after_first, fakehip_stream_creates(), lives);
    rc = 1;
  }
  if (lives > 1 && fakehip_event_creates() != events_after_first) {
    fprintf(stderr, "[drive] FAIL: events are created per plan (%ld after the first plan, %ld after %d)\n", events_after_first, fakehip_event_creates(), lives);
    rc = 1;
  }
  printf("%s\n", rc ? "DRIVE FAILED" : "DRIVE OK");
  return rc;
}
