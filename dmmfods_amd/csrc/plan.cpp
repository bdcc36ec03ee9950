// Plan builder for the Dense_U_Net_lidar training step on MI355X.
//
// Follows the reference topology  dmmfods/graphs/models/Dense_U_Net_lidar.py:29-267  (encoder = DenseNet without
// norm5/classifier, optional second stream, mid-fusion concat module, U-Net decoder, heat-map head) but lays it out
// for NHWC implicit-GEMM kernels:
//   * every dense block lives in ONE preallocated buffer; a layer's 3x3 conv writes its growth channels in place, so
//     torch.cat inside blocks disappears; blocks that feed a decoder skip carry the decoder's ConvTranspose output in
//     a front region of the same buffer, so the decoder concat disappears too;
//   * BatchNorm batch statistics are reduced in the producing kernel's epilogue (fp64 accumulators), BN+ReLU is
//     applied in the consuming kernel's operand gather; nothing normalised is ever materialised;
//   * transitions pool BEFORE their 1x1 conv (both are linear; 4x fewer FLOPs), nearest upsampling is index math;
//   * BatchNorm backward is deferred: the dgrad epilogue stores s*dz and reduces sum(dz), sum(dz*x); the remaining
//     per-channel affine term (q + r*x) is applied by whoever reads that gradient next.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>

namespace dmm {

// ------------------------------------------------------------------------------------------------ taps
static unsigned tw1(int mt) { return 0xffffff00u | (unsigned)mt; }

std::vector<Tap> taps_conv(int R, int S, int pad) {
  std::vector<Tap> t;
  for (int ky = 0; ky < R; ++ky)
    for (int kx = 0; kx < S; ++kx) t.push_back({ky - pad, kx - pad, tw1(ky * S + kx)});
  return t;
}
std::vector<Tap> taps_conv_dgrad(int R, int S, int pad) {
  std::vector<Tap> t;
  for (int ky = 0; ky < R; ++ky)
    for (int kx = 0; kx < S; ++kx) t.push_back({pad - ky, pad - kx, tw1(ky * S + kx)});
  return t;
}
// out[2i - 1 + k] += in[i] * w[k]:  even outputs use k = 1 (i = o/2), odd outputs k = 0 (i = (o+1)/2) and k = 2 (i = (o-1)/2)
std::vector<Tap> taps_convT_phase(int py, int px) {
  std::vector<std::pair<int, int>> ys, xs;  // (k, d)
  if (py == 0) ys = {{1, 0}}; else ys = {{0, 1}, {2, 0}};
  if (px == 0) xs = {{1, 0}}; else xs = {{0, 1}, {2, 0}};
  std::vector<Tap> t;
  for (auto& y : ys)
    for (auto& x : xs) t.push_back({y.second, x.second, tw1(y.first * 3 + x.first)});
  return t;
}
std::vector<Tap> taps_convT_dgrad() {
  std::vector<Tap> t;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) t.push_back({ky - 1, kx - 1, tw1(ky * 3 + kx)});
  return t;
}
// d a[z] = sum_{e in {0,1}} sum_ky dy[2z + e - ky + 1] w[ky]; offset o = e - ky + 1 in {-1,0,1,2} collects
// ky sets {2}, {1,2}, {0,1}, {0}.
std::vector<Tap> taps_up2_merged_dgrad() {
  static const int sets[4][2] = {{2, -1}, {1, 2}, {0, 1}, {0, -1}};
  std::vector<Tap> t;
  for (int oy = 0; oy < 4; ++oy)
    for (int ox = 0; ox < 4; ++ox) {
      unsigned w = 0;
      int cnt = 0;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
          const int ky = sets[oy][a], kx = sets[ox][b];
          if (ky < 0 || kx < 0) continue;
          w |= (unsigned)(ky * 3 + kx) << (8 * cnt++);
        }
      for (; cnt < 4; ++cnt) w |= 0xffu << (8 * cnt);
      t.push_back({oy - 1, ox - 1, w});
    }
  return t;
}

// Forward 3x3 (pad 1) conv over a nearest-x2 upsampled source, restricted to output pixels (2y+e, 2x+f): the source row of
// tap ky is y + floor((e + ky - 1) / 2), so e = 0 collects {ky=0} at -1 and {1,2} at 0; e = 1 collects {0,1} at 0 and {2} at +1.
std::vector<Tap> taps_up2_phase(int e, int f) {
  struct G { int off; int k[2]; };
  static const G rows[2][2] = {{{-1, {0, -1}}, {0, {1, 2}}}, {{0, {0, 1}}, {1, {2, -1}}}};
  std::vector<Tap> t;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const G &gy = rows[e][a], &gx = rows[f][b];
      unsigned w = 0;
      int cnt = 0;
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          if (gy.k[i] < 0 || gx.k[j] < 0) continue;
          w |= (unsigned)(gy.k[i] * 3 + gx.k[j]) << (8 * cnt++);
        }
      for (; cnt < 4; ++cnt) w |= 0xffu << (8 * cnt);
      t.push_back({gy.off, gx.off, w});
    }
  return t;
}
// The same output pixels read a full-resolution source at (2y + e + ky - 1, 2x + f + kx - 1): stride-2 rows, taps offset by (e,f).
std::vector<Tap> taps_conv_phase_s2(int e, int f) {
  std::vector<Tap> t;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) t.push_back({e + ky - 1, f + kx - 1, tw1(ky * 3 + kx)});
  return t;
}

void fill_seg_taps(Seg& s, const std::vector<Tap>& taps, int BK) {
  if ((int)taps.size() > MAX_TAPS) throw std::runtime_error("too many taps");
  s.ntaps = (int)taps.size();
  for (int i = 0; i < s.ntaps; ++i) s.taps[i] = (short)((taps[i].dy & 0xff) | ((taps[i].dx & 0xff) << 8));
  s.nchunks = (s.ntaps * s.Cpad + BK - 1) / BK;
}
void fill_pack_seg(PackSeg& p, const std::vector<Tap>& taps, int Creal, int Cpad, int koff, int BK) {
  p.Creal = Creal;
  p.Cpad = Cpad;
  p.koff = koff;
  p.ntaps = (int)taps.size();
  p.nchunks = (p.ntaps * Cpad + BK - 1) / BK;
  for (int i = 0; i < p.ntaps; ++i) p.tapw[i] = taps[i].tapw;
}

// ------------------------------------------------------------------------------------------------ state_dict table
namespace {

struct TableBuilder {
  std::vector<TensorInfo>& out;
  int64_t np = 0, nb = 0;
  void add(const std::string& name, int kind, std::initializer_list<int64_t> shape) {
    TensorInfo t;
    t.name = name;
    t.kind = kind;
    t.ndim = (int)shape.size();
    int64_t n = 1;
    int i = 0;
    for (int k = 0; k < 4; ++k) t.shape[k] = 0;
    for (auto s : shape) { t.shape[i++] = s; n *= s; }
    if (kind <= DMM_T_BN_BIAS) { t.off = np; np += n; }
    else if (kind <= DMM_T_BN_VAR) { t.off = nb; nb += n; }
    else t.off = -1;
    out.push_back(t);
  }
  void bn(const std::string& p, int64_t c) {
    add(p + ".weight", DMM_T_BN_WEIGHT, {c});
    add(p + ".bias", DMM_T_BN_BIAS, {c});
    add(p + ".running_mean", DMM_T_BN_MEAN, {c});
    add(p + ".running_var", DMM_T_BN_VAR, {c});
    add(p + ".num_batches_tracked", DMM_T_BN_TRACKED, {});
  }
};

struct Geo {  // channel algebra
  int nb, k, bs, nif, s1, s2, nc, cbb;
  int fusion;  // 0 no, 1 early, 2 mid
  int net_in;
  std::vector<int> L, cin, cout, nin, nf;
  explicit Geo(const dmm_model_desc& d) {
    nb = d.num_blocks; k = d.growth_rate; bs = d.bn_size; nif = d.num_init_features;
    s1 = d.stream_1_in_channels; s2 = d.stream_2_in_channels; nc = d.num_classes; cbb = d.concat_before_block_num;
    if (cbb == 1 && s2 == 0) fusion = 0;
    else if (cbb == 1 && s2 > 0) fusion = 1;
    else if (cbb > 1 && cbb <= nb) fusion = 2;
    else throw std::invalid_argument("invalid fusion configuration (AttributeError in the reference, M:65)");
    net_in = s1 + (fusion == 1 ? s2 : 0);
    int c = nif;
    for (int i = 0; i < nb; ++i) {
      L.push_back(d.block_config[i]);
      cin.push_back(c);
      c += L[i] * k;
      cout.push_back(c);
      if (i != nb - 1) c /= 2;
    }
    std::vector<int> stack;
    stack.push_back(nif + 2 * k);
    for (int i = 0; i < nb; ++i) stack.push_back(cout[i]);
    int num_in = stack.back(); stack.pop_back();
    for (int i = 0; i < nb; ++i) {
      const int f = stack.back(); stack.pop_back();
      nin.push_back(num_in);
      nf.push_back(f);
      num_in = 2 * f;
    }
  }
};

void encoder_table(TableBuilder& tb, const std::string& p, const Geo& g, int in_ch, int upto) {
  tb.add(p + ".conv0.weight", DMM_T_CONV, {g.nif, in_ch, 7, 7});
  tb.bn(p + ".norm0", g.nif);
  for (int b = 0; b < g.nb && b < upto; ++b) {
    for (int l = 0; l < g.L[b]; ++l) {
      const std::string q = p + ".denseblock" + std::to_string(b + 1) + ".denselayer" + std::to_string(l + 1);
      const int c = g.cin[b] + l * g.k;
      tb.bn(q + ".norm1", c);
      tb.add(q + ".conv1.weight", DMM_T_CONV, {g.bs * g.k, c, 1, 1});
      tb.bn(q + ".norm2", g.bs * g.k);
      tb.add(q + ".conv2.weight", DMM_T_CONV, {g.k, g.bs * g.k, 3, 3});
    }
    if (b != g.nb - 1) {
      const std::string q = p + ".transition" + std::to_string(b + 1);
      tb.bn(q + ".norm", g.cout[b]);
      tb.add(q + ".conv.weight", DMM_T_CONV, {g.cout[b] / 2, g.cout[b], 1, 1});
    }
  }
}

}  // namespace

static void build_tensor_table(const dmm_model_desc& d, std::vector<TensorInfo>& out, int64_t& np, int64_t& nbuf) {
  Geo g(d);
  TableBuilder tb{out};
  encoder_table(tb, "features", g, g.net_in, g.nb);
  for (int j = 0; j < g.nb; ++j) {
    const std::string p = "decoder.Transposed_Convolution_Sequence_" + std::to_string(j + 1);
    tb.bn(p + ".norm0", g.nin[j]);
    tb.add(p + ".conv_reduce.weight", DMM_T_CONV, {g.nf[j], g.nin[j], 1, 1});
    tb.bn(p + ".norm1", g.nf[j]);
    tb.add("decoder.Transposed_Convolution_" + std::to_string(j + 1) + ".weight", DMM_T_CONVT, {g.nf[j], g.nf[j], 3, 3});
  }
  const int nfl = g.nf[g.nb - 1], hin = nfl + g.s1 + g.s2;
  tb.bn("dec_out_to_heat_maps.norm0", hin);
  tb.add("dec_out_to_heat_maps.refine0.weight", DMM_T_CONV, {nfl / 2, hin, 3, 3});
  tb.bn("dec_out_to_heat_maps.norm1", nfl / 2);
  tb.add("dec_out_to_heat_maps.refine1.weight", DMM_T_CONV, {g.nc, nfl / 2, 5, 5});
  if (g.fusion == 2) {
    encoder_table(tb, "stream_2_features", g, g.s2, g.cbb - 1);
    const int c = g.cin[g.cbb - 1];
    tb.bn("concat_module.norm", 2 * c);
    tb.add("concat_module.conv.weight", DMM_T_CONV, {c, 2 * c, 1, 1});
  }
  np = tb.np;
  nbuf = tb.nb;
}

// ------------------------------------------------------------------------------------------------ builder
namespace {

struct Bump {
  size_t off = 0;
  size_t take(size_t bytes, size_t align = 256) {
    off = (off + align - 1) / align * align;
    const size_t o = off;
    off += bytes;
    return o;
  }
};

struct Buf {
  int B, H, W, ld;
  int cw = 0;             // logical channel count (ld may carry padding)
  uint8_t* x = nullptr;   // activations (T)
  uint8_t* g = nullptr;   // raw gradient (T), same layout
  double* ssum = nullptr; // per-channel sum / sum^2 of x
  double* ssq = nullptr;
  float* q = nullptr;     // deferred BN-backward correction of g: float hi parts ...
  float* r = nullptr;
  float* ql = nullptr;    // ... lo parts ...
  float* rl = nullptr;
  double* qd = nullptr;   // ... and the fp64 accumulators they are split from
  double* rd = nullptr;
  bool ginit = false;
  bool matz = false;          // single-consumer tensor whose gradient is re-gathered often: materialise q + r*x once
  bool materialized = false;  // (set during backward emission) gradient already holds the effective gradient
  int front = 0;              // leading channels written by a ConvTranspose (the decoder concat): one BatchNorm consumer, re-gathered by that
                              // ConvTranspose's data gradient (72x per element) and weight gradients -> their correction is applied once
  int mat_front = 0;          // (set during backward emission) channels [0, mat_front) already hold the effective gradient
};

struct BnRange { int buf, ch0, c0, n; double count, count_unb; bool want_qr; };
struct Bn {
  int C;
  float *gamma, *beta, *rm, *rv, *dgamma, *dbeta;
  float *scale, *shift, *mean, *invstd;
  double *red1, *red2;
  int cp = 0;  // allocated channels = replica stride of red1 / red2
  std::vector<BnRange> ranges;
};

enum DgradKind { DG_NONE = 0, DG_FLIP, DG_POOL2, DG_CONVT, DG_UP2 };

struct SegRec {
  int buf, ch0, C /*storage channels*/, Cw /*master channels*/, koff, mode, istride;
  int bn, bn_c0;
  int dgrad;
};
struct PhaseRec { int py, px; std::vector<Tap> taps; int pack; std::vector<Tap> taps1; /* segment 1, if different */ };
struct ConvRec {
  std::string wname;
  bool transposed;
  int N, Kin, R, S, pad;
  int B, Ho, Wo;
  int nseg;
  SegRec seg[2];
  int obuf, och0, ostride;
  int epi;
  bool stats;
  std::vector<PhaseRec> phases;
  int dpack[2];
  bool wgrad_transposed;  // taps on the gradient side (see wgrad.hip)
  bool shared_master;     // several phases' packed gradients add into the same master weights
  double flops_ref;       // 2*MACs in the reference's formulation when it differs from the launched geometry (else 0)
};
struct PoolRec { int y0buf, bn, obuf, och0; uint8_t* argmax; int C; };
struct Rec { int type; int idx; };  // 0 conv, 1 pool

struct Builder {
  dmm_plan& P;
  const dmm_model_desc& d;
  Geo g;
  const int dtype, esz, SLOT, BK;
  const bool sizing;
  uint8_t *zbase, *zbbase, *wbase;
  float *Pp, *Pg, *Pb;  // bases of the parameter / gradient / buffer arenas (see the constructor)
  Bump Z, ZB, W;  // Z: forward accumulators (BN statistics); ZB: backward accumulators (reductions, q/r, packed dW, metrics)
  std::map<std::string, const TensorInfo*> tmap;
  std::vector<Buf> bufs;
  std::vector<Bn> bns;
  std::vector<ConvRec> convs;
  std::vector<PoolRec> pools;
  std::vector<Rec> recs;
  std::vector<Op>* ops = nullptr;
  bool training = true;
  int in1 = -1, in2 = -1, inH = -1, dl = -1;
  double flops = 0;

  Builder(dmm_plan& p, bool sizing_, uint8_t* ws)
      : P(p), d(p.desc), g(p.desc), dtype(p.desc.dtype), esz((int)dtype_size(p.desc.dtype)), SLOT(16 / esz), BK(4 * (16 / esz)),
        sizing(sizing_) {
    // The sizing pass (dmm_plan_create: no workspace, no arenas yet) walks the SAME code as the bound pass and must take the same
    // decisions: several of them test a pointer for null (a buffer without gradient, a BatchNorm without statistics, a descriptor
    // without a master gradient).  With null bases the FIRST allocation of every region and the tensor at arena offset 0 looked null
    // in the sizing pass only (the tile count of the unpack table could then differ between the passes, and with it
    // every later workspace offset).  So the sizing pass gets distinct, never dereferenced, non-null bases; plan_bind compares the
    // bytes the bound pass took with the sized ones and refuses to bind on any difference.
    auto fake = [](int k) { return (uint8_t*)(uintptr_t)((0x100ull + (uintptr_t)k) << 40); };
    zbase = sizing ? fake(0) : ws;
    zbbase = sizing ? fake(1) : ws + p.zero_bytes;
    wbase = sizing ? fake(2) : ws + p.zero_bytes + p.zero_bwd_bytes;
    Pp = sizing ? (float*)fake(3) : p.params;
    Pg = sizing ? (float*)fake(4) : p.grads;
    Pb = sizing ? (float*)fake(5) : p.buffers;
    for (auto& t : P.tensors) tmap[t.name] = &t;
  }

  template <typename U> U* zptr(size_t n) { return (U*)((uintptr_t)zbase + Z.take(n * sizeof(U))); }
  template <typename U> U* zbptr(size_t n) { return (U*)((uintptr_t)zbbase + ZB.take(n * sizeof(U))); }
  template <typename U> U* wptr(size_t n) { return (U*)((uintptr_t)wbase + W.take(n * sizeof(U))); }
  int rup(int v, int m) const { return (v + m - 1) / m * m; }

  const TensorInfo& T(const std::string& n) {
    auto it = tmap.find(n);
    if (it == tmap.end()) throw std::runtime_error("unknown tensor " + n);
    return *it->second;
  }

  int new_buf(int B, int H, int W_, int cw, bool grad, bool stats, bool matz = false) {
    Buf b;
    b.matz = matz && !lab_flag("DMM_NO_MATZ");
    // Pixel pitch: power-of-two pitches make every workgroup hit the same HBM channels at the same time (all of them read
    // the same 64-byte column window of their rows as they walk K in step), so wide buffers get an odd multiple of 64 B.
    int ld = cw;
    if (pad_pitch && cw >= 64 && (cw * esz) % 256 == 0) ld = cw + 64 / esz;
    b.B = B; b.H = H; b.W = W_; b.ld = ld; b.cw = cw;
    const size_t n = (size_t)B * H * W_ * ld;
    b.x = wptr<uint8_t>(n * esz);
    if (grad) b.g = wptr<uint8_t>(n * esz);
    if (stats) { b.ssum = zptr<double>((size_t)ld * STAT_REPS); b.ssq = zptr<double>((size_t)ld * STAT_REPS); }  // replica stride = ld
    if (grad) {
      b.q = zbptr<float>(ld + 8); b.r = zbptr<float>(ld + 8); b.ql = zbptr<float>(ld + 8); b.rl = zbptr<float>(ld + 8);
      b.qd = zbptr<double>(ld + 8); b.rd = zbptr<double>(ld + 8);
    }
    bufs.push_back(b);
    return (int)bufs.size() - 1;
  }

  int new_bn(const std::string& prefix, int C) {
    Bn b;
    b.C = C;
    const TensorInfo &w = T(prefix + ".weight"), &bi = T(prefix + ".bias"), &rm = T(prefix + ".running_mean"),
                     &rv = T(prefix + ".running_var");
    b.gamma = Pp + w.off; b.beta = Pp + bi.off;
    b.dgamma = Pg + w.off; b.dbeta = Pg + bi.off;
    b.rm = Pb + rm.off; b.rv = Pb + rv.off;
    const int cp = rup(C, 8) + 8;
    b.scale = wptr<float>(cp); b.shift = wptr<float>(cp); b.mean = wptr<float>(cp); b.invstd = wptr<float>(cp);
    b.red1 = zbptr<double>((size_t)cp * STAT_REPS); b.red2 = zbptr<double>((size_t)cp * STAT_REPS);  // replica stride = cp
    b.cp = cp;
    bns.push_back(b);
    return (int)bns.size() - 1;
  }
  void bn_range(int bn, int buf, int ch0, int c0, int n, double unb_mult = 1.0, bool want_qr = true) {
    const Buf& b = bufs[buf];
    const double cnt = (double)b.B * b.H * b.W;
    bns[bn].ranges.push_back({buf, ch0, c0, n, cnt, cnt * unb_mult, want_qr});
  }

  // -------------------------------------------------------------------------------- pack registry
  int add_pack(const ConvRec& c, const std::vector<Tap>& taps, bool dgrad_seg, int seg_index, const std::vector<Tap>* taps1 = nullptr) {
    PackDesc pd;
    memset(&pd, 0, sizeof(pd));
    const TensorInfo& w = T(c.wname);
    pd.w = Pp + w.off;
    const long long RS = (long long)c.R * c.S;
    long long base_off = 0;
    if (!dgrad_seg) {
      pd.N = c.N;
      pd.nseg = c.nseg;
      if (!c.transposed) { pd.sn = c.Kin * RS; pd.sk = RS; }
      else { pd.sn = RS; pd.sk = (long long)c.N * RS; }
      for (int s = 0; s < c.nseg; ++s)
        fill_pack_seg(pd.seg[s], (s == 1 && taps1 && !taps1->empty()) ? *taps1 : taps, c.seg[s].Cw, c.seg[s].C, c.seg[s].koff, BK);
      pd.gw = Pg + w.off;
      pd.shared_master = c.shared_master ? 1 : 0;
    } else {
      const SegRec& sr = c.seg[seg_index];
      pd.N = sr.Cw;
      pd.nseg = 1;
      if (!c.transposed) { pd.sn = RS; pd.sk = c.Kin * RS; base_off = sr.koff * RS; }
      else { pd.sn = (long long)c.N * RS; pd.sk = RS; }
      const int kc = rup(c.N, 8);
      fill_pack_seg(pd.seg[0], taps, c.N, kc, 0, BK);
    }
    pd.st = 1;
    pd.rs = (int)RS;
    pd.w += base_off;
    {
      const int cs = dgrad_seg ? c.seg[seg_index].C : 0;
      // data-gradient output width: 128-column tiles once the padding waste is <= 25 %, else 64, else 32
      pd.Npad = dgrad_seg ? rup(cs, cs >= 384 || cs % 128 == 0 ? 128 : (cs >= 64 ? 64 : 32)) : rup(c.N, 32);
    }
    int chunks = 0;
    for (int s = 0; s < pd.nseg; ++s) chunks += pd.seg[s].nchunks;
    const size_t elems = (size_t)chunks * pd.Npad * BK;
    pd.dst = wptr<uint8_t>(elems * esz);
    if (!dgrad_seg) pd.dpack = zbptr<float>(elems);
    if (!dgrad_seg && pd.gw) pd.gw += 0;
    P.packs.push_back(pd);
    return (int)P.packs.size() - 1;
  }

  // -------------------------------------------------------------------------------- op emission
  bool leaf_scope = false;  // ops emitted now feed only parameter gradients (stem / raw-input branches)
  float* wg3_part = nullptr;  // wg3.hip's slots, shared by every launch of the family
  bool wg3_part_taken = false;
  const bool front_matz = !lab_flag("DMM_NO_FRONT_MATZ");  // lab knob
  // which multi-consumer gradients are materialised (q + r*y applied once by applycorr) instead of corrected by every consumer's
  // prologue: 1 the decoder's conv_reduce outputs, 2 the last ConvTranspose's output, 4 refine0's output (A/B knob)
  const int matz_mask = lab_int("DMM_MATZ_MASK", 7);
  // Deferred weight gradients (round 4).  The head's and the decoder's multi-tap weight gradients (wgp / wg5: 4 ms of work alone)
  // are the FIRST things backward can start on the weight-gradient stream - and they then run beside the head's and the decoder's
  // data gradients, the heaviest stretch of the main chain: measured with the launches skipped, they cost 3.2 ms of a 27.8 ms step,
  // i.e. they were hardly hidden at all, while the encoder's weight gradients (3.6 ms alone) cost 1.2 ms.  Nothing needs them before
  // their bucket's unpack, so they CAN be held back and enter the list where the main chain reaches the encoder (DMM_DEFER_WGRAD=1;
  // DMM_DEFER_AT=<substring of a weight name> moves the point): there the main chain is a sequence of small launches (blocks 4-3).
  // Their operands are final by then and stay so: a weight gradient reads forward activations and the gradient of its convolution's
  // OUTPUT, which nothing writes after that convolution's own backward has been emitted.
  // MEASURED (C2, one box): 27.71 ms in the old order, 28.14 deferred to the first encoder convolution, 28.65-28.77 deferred further
  // (transition 3, block 3, block 2).  The chip is work-conserving: the deferred launches cost their 3 ms wherever they run and the
  // encoder's own weight gradients then pile up behind them at the end of backward.  OFF by default; kept as a switch and a test.
  bool defer_scope = false;
  std::vector<Op> deferred_ops;
  std::vector<const ConvRec*> deferred_convs;
  Op& push(int kind) {
    if (defer_scope && kind == OP_WGRAD) {
      deferred_ops.reserve(256);   // (references returned earlier stay valid: one reservation, far above the ~30 launches held)
      deferred_ops.emplace_back();
      deferred_ops.back().kind = kind;
      deferred_ops.back().leaf = leaf_scope ? 1 : 0;
      return deferred_ops.back();
    }
    ops->emplace_back();
    ops->back().kind = kind;
    ops->back().leaf = leaf_scope ? 1 : 0;
    return ops->back();
  }
  void flush_deferred() {
    if (deferred_ops.empty() && deferred_convs.empty()) return;
    for (auto& o : deferred_ops) ops->push_back(o);
    deferred_ops.clear();
    const bool keep = defer_scope;
    defer_scope = false;
    for (const ConvRec* c : deferred_convs) conv_grad_done(*c);
    deferred_convs.clear();
    defer_scope = keep;
  }
  bool is_raw_input(int buf) const { return buf == in1 || buf == in2 || buf == inH; }
  void tag(Op& o, const char* cls, const std::string& layer, double flops, double bytes) {
    snprintf(o.label, sizeof(o.label), "%s/%s", cls, layer.c_str());
    o.flops = flops;
    o.bytes = bytes;
  }
  static const char* ncls(const char* base, int npad, char* buf) {
    snprintf(buf, 32, "%s.n%d", base, npad % 128 == 0 ? 128 : (npad % 64 == 0 ? 64 : 32));
    return buf;
  }
  std::string short_name(const std::string& w) const {
    std::string s = w;
    const char* cuts[] = {"features.", "stream_2_features.", "decoder.", "dec_out_to_heat_maps.", ".weight", "denseblock", "denselayer", "Transposed_Convolution"};
    const char* reps[] = {"f.", "s2.", "d.", "h.", "", "b", "l", "TC"};
    for (int i = 0; i < 8; ++i) {
      size_t p;
      while ((p = s.find(cuts[i])) != std::string::npos) s.replace(p, strlen(cuts[i]), reps[i]);
    }
    return s;
  }
  double conv_flops(const ConvRec& c, size_t nphases) const {
    if (c.flops_ref > 0) return c.flops_ref / (double)nphases;
    const double px = (double)c.B * c.Ho * c.Wo * (c.seg[0].mode == G_POOL2 ? 4.0 : 1.0);
    const double f = c.transposed ? 2.0 * px * c.N * c.Kin * 9.0 : 2.0 * px * c.N * c.Kin * c.R * c.S;
    return f / (double)nphases;
  }
  double src_bytes(const ConvRec& c) const {
    double b = 0;
    for (int s = 0; s < c.nseg; ++s) {
      const Buf& sb = bufs[c.seg[s].buf];
      b += (double)sb.B * sb.H * sb.W * c.seg[s].C * esz;
    }
    return b;
  }
  double out_bytes(const ConvRec& c) const {
    const Buf& ob = bufs[c.obuf];
    return (double)ob.B * ob.H * ob.W * rup(c.N, 8) * (c.epi == EPI_LOGITS ? 4 : esz);
  }
  double w_bytes(const ConvRec& c) const { return (double)c.N * c.Kin * c.R * c.S * esz; }
  const uint8_t* xat(int buf, int ch) const { return bufs[buf].x + (size_t)ch * esz; }
  uint8_t* gat(int buf, int ch) const { return bufs[buf].g ? bufs[buf].g + (size_t)ch * esz : nullptr; }

  void emit_bn_finalize(int bn) {
    Bn& b = bns[bn];
    for (auto& rg : b.ranges) {
      Op& o = push(OP_BNFIN);
      BnFinalizeArgs& a = o.bf;
      const Buf& sb = bufs[rg.buf];
      a.sum = sb.ssum ? sb.ssum + rg.ch0 : nullptr;
      a.sq = sb.ssq ? sb.ssq + rg.ch0 : nullptr;
      a.stat_stride = sb.ld;
      a.count = rg.count;
      a.count_unbiased = rg.count_unb;
      a.gamma = b.gamma + rg.c0; a.beta = b.beta + rg.c0;
      a.running_mean = b.rm + rg.c0; a.running_var = b.rv + rg.c0;
      a.scale = b.scale + rg.c0; a.shift = b.shift + rg.c0; a.mean = b.mean + rg.c0; a.invstd = b.invstd + rg.c0;
      a.C = rg.n;
      a.training = training ? 1 : 0;
      a.momentum = d.bn_momentum;
      a.eps = d.bn_eps;
    }
  }

  void fill_fwd_seg(Seg& s, const SegRec& sr, const std::vector<Tap>& taps) {
    memset(&s, 0, sizeof(s));
    const Buf& b = bufs[sr.buf];
    s.src = xat(sr.buf, sr.ch0);
    s.ld = b.ld;
    s.Hs = b.H; s.Ws = b.W;
    s.C = sr.C; s.Cpad = sr.C;
    s.mode = sr.mode;
    s.istride = sr.istride;
    if (sr.bn >= 0) { s.scale = bns[sr.bn].scale + sr.bn_c0; s.shift = bns[sr.bn].shift + sr.bn_c0; }
    fill_seg_taps(s, taps, BK);
  }

  void emit_conv_fwd(ConvRec& c) {
    int done[2] = {-1, -1};
    for (int s = 0; s < c.nseg; ++s) {
      const int bn = c.seg[s].bn;
      if (bn < 0 || bn == done[0]) continue;
      emit_bn_finalize(bn);
      done[s] = bn;
    }
    const Buf& ob = bufs[c.obuf];
    for (auto& ph : c.phases) {
      Op& o = push(OP_IGEMM);
      o.epi = c.epi;
      ConvArgs& a = o.c;
      memset(&a, 0, sizeof(a));
      a.nseg = c.nseg;
      for (int s = 0; s < c.nseg; ++s) fill_fwd_seg(a.seg[s], c.seg[s], (s == 1 && !ph.taps1.empty()) ? ph.taps1 : ph.taps);
      a.B = c.B; a.Ho = c.Ho; a.Wo = c.Wo; a.M = c.B * c.Ho * c.Wo;
      const PackDesc& pd = P.packs[ph.pack];
      a.wpack = pd.dst;
      a.N = c.N; a.Npad = pd.Npad;
      a.out = (void*)xat(c.obuf, c.och0);
      a.ldo = ob.ld; a.Hout = ob.H; a.Wout = ob.W;
      a.ostride = c.ostride; a.py = ph.py; a.px = ph.px;
      if (c.stats && ob.ssum) { a.stat_sum = ob.ssum + c.och0; a.stat_sq = ob.ssq + c.och0; a.stat_stride = ob.ld; }
      {
        char cb[32];
        const double np = (double)c.phases.size();
        o.impl = igemm_pick(a, dtype, c.epi, d.use_mfma != 0);
        const bool c3 = o.impl == IMPL_CONV3, cp = o.impl == IMPL_CVP;
        const char* lcls = o.impl == IMPL_THIN ? "thin.logits" : (o.impl == IMPL_HALO ? "halo.logits" : "igemm.logits");
        tag(o, ncls(c.epi == EPI_LOGITS ? lcls : (c3 ? "conv3.store" : (cp ? "cvp.store" : (o.impl == IMPL_CF ? "cf.store" : (o.impl == IMPL_HALO ? "halo.store" : (o.impl == IMPL_PIG ? "pig.store" : "igemm.store"))))), pd.Npad, cb), short_name(c.wname), conv_flops(c, c.phases.size()),
            (src_bytes(c) + out_bytes(c)) / np + w_bytes(c) / np);
      }
      if (c.epi == EPI_LOGITS) {
        if (training) P.logits_op_train = (int)ops->size() - 1; else P.logits_op_eval = (int)ops->size() - 1;
      }
    }
    // Four phases of the two-segment head convolution, all on conv3.hip: ONE launch that walks (tile, phase) pairs - the four phases of
    // a tile run side by side on one XCD and the half-resolution input (0.8 GB at C2) comes from HBM once instead of four times.
    if (c.phases.size() == 4 && c.nseg == 2 && ops->size() >= 4 && !P.sw.no_c3_merge) {
      const size_t first = ops->size() - 4;
      bool ok = true;
      for (size_t k = first; k < ops->size(); ++k) {
        const Op& po = (*ops)[k];
        ok = ok && po.kind == OP_IGEMM && po.epi == EPI_STORE && po.impl == IMPL_CONV3 && po.c.nseg == 2 && po.c.seg[0].ntaps == 4 &&
             po.c.seg[1].ntaps == 9 && po.leaf == (*ops)[first].leaf;
      }
      if (ok) {
        Op merged = (*ops)[first];
        merged.c.nphase = 4;
        merged.flops = 0; merged.bytes = 0;
        for (int ph = 0; ph < 4; ++ph) {
          const Op& po = (*ops)[first + ph];
          for (int t = 0; t < 4; ++t) merged.c.ph_taps0[ph][t] = po.c.seg[0].taps[t];
          for (int t = 0; t < 9; ++t) merged.c.ph_taps1[ph][t] = po.c.seg[1].taps[t];
          merged.c.ph_wpack[ph] = po.c.wpack;
          merged.c.ph_py[ph] = (signed char)po.c.py; merged.c.ph_px[ph] = (signed char)po.c.px;
          merged.flops += po.flops; merged.bytes += po.bytes;
        }
        const int took = igemm_pick(merged.c, dtype, EPI_STORE, d.use_mfma != 0);
        if (took == IMPL_CONV3 || took == IMPL_HF) {   // hf.hip (wave-specialised, one phase per workgroup) where its shape fits, else conv3.hip
          merged.impl = took;
          if (took == IMPL_HF) {
            char lb[56];
            snprintf(lb, sizeof(lb), "hf.store%s", strchr(merged.label, '.') && strstr(merged.label, ".n") ? strstr(merged.label, ".n") : "");
            snprintf(merged.label, sizeof(merged.label), "%s", lb);
          }
          ops->resize(first);
          ops->push_back(merged);
        }
      }
    }
    // The four parity phases of a ConvTranspose (1, 2, 2, 4 taps), all on cvp.hip: ONE launch.  A phase of the first decoder stage is 600
    // workgroups of 128 x 128 outputs for 512 slots on the chip (two per CU) - two rounds, the second 17 % full; 1200 in the second stage - three
    // rounds.  One launch deals the 2400 (4800) workgroups of all phases, the 4-tap phase first, and the slots stay full.
    if (c.phases.size() == 4 && c.nseg == 1 && c.transposed && ops->size() >= 4 && !P.sw.no_cvp_merge) {
      const size_t first = ops->size() - 4;
      bool ok = true;
      for (size_t k = first; k < ops->size(); ++k) {
        const Op& po = (*ops)[k];
        ok = ok && po.kind == OP_IGEMM && po.epi == EPI_STORE && po.impl == IMPL_CVP && po.c.nseg == 1 && po.c.seg[0].ntaps <= 4 &&
             po.c.nphase == 0 && po.leaf == (*ops)[first].leaf && po.c.out == (*ops)[first].c.out && po.c.seg[0].src == (*ops)[first].c.seg[0].src;
      }
      if (ok) {
        Op merged = (*ops)[first];
        merged.c.nphase = 4;
        merged.flops = 0; merged.bytes = 0;
        for (int ph = 0; ph < 4; ++ph) {
          const Op& po = (*ops)[first + ph];
          merged.c.ph_ntaps[ph] = (signed char)po.c.seg[0].ntaps;
          for (int t = 0; t < 4; ++t) merged.c.ph_taps0[ph][t] = po.c.seg[0].taps[t < po.c.seg[0].ntaps ? t : 0];
          merged.c.ph_wpack[ph] = po.c.wpack;
          merged.c.ph_py[ph] = (signed char)po.c.py; merged.c.ph_px[ph] = (signed char)po.c.px;
          merged.flops += po.flops; merged.bytes += po.bytes;
        }
        if (igemm_pick(merged.c, dtype, EPI_STORE, d.use_mfma != 0) == IMPL_CVP) {
          ops->resize(first);
          ops->push_back(merged);
        }
      }
    }
  }

  void fill_grad_seg(Seg& s, int buf, int ch0, int C, const std::vector<Tap>& taps, int istride) {
    memset(&s, 0, sizeof(s));
    const Buf& b = bufs[buf];
    s.src = gat(buf, ch0);
    s.ld = b.ld;
    s.Hs = b.H; s.Ws = b.W;
    s.C = C; s.Cpad = C;
    s.mode = G_PLAIN;
    s.istride = istride;
    if (b.q && !b.materialized && ch0 + C > b.mat_front) { s.src2 = xat(buf, ch0); s.ld2 = b.ld; s.q = b.q + ch0; s.r = b.r + ch0; s.ql = b.ql + ch0; s.rl = b.rl + ch0; }
    fill_seg_taps(s, taps, BK);
  }

  void emit_bn_bwd_finalize(int bn) {
    Bn& b = bns[bn];
    const size_t first_op = ops->size();
    for (auto& rg : b.ranges) {
      const bool outer = leaf_scope;
      leaf_scope = outer || is_raw_input(rg.buf);
      Op& o = push(OP_BNBWD);
      leaf_scope = outer;
      BnBwdFinalizeArgs& a = o.bb;
      a.red1 = b.red1 + rg.c0; a.red2 = b.red2 + rg.c0;
      a.stat_stride = b.cp;
      a.mean = b.mean + rg.c0; a.invstd = b.invstd + rg.c0; a.scale = b.scale + rg.c0;
      a.dgamma = b.dgamma + rg.c0; a.dbeta = b.dbeta + rg.c0;
      const Buf& sb = bufs[rg.buf];
      const bool qr = rg.want_qr && sb.q;
      a.qd = qr ? sb.qd + rg.ch0 : nullptr; a.rd = qr ? sb.rd + rg.ch0 : nullptr;
      a.q = qr ? sb.q + rg.ch0 : nullptr; a.r = qr ? sb.r + rg.ch0 : nullptr;
      a.ql = qr ? sb.ql + rg.ch0 : nullptr; a.rl = qr ? sb.rl + rg.ch0 : nullptr;
      a.count = rg.count;
      a.grad_scale = 1.0f / d.loss_scale;
      a.C = rg.n;
    }
    bn_grad_done(bn, first_op);
    for (auto& rg : b.ranges) {
      Buf& sb = bufs[rg.buf];
      if (!(rg.want_qr && sb.q && sb.matz) || sb.materialized) continue;
      Op& o = push(OP_APPLYCORR);
      ApplyCorrArgs& a = o.ac;
      a.g = gat(rg.buf, rg.ch0); a.y = xat(rg.buf, rg.ch0);
      a.q = sb.q + rg.ch0; a.r = sb.r + rg.ch0; a.ql = sb.ql + rg.ch0; a.rl = sb.rl + rg.ch0;
      a.npix = (size_t)sb.B * sb.H * sb.W;
      a.C = rup(rg.n, 8); a.ldg = sb.ld; a.ldy = sb.ld;
      tag(o, "applycorr", "grad", 0, 3.0 * a.npix * a.C * esz);
      sb.materialized = true;
    }
    for (auto& rg : b.ranges) {  // the ConvTranspose-written front of a decoder concat buffer: final after this BatchNorm
      Buf& sb = bufs[rg.buf];
      if (!(rg.want_qr && sb.q && sb.front > 0 && rg.ch0 == 0 && rg.n >= sb.front) || sb.materialized || sb.mat_front || !front_matz) continue;
      Op& o = push(OP_APPLYCORR);
      ApplyCorrArgs& a = o.ac;
      a.g = gat(rg.buf, 0); a.y = xat(rg.buf, 0);
      a.q = sb.q; a.r = sb.r; a.ql = sb.ql; a.rl = sb.rl;
      a.npix = (size_t)sb.B * sb.H * sb.W;
      a.C = sb.front; a.ldg = sb.ld; a.ldy = sb.ld;
      tag(o, "applycorr", "grad", 0, 3.0 * a.npix * a.C * esz);
      sb.mat_front = sb.front;
    }
  }

  void emit_conv_bwd(ConvRec& c) {
    {
      const bool defer_on = P.sw.defer_wgrad;   // (OFF by default, see above)
      const char* defer_at = lab_str("DMM_DEFER_AT");
      const bool late = c.wname.rfind("decoder.", 0) == 0 || c.wname.rfind("dec_out_to_heat_maps.", 0) == 0;
      const bool flush_here = defer_at ? c.wname.find(defer_at) != std::string::npos : !late;
      if (flush_here) flush_deferred();
      // multi-tap convolutions only: the decoder's 1x1 convolutions keep their place (the 128-wide one is fused with its data gradient)
      defer_scope = defer_on && late && d.use_mfma && dtype != DT_F32 && c.R * c.S > 1 && (deferred_ops.size() + 40 < 256) &&
                    !(defer_at && flush_here);
    }
    struct ScopeEnd { bool& f; ~ScopeEnd() { f = false; } } scope_end{defer_scope};
    const Buf& ob = bufs[c.obuf];
    const int Nst = rup(c.N, 8);  // storage channels of the output gradient
    // ---- weight gradient, one launch per phase ----
    auto one_tap_seg = [&](Seg& s, int py, int px) {
      s.ntaps = 1; s.nchunks = 1;
      s.taps[0] = (short)((py & 0xff) | ((px & 0xff) << 8));
    };
    // The dense 3x3 convolution in 16-bit storage: its data gradient (conv3.hip) holds the effective output gradient of every tile in
    // LDS and writes the interior out as a compact [pixel][32] tensor, which the weight gradient (wg3.hip) reads INSTEAD of gathering
    // 64 of every row's bytes from the block's gradient and activation buffers (128-byte lines: 4x the bytes).  One buffer per
    // convolution: the weight gradient runs on the other stream, later than the next layer's data gradient.  Reserved by the
    // shape alone (the sizing pass and the bound pass must take the same bytes); the weight-gradient launch then FOLLOWS the data
    // gradient in the list.
    void* eff_compact = nullptr;
    bool wt_deferred = false;
    Op wt_op;
    bool have_second = false;
    bool r1_stats = false;    // (round 5) norm1's sums come from the 5x5 weight gradient: the reductions-only pass is not emitted
    bool raw_stats = false;   // (round 5) the raw-input channels' BatchNorm sums come from the weight gradient: no data gradient towards them
    Op second_pass;
    int second_buf = -1;
    if (c.wgrad_transposed && dtype != DT_F32 && c.R == 3 && c.S == 3 && Nst == 32 && c.nseg == 1 && c.seg[0].C == 128 &&
        c.seg[0].dgrad == DG_FLIP && !P.sw.no_eff_compact)
      eff_compact = wptr<uint8_t>((size_t)c.B * c.Ho * c.Wo * 32 * esz);
    // Round 5 (second half): the head's 5x5 convolution onto the classes.  Its two-pass data gradient (below) ran a reductions-only
    // FIRST pass over the 64-channel full-resolution activation for norm1's two sums; its weight gradient (wg5.hip) reads the same
    // activation and the same logits gradient.  With the activation entered as its two factors (wg5.hip, PA = 3) the weight-gradient
    // launch yields dW AND both sums (wg5_fin64_kernel): the first pass is not emitted, and the weight gradient - now the producer of
    // what the chain needs next - runs on the main stream (Op::chain).  The buffer is reserved by shape alone.
    float* r1_sbuf = nullptr;
    int r1_idx = -1;
    if (c.wgrad_transposed && dtype != DT_F32 && c.R == 5 && c.S == 5 && Nst == 8 && c.nseg == 1 && c.seg[0].C == 64 &&
        c.seg[0].dgrad == DG_FLIP && c.N <= 4 && !P.sw.no_r1_stats && !P.sw.no_two_pass)
      r1_sbuf = zbptr<float>(W5_SBUF64_FLOATS);
    if (c.wgrad_transposed) {
      Op& o = push(OP_WGRAD);
      WgradArgs& a = o.w;
      memset(&a, 0, sizeof(a));
      a.nseg = 1;
      fill_grad_seg(a.seg[0], c.obuf, c.och0, Nst, taps_conv_dgrad(c.R, c.S, c.pad), 1);   // Q: dY with flipped taps
      a.B = c.B; a.Ho = c.Ho; a.Wo = c.Wo; a.M = c.B * c.Ho * c.Wo;
      fill_fwd_seg(a.dy, c.seg[0], taps_conv(1, 1, 0));                                      // P: activated input, once
      const PackDesc& pd = P.packs[c.dpack[0]];
      a.N = c.seg[0].C; a.Npad = pd.Npad;
      a.dpack = (float*)pd.dpack;
      // wg3.hip's per-workgroup slots: ONE buffer for all launches of the family (they run in order on one stream, each followed
      // by its reduction).  Reserved by the shape alone: the sizing pass and the bound pass must take the same bytes.
      if (dtype != DT_F32 && c.R == 3 && c.S == 3 && Nst == 32 && c.seg[0].C == 128 && !lab_flag("DMM_WG3_SLOTS_OFF")) {
        if (!wg3_part_taken) { wg3_part = wptr<float>((size_t)W3_MAX_SLOTS * W3_SLOT_FLOATS); wg3_part_taken = true; }
        a.part = wg3_part;
        a.part_slots = W3_MAX_SLOTS;
      }
      char cb[32];
      o.impl = wgrad_pick(a, dtype, d.use_mfma != 0);
      tag(o, ncls(o.impl == IMPL_WG3 ? "wg3" : (o.impl == IMPL_WG5 ? "wg5" : "wgradT"), pd.Npad, cb), short_name(c.wname), conv_flops(c, 1),
          src_bytes(c) + ((ob.q && !ob.materialized && c.och0 + rup(c.N, 8) > ob.mat_front) ? 2.0 : 1.0) * out_bytes(c) + w_bytes(c) * 4.0 / esz);
      if (eff_compact != nullptr && o.impl == IMPL_WG3 && !leaf_scope) {  // emitted behind the data gradient, reading its compact copy
        wt_op = o;
        ops->pop_back();
        wt_deferred = true;
      }
      // the head's 5x5 convolution: a candidate for norm1's sums from the factor correlations (decided with the data gradient below)
      if (r1_sbuf != nullptr && o.impl == IMPL_WG5 && !leaf_scope && !defer_scope && c.seg[0].bn >= 0) r1_idx = (int)ops->size() - 1;
    } else {
    // The head's first convolution (two segments, four output-parity phases): the phase split exists for the upsampled decoder
    // segment; the 8-channel raw-input segment is a plain 3x3 convolution over the full-resolution grid, whose weight gradient is ONE
    // pass over the output gradient (wg5.hip) instead of four generic launches that each re-gather a quarter of it tap by tap.  Its
    // result goes to phase 0's packed gradient (the phases' packed gradients are summed into the master weights by unpack).
    const bool raw_once = c.nseg == 2 && c.phases.size() == 4 && c.shared_master && c.seg[1].C == 8 && c.seg[1].istride == 2 &&
                          c.ostride == 2 && !lab_flag("DMM_NO_RAW_ONCE");
    const int wseg = raw_once ? 1 : c.nseg;
    for (auto& ph : c.phases) {
      Op& o = push(OP_WGRAD);
      WgradArgs& a = o.w;
      memset(&a, 0, sizeof(a));
      a.nseg = wseg;
      for (int s = 0; s < wseg; ++s) fill_fwd_seg(a.seg[s], c.seg[s], (s == 1 && !ph.taps1.empty()) ? ph.taps1 : ph.taps);
      a.B = c.B; a.Ho = c.Ho; a.Wo = c.Wo; a.M = c.B * c.Ho * c.Wo;
      fill_grad_seg(a.dy, c.obuf, c.och0, Nst, taps_conv(1, 1, 0), c.ostride);
      one_tap_seg(a.dy, ph.py, ph.px);
      const PackDesc& pd = P.packs[ph.pack];
      a.N = c.N; a.Npad = pd.Npad;
      a.dpack = (float*)pd.dpack;
      {
        char cb[32];
        const double np = (double)c.phases.size();
        // reads: forward operand, output gradient and (for the deferred correction) the forward output; writes dW
        o.impl = wgrad_pick(a, dtype, d.use_mfma != 0);
        tag(o, ncls(o.impl == IMPL_WGP ? "wgp" : (o.impl == IMPL_WG5 ? "wg5" : "wgrad"), pd.Npad, cb), short_name(c.wname),
            conv_flops(c, c.phases.size()) * (raw_once ? (double)c.seg[0].Cw / c.Kin : 1.0),
            (src_bytes(c) + ((ob.q && !ob.materialized && c.och0 + rup(c.N, 8) > ob.mat_front) ? 2.0 : 1.0) * out_bytes(c)) / np + w_bytes(c) * 4.0 / esz / np);
      }
    }
    // Four phases, all on wgp.hip: ONE launch, the phases of a tile range side by side on one XCD - the input comes from HBM once
    // instead of four times.  Round 4: the head's 3x3 over the upsampled decoder output (four taps in every phase; 0.8 GB of
    // half-resolution input at C2).  Round 5: the decoder's ConvTranspose stages as well (1, 2, 2, 4 taps: WgradArgs::ph_ntaps; the
    // wave-specialised form of the kernel, wgpw.hip, takes the phase's tap count per workgroup).
    if (!defer_scope && c.phases.size() == 4 && wseg == 1 && ops->size() >= 4 && !P.sw.no_wgp_merge) {
      const size_t first = ops->size() - 4;
      bool ok = true, all4 = true;
      for (size_t k = first; k < ops->size(); ++k) {
        const Op& po = (*ops)[k];
        const int nt = po.w.seg[0].ntaps;
        ok = ok && po.kind == OP_WGRAD && po.impl == IMPL_WGP && po.w.nseg == 1 && (nt == 1 || nt == 2 || nt == 4) && po.leaf == (*ops)[first].leaf &&
             po.w.seg[0].src == (*ops)[first].w.seg[0].src && po.w.dy.src == (*ops)[first].w.dy.src;
        all4 = all4 && nt == 4;
      }
      if (ok) {
        Op merged = (*ops)[first];
        merged.w.nphase = 4;
        merged.flops = 0; merged.bytes = 0;
        for (int ph = 0; ph < 4; ++ph) {
          const Op& po = (*ops)[first + ph];
          for (int t = 0; t < 4; ++t) merged.w.ph_xtaps[ph][t] = po.w.seg[0].taps[t < po.w.seg[0].ntaps ? t : 0];
          merged.w.ph_ntaps[ph] = all4 ? 0 : (signed char)po.w.seg[0].ntaps;
          merged.w.ph_ytap[ph] = po.w.dy.taps[0];
          merged.w.ph_dpack[ph] = po.w.dpack;
          merged.flops += po.flops; merged.bytes += po.bytes;
        }
        if (wgrad_pick(merged.w, dtype, d.use_mfma != 0) == IMPL_WGP) {
          ops->resize(first);
          ops->push_back(merged);
        }
      }
    }
    // Round 5: with the raw-input segment's weight gradient on wg5.hip, that launch correlates the output gradient with the two FACTORS
    // of the activation (wg5.hip, PY = 2) and a one-workgroup launch turns the result into the packed weight gradient AND the
    // BatchNorm-backward sums of the raw-input channels - the reductions-only data gradient towards the raw input (a leaf launch: one
    // more pass over the full-resolution gradient, 0.69 ms at C2) is not emitted.  The buffer is reserved by shape alone.
    float* raw_sbuf = nullptr;
    if (raw_once && dtype != DT_F32 && c.N == 64 && !P.sw.no_raw_stats) raw_sbuf = zbptr<float>(W5_SBUF_FLOATS);
    if (raw_once) {
      Op& o = push(OP_WGRAD);
      WgradArgs& a = o.w;
      memset(&a, 0, sizeof(a));
      a.nseg = 1;
      fill_fwd_seg(a.seg[0], c.seg[1], taps_conv(3, 3, 1));
      a.seg[0].istride = 1;                                  // the row grid is the full-resolution output grid
      a.B = c.B; a.Ho = ob.H; a.Wo = ob.W; a.M = c.B * ob.H * ob.W;
      fill_grad_seg(a.dy, c.obuf, c.och0, Nst, taps_conv(1, 1, 0), 1);
      one_tap_seg(a.dy, 0, 0);
      const PackDesc& pd = P.packs[c.phases[0].pack];
      a.N = c.N; a.Npad = pd.Npad;
      a.dpack = (float*)pd.dpack + (size_t)pd.seg[0].nchunks * pd.Npad * BK;   // behind segment 0's chunks
      char cb[32];
      o.impl = wgrad_pick(a, dtype, d.use_mfma != 0);
      if (raw_sbuf != nullptr && o.impl == IMPL_WG5 && c.seg[1].bn >= 0 && !defer_scope) {   // (a held-back launch would run behind its finish)
        WgradArgs f = a;
        f.sbuf = raw_sbuf;
        f.t_mean = bns[c.seg[1].bn].mean + c.seg[1].bn_c0;
        f.t_invstd = bns[c.seg[1].bn].invstd + c.seg[1].bn_c0;
        if (wgrad_pick(f, dtype, d.use_mfma != 0) == IMPL_WG5) { a = f; raw_stats = true; }
      }
      const Buf& rb = bufs[c.seg[1].buf];
      tag(o, ncls(o.impl == IMPL_WG5 ? "wg5" : "wgrad", pd.Npad, cb), short_name(c.wname) + ".raw", conv_flops(c, 1) * ((double)c.seg[1].Cw / c.Kin),
          (double)rb.B * rb.H * rb.W * 8 * esz + ((ob.q && !ob.materialized) ? 2.0 : 1.0) * out_bytes(c));
      if (raw_stats) {   // right behind it on the same stream: packed gradient of the segment + the norm's sums for these channels
        const WgradArgs wa = o.w;
        const bool leaf_of = o.leaf;
        Op& fo = push(OP_RAWFIN);
        fo.leaf = leaf_of ? leaf_of : 1;
        RawFinArgs& r = fo.rf;
        memset(&r, 0, sizeof(r));
        r.sbuf = wa.sbuf; r.dpack = wa.dpack; r.Npad = wa.Npad;
        r.w = Pp + T(c.wname).off; r.Kin = c.Kin; r.koff = c.seg[1].koff; r.nreal = c.seg[1].Cw;
        for (int t = 0; t < 9; ++t) r.tapw[t] = (int)(pd.seg[1].tapw[t] & 0xff);
        const Bn& bn = bns[c.seg[1].bn];
        r.gamma = bn.gamma + c.seg[1].bn_c0; r.beta = bn.beta + c.seg[1].bn_c0;
        r.red1 = bn.red1 + c.seg[1].bn_c0; r.red2 = bn.red2 + c.seg[1].bn_c0;
        tag(fo, "wg5.rawfin", short_name(c.wname) + ".raw", 0, W5_SBUF_FLOATS * 4.0);
      }
    }
    }
    // A dense layer's 1x1 bottleneck convolution: its weight gradient and its data gradient read the same three tensors; when the
    // pair qualifies (bw1.hip) the weight-gradient launch is taken back here and folded into the data-gradient launch below.
    bool pending_w = false;
    Op saved_w;
    if (d.use_mfma && !c.wgrad_transposed && c.phases.size() == 1 && c.nseg == 1 && c.seg[0].dgrad == DG_FLIP && c.R == 1 && c.S == 1 &&
        !ops->empty() && ops->back().kind == OP_WGRAD && !leaf_scope) {
      saved_w = ops->back();
      ops->pop_back();
      pending_w = true;
    }
    const bool r1_wait = r1_idx >= 0;   // (the finish launch may still add into this convolution's packed gradient)
    if (!pending_w && !wt_deferred && !r1_wait) conv_grad_done(c);
    // ---- data gradients with fused BN+ReLU backward ----
    for (int s = 0; s < c.nseg; ++s) {
      SegRec& sr = c.seg[s];
      if (sr.dgrad == DG_NONE) continue;
      if (raw_stats && s == 1 && is_raw_input(sr.buf)) continue;   // its only purpose were the two sums wg5.rawfin has just produced
      Buf& sb = bufs[sr.buf];
      std::vector<Tap> taps;
      int istride = 1, rB = sb.B, rH = sb.H, rW = sb.W, pool2 = 0, ostride = 1;
      switch (sr.dgrad) {
        case DG_FLIP: taps = taps_conv_dgrad(c.R, c.S, c.pad); break;
        case DG_POOL2: taps = taps_conv(1, 1, 0); rH = sb.H / 2; rW = sb.W / 2; pool2 = 1; ostride = 2; break;
        case DG_CONVT: taps = taps_convT_dgrad(); istride = 2; break;
        case DG_UP2: taps = taps_up2_merged_dgrad(); istride = 2; break;
      }
      // the gradient w.r.t. a raw input is only needed for the BatchNorm parameter gradients of the norm in front of the
      // conv: its launch reduces sum(dz), sum(dz*xhat), stores nothing, and is a leaf of the backward graph
      const bool raw = is_raw_input(sr.buf);
      leaf_scope = raw;
      Op& o = push(OP_IGEMM);
      leaf_scope = false;
      o.epi = EPI_BNBWD;
      ConvArgs& a = o.c;
      memset(&a, 0, sizeof(a));
      a.nseg = 1;
      fill_grad_seg(a.seg[0], c.obuf, c.och0, Nst, taps, istride);
      a.B = rB; a.Ho = rH; a.Wo = rW; a.M = rB * rH * rW;
      const PackDesc& pd = P.packs[c.dpack[s]];
      a.wpack = pd.dst;
      a.N = sr.C; a.Npad = pd.Npad;
      a.out = raw ? nullptr : gat(sr.buf, sr.ch0);
      a.ldo = sb.ld; a.Hout = sb.H; a.Wout = sb.W; a.ostride = ostride; a.py = 0; a.px = 0;
      a.bx = xat(sr.buf, sr.ch0);
      a.ldbx = sb.ld;
      const Bn& bn = bns[sr.bn];
      a.bscale = bn.scale + sr.bn_c0; a.bshift = bn.shift + sr.bn_c0;
      a.bmean = bn.mean + sr.bn_c0; a.binvstd = bn.invstd + sr.bn_c0;
      a.red1 = bn.red1 + sr.bn_c0; a.red2 = bn.red2 + sr.bn_c0;
      a.stat_stride = bn.cp;
      a.accumulate = (sb.ginit && !raw) ? 1 : 0;
      a.pool2 = pool2;
      {
        char cb[32];
        // reads: output gradient (+ forward output for the correction), x for the ReLU mask, (old gradient); writes gradient
        const double srcb = (double)sb.B * sb.H * sb.W * sr.C * esz;
        const double segf = conv_flops(c, 1) * ((double)sr.Cw / c.Kin) * (sr.dgrad == DG_UP2 ? 16.0 / 36.0 : 1.0);
        o.impl = igemm_pick(a, dtype, EPI_BNBWD, d.use_mfma != 0);
        const bool c3 = o.impl == IMPL_CONV3, cp = o.impl == IMPL_CVP;
        if (wt_deferred && c3) {  // hand the effective gradient over: the weight gradient reads the compact copy, no prologue
          a.eff_out = eff_compact;
          Seg& q = wt_op.w.seg[0];
          q.src = eff_compact; q.ld = 32;
          q.src2 = nullptr; q.ld2 = 0; q.q = q.r = q.ql = q.rl = nullptr;
        }
        tag(o, ncls(c3 ? "conv3.bnbwd" : (cp ? "cvp.bnbwd" : (o.impl == IMPL_HALO ? "halo.bnbwd" : "igemm.bnbwd")), pd.Npad, cb), short_name(c.wname), segf,
            ((ob.q && !ob.materialized && c.och0 + rup(c.N, 8) > ob.mat_front) ? 2.0 : 1.0) * out_bytes(c) + srcb * (sb.ginit ? 3.0 : 2.0) + w_bytes(c));
        // Two-pass BatchNorm backward where the data gradient is cheap to run twice and its result would otherwise be corrected by a
        // pass of its own (apply_corr: read g, read x, write g).  The head's 5x5 convolution onto the classes: its data gradient
        // reads the 3-channel logits gradient and, for the ReLU mask, the 64-channel full-resolution activation x.  First pass:
        // reductions only (nothing stored); finalize (q, r); second pass: the same launch stores the FINAL gradient
        // s*dz + q + r*x.  Traffic 2 x |x| + |g| instead of |x| + |g| (store) + 2 |g| + |x| (apply_corr).
        if (c3 && !raw && sb.matz && sb.q && !sb.ginit && !sb.materialized && c.nseg == 1 && c.R == 5 && c.S == 5 && dtype != DT_F32 &&
            !P.sw.no_two_pass) {
          Op first_pass = o;                     // reductions only: stores nothing
          first_pass.c.out = nullptr;
          second_pass = o;                       // stores; no reductions
          second_pass.c.red1 = nullptr; second_pass.c.red2 = nullptr;
          second_pass.c.eq = sb.q + sr.ch0; second_pass.c.er = sb.r + sr.ch0;
          // both MUTATED launches must still be conv3.hip's (the only kernel whose epilogue knows eq / er)
          if (igemm_pick(first_pass.c, dtype, EPI_BNBWD, d.use_mfma != 0) == IMPL_CONV3 &&
              igemm_pick(second_pass.c, dtype, EPI_BNBWD, d.use_mfma != 0) == IMPL_CONV3) {
            have_second = true;
            second_buf = sr.buf;
            a.out = nullptr;
            o.bytes = out_bytes(c) + srcb + w_bytes(c);          // reads dy and x, writes nothing
            second_pass.bytes = out_bytes(c) + srcb * 2.0 + w_bytes(c);
            if (r1_idx >= 0) {   // norm1's sums from the weight gradient's factor correlations: the first pass becomes the finish launch
              WgradArgs f = (*ops)[r1_idx].w;
              f.sbuf = r1_sbuf;
              f.t_mean = bn.mean + sr.bn_c0; f.t_invstd = bn.invstd + sr.bn_c0;   // (mark of the factor form; the kernel needs scale / shift only)
              if (wgrad_pick(f, dtype, d.use_mfma != 0) == IMPL_WG5) {
                Op& wo = (*ops)[r1_idx];
                wo.w = f;
                wo.chain = 1;
                const PackDesc& wpd = P.packs[c.dpack[0]];
                Fin64Args r;
                memset(&r, 0, sizeof(r));
                r.sbuf = r1_sbuf; r.dpack = f.dpack; r.Npad = f.Npad;
                r.w = Pp + T(c.wname).off; r.Kin = c.Kin; r.nreal = c.N; r.dtype = dtype;
                for (int t = 0; t < 25; ++t) r.tapw[t] = (unsigned char)(wpd.seg[0].tapw[t] & 0xff);
                r.scale = bn.scale + sr.bn_c0; r.shift = bn.shift + sr.bn_c0;
                r.mean = bn.mean + sr.bn_c0; r.invstd = bn.invstd + sr.bn_c0;
                r.red1 = bn.red1 + sr.bn_c0; r.red2 = bn.red2 + sr.bn_c0;
                o.kind = OP_FIN64; o.epi = 0; o.impl = IMPL_AUTO; o.chain = 1;
                o.f64 = r;     // (overwrites the first pass's arguments: `a` is dead from here on)
                tag(o, "wg5.fin64", short_name(c.wname), 0, W5_SBUF64_FLOATS * 4.0);
                r1_stats = true;
              }
            }
          }
        }
      }
      if (pending_w) {
        pending_w = false;
        Op dgrad_op = ops->back();
        // slots for the weight-gradient partials of the fused launch (pointwise.h: Bw1Args::part).  Reserved by the shape of the pair
        // alone - the sizing pass and the bound pass must take the same bytes, and bw1_eligible looks at pointers
        float* part = nullptr;
        int part_slots = 0;
        if (!raw && dtype != DT_F32 && c.R == 1 && c.S == 1 && c.N == 128 && c.nseg == 1) {
          const Bw1Geom q = bw1_geometry(dgrad_op.c);
          part_slots = q.nsplit * q.nct;
          part = wptr<float>((size_t)part_slots * B1_SLOT_FLOATS);
        }
        if (!raw && bw1_eligible(saved_w.w, dgrad_op.c, dtype)) {
          Op& f = ops->back();
          f.kind = OP_BW1;
          f.b1.dpack = saved_w.w.dpack;
          f.b1.dNpad = saved_w.w.Npad;
          f.b1.wC = saved_w.w.seg[0].C;
          f.b1.part = part;
          f.b1.part_slots = part_slots;
          f.b1.nct = f.b1.ntiles = f.b1.tiles_per_wg = f.b1.xcd_group = f.b1.nsplit = 0;
          char cb[32];
          // the pair's traffic with every operand read once: the data gradient's bytes + the packed weight gradient
          tag(f, ncls("bw1", f.c.Npad, cb), short_name(c.wname), dgrad_op.flops + saved_w.flops, dgrad_op.bytes + w_bytes(c) * 4.0 / esz);
          if (part != nullptr) {  // the slots are added up beside the data-gradient chain (weight-gradient stream)
            const Bw1Args b1 = f.b1;
            Op& rop = push(OP_BW1RED);
            rop.leaf = 1;
            rop.b1 = b1;
            tag(rop, "bw1.reduce", short_name(c.wname), 0, ((double)part_slots + b1.nct) * B1_SLOT_FLOATS * 4.0);
          }
          conv_grad_done(c);
        } else {  // not this time: weight gradient, bucket bookkeeping, data gradient - in the original order
          ops->pop_back();
          ops->push_back(saved_w);
          conv_grad_done(c);
          ops->push_back(dgrad_op);
        }
      }
      if (!raw) sb.ginit = true;
    }
    if (pending_w) { ops->push_back(saved_w); conv_grad_done(c); pending_w = false; }  // (no data gradient was emitted)
    if (wt_deferred) { ops->push_back(wt_op); conv_grad_done(c); }
    if (r1_wait) conv_grad_done(c);
    int done = -1;
    if (have_second) bufs[second_buf].materialized = true;   // (no apply_corr behind the finalize: the second pass stores the final gradient)
    for (int s = 0; s < c.nseg; ++s) {
      const int bn = c.seg[s].bn;
      if (bn < 0 || bn == done || c.seg[s].dgrad == DG_NONE) continue;
      emit_bn_bwd_finalize(bn);
      done = bn;
    }
    if (have_second) ops->push_back(second_pass);
  }

  void emit_pool_fwd(PoolRec& p) {
    emit_bn_finalize(p.bn);
    Op& o = push(OP_POOL);
    MaxpoolArgs& a = o.mp;
    const Buf &yb = bufs[p.y0buf], &ob = bufs[p.obuf];
    a.y0 = yb.x; a.ld0 = yb.ld; a.H0 = yb.H; a.W0 = yb.W; a.B = yb.B; a.C = p.C;
    a.scale = bns[p.bn].scale; a.shift = bns[p.bn].shift;
    a.out = (void*)xat(p.obuf, p.och0); a.ldo = ob.ld; a.Hp = ob.H; a.Wp = ob.W;
    a.argmax = p.argmax;
    a.stat_sum = ob.ssum + p.och0; a.stat_sq = ob.ssq + p.och0; a.stat_stride = ob.ld;
    tag(o, "maxpool.fwd", "pool0", 0, ((double)yb.B * yb.H * yb.W + (double)ob.B * ob.H * ob.W) * p.C * esz + (double)ob.B * ob.H * ob.W * p.C);
  }
  void emit_pool_bwd(PoolRec& p) {
    // the stem: pool0 backward, norm0's reductions and conv0's weight gradient feed parameter gradients only
    leaf_scope = true;
    Op& o = push(OP_POOLBWD);
    MaxpoolBwdArgs& a = o.mpb;
    Buf &yb = bufs[p.y0buf];
    const Buf &ob = bufs[p.obuf];
    a.y0 = yb.x; a.ld0 = yb.ld; a.H0 = yb.H; a.W0 = yb.W; a.B = yb.B; a.C = p.C;
    a.scale = bns[p.bn].scale; a.shift = bns[p.bn].shift;
    a.gpool = gat(p.obuf, p.och0); a.xpool = xat(p.obuf, p.och0);
    a.q = ob.q + p.och0; a.r = ob.r + p.och0; a.ql = ob.ql + p.och0; a.rl = ob.rl + p.och0;
    a.ldg = ob.ld; a.Hp = ob.H; a.Wp = ob.W;
    a.argmax = p.argmax;
    a.gy0 = yb.g;
    a.mean = bns[p.bn].mean; a.invstd = bns[p.bn].invstd;
    a.red1 = bns[p.bn].red1; a.red2 = bns[p.bn].red2; a.stat_stride = bns[p.bn].cp;
    tag(o, "maxpool.bwd", "pool0", 0, (2.0 * yb.B * yb.H * yb.W + 2.0 * ob.B * ob.H * ob.W) * p.C * esz + (double)ob.B * ob.H * ob.W * p.C);
    yb.ginit = true;
    emit_bn_bwd_finalize(p.bn);
    leaf_scope = false;
  }

  // -------------------------------------------------------------------------------- layer records
  ConvRec& new_conv(const std::string& wname, bool transposed, int N, int Kin, int R, int S, int pad) {
    ConvRec c;
    c.wname = wname; c.transposed = transposed; c.N = N; c.Kin = Kin; c.R = R; c.S = S; c.pad = pad;
    c.nseg = 1; c.ostride = 1; c.epi = EPI_STORE; c.stats = true;
    c.dpack[0] = c.dpack[1] = -1;
    c.wgrad_transposed = false;
    c.shared_master = false;
    c.flops_ref = 0;
    memset(c.seg, 0, sizeof(c.seg));
    c.seg[0].bn = c.seg[1].bn = -1;
    convs.push_back(c);
    recs.push_back({0, (int)convs.size() - 1});
    return convs.back();
  }
  void finish_conv(ConvRec& c) {
    // forward packs (one per phase) and dgrad packs (one per segment that needs a data gradient)
    for (auto& ph : c.phases) ph.pack = add_pack(c, ph.taps, false, 0, &ph.taps1);
    for (int s = 0; s < c.nseg; ++s) {
      std::vector<Tap> t;
      switch (c.seg[s].dgrad) {
        case DG_FLIP: t = taps_conv_dgrad(c.R, c.S, c.pad); break;
        case DG_POOL2: t = taps_conv(1, 1, 0); break;
        case DG_CONVT: t = taps_convT_dgrad(); break;
        case DG_UP2: t = taps_up2_merged_dgrad(); break;
        default: continue;
      }
      c.dpack[s] = add_pack(c, t, true, s);
    }
    // thin outputs: compute the weight gradient in the transposed form; it fills the dgrad-shaped packed gradient
    if (c.nseg == 1 && !c.transposed && c.seg[0].mode == G_PLAIN && c.seg[0].istride == 1 && c.seg[0].dgrad == DG_FLIP &&
        c.R * c.S > 1 && rup(c.N, 8) < c.seg[0].C && c.phases.size() == 1) {
      c.wgrad_transposed = true;
      PackDesc& dp = P.packs[c.dpack[0]];
      int chunks = 0;
      for (int s2 = 0; s2 < dp.nseg; ++s2) chunks += dp.seg[s2].nchunks;
      dp.dpack = zbptr<float>((size_t)chunks * dp.Npad * BK);
      dp.gw = Pg + (dp.w - Pp);
      P.packs[c.phases[0].pack].gw = nullptr;  // the forward-shaped packed gradient is not produced
    }
    // FLOPs (2*MACs) in the reference's formulation
    flops += conv_flops(c, 1);
  }
  void set_seg(ConvRec& c, int s, int buf, int ch0, int Cw, int koff, int mode, int istride, int bn, int bn_c0, int dgrad) {
    SegRec& sr = c.seg[s];
    sr.buf = buf; sr.ch0 = ch0; sr.Cw = Cw; sr.C = rup(Cw, 8); sr.koff = koff; sr.mode = mode; sr.istride = istride;
    sr.bn = bn; sr.bn_c0 = bn_c0; sr.dgrad = dgrad;
  }

  // stem: conv0 (7x7 s2) -> norm0 -> relu0 -> pool0, output into `obuf` channels [och0, och0+nif)
  void stem(const std::string& p, int inbuf, int in_ch, int obuf, int och0) {
    const Buf ib = bufs[inbuf];  // by value: new_buf() may reallocate `bufs`
    const int y0 = new_buf(ib.B, ib.H / 2, ib.W / 2, g.nif, true, true);
    ConvRec& c = new_conv(p + ".conv0.weight", false, g.nif, in_ch, 7, 7, 3);
    set_seg(c, 0, inbuf, 0, in_ch, 0, G_PLAIN, 2, -1, 0, DG_NONE);
    c.B = ib.B; c.Ho = ib.H / 2; c.Wo = ib.W / 2;
    c.obuf = y0; c.och0 = 0;
    c.phases.push_back({0, 0, taps_conv(7, 7, 3), -1});
    finish_conv(c);
    PoolRec pr;
    pr.y0buf = y0;
    pr.bn = new_bn(p + ".norm0", g.nif);
    bn_range(pr.bn, y0, 0, 0, g.nif);
    pr.obuf = obuf; pr.och0 = och0; pr.C = g.nif;
    pr.argmax = wptr<uint8_t>((size_t)ib.B * (ib.H / 4) * (ib.W / 4) * g.nif);
    pools.push_back(pr);
    recs.push_back({1, (int)pools.size() - 1});
  }

  void dense_block(const std::string& p, int xb, int base, int b) {
    const Buf X = bufs[xb];  // by value: new_buf() may reallocate `bufs`
    const int bw = g.bs * g.k;
    for (int l = 0; l < g.L[b]; ++l) {
      const std::string q = p + ".denselayer" + std::to_string(l + 1);
      const int K = g.cin[b] + l * g.k;
      // (round 2: the gradient of a wide layer's bottleneck was re-gathered by two launches x K/128 column tiles; since bw1 reads it
      // once per channel slice, with the slices of a row range on one XCD, materialising it first only adds a pass: DMM_MATZ_DENSE_K)
      // Measured (round 3, C2 b4, same box): threshold 384 / 640 / never: applycorr 2.08 / 1.81 / 1.69 ms, bw1 5.49 / 5.56 / 5.59 ms.
      static const int matz_k = lab_int("DMM_MATZ_DENSE_K", 1 << 30);
      const int y1 = new_buf(X.B, X.H, X.W, rup(bw, 8), true, true, /*matz=*/K >= matz_k);
      const int n1 = new_bn(q + ".norm1", K);
      bn_range(n1, xb, base, 0, K);
      {
        ConvRec& c = new_conv(q + ".conv1.weight", false, bw, K, 1, 1, 0);
        set_seg(c, 0, xb, base, K, 0, G_PLAIN, 1, n1, 0, DG_FLIP);
        c.B = X.B; c.Ho = X.H; c.Wo = X.W; c.obuf = y1; c.och0 = 0;
        c.phases.push_back({0, 0, taps_conv(1, 1, 0), -1});
        finish_conv(c);
      }
      const int n2 = new_bn(q + ".norm2", bw);
      bn_range(n2, y1, 0, 0, bw);
      {
        ConvRec& c = new_conv(q + ".conv2.weight", false, g.k, bw, 3, 3, 1);
        set_seg(c, 0, y1, 0, bw, 0, G_PLAIN, 1, n2, 0, DG_FLIP);
        c.B = X.B; c.Ho = X.H; c.Wo = X.W; c.obuf = xb; c.och0 = base + K;
        c.phases.push_back({0, 0, taps_conv(3, 3, 1), -1});
        finish_conv(c);
      }
    }
  }

  void transition(const std::string& p, int xb, int base, int C, int obuf, int och0) {
    const Buf X = bufs[xb];
    const int n = new_bn(p + ".norm", C);
    bn_range(n, xb, base, 0, C);
    ConvRec& c = new_conv(p + ".conv.weight", false, C / 2, C, 1, 1, 0);
    set_seg(c, 0, xb, base, C, 0, G_POOL2, 1, n, 0, DG_POOL2);
    c.B = X.B; c.Ho = X.H / 2; c.Wo = X.W / 2; c.obuf = obuf; c.och0 = och0;
    c.phases.push_back({0, 0, taps_conv(1, 1, 0), -1});
    finish_conv(c);
  }

  void build() {
    const int B = d.batch, H = d.height, Wd = d.width;
    if (H % 32 || Wd % 32 || H <= 0 || Wd <= 0) throw std::domain_error("spatial size must be a multiple of 32");
    if (g.k % 8 || g.nif % 8 || (g.bs * g.k) % 8) throw std::invalid_argument("channel counts must be multiples of 8");
    if (g.s1 + g.s2 > 8 || g.s1 < 1) throw std::invalid_argument("at most 8 raw input channels are supported");
    for (int b = 0; b + 1 < g.nb; ++b)
      if (g.cout[b] % 16) throw std::invalid_argument("block widths must be multiples of 16");
    if (g.nc > 8) throw std::invalid_argument("num_classes > 8 unsupported");

    // raw inputs (NHWC8).  inH = what the head concatenates (both streams in early / mid fusion).
    in1 = new_buf(B, H, Wd, 8, g.fusion != 2, true);
    if (g.fusion == 2) { in2 = new_buf(B, H, Wd, 8, false, false); inH = new_buf(B, H, Wd, 8, true, true); }
    else inH = in1;

    // block buffers of stream 1: [decoder ConvTranspose output | block channels]
    std::vector<int> X(g.nb), base(g.nb);
    for (int b = 0; b < g.nb; ++b) {
      base[b] = (b < g.nb - 1) ? g.cout[b] : 0;
      X[b] = new_buf(B, H >> (2 + b), Wd >> (2 + b), base[b] + g.cout[b], true, true);
    }
    int F = -1;
    const int cbb = g.cbb;
    if (g.fusion == 2) {
      F = new_buf(B, H >> (1 + cbb), Wd >> (1 + cbb), 2 * g.cin[cbb - 1], true, true);
      std::vector<int> X2(cbb - 1);
      for (int b = 0; b < cbb - 1; ++b) X2[b] = new_buf(B, H >> (2 + b), Wd >> (2 + b), g.cout[b], true, true);
      stem("stream_2_features", in2, g.s2, X2[0], 0);
      for (int b = 0; b < cbb - 1; ++b) {
        dense_block("stream_2_features.denseblock" + std::to_string(b + 1), X2[b], 0, b);
        const bool last = b == cbb - 2;
        transition("stream_2_features.transition" + std::to_string(b + 1), X2[b], 0, g.cout[b], last ? F : X2[b + 1],
                   last ? g.cin[cbb - 1] : 0);
      }
      s2_recs = recs.size();
    }
    stem("features", in1, g.net_in, X[0], base[0]);
    for (int b = 0; b < g.nb; ++b) {
      if (g.fusion == 2 && b == cbb - 1) {
        const int C = g.cin[b];
        concat_rec = recs.size();
        const int n = new_bn("concat_module.norm", 2 * C);
        bn_range(n, F, 0, 0, 2 * C);
        ConvRec& c = new_conv("concat_module.conv.weight", false, C, 2 * C, 1, 1, 0);
        set_seg(c, 0, F, 0, 2 * C, 0, G_PLAIN, 1, n, 0, DG_FLIP);
        c.B = B; c.Ho = bufs[F].H; c.Wo = bufs[F].W; c.obuf = X[b]; c.och0 = base[b];
        c.phases.push_back({0, 0, taps_conv(1, 1, 0), -1});
        finish_conv(c);
      }
      dense_block("features.denseblock" + std::to_string(b + 1), X[b], base[b], b);
      if (b < g.nb - 1) {
        const bool toF = g.fusion == 2 && b == cbb - 2;
        transition("features.transition" + std::to_string(b + 1), X[b], base[b], g.cout[b], toF ? F : X[b + 1],
                   toF ? 0 : base[b + 1]);
      }
    }
    // decoder
    int U = -1;
    for (int j = 0; j < g.nb; ++j) {
      const std::string p = "decoder.Transposed_Convolution_Sequence_" + std::to_string(j + 1);
      // stage j reads block nb-1 (j = 0) or the full [ConvTranspose_{j-1} | block nb-1-j] buffer
      const int inb = X[g.nb - 1 - j];
      const Buf I = bufs[inb];  // by value: new_buf() may reallocate `bufs`
      const int nin = g.nin[j], nf = g.nf[j];
      if (I.cw != nin) throw std::runtime_error("decoder width mismatch");
      const int n0 = new_bn(p + ".norm0", nin);
      bn_range(n0, inb, 0, 0, nin);
      const int Rb = new_buf(I.B, I.H, I.W, nf, true, true, /*matz=*/(matz_mask & 1) != 0);
      {
        ConvRec& c = new_conv(p + ".conv_reduce.weight", false, nf, nin, 1, 1, 0);
        set_seg(c, 0, inb, 0, nin, 0, G_PLAIN, 1, n0, 0, DG_FLIP);
        c.B = I.B; c.Ho = I.H; c.Wo = I.W; c.obuf = Rb; c.och0 = 0;
        c.phases.push_back({0, 0, taps_conv(1, 1, 0), -1});
        finish_conv(c);
      }
      const int n1 = new_bn(p + ".norm1", nf);
      bn_range(n1, Rb, 0, 0, nf);
      int ob;
      if (j < g.nb - 1) ob = X[g.nb - 2 - j];
      else ob = U = new_buf(I.B, 2 * I.H, 2 * I.W, nf, true, true, /*matz=*/(matz_mask & 2) != 0);
      if (bufs[ob].H != 2 * I.H) throw std::runtime_error("decoder size mismatch");
      if (j < g.nb - 1 && nf % 8 == 0) bufs[ob].front = nf;
      {
        ConvRec& c = new_conv("decoder.Transposed_Convolution_" + std::to_string(j + 1) + ".weight", true, nf, nf, 3, 3, 1);
        set_seg(c, 0, Rb, 0, nf, 0, G_PLAIN, 1, n1, 0, DG_CONVT);
        c.B = I.B; c.Ho = I.H; c.Wo = I.W; c.obuf = ob; c.och0 = 0; c.ostride = 2;
        for (int py = 0; py < 2; ++py)
          for (int px = 0; px < 2; ++px) c.phases.push_back({py, px, taps_convT_phase(py, px), -1});
        finish_conv(c);
      }
    }
    // head
    const int nfl = g.nf[g.nb - 1], raw = g.s1 + g.s2;
    const int hn0 = new_bn("dec_out_to_heat_maps.norm0", nfl + raw);
    bn_range(hn0, U, 0, 0, nfl, 4.0);
    bn_range(hn0, inH, 0, nfl, raw, 1.0, false);
    const int YR = new_buf(B, H, Wd, nfl / 2, true, true, /*matz=*/(matz_mask & 4) != 0);
    {
      ConvRec& c = new_conv("dec_out_to_heat_maps.refine0.weight", false, nfl / 2, nfl + raw, 3, 3, 1);
      // The decoder part of the input is a nearest-x2 upsample, so the conv is evaluated per output parity (e,f) on the
      // HALF-resolution row grid: the 3x3 taps collapse onto 2x2 half-res source pixels with pre-summed weights (2.25x fewer
      // MACs in forward and weight gradient; the 1.26 GB/img upsampled tensor never exists).  The raw-input channels are read
      // at stride 2 with taps shifted by (e,f).
      c.nseg = 2;
      set_seg(c, 0, U, 0, nfl, 0, G_PLAIN, 1, hn0, 0, DG_UP2);
      set_seg(c, 1, inH, 0, raw, nfl, G_PLAIN, 2, hn0, nfl, DG_FLIP);
      c.B = B; c.Ho = H / 2; c.Wo = Wd / 2; c.obuf = YR; c.och0 = 0; c.ostride = 2;
      c.shared_master = true;
      c.flops_ref = 2.0 * B * H * Wd * (double)(nfl / 2) * (nfl + raw) * 9.0;
      for (int e = 0; e < 2; ++e)
        for (int f = 0; f < 2; ++f) {
          PhaseRec ph{e, f, taps_up2_phase(e, f), -1, taps_conv_phase_s2(e, f)};
          c.phases.push_back(ph);
        }
      finish_conv(c);
    }
    const int hn1 = new_bn("dec_out_to_heat_maps.norm1", nfl / 2);
    bn_range(hn1, YR, 0, 0, nfl / 2);
    dl = new_buf(B, H, Wd, 8, false, false);  // x = d(loss)/d(logit) in NHWC8; used as a gradient source only
    bufs[dl].g = bufs[dl].x;
    {
      ConvRec& c = new_conv("dec_out_to_heat_maps.refine1.weight", false, g.nc, nfl / 2, 5, 5, 2);
      set_seg(c, 0, YR, 0, nfl / 2, 0, G_PLAIN, 1, hn1, 0, DG_FLIP);
      c.B = B; c.Ho = H; c.Wo = Wd; c.obuf = dl; c.och0 = 0;
      c.epi = EPI_LOGITS; c.stats = false;
      c.phases.push_back({0, 0, taps_conv(5, 5, 2), -1});
      finish_conv(c);
    }
  }

  // -------------------------------------------------------------------------------- pack / unpack tile lists (pointwise.hip)
  // Descriptors [begin, end) of `v` that the tile kernels can take, as groups of consecutive descriptors sharing one master tensor:
  // one K segment that spans the whole master, whole 32-channel groups, 1x1 or 3x3, the master contiguous along (k, tap) or (n, tap),
  // and every master tap covered by the group (the parity phases of a ConvTranspose together).  Marks PackDesc::tiled and appends the
  // tiles - 1x1 masters first, then 3x3 (one launch per instantiation) - with descriptor indices relative to `v`.
  const bool pack_tiles_on = !P.sw.no_pack_tiles;
  void make_tiles(std::vector<PackDesc>& v, size_t begin, size_t end, std::vector<PackTile>& out, int& nt1, int& nt9) {
    nt1 = nt9 = 0;
    for (size_t i = begin; i < end; ++i) v[i].tiled = 0;
    if (dtype == DT_F32 || !pack_tiles_on) return;
    std::vector<PackTile> t9;
    for (size_t i = begin; i < end;) {
      const PackDesc a = v[i];
      auto same = [&](const PackDesc& b) {
        return b.w == a.w && b.sn == a.sn && b.sk == a.sk && b.N == a.N && b.Npad == a.Npad && b.nseg == 1 && b.seg[0].Cpad == a.seg[0].Cpad &&
               b.seg[0].Creal == a.seg[0].Creal && b.seg[0].koff == a.seg[0].koff && (b.dpack != nullptr) == (a.dpack != nullptr) &&
               (b.gw != nullptr) == (a.gw != nullptr) && b.gw == a.gw;
      };
      size_t j = i + 1;
      while (j < end && j - i < 4 && a.nseg == 1 && same(v[j])) ++j;
      const int rs = a.rs;
      const int cls = a.sk == rs ? 1 : (a.sn == rs ? 2 : 0);
      bool ok = cls != 0 && a.nseg == 1 && a.seg[0].koff == 0 && a.seg[0].Cpad % 32 == 0 && (rs == 1 || rs == 9) && a.st == 1 && a.Npad % 32 == 0;
      // the segment spans the master: rows of Creal channels x rs taps (class 1), N rows per channel (class 2)
      ok = ok && (cls == 1 ? a.sn == (long long)a.seg[0].Creal * rs : a.sk == (long long)a.N * rs);
      unsigned covered = 0;
      bool exact = true;   // no master tap is written twice (unpack: plain LDS stores instead of atomics)
      for (size_t k = i; k < j && ok; ++k) {
        if (v[k].shared_master) ok = false;
        for (int t = 0; t < v[k].seg[0].ntaps; ++t)
          for (int q = 0; q < 4; ++q) {
            const unsigned mt = (v[k].seg[0].tapw[t] >> (8 * q)) & 0xff;
            if (mt == 0xff) continue;
            if ((int)mt >= rs) { ok = false; continue; }
            if (covered & (1u << mt)) exact = false;
            covered |= 1u << mt;
          }
      }
      ok = ok && covered == (1u << rs) - 1u;
      if (ok) {
        for (size_t k = i; k < j; ++k) v[k].tiled = cls;
        for (int n0 = 0; n0 < a.Npad; n0 += 32)
          for (int cg = 0; cg < a.seg[0].Cpad / 32; ++cg) {
            const PackTile pt{(int)i, (int)(j - i) | (exact ? 0x100 : 0), n0, cg};
            if (rs == 1) { out.push_back(pt); ++nt1; } else { t9.push_back(pt); ++nt9; }
          }
      }
      i = j;
    }
    out.insert(out.end(), t9.begin(), t9.end());
  }
  static int pack_rows(const PackDesc& pd) {  // rows the generic kernels walk for this descriptor (none once the tile kernels have it)
    if (pd.tiled) return 0;
    int chunks = 0;
    for (int s = 0; s < pd.nseg; ++s) chunks += pd.seg[s].nchunks;
    return chunks * pd.Npad;
  }
  PackTile* ptiles_dev = nullptr;
  PackTile* utiles_dev = nullptr;
  int pt_early1 = 0, pt_early9 = 0, pt_late1 = 0, pt_late9 = 0;

  // -------------------------------------------------------------------------------- launch lists
  void emit_convert(std::vector<int>& idx) {
    auto one = [&](int buf, int c1, int c2, int which) {
      Op& o = push(OP_CONVERT);
      ConvertArgs& a = o.cv;
      memset(&a, 0, sizeof(a));
      a.C1 = c1; a.C2 = c2;
      a.scale = 1.0f;
      a.dst = bufs[buf].x;
      a.B = bufs[buf].B; a.H = bufs[buf].H; a.W = bufs[buf].W;
      a.stat_sum = bufs[buf].ssum; a.stat_sq = bufs[buf].ssq;
      o.epi = which;  // 0: (s1), 1: (s1,s2), 2: (s2)
      tag(o, "convert", "input", 0, (double)a.B * a.H * a.W * ((c1 + c2) * 4.0 + 8.0 * esz));
      idx.push_back((int)ops->size() - 1);
    };
    if (g.fusion == 0) one(in1, g.s1, 0, 0);
    else if (g.fusion == 1) one(in1, g.s1, g.s2, 1);
    else { one(in1, g.s1, 0, 0); one(in2, g.s2, 0, 2); one(inH, g.s1, g.s2, 1); }
  }

  void emit_pack_op(int kind) {
    if (pack_split > 0) {  // the early layers' weights, on the launch stream
      Op& o = push(kind);
      o.pk.descs = pack_dev;
      o.pk.prefix = prefix_dev;
      o.pk.ndesc = pack_split;
      o.pk.total_rows = pack_rows0;
      o.pk.grad_scale = 1.0f / d.loss_scale;
      o.pk.tdescs = pack_dev; o.pk.tiles = ptiles_dev; o.pk.nt1 = pt_early1; o.pk.nt9 = pt_early9;
      tag(o, "pack", "early", 0, 0);
    }
    Op& o = push(kind);
    if (pack_split > 0) o.leaf = 2;  // pack stream; OP_JOIN (epi 1) in front of record pack_cut_rec (emit_forward_records)
    o.pk.descs = pack_dev + pack_split;
    o.pk.prefix = prefix_dev + pack_split;
    o.pk.ndesc = (int)P.packs.size() - pack_split;
    o.pk.total_rows = total_rows;
    o.pk.grad_scale = 1.0f / d.loss_scale;
    o.pk.tdescs = pack_dev; o.pk.tiles = ptiles_dev + pt_early1 + pt_early9; o.pk.nt1 = pt_late1; o.pk.nt9 = pt_late9;
    tag(o, kind == OP_PACK ? "pack" : "unpack", "weights", 0, (double)P.nparams * (4.0 + esz) * 2.0);
  }
  // Mid fusion: the second stream's encoder (records [0, s2_recs)) shares nothing with the first stream's until the concat module
  // (record concat_rec) reads both.  Its forward launches go to the side stream beside the first stream's: each chain alone leaves
  // the chip idle between its dependent launches (finalize steps, small late-block grids).
  // Round 4: the two encoders' records are emitted ALTERNATELY and the second stream's launches continue one chain on the side stream
  // (leaf 3: no fork event per launch).  Emitted one encoder after the other, the host fed ~1 ms of side-stream launches before the
  // first stream's first launch reached the queue; together with the join behind the full weight pack the main queue sat idle for
  // 2.2 ms at the start of every C3/C4/C5 step (tools/probes/dump_first.py; profiles/r04/ablations.txt section 8).
  size_t s2_recs = 0, concat_rec = 0;
  const bool s2_overlap = !lab_flag("DMM_NO_S2_OVERLAP");  // lab knob
  const bool s2_interleave = !P.sw.no_s2_interleave;
  void emit_record(size_t ri, bool on_side) {
    const Rec& r = recs[ri];
    if (pack_split > 0 && ri == pack_cut_rec) { Op& o = push(OP_JOIN); o.epi = 1; tag(o, "other", "join.pack", 0, 0); }
    const size_t first = ops->size();
    leaf_scope = on_side;
    if (r.type == 0) emit_conv_fwd(convs[r.idx]); else emit_pool_fwd(pools[r.idx]);
    leaf_scope = false;
    if (on_side && s2_interleave)
      for (size_t i = first; i < ops->size(); ++i)
        if ((*ops)[i].leaf == 1) { if (s2_chain_started) (*ops)[i].leaf = 3; s2_chain_started = true; }
  }
  bool s2_chain_started = false;
  void emit_forward_records() {
    const bool beside = g.fusion == 2 && s2_overlap && s2_recs > 0;
    s2_chain_started = false;
    size_t ri = 0;
    if (beside && s2_interleave && pack_cut_rec >= concat_rec) {
      size_t a = s2_recs, b = 0;  // the first stream's next record, the second stream's
      while (a < concat_rec || b < s2_recs) {
        if (a < concat_rec) emit_record(a++, false);
        if (b < s2_recs) emit_record(b++, true);
      }
      ri = concat_rec;
    }
    for (; ri < recs.size(); ++ri) {
      if (beside && ri == concat_rec) { Op& o = push(OP_JOIN); tag(o, "other", "join", 0, 0); }
      emit_record(ri, beside && ri < s2_recs);
    }
  }

  // ---------------------------------------------------------------- gradient buckets (data-parallel overlap)
  struct BucketRec {
    int64_t off = 0, n = 0;
    int convs_left = 0, bns_left = 0;
    int first_desc = 0, ndesc = 0, rows = 0;
    int tile0 = 0, nt1 = 0, nt9 = 0;   // the bucket's range of P.unpack_tiles (descriptor indices relative to P.unpacks)
    int last_main = -1, last_side = -1;
  };
  std::vector<BucketRec> brecs;
  PackDesc* unpack_dev = nullptr;
  int* unprefix_dev = nullptr;
  int bucket_of(int64_t off) const {
    for (size_t b = 0; b < brecs.size(); ++b)
      if (off >= brecs[b].off && off < brecs[b].off + brecs[b].n) return (int)b;
    throw std::runtime_error("gradient offset outside every bucket");
  }
  void make_buckets() {
    // Whole tensors in .parameters() order (a bucket is one contiguous slice of the arena), closed
    //   * once they reach the target size;
    //   * where backward finishes a part of the network long before the next one: at the boundaries of the top-level modules
    //     (features, stream_2_features, concat_module, decoder, head) and, INSIDE an encoder, of its dense blocks / transitions, as
    //     soon as >= 4 MB (1 MB at a top-level boundary) have gathered.  Backward walks an encoder from block 4 down to conv0, i.e. from the END of its slice to the
    //     front: without these cuts conv0 ... most of block 4 of DenseNet-121 were ONE bucket that became ready with conv0's weight
    //     gradient, the very last launch (round 3: bucket_mb [0.3, 17.1, 41.4, 25.2], 25 MB behind the end of backward); with them
    //     the last bucket is stem + blocks 1-2 (5 MB);
    //   * in front of and behind a single tensor of at least the target size (the first ConvTranspose of the decoder is 38 MB).
    brecs.clear();
    const int64_t target = (int64_t)(P.bucket_bytes / 4);
    const int64_t cut_min = (4 << 20) / 4;
    BucketRec cur;
    std::string group;
    auto close = [&]() { if (cur.n > 0) { brecs.push_back(cur); cur = BucketRec(); } };
    for (auto& t : P.tensors) {
      if (t.kind > DMM_T_BN_BIAS) continue;
      int64_t n = 1;
      for (int k = 0; k < t.ndim; ++k) n *= t.shape[k];
      const size_t d1 = t.name.find('.');
      std::string grp = t.name.substr(0, d1);
      if ((grp == "features" || grp == "stream_2_features") && d1 != std::string::npos)
        grp = t.name.substr(0, t.name.find('.', d1 + 1));  // features.denseblock3, features.transition2, features.conv0 ...
      const bool top_change = grp.substr(0, grp.find('.')) != group.substr(0, group.find('.'));
      if (target > 0 && grp != group && cur.n >= (top_change ? cut_min / 4 : cut_min)) close();
      if (target > 0 && n >= target) close();
      group = grp;
      if (cur.n == 0) cur.off = t.off;
      cur.n += n;
      if (target > 0 && cur.n >= target) close();
    }
    if (cur.n > 0) brecs.push_back(cur);
    for (auto& c : convs) brecs[bucket_of(T(c.wname).off)].convs_left++;
    for (auto& b : bns) brecs[bucket_of(b.dgamma - Pg)].bns_left++;
    // unpack tables: the descriptors that scatter into a master gradient, grouped by bucket
    P.unpacks.clear(); P.unpack_prefix.clear(); P.unpack_tiles.clear();
    for (size_t bi = 0; bi < brecs.size(); ++bi) {
      BucketRec& bk = brecs[bi];
      bk.first_desc = (int)P.unpacks.size();
      bk.rows = 0;
      for (auto& pd : P.packs) {
        if (pd.gw == nullptr || pd.dpack == nullptr || bucket_of(pd.gw - Pg) != (int)bi) continue;
        P.unpacks.push_back(pd);
      }
      bk.ndesc = (int)P.unpacks.size() - bk.first_desc;
      bk.tile0 = (int)P.unpack_tiles.size();
      make_tiles(P.unpacks, (size_t)bk.first_desc, P.unpacks.size(), P.unpack_tiles, bk.nt1, bk.nt9);
      for (size_t k = (size_t)bk.first_desc; k < P.unpacks.size(); ++k) {
        P.unpack_prefix.push_back(bk.rows);
        bk.rows += pack_rows(P.unpacks[k]);
      }
    }
    unpack_dev = wptr<PackDesc>(P.packs.size() + 1);  // sized by the pack count: identical in the sizing and the bound pass
    unprefix_dev = wptr<int>(P.packs.size() + 1);
    utiles_dev = wptr<PackTile>(P.unpack_tiles.size() + 1);
  }
  void note_write(int b, int op_index) {
    const Op& o = (*ops)[op_index];
    if (o.kind == OP_WGRAD || o.kind == OP_UNPACK || o.leaf) brecs[b].last_side = op_index; else brecs[b].last_main = op_index;
  }
  void conv_grad_done(const ConvRec& c) {  // called behind the weight-gradient launch(es) of a convolution
    if (defer_scope) { deferred_convs.push_back(&c); return; }
    const int b = bucket_of(T(c.wname).off);
    BucketRec& bk = brecs[b];
    if (--bk.convs_left > 0 || bk.ndesc == 0) return;   // (ndesc counts the tile kernels' descriptors as well)
    // the bucket's last weight gradient: scatter its packed gradients into the arena right behind it (same stream)
    Op& o = push(OP_UNPACK);
    o.leaf = 1;
    o.pk.descs = unpack_dev + bk.first_desc;
    o.pk.prefix = unprefix_dev + bk.first_desc;
    o.pk.ndesc = bk.ndesc;
    o.pk.total_rows = bk.rows;
    o.pk.grad_scale = 1.0f / d.loss_scale;
    o.pk.tdescs = unpack_dev; o.pk.tiles = utiles_dev + bk.tile0; o.pk.nt1 = bk.nt1; o.pk.nt9 = bk.nt9;
    tag(o, "unpack", "weights", 0, (double)bk.n * (4.0 + 4.0));
    note_write(b, (int)ops->size() - 1);
  }
  void bn_grad_done(int bn, size_t first_op) {  // called behind the bn_bwd_finalize launches [first_op, end) of a BatchNorm
    const int b = bucket_of(bns[bn].dgamma - Pg);
    brecs[b].bns_left--;
    for (size_t i = first_op; i < ops->size(); ++i) note_write(b, (int)i);
  }
  void finish_buckets() {
    P.buckets.clear();
    int nev = 0;
    for (auto& bk : brecs) {
      if (bk.convs_left != 0 || bk.bns_left != 0) throw std::runtime_error("gradient bucket bookkeeping is inconsistent");
      GradBucket gb;
      gb.off = bk.off; gb.n = bk.n;
      if (bk.last_main >= 0) { gb.ev_main = nev++; (*ops)[bk.last_main].signal = gb.ev_main; }
      if (bk.last_side >= 0) { gb.ev_side = nev++; (*ops)[bk.last_side].signal = gb.ev_side; }
      gb.ready = std::max(bk.last_main, bk.last_side);
      P.buckets.push_back(gb);
    }
    std::stable_sort(P.buckets.begin(), P.buckets.end(), [](const GradBucket& a, const GradBucket& b) { return a.ready < b.ready; });
  }

  bool pad_pitch = lab_flag("DMM_PITCH_PAD");  // measured: no effect on MI355X for this access pattern; off
  PackDesc* pack_dev = nullptr;
  int* prefix_dev = nullptr;
  int total_rows = 0;          // rows of the pack launch behind the split (all rows without a split)
  int pack_split = 0, pack_rows0 = 0;
  size_t pack_cut_rec = 0;     // the first forward record whose weights the second pack launch holds
  static constexpr double PACK_EARLY_ELEMS = 3.0e6;

  void emit_all() {
    // pack tables live in the workspace
    // The weights are packed by two launches: the first convolution's (needed at once) on the launch stream, everything else on
    // the side stream beside the input conversion and the stem convolution (pack_split descriptors / pack_rows0 rows in front;
    // the row prefix restarts at 0 behind the split).
    pack_split = 0;
    pack_cut_rec = recs.size();
    if (!recs.empty() && !lab_flag("DMM_NO_PACK_SPLIT")) {
      // The cut: the first record behind the point where PACK_EARLY_ELEMS weight elements have been seen (d121/d201: inside dense
      // block 3), not in front of the concat module.  DMM_PACK_CUT=<record index> moves it (1 = the round-3 split behind the stem).
      size_t cut = recs.size();
      if (P.sw.pack_cut > 0) cut = (size_t)P.sw.pack_cut;
      else {
        double elems = 0;
        for (size_t ri = 0; ri < recs.size(); ++ri) {
          if (recs[ri].type != 0) continue;
          const ConvRec& c = convs[recs[ri].idx];
          elems += (double)c.N * c.Kin * c.R * c.S;
          if (elems > PACK_EARLY_ELEMS && ri + 1 > concat_rec) { cut = ri + 1; break; }
        }
      }
      cut = std::min(cut, recs.size());
      auto packs_of = [&](const ConvRec& c, int& lo, int& hi) {
        for (auto& ph : c.phases) { lo = std::min(lo, ph.pack); hi = std::max(hi, ph.pack); }
        for (int s = 0; s < c.nseg; ++s) if (c.seg[s].dgrad != DG_NONE) { lo = std::min(lo, c.dpack[s]); hi = std::max(hi, c.dpack[s]); }
      };
      int lo0 = 1 << 30, hi0 = -1, lo1 = 1 << 30, hi1 = -1;
      for (size_t ri = 0; ri < recs.size(); ++ri)
        if (recs[ri].type == 0) { if (ri < cut) packs_of(convs[recs[ri].idx], lo0, hi0); else packs_of(convs[recs[ri].idx], lo1, hi1); }
      // (the descriptors of the early records must be a prefix of the table: they are, records and packs are created in one order)
      if (hi0 >= 0 && hi1 >= 0 && hi0 < lo1 && hi0 + 1 < (int)P.packs.size()) { pack_split = hi0 + 1; pack_cut_rec = cut; }
    }
    P.pack_tiles.clear();
    make_tiles(P.packs, 0, (size_t)pack_split, P.pack_tiles, pt_early1, pt_early9);
    make_tiles(P.packs, (size_t)pack_split, P.packs.size(), P.pack_tiles, pt_late1, pt_late9);
    P.pack_prefix.clear();
    total_rows = 0;
    pack_rows0 = 0;
    for (size_t i = 0; i < P.packs.size(); ++i) {
      const PackDesc& pd = P.packs[i];
      if ((int)i == pack_split && pack_split > 0) { pack_rows0 = total_rows; total_rows = 0; }
      P.pack_prefix.push_back(total_rows);
      total_rows += pack_rows(pd);
    }
    pack_dev = wptr<PackDesc>(P.packs.size());
    prefix_dev = wptr<int>(P.packs.size());
    ptiles_dev = wptr<PackTile>(P.pack_tiles.size() + 1);
    P.metrics_bytes = (size_t)(2 * g.nc + (size_t)d.batch * 2 * g.nc) * sizeof(double);
    P.metrics = zbptr<double>(P.metrics_bytes / sizeof(double));

    make_buckets();
    std::vector<Op> dummy;
    // ---- training forward ----
    ops = sizing ? &dummy : &P.fwd_train;
    ops->clear();
    training = true;
    { Op& o = push(OP_MEMSET); o.ms.p = zbase; o.ms.bytes = 0; /* patched below */ }
    emit_pack_op(OP_PACK);
    P.convert_ops_train.clear();
    emit_convert(P.convert_ops_train);
    emit_forward_records();
    // ---- eval forward ----
    ops = sizing ? &dummy : &P.fwd_eval;
    ops->clear();
    training = false;
    emit_pack_op(OP_PACK);
    P.convert_ops_eval.clear();
    emit_convert(P.convert_ops_eval);
    emit_forward_records();
    // ---- loss + backward ----
    ops = sizing ? &dummy : &P.bwd;
    ops->clear();
    training = true;
    { Op& o = push(OP_MEMSET); o.ms.p = zbbase; o.ms.bytes = 0; /* patched in plan_bind */ }
    { Op& o = push(OP_MEMSET); o.ms.p = Pg; o.ms.bytes = (size_t)P.nparams * sizeof(float); }  // unpack may accumulate
    {
      Op& o = push(OP_BCE);
      BceArgs& a = o.bce;
      memset(&a, 0, sizeof(a));
      a.dlogits = bufs[dl].x;
      a.out = P.metrics;
      a.B = d.batch; a.NC = g.nc; a.H = d.height; a.W = d.width;
      a.thr = d.iou_threshold; a.loss_scale = d.loss_scale;
      a.metrics = 1;
      tag(o, "bce", "loss", 0, (double)a.B * a.H * a.W * (a.NC * 8.0 + 8.0 * esz));
      P.bce_op = (int)ops->size() - 1;
      P.bce_only = o;
      P.bce_only.bce.dlogits = nullptr;
    }
    for (auto& b : bufs) { b.ginit = false; b.materialized = false; b.mat_front = 0; }
    for (int i = (int)recs.size() - 1; i >= 0; --i) {
      if (recs[i].type == 0) emit_conv_bwd(convs[recs[i].idx]); else emit_pool_bwd(pools[recs[i].idx]);
    }
    flush_deferred();
    finish_buckets();
  }
};

}  // namespace
}  // namespace dmm

// ---------------------------------------------------------------------------------------------------- C-level plan API
using namespace dmm;

PlanSwitches PlanSwitches::from_environment() {
  PlanSwitches s;
  auto on = [](const char* n) { const char* v = getenv(n); return v != nullptr && !(v[0] == '0' && v[1] == 0); };
  s.no_pack_tiles = on("DMM_NO_PACK_TILES");
  s.no_hf = on("DMM_NO_HF");
  s.no_c3_merge = on("DMM_NO_C3_MERGE");
  s.no_cvp_merge = on("DMM_NO_CVP_MERGE");
  s.no_wgp_merge = on("DMM_NO_WGP_MERGE");
  s.no_two_pass = on("DMM_NO_TWO_PASS");
  s.no_eff_compact = on("DMM_NO_EFF_COMPACT");
  s.no_s2_interleave = on("DMM_NO_S2_INTERLEAVE");
  s.defer_wgrad = on("DMM_DEFER_WGRAD");
  s.no_raw_stats = on("DMM_NO_RAW_STATS");
  s.no_r1_stats = on("DMM_NO_R1_STATS");
  const char* pc = getenv("DMM_PACK_CUT");
  s.pack_cut = pc ? std::max(1, atoi(pc)) : 0;
  return s;
}

namespace {
// families the plan's switches rule out, in force (thread-local launch control) while a Builder picks kernel families
struct DenyScope {
  unsigned saved;
  explicit DenyScope(const PlanSwitches& sw) : saved(g_ctl.deny) { if (sw.no_hf) g_ctl.deny |= 1u << IMPL_HF; }
  ~DenyScope() { g_ctl.deny = saved; }
};
}  // namespace

void plan_build_tables(dmm_plan* p) {
  build_tensor_table(p->desc, p->tensors, p->nparams, p->nbuf);
  DenyScope deny(p->sw);
  Builder b(*p, true, nullptr);
  b.build();
  b.emit_all();
  p->zero_bytes = (b.Z.off + 255) / 256 * 256;
  p->zero_bwd_bytes = (b.ZB.off + 255) / 256 * 256;
  p->main_bytes = (b.W.off + 255) / 256 * 256;
  p->fwd_flops = b.flops;
  p->packs.clear();
}

void plan_bind(dmm_plan* p, void* ws) {
  p->packs.clear();
  p->fwd_train.clear(); p->fwd_eval.clear(); p->bwd.clear();
  DenyScope deny(p->sw);
  Builder b(*p, false, (uint8_t*)ws);
  b.build();
  b.emit_all();
  {  // the bound pass must have taken exactly what the sizing pass reported: anything else means launches that write outside the workspace
    const size_t z = (b.Z.off + 255) / 256 * 256, zb = (b.ZB.off + 255) / 256 * 256, w = (b.W.off + 255) / 256 * 256;
    if (z != p->zero_bytes || zb != p->zero_bwd_bytes || w != p->main_bytes) {
      p->fwd_train.clear(); p->fwd_eval.clear(); p->bwd.clear();
      throw dmm::plan_sizing_error("plan_bind: the bound pass took " + std::to_string(z) + " / " + std::to_string(zb) + " / " + std::to_string(w) +
                             " bytes of the three workspace regions, the sizing pass " + std::to_string(p->zero_bytes) + " / " +
                             std::to_string(p->zero_bwd_bytes) + " / " + std::to_string(p->main_bytes) +
                             " (an environment switch changed between dmm_plan_create and dmm_plan_bind?)");
    }
  }
  // the training forward starts by zeroing the whole accumulator region
  p->fwd_train[0].ms.p = ws;
  p->fwd_train[0].ms.bytes = p->zero_bytes;
  // ... and the backward by zeroing its own accumulators, so it can be repeated after one forward
  p->bwd[0].ms.p = (uint8_t*)ws + p->zero_bytes;
  p->bwd[0].ms.bytes = p->zero_bwd_bytes;
  p->ws = ws;
  // zero everything once: padding lanes of scale/shift tables and NHWC8 pad channels must read as 0
  hipMemset(ws, 0, p->zero_bytes + p->zero_bwd_bytes + p->main_bytes);
  // upload the pack tables
  hipMemcpy(b.pack_dev, p->packs.data(), p->packs.size() * sizeof(PackDesc), hipMemcpyHostToDevice);
  hipMemcpy(b.prefix_dev, p->pack_prefix.data(), p->pack_prefix.size() * sizeof(int), hipMemcpyHostToDevice);
  if (!p->pack_tiles.empty()) hipMemcpy(b.ptiles_dev, p->pack_tiles.data(), p->pack_tiles.size() * sizeof(PackTile), hipMemcpyHostToDevice);
  if (!p->unpack_tiles.empty()) hipMemcpy(b.utiles_dev, p->unpack_tiles.data(), p->unpack_tiles.size() * sizeof(PackTile), hipMemcpyHostToDevice);
  if (!p->unpacks.empty()) {
    hipMemcpy(b.unpack_dev, p->unpacks.data(), p->unpacks.size() * sizeof(PackDesc), hipMemcpyHostToDevice);
    hipMemcpy(b.unprefix_dev, p->unpack_prefix.data(), p->unpack_prefix.size() * sizeof(int), hipMemcpyHostToDevice);
  }
}
