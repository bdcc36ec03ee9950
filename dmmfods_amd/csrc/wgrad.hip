// Weight-gradient GEMM for gfx950:  dP[chunk][n][k] += sum_m P[m][n] * Q[m][k]
//   Both operands come through gather_slot (gather.h): P is pixel-aligned (one tap), Q carries the taps.
//   standard form   : P = dYeff (output gradient + deferred BatchNorm-backward correction), Q = the forward conv's
//                     gathered operand A (same taps, BN+ReLU prologue, modes); dP is laid out like the forward pack.
//   transposed form : P = A at the pixel (BN+ReLU applied ONCE per pixel), Q = dYeff gathered with the flipped taps,
//                     dW[n][c][tap] = sum_m' dY[m'-tap][n] * A[m'][c]; dP is laid out like the dgrad pack.  Chosen by the
//                     plan when the output is thin (N*taps < Cin*taps), e.g. the 3x3 growth convs (N = 32) and the
//                     5x5 logits conv (N = 3): the per-tap prologue work moves from Cin to N channels.
//   dP   = gradient w.r.t. the PACKED weights (fp32, layout [chunk][Npad][BK]); unpack_grads scatters it back.
// The contraction index is the pixel m, which is the slow (strided) index of both NHWC operands, so both MFMA
// operands are read from row-major LDS tiles with the hardware transposing read ds_read_b64_tr_b16 (16-bit
// types) or plain ds_read_b32 (fp32).  A workgroup owns an n-tile (<=128) x 128 k-elements of dP and a range
// of rows; partial sums are added to dP with fp32 atomics (128-byte contiguous segments per wave instruction).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gather.h"

namespace dmm {

#ifndef WGRAD_DIST_HEAVY
#define WGRAD_DIST_HEAVY 1
#endif
#ifndef WGRAD_DBG
#define WGRAD_DBG 0  // timing experiments only: 1 skip atomics, 2 skip MFMA, 4 skip tile loads
#endif
constexpr int KW = 128;  // k elements per workgroup

template <typename T> struct WgCfg;
template <> struct WgCfg<f16> { static constexpr int BMW = 64; };
template <> struct WgCfg<bf16> { static constexpr int BMW = 64; };
template <> struct WgCfg<float> { static constexpr int BMW = 32; };

template <typename T, int WBN>
struct WgradSmem {
  static constexpr int BMW = WgCfg<T>::BMW;
  // 256-byte rows of 16-bit elements (128 columns) are stored unpadded with the 64-byte granule index XOR-ed with (row & 3); other row
  // lengths are padded so that (pitch/4) mod 64 is 16 or 48.  Either way 4 consecutive rows of a transposed read tile
  // the banks.
  static constexpr int pitch_for(int row_bytes) {
    if (row_bytes == 256 && sizeof(T) == 2) return 256;
    int p = row_bytes;
    while (((p / 4) % 64) != 16 && ((p / 4) % 64) != 48) p += 16;
    return p;
  }
  static constexpr int PD = pitch_for(WBN * (int)sizeof(T));
  static constexpr int PA = pitch_for(KW * (int)sizeof(T));
  static constexpr bool SWZ_D = PD == 256 && sizeof(T) == 2, SWZ_A = PA == 256 && sizeof(T) == 2;
  static constexpr int D_BYTES = BMW * PD;
  static constexpr int A_BYTES = BMW * PA;
  static constexpr int BUF = D_BYTES + A_BYTES;  // one of the two operand buffers
  static constexpr int TAB = 2 * BMW * 16;       // two row tables of int4 {b, y, x, valid}
  static constexpr int bytes = 2 * BUF + TAB;
};

// ds_read_b64_tr_b16 moves 16-bit elements whatever they encode: the result is handed back as two dwords
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 lds_tr16(const unsigned char* p) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(p));
  return __builtin_bit_cast(u32x2, r);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::vec frag16(const u32x2& lo, const u32x2& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(typename TT<T>::vec, v);
}

// PP / PQ = prologue kind of the pixel-aligned operand P and of the tapped operand Q (-1 run time, 0 none, 1 BN+ReLU,
// 2 effective gradient); LIN = both operands are plain one-tap unit-stride tensors on the row grid (1x1 layers): the
// source pixel is the row index, no row table and no coordinate arithmetic.
// Pipeline: two LDS operand buffers (one barrier per row tile) and DIST register sets: the loads of tile t + DIST are
// issued right after the barrier of tile t and are consumed (prologue + ds_write) DIST iterations later.
template <typename T, int WBN, bool MFMA, int PP, int PQ, bool LIN, int DIST>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int SLOT = TT<T>::SLOT;
  constexpr int BK = 4 * SLOT;
  constexpr int KCH = KW / BK;  // chunks per workgroup
  typedef typename TT<T>::vec V;
  typedef WgradSmem<T, WBN> SM;
  constexpr int BMW = SM::BMW;
  constexpr int WAVES_N = WBN / 32, WAVES_K = 4 / WAVES_N;
  constexpr int TPW = (KW / 32) / WAVES_K;  // 32x32 k-tiles per wave
  constexpr int NCA = KW / SLOT, RGA = NTHREADS / NCA, LA = BMW / RGA;
  constexpr int NCD = WBN / SLOT, RGD = NTHREADS / NCD, LD = (BMW + RGD - 1) / RGD;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int4* rowtab = (int4*)(smem + 2 * SM::BUF);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = a.Npad / WBN;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = bid % ntiles; bid /= ntiles;
  const int kg = bid % a.kgroups;
  const int split = bid / a.kgroups;
  const int n0 = ntile * WBN;
  const int mbeg = split * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);
  if (mbeg >= mend) return;

  // ---- this thread's fixed K position for the A tile ----
  const int ca = tid % NCA, rga = tid / NCA;
  int total = 0;
  for (int s = 0; s < a.nseg; ++s) total += a.seg[s].nchunks;
  int ks = 0, ktap = 1 << 20, kc = 0;
  {
    int gc = kg * KCH + (ca * SLOT) / BK;
    if (gc < total) {
      int s = 0;
      while (gc >= a.seg[s].nchunks) { gc -= a.seg[s].nchunks; ++s; }
      const int e = gc * BK + (ca * SLOT) % BK;
      ks = s;
      ktap = e / a.seg[s].Cpad;
      kc = e - ktap * a.seg[s].Cpad;
    }
  }
  const int cd = tid % NCD, rgd = tid / NCD;
  const int nD = n0 + cd * SLOT;
  // every thread keeps its channel position for the whole kernel: load the prologue constants once
  const SlotK<SLOT> preQ = load_slot_consts<SLOT>(a.seg[ks], ktap < a.seg[ks].ntaps ? kc : -1);
  const SlotK<SLOT> preP = load_slot_consts<SLOT>(a.dy, nD < a.N ? nD : -1);

  // row tables (non-LIN): lane tid < BMW walks its row (b, y, x) from tile to tile without divisions
  int wb = 0, wy = 0, wx = 0, wm = mbeg + tid;
  if (!LIN && tid < BMW) row_to_byx(min(wm, a.M - 1), a.Ho, a.Wo, wb, wy, wx);
  auto fill_rowtab = [&](int slot) {  // writes the table of the walker's current tile, then advances one tile
    if (tid < BMW) {
      rowtab[slot * BMW + tid] = wm < mend ? make_int4(wb, wy, wx, 1) : make_int4(0, 0, 0, -1);
      wm += BMW;
      wx += BMW;
      while (wx >= a.Wo) {
        wx -= a.Wo;
        if (++wy >= a.Ho) { wy = 0; ++wb; }
      }
    }
  };

  // issue-early / write-late (see igemm.hip): raw loads now, prologue when the tile is written to LDS DIST steps later
  struct Ring {
    RawSlot<T> q[LA], p[LD];
  };
  Ring R0, R1;
  // The tapped operand's segment depends on the thread's K position, so a.seg[ks] is a per-LANE address: everything the
  // tile loop needs from it is copied into registers here (left in the loop, every field access is a vector load from
  // the kernel-argument segment in front of the data load that depends on it).
  const Seg& sgq = a.seg[ks];
  const bool qinside = ktap < sgq.ntaps && kc < sgq.C;
  const int qnarr = sgq.scale ? 2 : (sgq.q ? 4 : 0);
  const bool qtwo = PQ == 2 || (PQ < 0 && qnarr == 4);
  const T* const qsrc = (const T*)sgq.src + kc;
  const T* const qsrc2 = (const T*)sgq.src2 + kc;
  const int qld = sgq.ld, qld2 = sgq.ld2, qHs = sgq.Hs, qWs = sgq.Ws, qistr = sgq.istride;
  const bool qpool = sgq.mode == G_POOL2;
  const int qup = sgq.mode == G_UP2 ? 1 : 0;
  int qdy = 0, qdx = 0;
  if (qinside) {
    const int t = sgq.taps[ktap];
    qdy = (int)(signed char)(t & 0xff);
    qdx = (int)(signed char)((t >> 8) & 0xff);
  }
  // the pixel-aligned operand: one segment for all threads (wave-uniform fields)
  const int pnarr = a.dy.scale ? 2 : (a.dy.q ? 4 : 0);
  const bool ptwo = PP == 2 || (PP < 0 && pnarr == 4);
  const bool pvalid = nD < a.N;
  const T* const psrc = (const T*)a.dy.src + (pvalid ? nD : 0);
  const T* const psrc2 = (const T*)a.dy.src2 + (pvalid ? nD : 0);
  int pdy, pdx;
  {
    const int t = a.dy.taps[0];
    pdy = (int)(signed char)(t & 0xff);
    pdx = (int)(signed char)((t >> 8) & 0xff);
  }

  auto zero_slot = [&](RawSlot<T>& rs) {
#pragma unroll
    for (int e = 0; e < SLOT; ++e) { rs.v[e] = (T)0; rs.v2[e] = (T)0; }
    rs.state = 0;
  };
  auto issue = [&](Ring& R, int mt, int slot) {
    if constexpr (LIN) {
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        const int m = mt + rga + i * RGA;
        zero_slot(R.q[i]);
        if (qinside && m < mend) {
          R.q[i].v = *(const V*)(qsrc + (size_t)m * qld);
          if (qtwo) R.q[i].v2 = *(const V*)(qsrc2 + (size_t)m * qld2);
          R.q[i].state = 1;
        }
      }
#pragma unroll
      for (int i = 0; i < LD; ++i) {
        const int row = rgd + i * RGD, m = mt + row;
        zero_slot(R.p[i]);
        if (pvalid && row < BMW && m < mend) {
          R.p[i].v = *(const V*)(psrc + (size_t)m * a.dy.ld);
          if (ptwo) R.p[i].v2 = *(const V*)(psrc2 + (size_t)m * a.dy.ld2);
          R.p[i].state = 1;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int4 e = rowtab[slot * BMW + rga + i * RGA];
      if (qpool) {  // pooled transition input: four loads + BN + averaging, done synchronously (rare)
        R.q[i] = gather_issue<T>(sgq, e.x, e.y, e.z, e.w > 0, ktap, kc, preQ);
        continue;
      }
      zero_slot(R.q[i]);
      const int sy = e.y * qistr + qdy, sx = e.z * qistr + qdx;
      if (qinside && e.w > 0 && (unsigned)sy < ((unsigned)qHs << qup) && (unsigned)sx < ((unsigned)qWs << qup)) {
        const size_t pix = (size_t)((e.x * qHs + (sy >> qup)) * qWs + (sx >> qup));
        R.q[i].v = *(const V*)(qsrc + pix * qld);
        if (qtwo) R.q[i].v2 = *(const V*)(qsrc2 + pix * qld2);
        R.q[i].state = 1;
      }
    }
    const int pup = a.dy.mode == G_UP2 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int row = rgd + i * RGD;
      int4 e = {0, 0, 0, 0};
      if (row < BMW) e = rowtab[slot * BMW + row];
      zero_slot(R.p[i]);
      const int sy = e.y * a.dy.istride + pdy, sx = e.z * a.dy.istride + pdx;
      if (pvalid && e.w > 0 && (unsigned)sy < ((unsigned)a.dy.Hs << pup) && (unsigned)sx < ((unsigned)a.dy.Ws << pup)) {
        const size_t pix = (size_t)((e.x * a.dy.Hs + (sy >> pup)) * a.dy.Ws + (sx >> pup));
        R.p[i].v = *(const V*)(psrc + pix * a.dy.ld);
        if (ptwo) R.p[i].v2 = *(const V*)(psrc2 + pix * a.dy.ld2);
        R.p[i].state = 1;
      }
    }
  };
  auto store = [&](const Ring& R, int buf) {
    unsigned char* Ds = smem + buf * SM::BUF;
    unsigned char* As = Ds + SM::D_BYTES;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int row = rga + i * RGA;
      const int col = SM::SWZ_A ? ((ca * 16) ^ ((row & 3) << 6)) : ca * 16;
      *(V*)(As + row * SM::PA + col) = (WGRAD_DBG & 8) ? R.q[i].v : finish_slot<T, PQ>(qnarr, R.q[i], preQ);
    }
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int row = rgd + i * RGD;
      const int col = SM::SWZ_D ? ((cd * 16) ^ ((row & 3) << 6)) : cd * 16;
      if (row < BMW) *(V*)(Ds + row * SM::PD + col) = (WGRAD_DBG & 8) ? R.p[i].v : finish_slot<T, PP>(pnarr, R.p[i], preP);
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int wn = wave % WAVES_N, wk = wave / WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  // transposed-read lane geometry (16-bit types): group g = lane>>4 covers columns 16*(g&1).., rows 8*(g>>1)..
  const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;

  auto mma = [&](int buf) {
    const unsigned char* Ds = smem + buf * SM::BUF;
    const unsigned char* As = Ds + SM::D_BYTES;
    if (MFMA && !(WGRAD_DBG & (2 | 32))) {
      if constexpr (sizeof(T) == 2) {
        // rows read by this lane are 16 ms + 8 (tg >> 1) + tq (+4): row & 3 == tq, so the swizzle term is a lane constant
        const int dcol = (32 * wn + 16 * (tg & 1) + 4 * tp) * 2;
        const int dsw = SM::SWZ_D ? (dcol ^ (tq << 6)) : dcol;
#pragma unroll
        for (int ms = 0; ms < BMW / 16; ++ms) {
          const int rowb = 16 * ms + 8 * (tg >> 1) + tq;
          const unsigned char* dp = Ds + rowb * SM::PD + dsw;
          const V af = frag16<T>(lds_tr16(dp), lds_tr16(dp + 4 * SM::PD));
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int kt = wk * TPW + t;
            const int acol = (32 * kt + 16 * (tg & 1) + 4 * tp) * 2;
            const unsigned char* ap = As + rowb * SM::PA + (SM::SWZ_A ? (acol ^ (tq << 6)) : acol);
            const V bf = frag16<T>(lds_tr16(ap), lds_tr16(ap + 4 * SM::PA));
            acc[t] = mma16(af, bf, acc[t]);
          }
        }
      } else {
#pragma unroll 4
        for (int ms = 0; ms < BMW / 2; ++ms) {
          const int row = 2 * ms + h;
          const float av = *(const float*)(Ds + row * SM::PD + (32 * wn + r) * 4);
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int kt = wk * TPW + t;
            const float bv = *(const float*)(As + row * SM::PA + (32 * kt + r) * 4);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    } else if (!MFMA) {
      // scalar check path, same accumulator layout: acc[t][i] <-> (n = 32*wn + rowmap(i), k = 32*kt + r)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int kt = wk * TPW + t;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int nrow = 32 * wn + (i & 3) + 8 * (i >> 2) + 4 * h;
          float s = 0.f;
          for (int m = 0; m < BMW; ++m) {
            const int dc = nrow * (int)sizeof(T), ac = (32 * kt + r) * (int)sizeof(T);
            s = fmaf(to_f32(*(const T*)(Ds + m * SM::PD + (SM::SWZ_D ? (dc ^ ((m & 3) << 6)) : dc))),
                     to_f32(*(const T*)(As + m * SM::PA + (SM::SWZ_A ? (ac ^ ((m & 3) << 6)) : ac))), s);
          }
          acc[t][i] += s;
        }
      }
    }
  };

  // tiles are processed in pairs (one per register set / LDS buffer); a tile past mend gathers zeros
  const int npairs = ((mend - mbeg) + 2 * BMW - 1) / (2 * BMW);
  Ring& RA = R0;
  Ring& RB = DIST == 2 ? R1 : R0;
  if constexpr (!LIN) {
    fill_rowtab(0);
    if (DIST == 2) fill_rowtab(1);
    __syncthreads();
  }
  issue(RA, mbeg, 0);
  if (DIST == 2) issue(RB, mbeg + BMW, 1);
  for (int pr = 0; pr < npairs; ++pr) {
    const int mt = (WGRAD_DBG & 16) ? mbeg : mbeg + pr * 2 * BMW;
    // the tile whose loads are issued in this half goes to table slot (tile index & 1); the walker is DIST tiles ahead
    store(RA, 0);
    if constexpr (!LIN) fill_rowtab(DIST == 2 ? 0 : 1);
    __syncthreads();
    if (!(WGRAD_DBG & 4)) issue(RA, mt + DIST * BMW, DIST == 2 ? 0 : 1);
    mma(0);
    if (WGRAD_DBG & 32) acc[0][0] += *(const float*)(smem + lane * 4);
    store(RB, 1);
    if constexpr (!LIN) fill_rowtab(DIST == 2 ? 1 : 0);
    __syncthreads();
    if (!(WGRAD_DBG & 4)) issue(RB, mt + (1 + DIST) * BMW, DIST == 2 ? 1 : 0);
    mma(1);
  }

  // ---- add the partial tile to the packed gradient ----
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int kt = wk * TPW + t;
    const int ke = 32 * kt + r;
    const int chunk = kg * KCH + ke / BK;
    const int kk = ke % BK;
    if (chunk >= total || ((WGRAD_DBG & 1) && acc[t][0] != 123.f)) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + 32 * wn + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (n < a.N) atomic_add_f32(a.dpack + ((size_t)chunk * a.Npad + n) * BK + kk, acc[t][i]);
    }
  }
}

// register prefetch distance per variant: two sets unless the raw operands (two tensors each for the effective
// gradient) would not fit the register budget of two workgroups per CU
constexpr int WDIST(int wbn, int pp, int pq) { return ((pp == 2 || pq == 2 || pp < 0) && wbn == 128) ? WGRAD_DIST_HEAVY : 2; }

static int seg_pro(const Seg& sg) { return sg.mode == G_POOL2 ? -1 : (sg.scale ? 1 : (sg.q ? 2 : 0)); }
static bool seg_lin(const Seg& sg, const WgradArgs& a) {
  return sg.ntaps == 1 && sg.taps[0] == 0 && sg.mode == G_PLAIN && sg.istride == 1 && sg.Hs == a.Ho && sg.Ws == a.Wo;
}

template <typename T, int WBN>
static hipError_t launch_w(const WgradArgs& a, bool mfma, hipStream_t st) {
  typedef WgradSmem<T, WBN> SM;
  const int ntiles = a.Npad / WBN;
  const int splits = (a.M + a.rows_per_split - 1) / a.rows_per_split;
  dim3 grid(ntiles * a.kgroups * splits), block(NTHREADS);
  int pq = seg_pro(a.seg[0]);
  for (int s = 1; s < a.nseg; ++s) if (seg_pro(a.seg[s]) != pq) pq = -1;
  const int pp = seg_pro(a.dy);
  const bool lin = a.nseg == 1 && seg_lin(a.seg[0], a) && seg_lin(a.dy, a);
  void (*kern)(const WgradArgs);
  int ai;
  if (!mfma) {
    if constexpr (std::is_same<T, bf16>::value) return hipErrorNotSupported;  // no scalar check kernels for bf16 (see igemm.hip)
    else { kern = wgrad_kernel<T, WBN, false, -1, -1, false, WDIST(WBN, -1, -1)>; ai = 0; }
  }
  else if (lin && pp == 2 && pq == 1) { kern = wgrad_kernel<T, WBN, true, 2, 1, true, WDIST(WBN, 2, 1)>; ai = 1; }
  else if (lin && pp == 0 && pq == 1) { kern = wgrad_kernel<T, WBN, true, 0, 1, true, WDIST(WBN, 0, 1)>; ai = 7; }
  else if (pp == 0 && pq == 1) { kern = wgrad_kernel<T, WBN, true, 0, 1, false, WDIST(WBN, 0, 1)>; ai = 8; }
  else if (pp == 2 && pq == 1) { kern = wgrad_kernel<T, WBN, true, 2, 1, false, WDIST(WBN, 2, 1)>; ai = 2; }
  else if (pp == 1 && pq == 2) { kern = wgrad_kernel<T, WBN, true, 1, 2, false, WDIST(WBN, 1, 2)>; ai = 3; }
  else if (pp == 2 && pq == 0) { kern = wgrad_kernel<T, WBN, true, 2, 0, false, WDIST(WBN, 2, 0)>; ai = 4; }
  else if (pp == 1 && pq == 0) { kern = wgrad_kernel<T, WBN, true, 1, 0, false, WDIST(WBN, 1, 0)>; ai = 5; }
  else { kern = wgrad_kernel<T, WBN, true, -1, -1, false, WDIST(WBN, -1, -1)>; ai = 6; }
  static bool attr_done[9] = {false, false, false, false, false, false, false, false, false};
  if (SM::bytes > 48 * 1024 && !attr_done[ai]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SM::bytes);
    if (e != hipSuccess) return e;
    attr_done[ai] = true;
  }
  hipLaunchKernelGGL(kern, grid, block, SM::bytes, st, a);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_wt(const WgradArgs& a, bool mfma, hipStream_t st) {
  if (a.Npad % 128 == 0) return launch_w<T, 128>(a, mfma, st);
  if (a.Npad % 64 == 0) return launch_w<T, 64>(a, mfma, st);
  return launch_w<T, 32>(a, mfma, st);
}

// One translation unit per storage type (WGRAD_PART = 0 fp32, 1 f16, 2 bf16; see the Makefile).
#if !defined(WGRAD_PART) || WGRAD_PART == 0
template hipError_t launch_wt<float>(const WgradArgs&, bool, hipStream_t);
#endif
#if !defined(WGRAD_PART) || WGRAD_PART == 1
template hipError_t launch_wt<f16>(const WgradArgs&, bool, hipStream_t);
#endif
#if !defined(WGRAD_PART) || WGRAD_PART == 2
template hipError_t launch_wt<bf16>(const WgradArgs&, bool, hipStream_t);
#endif

#if !defined(WGRAD_PART) || WGRAD_PART == 1
extern template hipError_t launch_wt<float>(const WgradArgs&, bool, hipStream_t);
extern template hipError_t launch_wt<bf16>(const WgradArgs&, bool, hipStream_t);

hipError_t launch_wg3(const WgradArgs& a, int dtype, hipStream_t st);  // wg3.hip
hipError_t launch_wgp(const WgradArgs& a, int dtype, hipStream_t st);  // wgp.hip
hipError_t launch_wg5(const WgradArgs& a, int dtype, hipStream_t st);  // wg5.hip

// Fills rows_per_split / kgroups (if zero) and launches.
// Which family launch_wgrad(..., IMPL_AUTO) would run right now (nothing is launched).
int wgrad_pick(const WgradArgs& a, int dtype, bool mfma) {
  if (!mfma || a.M <= 0) return IMPL_GENERIC;
  const LaunchCtl keep = g_ctl;
  g_ctl.dry = true;
  g_ctl.impl = IMPL_AUTO;
  int took = IMPL_GENERIC;
  if (launch_wg3(a, dtype, nullptr) == hipSuccess) took = IMPL_WG3;
  else if (launch_wg5(a, dtype, nullptr) == hipSuccess) took = IMPL_WG5;
  else if (launch_wgp(a, dtype, nullptr) == hipSuccess) took = IMPL_WGP;
  g_ctl = keep;
  return took;
}

struct CtlScope {  // the family a plan recorded is the only one allowed to take the launch while this is alive
  LaunchCtl keep;
  explicit CtlScope(int impl) : keep(g_ctl) { g_ctl.dry = false; g_ctl.impl = impl; }
  ~CtlScope() { g_ctl = keep; }
};

hipError_t launch_wgrad(WgradArgs a, int dtype, bool mfma, hipStream_t st, int impl) {
  if (a.M <= 0) return hipSuccess;
  const CtlScope scope(impl);
  const bool special = mfma && impl != IMPL_GENERIC;
  bool wgp_took = false;
  if (special) {  // the dense layers' 3x3 growth convolution: persistent tiles, the whole result in registers
    const hipError_t e = launch_wg3(a, dtype, st);
    if (e != hipErrorNotSupported) { note_impl(IMPL_WG3); return e; }
  }
  if (special) {  // the head's 5x5 convolution onto 3 classes: persistent tiles, the 25 x 8 (tap, class) columns as per-lane addresses
    const hipError_t e = launch_wg5(a, dtype, st);
    if (e != hipErrorNotSupported) { note_impl(IMPL_WG5); return e; }
  }
  if (special) {  // parity-phase convolutions (ConvTranspose stages, the head's 3x3 over the upsampled map): all taps of a phase per tile
    const hipError_t e = launch_wgp(a, dtype, st);
    if (e == hipSuccess) {
      note_impl(IMPL_WGP);
      wgp_took = true;
      if (a.nseg == 1) return e;
      // the 8-channel raw-input segment of the head convolution stays with the generic kernel: its chunks follow segment 0's
      a.dpack += (size_t)a.seg[0].nchunks * a.Npad * 32;
      a.seg[0] = a.seg[1];
      a.nseg = 1;
      a.rows_per_split = 0;
    } else if (e != hipErrorNotSupported) return e;
  }
  if (!wgp_took) note_impl(IMPL_GENERIC);  // (wgp for segment 0 + the generic kernel for the raw-input remainder reports wgp)
  const int BK = dtype == DT_F32 ? 16 : 32;
  const int bmw = dtype == DT_F32 ? 32 : 64;
  int total = 0;
  for (int s = 0; s < a.nseg; ++s) total += a.seg[s].nchunks;
  a.kgroups = (total * BK + KW - 1) / KW;
  if (a.rows_per_split <= 0) {
    const int wbn = a.Npad % 128 == 0 ? 128 : (a.Npad % 64 == 0 ? 64 : 32);
    const int base = a.kgroups * (a.Npad / wbn);
    // ~256 workgroups for short row ranges (fewer, longer workgroups also leave the other stream more room) (every workgroup ends with WBN x 128 atomics); long ranges are cut finer, down
    // to 32 row steps per workgroup, which evens out the tail of the launch
    static const int target = lab_int("DMM_WGRAD_WGS", 256);
    const int steps = (a.M + bmw - 1) / bmw;
    const int want = (target + base - 1) / base, want_hi = (8 * target + base - 1) / base;
    int per = (steps + want - 1) / want;
    if (per > 32) per = std::max(32, (steps + want_hi - 1) / want_hi);
    if (per < 4) per = 4;
    a.rows_per_split = per * bmw;
  }
  if (dtype == DT_F16) return launch_wt<f16>(a, mfma, st);
  if (dtype == DT_BF16) return launch_wt<bf16>(a, mfma, st);
  return launch_wt<float>(a, mfma, st);
}
#endif

}  // namespace dmm
