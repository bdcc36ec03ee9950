// Forward of the decoder's ConvTranspose2d 3x3 stride 2 stages (reference M:155-160; four output-parity phases of 1, 2, 2 and 4 taps
// over the input grid), WAVE-SPECIALISED (round 5).  cvp.hip's tile (8 x 16 pixels x 128 output channels), halo image (9 x 17
// pixels x 128 input channels per channel group, a tap is an address offset of the A fragment) and weight stages (two 64-byte K chunks x
// 128 columns = 16 KB) - run by EIGHT waves per workgroup, one workgroup per CU, PERSISTENT over the launch:
//   * waves 0-3 (matrix waves), arranged 2 x 2 over the tile: 64 pixels x 64 columns each (2 x 2 MFMA tiles 32x32x16: one A and one B
//     fragment read per MFMA instead of cvp.hip's 1.25) - nothing but fragment reads and MFMAs between two barriers; the epilogue of a
//     tile is wave-local (a wave stages, reads back and stores its own 64 x 64 block: no barrier) and the BatchNorm sums stay in fp64
//     registers until the column tile changes;
//   * waves 4-7 (loader waves) own every global load of the K loop, issued from inline assembly with counted waits (wg3.hip explains
//     why): the halo of the NEXT channel group (10 slots per thread, BN+ReLU once per element on the way into LDS) and the weight stages
//     two ahead (4 pieces per thread), two halo images and a ring of two weight stages in LDS;
//   * ONE raw s_barrier per weight stage (16 MFMAs per matrix wave) for all eight waves: stage s is complete / stage s - 1 has been
//     read.  The stream of stages runs on across channel groups AND tiles: while the matrix waves are in a tile's epilogue the loaders
//     are already a stage ahead in the next tile.
// cvp.hip spent 10 vector instructions per MFMA (addresses of the weight loads, of the LDS writes, the halo prologue, the epilogue) in
// the same four waves that issue the MFMAs, two workgroups per CU: 570 - 750 TF/s on stages that are 181 GFLOP each (0.23 of the dense
// fp16 peak, profiles/r04).  Same arithmetic and K order as cvp.hip per output element: results are bit-equal to cvp.hip's.
//
// Work order: the launch's (phase, column tile, pixel tile) items in phase-major order, most taps first; workgroup w takes items
// w, w + W, w + 2 W, ... - every workgroup gets its share of every phase (the phases differ 4 : 1 in MFMAs per item), and inside a
// phase walks ONE template instantiation (the loaders' wait counts are compile-time functions of the tap count).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

#ifndef CW_DBG
#define CW_DBG 0  // timing experiments only (tools/build_variant.sh): 1 no MFMA, 2 no weight requests, 4 no halo requests, 8 no epilogue, 16 no fragment reads
#endif
namespace dmm {

constexpr int CW_TH = 8, CW_TW = 16, CW_HH = 9, CW_HW = 17;
constexpr int CW_CA = 128, CW_BN = 128;
constexpr int CW_PP = CW_CA * 2 + 16;                       // pixel pitch of the halo image (an odd number of 16-byte slots)
constexpr int CW_RP = (CW_HW * CW_PP + 255) / 256 * 256;    // 4864
constexpr int CW_X_BYTES = CW_HH * CW_RP;                   // 43 776
constexpr int CW_B_STAGE = 2 * CW_BN * 64;                  // 16 KB: two 64-byte chunks of K for 128 columns
constexpr int CW_SPITCH = 64 + 8;                           // staging pitch of a wave's 32 x 64 half block (elements)
constexpr int CW_STAGE_W = 32 * CW_SPITCH * 2;              // 4608 bytes per matrix wave (its 64 x 64 block goes out in two halves)
constexpr int CW_OFF_B = 2 * CW_X_BYTES;
constexpr int CW_OFF_S = CW_OFF_B + 2 * CW_B_STAGE;
constexpr int CW_OFF_K = CW_OFF_S + 4 * CW_STAGE_W;         // 138 752: the norm's scale | shift of ALL input channels, staged once per workgroup
constexpr int CW_NT = 512;
constexpr int CW_MAX_C = (160 * 1024 - CW_OFF_K) / 8;       // 3136 input channels

struct CwPhase { const void* wpack; short taps[4]; int ntaps, py, px, dymin, dxmin; };
struct CvwArgs {
  ConvArgs c;
  int tiles_y, tiles_x, ntn, ntl;   // pixel tiles per image row / column, 128-column tiles, pixel tiles of the launch
  int per;                          // items per phase = ntl * ntn
  int nphase;                       // 1 or 4
  CwPhase ph[4];                    // in work order: most taps first
};

__device__ __forceinline__ void cw_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename V>
__device__ __forceinline__ void cw_load(V& dst, const void* base, unsigned off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}
#define CW_X10(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])
#define CW_B4(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])
// waits of the loader waves: N = requests that may stay in flight behind the set being waited for (see the schedule in cw_walk)
template <typename V, int N>
__device__ __forceinline__ void cw_wait_halo(V (&rx)[10]) {
  asm volatile("s_waitcnt vmcnt(%10)" : CW_X10(rx) : "n"(N));
}
template <typename V, int N>
__device__ __forceinline__ void cw_wait_b(V (&rb)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : CW_B4(rb) : "n"(N));
}
template <typename V>
__device__ __forceinline__ void cw_hold_x(V (&rx)[10]) { asm volatile("s_waitcnt vmcnt(0)" : CW_X10(rx)); }
template <typename V>
__device__ __forceinline__ void cw_hold_b(V (&rb)[4]) { asm volatile("; hold" : CW_B4(rb)); }

// One phase's share of the launch for this workgroup: items first, first + stride, ... < P.per of a phase with NTAP taps.
template <typename T, int NTAP>
__device__ __forceinline__ void cw_walk(const CvwArgs& g, const CwPhase& P, const int first, const int stride, double (&dsum)[2], int& stat_ntile) {
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  constexpr int NST = 2 * NTAP;   // weight stages per channel group
  const ConvArgs& a = g.c;
  const Seg& sx = a.seg[0];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ngrp = sx.C / CW_CA;
  const int nitems = first < g.per ? (g.per - first + stride - 1) / stride : 0;
  if (nitems == 0) return;                    // (workgroup-uniform)
  const int cpt = sx.Cpad / 32;               // K chunks per tap
  const long nstages = (long)nitems * ngrp * NST;

  if (wave >= 4) {
    // ================================ loader waves ================================
    const int lt = tid - 256;
    const int cx = lt & 15, px0 = lt >> 4;
    int xlds[10];
    unsigned xin = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int hp = px0 + 16 * i;
      const int hpc = min(hp, CW_HH * CW_HW - 1);
      const int hy = hpc / CW_HW, hx = hpc - hy * CW_HW;
      if (hp < CW_HH * CW_HW) xin |= 1u << i;
      xlds[i] = hy * CW_RP + hx * CW_PP + cx * 16;
    }
    int blds[4];
    unsigned bsrc[4];   // byte offset of the thread's pieces inside a stage's two chunks (relative to chunk c0, column n0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = lt + 256 * j, uu = piece >> 9, q = piece & 511, row = q >> 2, slot = q & 3;
      blds[j] = uu * (CW_BN * 64) + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
      bsrc[j] = (unsigned)uu * (unsigned)a.Npad * 64u + (unsigned)q * 16u;
    }
    const unsigned xpitch = (unsigned)sx.ld * 2u;
    constexpr int R = NST < 4 ? NST : 4;   // weight stages in flight (register sets); NST % R == 0
    V rx[10], rb[R][4];
    unsigned okx_cur = 0, okx_next = 0;
    // Cursors of the two request streams: the halo of (item, group), the weights of (item, group, stage).  An item's geometry is taken
    // apart (three integer divisions) when a cursor ENTERS the item, not at every request.
    int hk = 0, hg = 0, hb = 0, hy0 = 0, hx0 = 0;   // next halo to request: item hk, group hg; the item's image and halo origin
    int bk = 0, bg = 0, bs = 0;                     // next weight stage to request
    unsigned bn0 = 0;                               // ... and its item's first output column
    auto item_geo = [&](int k, int& b, int& y0, int& x0, int& n0) {
      const int idx = first + min(k, nitems - 1) * stride;   // past the end: the last item again (requested, never used)
      const int ntile = idx % g.ntn;   // column tile FASTEST: with W workgroups a multiple of ntn, workgroup w always works on column tile
      int tile = idx / g.ntn;          // w % ntn - an XCD (w & 7) streams the weights of one or two column tiles, which stay in ITS L2
      const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
      const int ty_i = tile % g.tiles_y;
      b = tile / g.tiles_y;
      y0 = ty_i * CW_TH; x0 = tx_i * CW_TW; n0 = ntile * CW_BN;
    };
    auto enter_h = [&]() { int y0, x0, n0; item_geo(hk, hb, y0, x0, n0); hy0 = y0 + P.dymin; hx0 = x0 + P.dxmin; };
    auto enter_b = [&]() { int b, y0, x0, n0; item_geo(bk, b, y0, x0, n0); bn0 = (unsigned)n0; };
    enter_h();
    enter_b();
    auto issue_halo = [&]() {   // requests the halo of (hk, hg) into rx and advances the cursor; validity bits into okx_next
      okx_next = 0;
      const unsigned col = (unsigned)(hg * CW_CA + cx * SLOT) * 2u;
      const int rowbase = hb * sx.Hs;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int hp = min(px0 + 16 * i, CW_HH * CW_HW - 1);
        const int hy = hp / CW_HW, hx = hp - hy * CW_HW;
        const int y = hy0 + hy, x = hx0 + hx;
        if (((xin >> i) & 1) && (unsigned)y < (unsigned)sx.Hs && (unsigned)x < (unsigned)sx.Ws) okx_next |= 1u << i;
        const unsigned pix = (unsigned)((rowbase + min(max(y, 0), sx.Hs - 1)) * sx.Ws + min(max(x, 0), sx.Ws - 1));
        if (!(CW_DBG & 4)) cw_load(rx[i], sx.src, pix * xpitch + col);
      }
      if (++hg == ngrp) { hg = 0; ++hk; enter_h(); }
    };
    auto issue_b = [&](V (&rb)[4]) {   // requests weight stage (bk, bg, bs) and advances the cursor
      const int c0 = (bs >> 1) * cpt + bg * 4 + 2 * (bs & 1);
      const unsigned base = ((unsigned)c0 * (unsigned)a.Npad + bn0) * 64u;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (!(CW_DBG & 2)) cw_load(rb[j], P.wpack, base + bsrc[j]);
      if (++bs == NST) { bs = 0; if (++bg == ngrp) { bg = 0; ++bk; enter_b(); } }
    };
    SlotK<SLOT> kx;
    const float* kscale = (const float*)(smem + CW_OFF_K);   // staged by cvw_kernel: a global load here sat in the K loop's critical path
    const float* kshift = kscale + sx.C;                     // once per channel group (one memory latency per 4 - 8 stages: first version)
    auto store_halo = [&](int buf, int grp) {
      kx.k0 = load_fv<SLOT>(kscale + grp * CW_CA + cx * SLOT); kx.k1 = load_fv<SLOT>(kshift + grp * CW_CA + cx * SLOT); kx.k2 = 0.f; kx.k3 = 0.f;
      unsigned char* Xs = smem + buf * CW_X_BYTES;
      V z;
#pragma unroll
      for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
      for (int i = 0; i < 10; ++i)
        if ((xin >> i) & 1) *(V*)(Xs + xlds[i]) = ((okx_cur >> i) & 1) ? bn_relu_slot(rx[i], kx) : z;
    };
    auto store_b = [&](V (&rb)[4], int slot) {
      unsigned char* B = smem + CW_OFF_B + slot * CW_B_STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j) *(V*)(B + blds[j]) = rb[j];
    };
    // Schedule.  R = 4 weight stages (64 KB per workgroup; 2 for the one-tap phase) are in flight: stage s is requested behind the
    // barrier of stage s - R into register set s % R and waits there until its ring slot (s & 1) is free.  (First version: two stages
    // in flight - every stage then waited for an L2 / fabric round trip of ~1.5 us under load: 21 GB/s per CU, 650 TF/s on the
    // 1024-channel stage; cvp.hip sits at the same limit with its two workgroups per CU.  Eight sets spill.)  Wait counts = requests
    // issued BEHIND the one waited for (loads return in order).  Per channel group, stage position st = 0 .. NST-1:
    //   st == 0: wait halo(g)      [behind it: at least the R weight sets in flight = 4 R requests]         -> prologue, write halo image g & 1
    //            wait weights(s)   [behind: stages s+1 .. s+R-1 = 4 (R - 1)]                              -> write ring slot 0
    //            request halo(g+1) [10]                      barrier                    request weights(s + R) [4]
    //   st >= 1: wait weights(s)   [behind: 4 (R - 1), + halo(g+1) = 10 while st < R: this stage was requested before that halo]
    //            -> write slot st & 1;  barrier;  request weights(s + R)
    // Requests past the end of the walk re-request the last item (never used); everything lands behind the loop.
    issue_halo();
    okx_cur = okx_next;
#pragma unroll
    for (int st = 0; st < R; ++st) issue_b(rb[st]);
    int gcount = 0;   // groups done: halo buffer = gcount & 1
    for (int k = 0; k < nitems; ++k) {
      for (int grp = 0; grp < ngrp; ++grp, ++gcount) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          if (st == 0) {
            cw_wait_halo<V, 4 * R>(rx);
            store_halo(gcount & 1, grp);
            cw_wait_b<V, 4 * (R - 1)>(rb[0]);
            store_b(rb[0], 0);
            issue_halo();
            cw_bar();
            okx_cur = okx_next;
            issue_b(rb[0]);
          } else {
            if (st < R) cw_wait_b<V, 4 * (R - 1) + 10>(rb[st % R]); else cw_wait_b<V, 4 * (R - 1)>(rb[st % R]);
            store_b(rb[st % R], st & 1);
            cw_bar();
            issue_b(rb[st % R]);
          }
        }
      }
    }
    cw_hold_x<V>(rx);   // vmcnt(0): everything has landed; every set stays alive up to here
#pragma unroll
    for (int st = 0; st < R; ++st) cw_hold_b<V>(rb[st]);
    (void)nstages;
    return;
  }

  // ================================ matrix waves ================================
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  int aoff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tw = P.taps[t];
    const int dy = (int)(signed char)(tw & 0xff) - P.dymin, dx = (int)(signed char)((tw >> 8) & 0xff) - P.dxmin;
    aoff[t] = (4 * wr + (r >> 4) + dy) * CW_RP + ((r & 15) + dx) * CW_PP + h * 16;
  }
  const int bsw = (r >> 2) & 3;
  const int boff = (64 * wc + r) * 64;
  unsigned char* stg = smem + CW_OFF_S + wave * CW_STAGE_W;
  T* out = (T*)a.out;
  int gcount = 0;
  for (int k = 0; k < nitems; ++k) {
    const int idx = first + k * stride;
    const int ntile = idx % g.ntn;
    int tile = idx / g.ntn;
    const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty_i = tile % g.tiles_y;
    const int b = tile / g.tiles_y;
    const int y0 = ty_i * CW_TH, x0 = tx_i * CW_TW, n0 = ntile * CW_BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;
    for (int grp = 0; grp < ngrp; ++grp, ++gcount) {
      const unsigned char* Xs = smem + (gcount & 1) * CW_X_BYTES;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        cw_bar();   // stage complete (and, at st == 0, the halo image of this group)
        const unsigned char* B = smem + CW_OFF_B + (st & 1) * CW_B_STAGE + boff;
        const unsigned char* A = Xs + aoff[st >> 1] + (st & 1) * 128;
        // ALL sixteen fragments of the stage are requested first, into registers of their own, and the sixteen MFMAs follow in request
        // order: the first version left the order to the compiler, which kept three fragment registers and put an `s_waitcnt
        // lgkmcnt(0)` - one LDS latency - in front of every second MFMA (a stage took ~1600 cycles for 512 of MFMA).
        if (!(CW_DBG & 16)) {
          V af[4][2], bf[4][2];
#pragma unroll
          for (int q = 0; q < 4; ++q) {   // k-step q = 2 uu + s
            const int uu = q >> 1, s = q & 1;
            af[q][0] = *(const V*)(A + uu * 64 + s * 32);
            af[q][1] = *(const V*)(A + 2 * CW_RP + uu * 64 + s * 32);
            bf[q][0] = *(const V*)(B + uu * (CW_BN * 64) + (((2 * s + h) ^ bsw) << 4));
            bf[q][1] = *(const V*)(B + uu * (CW_BN * 64) + 32 * 64 + (((2 * s + h) ^ bsw) << 4));
          }
          __builtin_amdgcn_sched_barrier(0);   // (keep the requests in front of the arithmetic)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (CW_DBG & 1) { acc[0][0][0] += (float)af[q][0][0] + (float)bf[q][0][0] + (float)af[q][1][0] + (float)bf[q][1][0]; continue; }
            acc[0][0] = mma16(af[q][0], bf[q][0], acc[0][0]);
            acc[0][1] = mma16(af[q][0], bf[q][1], acc[0][1]);
            acc[1][0] = mma16(af[q][1], bf[q][0], acc[1][0]);
            acc[1][1] = mma16(af[q][1], bf[q][1], acc[1][1]);
          }
          __builtin_amdgcn_sched_barrier(0);   // (... and the arithmetic in front of the next barrier's wait: nothing is hoisted across)
        }
      }
    }
    // ---- epilogue, wave-local: stage as T, BatchNorm sums of the stored values straight from the accumulator layout ----
    if (a.stat_sum != nullptr && stat_ntile != ntile) {   // (uniform) another column tile: hand the sums of the previous one over
      if (stat_ntile >= 0) {
        const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int col = stat_ntile * CW_BN + 64 * wc + 32 * cb + r;
          if (col < a.N) atomic_add_f64((h ? a.stat_sq : a.stat_sum) + rep + col, dsum[cb]);
          dsum[cb] = 0.0;
        }
      }
      stat_ntile = ntile;
    }
    if ((CW_DBG & 8) && acc[0][0][0] != 1.2345e33f) continue;
    T* Cs = (T*)stg;
    float ps1[2] = {0.f, 0.f}, ps2[2] = {0.f, 0.f};
    const int slot = lane & 7, rsub = lane >> 3;
    const int n = n0 + 64 * wc + slot * SLOT;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {   // the wave's 64 x 64 block in two halves of 32 rows (tile rows 4 wr + 2 rb, + 1) through its 4.5 KB of staging
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * h;   // 0..31: tile row 4 wr + 2 rb + (row >> 4), column row & 15
          const T v = from_f32<T>(acc[rb][cb][i]);
          Cs[row * CW_SPITCH + 32 * cb + r] = v;
          const int y = y0 + 4 * wr + 2 * rb + (row >> 4), x = x0 + (row & 15);
          if (y < a.Ho && x < a.Wo) { const float f = to_f32(v); ps1[cb] += f; ps2[cb] = fmaf(f, f, ps2[cb]); }
        }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rsub;
        const int y = y0 + 4 * wr + 2 * rb + (row >> 4), x = x0 + (row & 15);
        if (y < a.Ho && x < a.Wo && n < a.N) {
          const size_t pix = (size_t)(b * a.Hout + y * a.ostride + P.py) * a.Wout + x * a.ostride + P.px;
          *(V*)(out + pix * a.ldo + n) = *(const V*)(Cs + row * CW_SPITCH + slot * SLOT);
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) dsum[cb] += (double)fold_swap32(ps1[cb], ps2[cb]);   // lane half 0: the sum, half 1: the sum of squares
  }
}

template <typename T>
__global__ __launch_bounds__(CW_NT, 1) void cvw_kernel(const CvwArgs g) {
  double dsum[2] = {0.0, 0.0};
  int stat_ntile = -1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {  // the norm's scale | shift tables of all input channels: to LDS once (the loaders read them per channel group)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* k = (float*)(smem + CW_OFF_K);
    const int C = g.c.seg[0].C;
    for (int i = threadIdx.x; i < C; i += CW_NT) { k[i] = g.c.seg[0].scale[i]; k[C + i] = g.c.seg[0].shift[i]; }
    __syncthreads();
  }
  for (int p = 0; p < g.nphase; ++p) {
    // (constant-index copies of the phase record: a kernel-argument array indexed by a run-time scalar is hf.hip's s_load trap)
    CwPhase P = g.ph[0];
    if (p == 1) P = g.ph[1];
    if (p == 2) P = g.ph[2];
    if (p == 3) P = g.ph[3];
    // the stream of (phase, item) pairs is dealt round-robin: this workgroup's first item of phase p
    const int done = p * g.per;                                   // items of the phases in front
    const int first = (int)(((long)blockIdx.x - done % (int)gridDim.x + (long)gridDim.x * 2) % gridDim.x);
    if (P.ntaps == 4) cw_walk<T, 4>(g, P, first, gridDim.x, dsum, stat_ntile);
    else if (P.ntaps == 2) cw_walk<T, 2>(g, P, first, gridDim.x, dsum, stat_ntile);
    else cw_walk<T, 1>(g, P, first, gridDim.x, dsum, stat_ntile);
    // BatchNorm sums of the last column tile of this phase (the next phase starts at column tile 0 again; all phases write the same channels)
    if (wave < 4 && g.c.stat_sum != nullptr && stat_ntile >= 0) {
      const int r = lane & 31, h = lane >> 5, wc = wave & 1;
      const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * g.c.stat_stride;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int col = stat_ntile * CW_BN + 64 * wc + 32 * cb + r;
        if (col < g.c.N) atomic_add_f64((h ? g.c.stat_sq : g.c.stat_sum) + rep + col, dsum[cb]);
        dsum[cb] = 0.0;
      }
      stat_ntile = -1;
    }
  }
}

template <typename T>
static hipError_t launch_cvw_t(const CvwArgs& g, int nwg, hipStream_t st) {
  auto kern = cvw_kernel<T>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int lds = CW_OFF_K + 8 * g.c.seg[0].C;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(CW_NT), lds, st, g);
  return hipGetLastError();
}

// Called by launch_cvp (cvp.hip) with a forward launch it has validated: one plain BN+ReLU segment of a multiple of 128 channels on the
// row grid, a multiple of 128 padded output columns, 1 / 2 / 4 taps per phase inside a 2 x 2 box, output stride 2.  nphase = 0: the
// launch's own taps / weights / parity; 4: ConvArgs::ph_*.  Returns hipErrorNotSupported for what this form does not cover.
hipError_t launch_cvw(const ConvArgs& a, int dtype, const int* ph_dymin, const int* ph_dxmin, hipStream_t st) {
  const Seg& x = a.seg[0];
  if ((double)a.B * x.Hs * x.Ws * x.ld * 2.0 >= 4294967296.0) return hipErrorNotSupported;          // 32-bit byte offsets in the loaders
  if ((double)x.Cpad / 32 * 4 * a.Npad * 64.0 >= 4294967296.0 || x.C > CW_MAX_C) return hipErrorNotSupported;
  // Measured (round 5, C2's four stages, ms alone, cvp.hip -> this kernel): 128 channels @320x480 0.47 -> 0.36, 256 @160x240 0.30 -> 0.30,
  // 512 @80x120 0.26 -> 0.28, 1024 @40x60 0.24 -> 0.27.  With many channel groups per tile the four loader waves' halo prologue (10 slots
  // x ~25 instructions per group, in front of the group's first barrier) sets the pace - ablations in profiles/r05/ablations.txt: without ANY
  // global request the deep stages take 0.25 ms, with idle matrix waves 0.19 - where cvp.hip's two workgroups per CU share that work
  // among eight waves.  So: the stages with one or two channel groups; DMM_CVW_MAX_GROUPS (lab) moves the limit.
  static const int max_groups = lab_int("DMM_CVW_MAX_GROUPS", 2);
  if (x.C / CW_CA > max_groups) return hipErrorNotSupported;
  CvwArgs g;
  g.c = a;
  g.tiles_y = (a.Ho + CW_TH - 1) / CW_TH;
  g.tiles_x = (a.Wo + CW_TW - 1) / CW_TW;
  g.ntn = a.Npad / CW_BN;
  g.ntl = a.B * g.tiles_y * g.tiles_x;
  g.per = g.ntl * g.ntn;
  g.nphase = a.nphase == 4 ? 4 : 1;
  int order[4] = {0, 1, 2, 3};
  if (a.nphase == 4) std::stable_sort(order, order + 4, [&](int p, int q) { return a.ph_ntaps[p] > a.ph_ntaps[q]; });   // most taps first
  for (int k = 0; k < g.nphase; ++k) {
    const int p = order[k];
    CwPhase& P = g.ph[k];
    if (a.nphase == 4) {
      P.wpack = a.ph_wpack[p]; P.ntaps = a.ph_ntaps[p]; P.py = a.ph_py[p]; P.px = a.ph_px[p];
      for (int t = 0; t < 4; ++t) P.taps[t] = a.ph_taps0[p][t];
    } else {
      P.wpack = a.wpack; P.ntaps = x.ntaps; P.py = a.py; P.px = a.px;
      for (int t = 0; t < 4; ++t) P.taps[t] = x.taps[t < x.ntaps ? t : 0];
    }
    P.dymin = ph_dymin[p]; P.dxmin = ph_dxmin[p];
    if (P.ntaps != 1 && P.ntaps != 2 && P.ntaps != 4) return hipErrorNotSupported;
  }
  for (int k = g.nphase; k < 4; ++k) g.ph[k] = g.ph[0];
  const int nwg = std::min(DESIGN_CUS, g.per * g.nphase);
  note_impl(IMPL_CVW);   // (beside IMPL_CVP, which the dispatcher notes: the family is cvp, this says which form ran)
  return dtype == DT_F16 ? launch_cvw_t<f16>(g, nwg, st) : launch_cvw_t<bf16>(g, nwg, st);
}

}  // namespace dmm
