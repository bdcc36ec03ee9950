// Forward of the heat-map head's first convolution `refine0` (reference M:126-127: 3x3 over [nearest-x2 upsampled decoder output | raw
// input], behind norm0 + ReLU) in the per-output-parity form of plan.cpp - 2x2 merged taps over the HALF-resolution decoder output
// (128 channels) plus 3x3 stride-2 taps over the raw input (8 channels), 64 output channels, four phases - as a WAVE-SPECIALISED
// kernel.  gfx950, 16-bit storage types.
//
// conv3.hip ran this (one launch walking (tile, phase) pairs) at 1.5-1.6 ms for 2 GB of traffic and 0.17 ms of MFMA time.  Its -DC3_DBG
// ablations (profiles/r04/ablations.txt section 11): the halo stage of an item (8 loads per thread, BN+ReLU, LDS image) 0.64 ms, the K loop
// +0.58 (19 chunks of K stream through a 3-slot weight ring of 2 chunks per slot - what fits beside a 53 KB halo at two workgroups per CU -
// i.e. ten barriers per item with 8 MFMAs per wave between them), the epilogue +0.43: the three stretches of an item ADD UP.  Here
//   * a workgroup is EIGHT waves (one per CU) and owns ONE phase for the whole launch; the four phase-workgroups of a tile group sit on
//     one XCD and walk the same tiles: the half-resolution input comes from HBM once;
//   * the phase's weights live in REGISTERS: matrix wave (rb, ct) owns 64 rows x 32 columns of the 128 x 64 tile and holds its 38 B
//     fragments (19 chunks x 2 k-steps of one 32-column tile: 152 registers) for the whole launch - a k-step is two A reads and two
//     MFMAs, no weight ring, no barrier inside the K stretch;
//   * waves 4-7 (loader waves) own the global loads (inline assembly, uniform base + 32-bit offsets, two register sets, counted
//     waits - see wg3.hip), the BN+ReLU prologue and the LDS images; TWO image sets (53 KB each: the 9 x 17-pixel halo of 128 channels
//     + the 17 x 33 raw-input slots), ONE raw barrier per item: the loaders write set (i + 1) & 1 while the matrix waves read set i & 1;
//   * the epilogue is wave-local (a wave stages, reads back and stores its own 64 x 32 tile: no barrier), the BatchNorm sums of the
//     stored values stay in an fp64 register for the whole walk (one round of atomics per workgroup).
// K order = conv3.hip's (tap, chunk; then the raw taps): the results are bit-equal to conv3.hip's.
// Measured (C2: 19200 tiles x 4 phases, 2 GB): conv3.hip 1.52-1.64 ms, this file 0.85.  Its -DHF_DBG ablations (profiles/r04/ablations.txt
// section 15): the 76 MFMAs of an item cost 0.05 ms, the matrix waves' epilogue 0.3, the loader side alone 0.56 - on every SIMD one loader
// wave and one matrix wave share the vector issue, and the matrix wave runs its K stretch and its epilogue one after the other.  Two
// other forms were built and measured: version 1 (weights resident in LDS, an item as two K stretches over 64-channel half images, loads
// two items ahead: 0.93 ms, in the history) and the epilogue handed to the loaders (stores in the loaders' in-order vmcnt queue make every
// wait for a set of loads wait for the L2's write acknowledgements as well: 1.51 ms).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

#ifndef HF_DBG
#define HF_DBG 0   // timing experiments only (tools/build_variant.sh): 1 the loaders request nothing, 2 nothing is stored, 8 no MFMA stretch, 64 no epilogue
#endif
namespace dmm {

constexpr int HF_TH = 8, HF_TW = 16, HF_HH = 9, HF_HW = 17;
constexpr int HF_HH1 = 2 * (HF_TH - 1) + 3, HF_HW1 = 2 * (HF_TW - 1) + 3;   // 17 x 33 slots of the raw input (stride 2, 3x3)
constexpr int HF_RP1 = HF_HW1 * 16;
constexpr int HF_NS1 = HF_HH1 * HF_HW1;                     // 561
constexpr int HF_THIN = (HF_NS1 * 16 + 255) / 256 * 256;    // 9216
constexpr int HF_NCH0 = 16, HF_NCH1 = 3, HF_NCH = HF_NCH0 + HF_NCH1, HF_BN = 64;
constexpr int HF_NT = 512, HF_NL = 256;
constexpr int HF_N1 = (HF_NS1 + HF_NL - 1) / HF_NL;              // 3 raw-input slots per loader thread
static_assert(HF_N1 == 3, "operand lists of the waits");

struct HfArgs {
  ConvArgs c;
  int tiles_y, tiles_x, ntiles;
  signed char ph_dymin0[4], ph_dxmin0[4], ph_dymin1[4], ph_dxmin1[4];  // origin of each phase's tap boxes
};

__device__ __forceinline__ void hf_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// uniform 64-bit base (scalar registers) + 32-bit per-lane byte offset: no 64-bit address arithmetic per request
template <typename V>
__device__ __forceinline__ void hf_load(V& dst, unsigned off, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}
// A loader wave's only vector-memory operations inside the walk are its own loads, issued set by set (13 per set): "all but the newest 13 have
// returned" is exactly "the older set has landed" (wg3.hip).
constexpr int H2_PP = 128 * 2 + 16;                          // 272: pixel pitch of the full 128-channel image (17 slots: odd)
constexpr int H2_RP = (HF_HW * H2_PP + 255) / 256 * 256;    // 4864
constexpr int H2_U = HF_HH * H2_RP;                         // 43776
constexpr int H2_SET = H2_U + HF_THIN;                      // 52992: one image set (decoder halo + raw-input slots)
constexpr int H2_CP = 32 + 8;                               // staging pitch (elements) of a wave's 64 x 32 tile
constexpr int H2_STG = 64 * H2_CP * 2;                      // 5120 bytes per matrix wave
constexpr int H2_OFF_STG = 2 * H2_SET;
constexpr int H2_LDS = H2_OFF_STG + 4 * H2_STG;             // 126464
constexpr int H2_NU = (HF_HH * HF_HW * 16 + HF_NL - 1) / HF_NL;   // 10 halo slots per loader thread
constexpr int H2_KS = 2 * HF_NCH;                           // 38 k-steps
static_assert(H2_NU == 10 && H2_LDS <= 160 * 1024, "operand lists of the waits / LDS budget");

template <typename V>
__device__ __forceinline__ void h2_wait(V (&u)[H2_NU], V (&t)[HF_N1]) {   // this set has landed; the other set's 13 requests stay in flight
  asm volatile("s_waitcnt vmcnt(13)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]),
                                      "+v"(t[0]), "+v"(t[1]), "+v"(t[2]));
}
template <typename V>
__device__ __forceinline__ void h2_hold(V (&u)[H2_NU], V (&t)[HF_N1]) {   // everything lands; the set is alive until here
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]),
                                     "+v"(t[0]), "+v"(t[1]), "+v"(t[2]));
}

template <typename T>
__global__ __launch_bounds__(HF_NT, 2) void hf_kernel(const HfArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  const ConvArgs& a = g.c;
  const Seg& sg = a.seg[0];
  const Seg& sg1 = a.seg[1];

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ph = q & 3, grp = (q >> 2) * 8 + xcd, ngrp = gridDim.x >> 2;
  if (grp >= g.ntiles) return;                          // (workgroup-uniform)
  const int nit = (g.ntiles - grp + ngrp - 1) / ngrp;   // items (tiles grp, grp + ngrp, ...) of this workgroup
  // The phase's parameters are SELECTED from the four constant-index copies: indexed by `ph` directly, hipcc (ROCm 7.2) loaded the weight
  // pointer with s_load_dwordx2 from base = kernarg + ph, offset = 7 ph + 0x278 - a base that is not dword-aligned, whose low bits the scalar
  // memory unit ignores: phases 1-3 read a wrong pointer and the launch ended in HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (bring-up, round 4).
  auto pick = [&](auto v0, auto v1, auto v2, auto v3) { return ph == 0 ? v0 : (ph == 1 ? v1 : (ph == 2 ? v2 : v3)); };
  const int dymin0 = pick(g.ph_dymin0[0], g.ph_dymin0[1], g.ph_dymin0[2], g.ph_dymin0[3]);
  const int dxmin0 = pick(g.ph_dxmin0[0], g.ph_dxmin0[1], g.ph_dxmin0[2], g.ph_dxmin0[3]);
  const int dymin1 = pick(g.ph_dymin1[0], g.ph_dymin1[1], g.ph_dymin1[2], g.ph_dymin1[3]);
  const int dxmin1 = pick(g.ph_dxmin1[0], g.ph_dxmin1[1], g.ph_dxmin1[2], g.ph_dxmin1[3]);
  const void* wphase = pick(a.ph_wpack[0], a.ph_wpack[1], a.ph_wpack[2], a.ph_wpack[3]);
  const int opy = pick(a.ph_py[0], a.ph_py[1], a.ph_py[2], a.ph_py[3]), opx = pick(a.ph_px[0], a.ph_px[1], a.ph_px[2], a.ph_px[3]);
  auto origin = [&](int item, int& b, int& y0, int& x0) {
    int tile = grp + min(item, nit - 1) * ngrp;
    const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty_i = tile % g.tiles_y;
    b = tile / g.tiles_y; y0 = ty_i * HF_TH; x0 = tx_i * HF_TW;
  };

  if (wave >= 4) {
    // ================================ loader waves ================================
    const int lt = tid - HF_NL;
    const int cs = lt & 15, px0 = lt >> 4;   // slot column cs (8 channels of 128), halo pixels px0 + 16 i
    SlotK<SLOT> ku, kt;
    ku.k0 = load_fv<SLOT>(sg.scale + cs * SLOT); ku.k1 = load_fv<SLOT>(sg.shift + cs * SLOT); ku.k2 = 0.f; ku.k3 = 0.f;
    kt.k0 = load_fv<SLOT>(sg1.scale); kt.k1 = load_fv<SLOT>(sg1.shift); kt.k2 = 0.f; kt.k3 = 0.f;
    int hyu[H2_NU], hxu[H2_NU], ldsu[H2_NU];
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) {
      const int hp = min(px0 + 16 * i, HF_HH * HF_HW - 1);
      hyu[i] = hp / HF_HW; hxu[i] = hp - hyu[i] * HF_HW;
      ldsu[i] = hyu[i] * H2_RP + hxu[i] * H2_PP + cs * 16;
    }
    int hy1[HF_N1], hx1[HF_N1];
#pragma unroll
    for (int i = 0; i < HF_N1; ++i) {
      const int hp = min(lt + HF_NL * i, HF_NS1 - 1);
      hy1[i] = hp / HF_HW1; hx1[i] = hp - hy1[i] * HF_HW1;
    }
    const unsigned char* ubase = (const unsigned char*)sg.src;
    const unsigned char* tbase = (const unsigned char*)sg1.src;
    const unsigned upix = (unsigned)sg.ld * 2u, tpix = (unsigned)sg1.ld * 2u, ucol = (unsigned)cs * 16u;
    struct Set { V u[H2_NU], t[HF_N1]; unsigned oku, okt; bool inner; };
    // Interior tiles (both halos inside the picture: 94 % of C2's) take a path without the per-slot clamps, validity tests and zero
    // selects - one add per request, the prologue straight to LDS: the loader and the matrix wave of a SIMD share its VALU, and the
    // ablations (profiles/r04/ablations.txt section 15) put the loader side alone at 0.56 ms of the launch's 0.9.
    unsigned cu[H2_NU], c1[HF_N1];   // byte offsets of this thread's slots relative to the halo's first pixel
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) cu[i] = (unsigned)(hyu[i] * sg.Ws + hxu[i]) * upix + ucol;
#pragma unroll
    for (int i = 0; i < HF_N1; ++i) c1[i] = (unsigned)(hy1[i] * sg1.Ws + hx1[i]) * tpix;
    auto issue = [&](Set& R, int item) {   // branch-free: clamped addresses, zeroed at the write if outside; past the end: the last item again
      int b, y0, x0;
      origin(item, b, y0, x0);
      const int yb = y0 + dymin0, xb = x0 + dxmin0, yt = 2 * y0 + dymin1, xt = 2 * x0 + dxmin1;
      // (bitwise: ONE condition, one branch - with short-circuit tests the compiled code carried a flag from several exits into a second
      // branch, a shape in which tools/check_asm_loads.py cannot tell the two paths apart)
      R.inner = (int)(yb >= 0) & (int)(xb >= 0) & (int)(yb + HF_HH <= sg.Hs) & (int)(xb + HF_HW <= sg.Ws) & (int)(yt >= 0) & (int)(xt >= 0) &
                (int)(yt + HF_HH1 <= sg1.Hs) & (int)(xt + HF_HW1 <= sg1.Ws);
      // the offsets are chosen in the branch, the requests are ONE sequence behind it (no inline-assembly load inside a branch: the
      // compiled if / else carries a flag between blocks, which the ISA checker cannot follow)
      unsigned ou[H2_NU], ot[HF_N1];
      R.oku = 0; R.okt = 0;
      if (R.inner) {   // (workgroup-uniform)
        const unsigned ub = (unsigned)((b * sg.Hs + yb) * sg.Ws + xb) * upix, tb = (unsigned)((b * sg1.Hs + yt) * sg1.Ws + xt) * tpix;
#pragma unroll
        for (int i = 0; i < H2_NU; ++i) ou[i] = ub + cu[i];
#pragma unroll
        for (int i = 0; i < HF_N1; ++i) ot[i] = tb + c1[i];
      } else {
        const int row0 = b * sg.Hs, row1 = b * sg1.Hs;
#pragma unroll
        for (int i = 0; i < H2_NU; ++i) {
          const int sy = yb + hyu[i], sx = xb + hxu[i];
          if (px0 + 16 * i < HF_HH * HF_HW && (unsigned)sy < (unsigned)sg.Hs && (unsigned)sx < (unsigned)sg.Ws) R.oku |= 1u << i;
          ou[i] = (unsigned)((row0 + min(max(sy, 0), sg.Hs - 1)) * sg.Ws + min(max(sx, 0), sg.Ws - 1)) * upix + ucol;
        }
#pragma unroll
        for (int i = 0; i < HF_N1; ++i) {
          const int sy = yt + hy1[i], sx = xt + hx1[i];
          if (lt + HF_NL * i < HF_NS1 && (unsigned)sy < (unsigned)sg1.Hs && (unsigned)sx < (unsigned)sg1.Ws) R.okt |= 1u << i;
          ot[i] = (unsigned)((row1 + min(max(sy, 0), sg1.Hs - 1)) * sg1.Ws + min(max(sx, 0), sg1.Ws - 1)) * tpix;
        }
      }
#pragma unroll
      for (int i = 0; i < H2_NU; ++i) if (!(HF_DBG & 1)) hf_load(R.u[i], ou[i], ubase);
#pragma unroll
      for (int i = 0; i < HF_N1; ++i) if (!(HF_DBG & 1)) hf_load(R.t[i], ot[i], tbase);
    };
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
    auto store = [&](Set& R, int set, bool wait = true) {
      if (wait && !(HF_DBG & 1)) h2_wait(R.u, R.t);
      unsigned char* img = smem + set * H2_SET;
      if (R.inner) {
#pragma unroll
        for (int i = 0; i < H2_NU; ++i)
          if (px0 + 16 * i < HF_HH * HF_HW) *(V*)(img + ldsu[i]) = bn_relu_slot(R.u[i], ku);
#pragma unroll
        for (int i = 0; i < HF_N1; ++i)
          if (lt + HF_NL * i < HF_NS1) *(V*)(img + H2_U + (lt + HF_NL * i) * 16) = bn_relu_slot(R.t[i], kt);
      } else {
#pragma unroll
      for (int i = 0; i < H2_NU; ++i)
        if (px0 + 16 * i < HF_HH * HF_HW) *(V*)(img + ldsu[i]) = ((R.oku >> i) & 1) ? bn_relu_slot(R.u[i], ku) : z;   // zero padding AFTER the prologue
#pragma unroll
      for (int i = 0; i < HF_N1; ++i)
        if (lt + HF_NL * i < HF_NS1) *(V*)(img + H2_U + (lt + HF_NL * i) * 16) = ((R.okt >> i) & 1) ? bn_relu_slot(R.t[i], kt) : z;
      }
    };
#pragma unroll
    for (int e = 0; e < SLOT; ++e) asm volatile("" : "+v"(ku.k0[e]), "+v"(ku.k1[e]), "+v"(kt.k0[e]), "+v"(kt.k1[e]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Set R0, R1;
    issue(R0, 0);
    issue(R1, 1);
    int k = 0;
    for (; k + 1 < nit; k += 2) {   // (wg3.hip's loop: both halves unconditional, the odd last item behind it)
      store(R0, 0);      // waits for R0's loads only: R1's stay in flight
      hf_bar();          // barrier k: image set 0 complete / the matrix waves have left set 1
      issue(R0, k + 2);
      store(R1, 1);
      hf_bar();          // barrier k + 1
      issue(R1, k + 3);
    }
    if (!(HF_DBG & 1)) { h2_hold(R0.u, R0.t); h2_hold(R1.u, R1.t); }
    if (nit & 1) {
      store(R0, 0, false);
      hf_bar();
    }
    return;
  }

  // ================================ matrix waves ================================
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, ct = wave >> 1;           // row half (64 rows = 4 tile rows), column tile (32 columns)
  int abase[2], abase1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ty = 4 * rb + 2 * j + (r >> 4), tx = r & 15;
    abase[j] = (ty - dymin0) * H2_RP + (tx - dxmin0) * H2_PP + h * 16;
    abase1[j] = H2_U + (2 * ty - dymin1) * HF_RP1 + (2 * tx - dxmin1) * 16;
  }
  int toffs[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tw = pick(a.ph_taps0[0][t], a.ph_taps0[1][t], a.ph_taps0[2][t], a.ph_taps0[3][t]);
    toffs[t] = (int)(signed char)(tw & 0xff) * H2_RP + (int)(signed char)((tw >> 8) & 0xff) * H2_PP;
  }
  // raw input: k-step (chunk c, half s) of lane half h reads tap j = 4c + 2s + h (j >= 9: zeros); offset relative to abase1
  int toff1[2 * HF_NCH1];
  bool tok1[2 * HF_NCH1];
#pragma unroll
  for (int c = 0; c < HF_NCH1; ++c)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int j = 4 * c + 2 * s2 + h;
      const int jj = j < 9 ? j : 0;
      int tw = 0;
#pragma unroll
      for (int k9 = 0; k9 < 9; ++k9) { const int tk = pick(a.ph_taps1[0][k9], a.ph_taps1[1][k9], a.ph_taps1[2][k9], a.ph_taps1[3][k9]); tw = jj == k9 ? tk : tw; }
      toff1[2 * c + s2] = (int)(signed char)(tw & 0xff) * HF_RP1 + (int)(signed char)((tw >> 8) & 0xff) * 16;
      tok1[2 * c + s2] = j < 9;
    }
  // the wave's B fragments, once: k-step ks = 2 chunk + s of column 32 ct + r, K slice (2 s + h) of the chunk's 32
  V Bf[H2_KS];
  {
    const T* wp = (const T*)wphase + (size_t)(32 * ct + r) * 32 + h * SLOT;
#pragma unroll
    for (int ks = 0; ks < H2_KS; ++ks) Bf[ks] = *(const V*)(wp + (size_t)(ks >> 1) * (HF_BN * 32) + (ks & 1) * 16);
  }
  V z;
#pragma unroll
  for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
  T* stg = (T*)(smem + H2_OFF_STG + wave * H2_STG);
  T* out = (T*)a.out;
  double dsum = 0.0;   // lane (r, h): column 32 ct + r - the sum (h = 0) / the sum of squares (h = 1) of the stored values

  for (int it = 0; it < nit; ++it) {
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    hf_bar();   // barrier it: image set it & 1 is complete
    const unsigned char* img = smem + (it & 1) * H2_SET;
    if (!(HF_DBG & 8)) {
#pragma unroll
    for (int ck = 0; ck < HF_NCH0; ++ck)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const V av = *(const V*)(img + abase[j] + toffs[ck >> 2] + (ck & 3) * 64 + s2 * 32);
          acc[j] = mma16(av, Bf[2 * ck + s2], acc[j]);
        }
#pragma unroll
    for (int c = 0; c < HF_NCH1; ++c)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const V ld = *(const V*)(img + abase1[j] + (tok1[2 * c + s2] ? toff1[2 * c + s2] : 0));
          const V av = tok1[2 * c + s2] ? ld : z;
          acc[j] = mma16(av, Bf[2 * (HF_NCH0 + c) + s2], acc[j]);
        }
    }
    if (HF_DBG & 64) continue;
    // ---- epilogue, wave-local: stage the wave's 64 x 32 tile as T, sums of the stored values from the accumulator layout, 16-byte stores ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float ps1 = 0.f, ps2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 32 * j + (i & 3) + 8 * (i >> 2) + 4 * h;
        const T v = from_f32<T>(acc[j][i]);
        stg[row * H2_CP + r] = v;
        const float f = to_f32(v);
        ps1 += f; ps2 = fmaf(f, f, ps2);
      }
      dsum += (double)fold_swap32(ps1, ps2);
    }
    int b, y0, x0;
    origin(it, b, y0, x0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int piece = lane + 64 * p, row = piece >> 2, slot = piece & 3;
      const V v = *(const V*)(stg + row * H2_CP + slot * SLOT);
      const int y = y0 + 4 * rb + (row >> 4), x = x0 + (row & 15);
      const size_t pix = ((size_t)b * a.Hout + (size_t)(2 * y + opy)) * a.Wout + (size_t)(2 * x + opx);
      if (!(HF_DBG & 2) || v[0] == (T)12345.f) *(V*)(out + pix * a.ldo + 32 * ct + slot * SLOT) = v;
    }
  }
  if (a.stat_sum != nullptr) {
    const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
    atomic_add_f64((h ? a.stat_sq : a.stat_sum) + rep + 32 * ct + r, dsum);
  }
}

template <typename T>
static hipError_t launch_hf_t(const HfArgs& g, int nwg, hipStream_t st) {
  auto kern = hf_kernel<T>;
  const int lds = H2_LDS;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(HF_NT), lds, st, g);
  return hipGetLastError();
}

static bool hf_tap_box(const short* taps, int ntaps, int want_span, int& dymin, int& dxmin) {
  int dymax = -128, dxmax = -128;
  dymin = 127; dxmin = 127;
  for (int t = 0; t < ntaps; ++t) {
    const int dy = (int)(signed char)(taps[t] & 0xff), dx = (int)(signed char)((taps[t] >> 8) & 0xff);
    dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax; dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
  }
  if (dymax - dymin != want_span || dxmax - dxmin != want_span) return false;
  bool seen[9] = {false};
  for (int t = 0; t < ntaps; ++t) {   // every offset of the box exactly once
    const int i = ((int)(signed char)(taps[t] & 0xff) - dymin) * (want_span + 1) + ((int)(signed char)((taps[t] >> 8) & 0xff) - dxmin);
    if (seen[i]) return false;
    seen[i] = true;
  }
  return true;
}

// Takes the four-phase forward launch of the head's first convolution (ConvArgs::nphase = 4): a 128-channel BN+ReLU-normalised
// half-resolution segment with 2x2 taps per phase, an 8-channel raw segment with 3x3 taps at stride 2, 64 output channels stored at
// stride 2, whole 8 x 16 tiles, 16-bit storage.  hipErrorNotSupported otherwise (conv3.hip takes the launch then).
hipError_t launch_hf(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  // (DMM_NO_HF: read when a plan is created - PlanSwitches::no_hf, plan.h - and applied through g_ctl.deny while it is built)
  if (!family_on(true, IMPL_HF) || dtype == DT_F32 || epi != EPI_STORE || a.nphase != 4 || a.nseg != 2 || a.pool2) return hipErrorNotSupported;
  const Seg& u = a.seg[0];
  const Seg& t = a.seg[1];
  if (u.mode != G_PLAIN || u.istride != 1 || u.C != 128 || u.Cpad != 128 || u.Hs != a.Ho || u.Ws != a.Wo || u.scale == nullptr || u.ntaps != 4) return hipErrorNotSupported;
  if (t.mode != G_PLAIN || t.istride != 2 || t.C != 8 || t.Cpad != 8 || t.Hs != 2 * a.Ho || t.Ws != 2 * a.Wo || t.scale == nullptr || t.ntaps != 9 || t.q != nullptr) return hipErrorNotSupported;
  if (a.N != HF_BN || a.Npad != HF_BN || a.out == nullptr || a.ostride != 2 || a.Hout != 2 * a.Ho || a.Wout != 2 * a.Wo) return hipErrorNotSupported;
  if (a.Ho % HF_TH || a.Wo % HF_TW || a.ldo % 8) return hipErrorNotSupported;
  if (2.0 * a.B * u.Hs * u.Ws * u.ld >= 4294967296.0 || 2.0 * a.B * t.Hs * t.Ws * t.ld >= 4294967296.0) return hipErrorNotSupported;   // 32-bit byte offsets
  HfArgs g;
  for (int ph = 0; ph < 4; ++ph) {
    int y0, x0, y1, x1;
    if (a.ph_wpack[ph] == nullptr || !hf_tap_box(a.ph_taps0[ph], 4, 1, y0, x0) || !hf_tap_box(a.ph_taps1[ph], 9, 2, y1, x1) ||
        a.ph_py[ph] < 0 || a.ph_py[ph] > 1 || a.ph_px[ph] < 0 || a.ph_px[ph] > 1)
      return hipErrorNotSupported;
    g.ph_dymin0[ph] = (signed char)y0; g.ph_dxmin0[ph] = (signed char)x0; g.ph_dymin1[ph] = (signed char)y1; g.ph_dxmin1[ph] = (signed char)x1;
  }
  if (g_ctl.dry) return hipSuccess;
  g.c = a;
  g.tiles_y = a.Ho / HF_TH;
  g.tiles_x = a.Wo / HF_TW;
  g.ntiles = a.B * g.tiles_y * g.tiles_x;
  // one workgroup per CU (152 KB of LDS), in whole groups of 4 phases x 8 XCDs; launches with fewer tile groups than that: one item each
  static const int cus = lab_int("DMM_HF_WGS", DESIGN_CUS);
  int nwg = std::max(32, cus / 32 * 32);
  const int need = (g.ntiles + 7) / 8 * 32;   // ntiles groups rounded up to whole XCD rows
  nwg = std::min(nwg, need);
  return dtype == DT_F16 ? launch_hf_t<f16>(g, nwg, st) : launch_hf_t<bf16>(g, nwg, st);
}

}  // namespace dmm
