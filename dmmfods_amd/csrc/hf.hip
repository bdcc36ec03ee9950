// Forward of the heat-map head's first convolution `refine0` (reference M:126-127: 3x3 over [nearest-x2 upsampled decoder output | raw
// input], behind norm0 + ReLU) in the per-output-parity form of plan.cpp - 2x2 merged taps over the HALF-resolution decoder output
// (128 channels) plus 3x3 stride-2 taps over the raw input (8 channels), 64 output channels, four phases - as a WAVE-SPECIALISED
// kernel.  gfx950, 16-bit storage types.
//
// conv3.hip ran this (one launch walking (tile, phase) pairs) at 1.5 ms for 2 GB of traffic and 0.17 ms of MFMA time.  Its -DC3_DBG
// ablations (profiles/r04/ablations.txt section 11): the halo stage of an item (8 loads per thread, BN+ReLU, LDS image) 0.64 ms, the K loop
// +0.58 (19 chunks of K stream through a 3-slot weight ring of 2 chunks per slot - what fits beside a 53 KB halo at two workgroups per CU -
// i.e. ten barriers per item with 8 MFMAs per wave between them), the epilogue +0.43: the three stretches of an item ADD UP.  Here
//   * a workgroup is EIGHT waves and owns ONE phase for the whole launch: the phase's packed weights (19 chunks x 64 columns = 76 KB) are
//     copied to LDS once and stay - no ring, no barrier inside a K stretch;
//   * waves 4-7 (loader waves) own the global loads (inline assembly, two register sets, counted waits - see wg3.hip), the BN+ReLU
//     prologue and the LDS images; waves 0-3 (matrix waves) do fragment reads, MFMAs and the epilogue;
//   * the 128-channel halo does not fit twice beside the weights, so an item is TWO K stretches over 64-channel half images
//     (9 x 17 pixels x 144 bytes): stretch 0 = channels 0-63 of the four taps + the raw-input taps (its 17 x 33-slot image travels with
//     half 0), stretch 1 = channels 64-127.  While the matrix waves multiply one half the loaders fill the other: two raw barriers per
//     item, each meaning "your next image is complete / the one you just left is free";
//   * the epilogue is wave-local (a wave stages, reads back and stores its own 32 rows: no barrier), the BatchNorm sums of the stored
//     values stay in fp64 registers for the whole walk (one round of atomics per workgroup).
// The four phase-workgroups of a tile group sit on one XCD and walk the same tiles: the half-resolution input comes from HBM once.
// K order: (half, tap, chunk) - not conv3's (tap, chunk): results agree with conv3's to the rounding of the fp32 accumulation order.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

#ifndef HF_DBG
#define HF_DBG 0   // bring-up / timing experiments only: 1 the loaders request nothing, 2 nothing is stored, 4 no statistics atomics, 8 no MFMA stretches
#endif
namespace dmm {

constexpr int HF_TH = 8, HF_TW = 16, HF_HH = 9, HF_HW = 17;
constexpr int HF_PP = 64 * 2 + 16;                          // pixel pitch of a 64-channel half image: 9 slots (odd)
constexpr int HF_RP = (HF_HW * HF_PP + 255) / 256 * 256;    // 2560
constexpr int HF_HALF = HF_HH * HF_RP;                      // 23040 bytes
constexpr int HF_HH1 = 2 * (HF_TH - 1) + 3, HF_HW1 = 2 * (HF_TW - 1) + 3;   // 17 x 33 slots of the raw input (stride 2, 3x3)
constexpr int HF_RP1 = HF_HW1 * 16;
constexpr int HF_NS1 = HF_HH1 * HF_HW1;                     // 561
constexpr int HF_THIN = (HF_NS1 * 16 + 255) / 256 * 256;    // 9216
constexpr int HF_NCH0 = 16, HF_NCH1 = 3, HF_NCH = HF_NCH0 + HF_NCH1, HF_BN = 64;
constexpr int HF_W = HF_NCH * HF_BN * 64;                   // 77824: the phase's packed weights
constexpr int HF_CP = HF_BN + 8;                            // staging pitch (elements)
constexpr int HF_STG = 32 * HF_CP * 2;                      // 4608 bytes per matrix wave
constexpr int HF_OFF_A0 = HF_W, HF_OFF_TH = HF_OFF_A0 + HF_HALF, HF_OFF_A1 = HF_OFF_TH + HF_THIN, HF_OFF_STG = HF_OFF_A1 + HF_HALF;
constexpr int HF_LDS = HF_OFF_STG + 4 * HF_STG;             // 151552
constexpr int HF_NT = 512, HF_NL = 256;
constexpr int HF_NU = (HF_HH * HF_HW * 8 + HF_NL - 1) / HF_NL;   // 5 half-image slots per loader thread
constexpr int HF_N1 = (HF_NS1 + HF_NL - 1) / HF_NL;              // 3 raw-input slots per loader thread
static_assert(HF_LDS <= 160 * 1024 && HF_NU == 5 && HF_N1 == 3, "LDS budget / operand lists of the waits");

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
// Waits of the loader waves (see wg3.hip): a loader wave's only vector-memory operations inside the walk are its own loads, issued
// set by set, so "all but the newest N have returned" is exactly "the oldest set has landed".  A half-0 set is 5 + 3 requests (the
// raw-input slots travel with it), a half-1 set 5; FOUR sets are in flight (two items ahead), so the oldest set is followed by
// 8 + 5 + 8 = 21 (half 1) or 5 + 8 + 5 = 18 (half 0) younger requests.
template <typename V>
__device__ __forceinline__ void hf_wait0(V (&u)[HF_NU], V (&t)[HF_N1]) {
  asm volatile("s_waitcnt vmcnt(18)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]));
}
template <typename V>
__device__ __forceinline__ void hf_wait1(V (&u)[HF_NU]) {
  asm volatile("s_waitcnt vmcnt(21)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]));
}
template <typename V>
__device__ __forceinline__ void hf_hold0(V (&u)[HF_NU], V (&t)[HF_N1]) {   // everything lands; the set is alive until here
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]));
}
template <typename V>
__device__ __forceinline__ void hf_hold1(V (&u)[HF_NU]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]));
}

template <typename T>
__global__ __launch_bounds__(HF_NT, 2) void hf_kernel(const HfArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  const ConvArgs& a = g.c;
  const Seg& sg = a.seg[0];    // the half-resolution decoder output: 128 channels, 2x2 taps per phase
  const Seg& sg1 = a.seg[1];   // the raw input: 8 channels, 3x3 taps at stride 2

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup b runs on XCD b % 8 and is the (b / 8)-th there: four neighbours are the four phases of one tile group.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ph = q & 3, grp = (q >> 2) * 8 + xcd, ngrp = gridDim.x >> 2;
  if (grp >= g.ntiles) return;                       // (workgroup-uniform)
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

  // ---- the phase's packed weights [chunk][64 columns][32 K] -> LDS, XOR swizzle of igemm.hip's B image; they stay for the whole walk ----
  {
    const T* wp = (const T*)wphase;
    if (!(HF_DBG & 32))
    for (int p = tid; p < HF_W / 16; p += HF_NT) {
      const int chunk = p >> 8, col = (p >> 2) & 63, slot = p & 3;
      const V v = *(const V*)(wp + ((size_t)chunk * HF_BN + col) * 32 + slot * SLOT);
      *(V*)(smem + chunk * (HF_BN * 64) + col * 64 + ((slot ^ ((col >> 2) & 3)) << 4)) = v;
    }
  }
  __syncthreads();

  if (wave >= 4) {
    // ================================ loader waves ================================
    const int lt = tid - HF_NL;
    const int cs = lt & 7, px0 = lt >> 3;   // half image: slot column cs (8 channels), halo pixels px0 + 32 i
    SlotK<SLOT> k0, k1, kt;
    k0.k0 = load_fv<SLOT>(sg.scale + cs * SLOT); k0.k1 = load_fv<SLOT>(sg.shift + cs * SLOT); k0.k2 = 0.f; k0.k3 = 0.f;
    k1.k0 = load_fv<SLOT>(sg.scale + 64 + cs * SLOT); k1.k1 = load_fv<SLOT>(sg.shift + 64 + cs * SLOT); k1.k2 = 0.f; k1.k3 = 0.f;
    kt.k0 = load_fv<SLOT>(sg1.scale); kt.k1 = load_fv<SLOT>(sg1.shift); kt.k2 = 0.f; kt.k3 = 0.f;
    int hyu[HF_NU], hxu[HF_NU], ldsu[HF_NU];
#pragma unroll
    for (int i = 0; i < HF_NU; ++i) {
      const int hp = min(px0 + 32 * i, HF_HH * HF_HW - 1);
      hyu[i] = hp / HF_HW; hxu[i] = hp - hyu[i] * HF_HW;
      ldsu[i] = hyu[i] * HF_RP + hxu[i] * HF_PP + cs * 16;
    }
    int hy1[HF_N1], hx1[HF_N1];
#pragma unroll
    for (int i = 0; i < HF_N1; ++i) {
      const int hp = min(lt + HF_NL * i, HF_NS1 - 1);
      hy1[i] = hp / HF_HW1; hx1[i] = hp - hy1[i] * HF_HW1;
    }
    // 32-bit byte offsets from uniform bases (the launcher checks that both tensors span < 4 GiB)
    const unsigned char* ubase = (const unsigned char*)sg.src;
    const unsigned char* tbase = (const unsigned char*)sg1.src;
    const unsigned upix = (unsigned)sg.ld * 2u, tpix = (unsigned)sg1.ld * 2u, ucol = (unsigned)cs * 16u;
    struct Set0 { V u[HF_NU], t[HF_N1]; unsigned oku, okt; };
    struct Set1 { V u[HF_NU]; unsigned oku; };
    // branch-free: clamped addresses, zeroed at the write if outside the picture.  (Past the end of the walk: the last item again - never used.)
    auto issue0 = [&](Set0& R, int item) {
      int b, y0, x0;
      origin(item, b, y0, x0);
      R.oku = 0; R.okt = 0;
      const int yb = y0 + dymin0, xb = x0 + dxmin0, row0 = b * sg.Hs;
#pragma unroll
      for (int i = 0; i < HF_NU; ++i) {
        const int sy = yb + hyu[i], sx = xb + hxu[i];
        if (px0 + 32 * i < HF_HH * HF_HW && (unsigned)sy < (unsigned)sg.Hs && (unsigned)sx < (unsigned)sg.Ws) R.oku |= 1u << i;
        const unsigned pix = (unsigned)((row0 + min(max(sy, 0), sg.Hs - 1)) * sg.Ws + min(max(sx, 0), sg.Ws - 1));
        if (!(HF_DBG & 1)) hf_load(R.u[i], pix * upix + ucol, ubase);
      }
      const int yt = 2 * y0 + dymin1, xt = 2 * x0 + dxmin1, row1 = b * sg1.Hs;
#pragma unroll
      for (int i = 0; i < HF_N1; ++i) {
        const int sy = yt + hy1[i], sx = xt + hx1[i];
        if (lt + HF_NL * i < HF_NS1 && (unsigned)sy < (unsigned)sg1.Hs && (unsigned)sx < (unsigned)sg1.Ws) R.okt |= 1u << i;
        const unsigned pix = (unsigned)((row1 + min(max(sy, 0), sg1.Hs - 1)) * sg1.Ws + min(max(sx, 0), sg1.Ws - 1));
        if (!(HF_DBG & 1)) hf_load(R.t[i], pix * tpix, tbase);
      }
    };
    auto issue1 = [&](Set1& R, int item) {
      int b, y0, x0;
      origin(item, b, y0, x0);
      R.oku = 0;
      const int yb = y0 + dymin0, xb = x0 + dxmin0, row0 = b * sg.Hs;
#pragma unroll
      for (int i = 0; i < HF_NU; ++i) {
        const int sy = yb + hyu[i], sx = xb + hxu[i];
        if (px0 + 32 * i < HF_HH * HF_HW && (unsigned)sy < (unsigned)sg.Hs && (unsigned)sx < (unsigned)sg.Ws) R.oku |= 1u << i;
        const unsigned pix = (unsigned)((row0 + min(max(sy, 0), sg.Hs - 1)) * sg.Ws + min(max(sx, 0), sg.Ws - 1));
        if (!(HF_DBG & 1)) hf_load(R.u[i], pix * upix + ucol + 128u, ubase);
      }
    };
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
    auto store_u = [&](const V (&u)[HF_NU], unsigned ok, const SlotK<SLOT>& k, unsigned char* img) {
#pragma unroll
      for (int i = 0; i < HF_NU; ++i)
        if (px0 + 32 * i < HF_HH * HF_HW && !(HF_DBG & 16)) *(V*)(img + ldsu[i]) = ((ok >> i) & 1) ? bn_relu_slot(u[i], k) : z;   // zero padding AFTER the prologue
    };
    auto store_t = [&](const V (&t)[HF_N1], unsigned ok) {
#pragma unroll
      for (int i = 0; i < HF_N1; ++i)
        if (lt + HF_NL * i < HF_NS1 && !(HF_DBG & 16)) *(V*)(smem + HF_OFF_TH + (lt + HF_NL * i) * 16) = ((ok >> i) & 1) ? bn_relu_slot(t[i], kt) : z;
    };
    auto store0 = [&](Set0& R) {   // half 0 of an item + its raw-input slots
      if (!(HF_DBG & 1)) hf_wait0(R.u, R.t);
      store_u(R.u, R.oku, k0, smem + HF_OFF_A0); store_t(R.t, R.okt);
    };
    auto store1 = [&](Set1& R) {
      if (!(HF_DBG & 1)) hf_wait1(R.u);
      store_u(R.u, R.oku, k1, smem + HF_OFF_A1);
    };
    // the constants have ARRIVED before the ring starts (see wg3.hip: pending compiler-counted loads at the loop header cost a drain per turn)
#pragma unroll
    for (int e = 0; e < SLOT; ++e) asm volatile("" : "+v"(k0.k0[e]), "+v"(k0.k1[e]), "+v"(k1.k0[e]), "+v"(k1.k1[e]), "+v"(kt.k0[e]), "+v"(kt.k1[e]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // FOUR register sets: the loads of the two halves of items it + 1 and it + 2 are in flight while item it is multiplied (one item
    // ahead - two sets - left them half an item, ~1500 cycles, to land: the loader side alone took 0.66 ms of the launch's 1.05).
    Set0 Pa, Pb;   // half 0 (+ raw input) of even / odd items
    Set1 Qa, Qb;   // half 1
    issue0(Pa, 0); issue1(Qa, 0); issue0(Pb, 1); issue1(Qb, 1);
    store0(Pa);
    // one item: barrier X (half 0 complete / half 1 free), request half 0 of item it + 2 into the set just emptied, write half 1;
    // barrier Y (half 1 complete / half 0 free), request half 1 of item it + 2, write half 0 of item it + 1
    auto item = [&](int it, Set0& P, Set1& Q, Set0& Pn) {
      hf_bar();
      issue0(P, it + 2);
      store1(Q);
      hf_bar();
      issue1(Q, it + 2);
      store0(Pn);
    };
    int it = 0;
    for (; it + 1 < nit; it += 2) {   // (both items unconditional in the loop, the odd last item behind it: see wg3.hip)
      item(it, Pa, Qa, Pb);
      item(it + 1, Pb, Qb, Pa);
    }
    if (nit & 1) item(nit - 1, Pa, Qa, Pb);
    if (!(HF_DBG & 1)) { hf_hold0(Pa.u, Pa.t); hf_hold0(Pb.u, Pb.t); hf_hold1(Qa.u); hf_hold1(Qb.u); }
    return;
  }

  // ================================ matrix waves ================================
  const int r = lane & 31, h = lane >> 5;
  const int ty = 2 * wave + (r >> 4), tx = r & 15;   // this lane's pixel of the tile
  const int abase = (ty - dymin0) * HF_RP + (tx - dxmin0) * HF_PP + h * 16;
  const int bsw = (r >> 2) & 3;
  int toffs[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tw = pick(a.ph_taps0[0][t], a.ph_taps0[1][t], a.ph_taps0[2][t], a.ph_taps0[3][t]);
    toffs[t] = (int)(signed char)(tw & 0xff) * HF_RP + (int)(signed char)((tw >> 8) & 0xff) * HF_PP;
  }
  // raw input: one 16-byte slot per tap; k-step (chunk c, half s) of lane half h reads tap j = 4c + 2s + h (j >= 9: zeros)
  int off1[2 * HF_NCH1];
  {
    const int abase1 = (2 * ty - dymin1) * HF_RP1 + (2 * tx - dxmin1) * 16;
#pragma unroll
    for (int c = 0; c < HF_NCH1; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = 4 * c + 2 * s + h;
        const int jj = j < 9 ? j : 0;   // (j depends on the lane half: a per-lane index into the selected phase's nine taps)
        int tw = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) { const int tk = pick(a.ph_taps1[0][k], a.ph_taps1[1][k], a.ph_taps1[2][k], a.ph_taps1[3][k]); tw = jj == k ? tk : tw; }
        off1[2 * c + s] = j < 9 ? abase1 + (int)(signed char)(tw & 0xff) * HF_RP1 + (int)(signed char)((tw >> 8) & 0xff) * 16 : -1;
      }
  }
  V z;
#pragma unroll
  for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
  const unsigned char* Wl = smem + (32 * 0 + r) * 64;   // + chunk * 4096 + 32 t * 64 + (((2 s + h) ^ bsw) << 4)
  T* stg = (T*)(smem + HF_OFF_STG + wave * HF_STG);
  T* out = (T*)a.out;
  double dsum[2] = {0.0, 0.0};   // lane (r, h): column 32 t + r - the sum (h = 0) / the sum of squares (h = 1) of the stored values

  for (int it = 0; it < nit; ++it) {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    hf_bar();   // barrier X(it)
    if (!(HF_DBG & 8)) {
      const unsigned char* A = smem + HF_OFF_A0 + abase;
#pragma unroll
      for (int tap = 0; tap < 4; ++tap)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const V av = *(const V*)(A + toffs[tap] + cg * 64 + s * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const V bv = *(const V*)(Wl + (tap * 4 + cg) * (HF_BN * 64) + 32 * t * 64 + (((2 * s + h) ^ bsw) << 4));
              acc[t] = mma16(av, bv, acc[t]);
            }
          }
#pragma unroll
      for (int c = 0; c < HF_NCH1; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int o = off1[2 * c + s];
          const V ld = *(const V*)(smem + HF_OFF_TH + (o >= 0 ? o : 0));
          const V av = o >= 0 ? ld : z;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const V bv = *(const V*)(Wl + (HF_NCH0 + c) * (HF_BN * 64) + 32 * t * 64 + (((2 * s + h) ^ bsw) << 4));
            acc[t] = mma16(av, bv, acc[t]);
          }
        }
    }
    hf_bar();   // barrier Y(it)
    if (!(HF_DBG & 8)) {
      const unsigned char* A = smem + HF_OFF_A1 + abase;
#pragma unroll
      for (int tap = 0; tap < 4; ++tap)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const V av = *(const V*)(A + toffs[tap] + cg * 64 + s * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const V bv = *(const V*)(Wl + (tap * 4 + 2 + cg) * (HF_BN * 64) + 32 * t * 64 + (((2 * s + h) ^ bsw) << 4));
              acc[t] = mma16(av, bv, acc[t]);
            }
          }
    }
    if (HF_DBG & 64) continue;
    // ---- epilogue, wave-local: stage the wave's 32 rows as T, sums of the stored values from the accumulator layout, 16-byte stores ----
    float ps1[2], ps2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ps1[t] = 0.f; ps2[t] = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const T v = from_f32<T>(acc[t][i]);
        stg[row * HF_CP + 32 * t + r] = v;
        const float f = to_f32(v);
        ps1[t] += f; ps2[t] = fmaf(f, f, ps2[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) dsum[t] += (double)fold_swap32(ps1[t], ps2[t]);
    int b, y0, x0;
    origin(it, b, y0, x0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = lane + 64 * j, row = piece >> 3, slot = piece & 7;
      const V v = *(const V*)(stg + row * HF_CP + slot * SLOT);
      const int y = y0 + 2 * wave + (row >> 4), x = x0 + (row & 15);
      const size_t pix = ((size_t)b * a.Hout + (size_t)(2 * y + opy)) * a.Wout + (size_t)(2 * x + opx);
      if (!(HF_DBG & 2) || v[0] == (T)12345.f) *(V*)(out + pix * a.ldo + slot * SLOT) = v;
    }
  }
  if (a.stat_sum != nullptr && !(HF_DBG & 4)) {
    const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) atomic_add_f64((h ? a.stat_sq : a.stat_sum) + rep + 32 * t + r, dsum[t]);
  }
}


template <typename T>
static hipError_t launch_hf_t(const HfArgs& g, int nwg, hipStream_t st) {
  auto kern = hf_kernel<T>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, HF_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(HF_NT), HF_LDS, st, g);
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
  // (DMM_NO_HF is read per call: the plan decides the family when it is built, and the tests switch it between plans)
  if (!family_on(getenv("DMM_NO_HF") == nullptr, IMPL_HF) || dtype == DT_F32 || epi != EPI_STORE || a.nphase != 4 || a.nseg != 2 || a.pool2) return hipErrorNotSupported;
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
  static const int cus = getenv("DMM_HF_WGS") ? atoi(getenv("DMM_HF_WGS")) : DESIGN_CUS;
  int nwg = std::max(32, cus / 32 * 32);
  const int need = (g.ntiles + 7) / 8 * 32;   // ntiles groups rounded up to whole XCD rows
  nwg = std::min(nwg, need);
  return dtype == DT_F16 ? launch_hf_t<f16>(g, nwg, st) : launch_hf_t<bf16>(g, nwg, st);
}

}  // namespace dmm
