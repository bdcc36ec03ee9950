// Weight gradients of the parity-phase convolutions, WAVE-SPECIALISED (round 5): wgp.hip's tiles, LDS images, fragment reads and
// result layout - the decoder's ConvTranspose2d 3x3 stride 2 (reference M:155-160, four output-parity phases of 1, 2, 2, 4 taps) and
// the head's 3x3 over the nearest-x2 upsampled decoder output (M:126-127, four phases of 4 pre-summed taps) - run by EIGHT waves
// per workgroup in wg3.hip's form:
//   * waves 0-3 (matrix waves) own the accumulators (wave w: input channels 32w .. 32w+31, NTAP x NJ tiles of 32 x 32) and do nothing
//     but transposing fragment reads and MFMAs on the image set of the current 8 x 16 pixel tile;
//   * waves 4-7 (loader waves) own the global loads - TWO register sets, the loads of tiles t+1 and t+2 in flight while tile t is
//     multiplied -, the BN+ReLU prologue of the 9 x 17 pixel input halo and the LDS writes of the NEXT tile's image set;
//   * two image sets (2 x 77 KB), ONE raw s_barrier per tile behind s_waitcnt lgkmcnt(0).
// wgp.hip's four waves did load -> prologue -> LDS write -> 64 MFMAs one after the other at one wave per SIMD: 356 TF/s on the
// ConvTranspose stages (0.14 of the dense fp16 peak) with SQ_WAIT_INST_ANY 0.43-0.50.  Same arithmetic, same K order
// per workgroup: results equal wgp.hip's up to the order of the fp32 atomics at the end of the walk.
// Covers the materialised output gradient (no prologue on dY: every launch of the benchmarked plans); the deferred-correction form
// (PQ = 2) stays on wgp.hip.  1-tap phases (the ConvTranspose's (0, 0) parity) are taken as well (NTAP = 1).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int WW_TH = 8, WW_TW = 16;
constexpr int WW_HH = 9, WW_HWR = 17, WW_HW = 20;   // halo rows, real halo columns, halo row pitch in pixels (wgp.hip)
constexpr int WW_CA = 128;
constexpr int WW_X_BYTES = WW_HH * WW_HW * 256;     // 45 KB: 256-byte pixel rows, 64-byte granule XOR-ed with (pixel index & 3)
constexpr int WW_Y_BYTES = BM * 256;                // 32 KB
constexpr int WW_LDS = 2 * (WW_X_BYTES + WW_Y_BYTES);   // 154 KB: one workgroup (8 waves) per CU
// layout [X set 0 | X set 1 | Y set 0 | Y set 1]: the second set of either image is a compile-time offset below 64 KB, i.e. the
// immediate of the LDS instruction - one address register per slot serves both sets
constexpr int WW_Y0 = 2 * WW_X_BYTES;
static_assert(WW_X_BYTES < 65536 && WW_Y_BYTES < 65536, "set offsets fit the instruction's immediate");
constexpr int WW_NT = 512;
static_assert(WW_LDS <= 160 * 1024, "two image sets fit the LDS of a compute unit");

struct WgpwArgs {
  WgradArgs w;
  int tiles_y, tiles_x, ntiles, tiles_per_wg, nsplit;
  int nct, ncot;
  int dymin, dxmin;
  int ph_dymin[4], ph_dxmin[4];
};

typedef unsigned ww_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ww_u32x2 ww_tr16(const unsigned char* p) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(p));
  return __builtin_bit_cast(ww_u32x2, r);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::vec ww_frag(const ww_u32x2& lo, const ww_u32x2& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(typename TT<T>::vec, v);
}
__device__ __forceinline__ void ww_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The loaders' global loads and waits are inline assembly (wg3.hip explains why: hipcc's merged wait-count state drains the second
// register set at the loop header).  A loader wave's only vector-memory operations are these loads, issued set by set in program
// order: "all but the newest NLD have returned" = "the older set has landed".  tools/check_asm_loads.py checks the ISA.
// Address = uniform base (an SGPR pair) + a 32-bit per-lane byte offset: one VALU register and 32-bit arithmetic per load instead of a
// 64-bit pointer (the register budget above); the launcher refuses tensors of 4 GB or more.
template <typename V>
__device__ __forceinline__ void ww_load(V& dst, const void* base, unsigned off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}
#define WW_X10(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])
#define WW_Y4(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])
#define WW_Y8(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
// MODE 0: wait until this set has landed (the other set's NX + NY requests stay in flight); 1: wait for everything; 2: keep alive only
template <typename V, int NY, int MODE>
__device__ __forceinline__ void ww_sync(V (&rx)[10], V (&ry)[NY]) {
  static_assert(NY == 4 || NY == 8, "operand lists below");
  if constexpr (NY == 4) {
    if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(14)" : WW_X10(rx), WW_Y4(ry));
    else if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" : WW_X10(rx), WW_Y4(ry));
    else asm volatile("; hold" : WW_X10(rx), WW_Y4(ry));
  } else {
    if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(18)" : WW_X10(rx), WW_Y8(ry));
    else if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" : WW_X10(rx), WW_Y8(ry));
    else asm volatile("; hold" : WW_X10(rx), WW_Y8(ry));
  }
}

// What distinguishes the phases of a multi-phase launch (a single-phase launch: the fields of WgradArgs / WgpwArgs themselves)
struct WgpwPhase { int dymin, dxmin, ytap; float* dpack; short xt0, xt1, xt2, xt3; };

// NTAP taps of the phase (1, 2 or 4), NJ = NCO / 32 accumulator columns of 32 output channels
template <typename T, int NTAP, int NJ>
__device__ __forceinline__ void wgpw_body(const WgpwArgs& g, const int unit, const WgpwPhase P) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  constexpr int NL = 256;                         // loader threads
  constexpr int NCO = 32 * NJ;
  constexpr int NX = 10;                          // halo slots per loader thread: 9 x 17 pixels x 16 slot columns / 256
  constexpr int YS = NCO / SLOT;                  // dY slot columns
  constexpr int NY = BM * YS / NL;                // 4 (NCO 64) or 8 (NCO 128) dY slots per loader thread
  constexpr int YRS = NL / YS;                    // dY pixel step between a thread's slots
  static_assert(WW_HH * WW_HWR * (WW_CA / SLOT) <= NX * NL, "halo slots covered");
  const WgradArgs& a = g.w;
  const Seg& sx = a.seg[0];  // A with the phase's taps
  const Seg& sy = a.dy;      // dY, one tap = the parity

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (unit >= g.nsplit * g.nct * g.ncot) return;   // (workgroup-uniform)
  const int pair = unit % (g.nct * g.ncot), split = unit / (g.nct * g.ncot);
  const int ct = pair % g.nct, cot = pair / g.nct;
  const int dymin_ = P.dymin, dxmin_ = P.dxmin, ytap_ = P.ytap;
  float* const dpack_ = P.dpack;
  const short xt0 = P.xt0, xt1 = P.xt1, xt2 = P.xt2, xt3 = P.xt3;
  const int t_beg = split * g.tiles_per_wg, t_end = min(g.ntiles, t_beg + g.tiles_per_wg);
  if (t_beg >= t_end) return;                       // (workgroup-uniform)
  const int nt = t_end - t_beg;
  const int tiles_img = g.tiles_y * g.tiles_x;

  if (wave >= 4) {
    // ================================ loader waves ================================
    const int lt = tid - NL;
    const int cx = lt & 15, px0 = lt >> 4;     // A: slot column, halo pixels px0 + 16 i
    const int cy = lt % YS, py0 = lt / YS;     // dY: slot column, tile pixels py0 + YRS i
    SlotK<SLOT> kx;
    kx.k0 = load_fv<SLOT>(sx.scale + ct * WW_CA + cx * SLOT); kx.k1 = load_fv<SLOT>(sx.shift + ct * WW_CA + cx * SLOT); kx.k2 = 0.f; kx.k3 = 0.f;
    const unsigned xcol = (unsigned)(ct * WW_CA + cx * SLOT) * 2u, ycol = (unsigned)(cot * NCO + cy * SLOT) * 2u;   // bytes
    const unsigned xpitch = (unsigned)sx.ld * 2u, ypitch = (unsigned)sy.ld * 2u;
    const int ypy = (int)(signed char)(ytap_ & 0xff), ypx = (int)(signed char)((ytap_ >> 8) & 0xff);
    // (register budget: two sets of NX + NY 16-byte registers leave ~60 for everything else at two waves per SIMD: per halo slot ONE
    // LDS address - the second image set is an immediate offset - and ONE global offset relative to the tile's first halo pixel; a
    // thread's dY slots are YRS whole pixels apart)
    int xlds[NX];       // LDS byte offset inside an X image; >> 8 = halo pixel index (row * WW_HW + column)
    unsigned xrel[NX];  // global byte offset of the slot relative to the halo origin of an INTERIOR tile
    unsigned xin = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hp = px0 + 16 * i;
      const int hpc = min(hp, WW_HH * WW_HWR - 1);
      const int hy = hpc / WW_HWR, hx = hpc - hy * WW_HWR;
      if (hp < WW_HH * WW_HWR) xin |= 1u << i;
      const int idx = hy * WW_HW + hx;
      xlds[i] = idx * 256 + ((cx * 16) ^ ((idx & 3) << 6));
      xrel[i] = (unsigned)(hy * sx.Ws + hx) * xpitch + xcol;
    }
    static_assert(YRS % 4 == 0, "the swizzle key (pixel & 3) of a thread's dY slots is that of its first");
    const int ylds0 = py0 * 256 + ((cy * 16) ^ ((py0 & 3) << 6));
    const unsigned yrel0 = (unsigned)(((py0 >> 4) * sy.Ws + (py0 & 15)) * sy.istride) * ypitch + ycol;
    const unsigned ystep = (unsigned)((YRS / 16) * sy.Ws * sy.istride) * ypitch;   // one slot of the thread to the next
    struct LSet {
      V rx[NX], ry[NY];
      unsigned okx, oky;
    };
    int cb = t_beg / tiles_img, cty, ctx, cleft = nt;
    { const int tr = t_beg - cb * tiles_img; cty = tr / g.tiles_x; ctx = tr - cty * g.tiles_x; }
    // The loads themselves are issued in ONE place, behind the branch that computes their offsets: an inline-assembly load in either
    // arm of a branch makes the register set a phi of two registers - a copy of data that has not landed (tools/check_asm_loads.py).
    auto issue = [&](LSet& R) {  // past the end: the last tile again
      const int y0 = cty * WW_TH, x0 = ctx * WW_TW;
      const int hy0 = y0 + dymin_, hx0 = x0 + dxmin_;
      const int xrow = cb * sx.Hs, yrow = cb * sy.Hs;
      unsigned offx[NX], offy[NY];
      const bool interior = hy0 >= 0 && hx0 >= 0 && hy0 + WW_HH <= sx.Hs && hx0 + WW_HWR <= sx.Ws && y0 + WW_TH <= a.Ho && x0 + WW_TW <= a.Wo;
      if (interior) {  // (workgroup-uniform) every slot is inside the picture: tile base + constant offset, one add per load
        const unsigned xb = (unsigned)((xrow + hy0) * sx.Ws + hx0) * xpitch;
        const unsigned yb = (unsigned)((yrow + y0 * sy.istride + ypy) * sy.Ws + x0 * sy.istride + ypx) * ypitch + yrel0;
        R.okx = xin; R.oky = (1u << NY) - 1u;
#pragma unroll
        for (int i = 0; i < NX; ++i) offx[i] = xb + xrel[i];
#pragma unroll
        for (int i = 0; i < NY; ++i) offy[i] = yb + (unsigned)i * ystep;
      } else {         // border tiles: clamped addresses, zeroed at the write where outside
        R.okx = 0; R.oky = 0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int idx = xlds[i] >> 8;
          const int hy = (idx * 205) >> 12, hx = idx - hy * WW_HW;   // idx / 20 for idx < 180
          const int y = hy0 + hy, x = hx0 + hx;
          if (((xin >> i) & 1) && (unsigned)y < (unsigned)sx.Hs && (unsigned)x < (unsigned)sx.Ws) R.okx |= 1u << i;
          const unsigned pix = (unsigned)((xrow + min(max(y, 0), sx.Hs - 1)) * sx.Ws + min(max(x, 0), sx.Ws - 1));
          offx[i] = pix * xpitch + xcol;
        }
        const int yb = y0 + (py0 >> 4), xb = x0 + (py0 & 15);
        const bool xok = xb < a.Wo;
        const int sxx = min(xb, a.Wo - 1) * sy.istride + ypx;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
          const int y = yb + i * (YRS / 16);
          if (xok && y < a.Ho) R.oky |= 1u << i;
          const int syy = min(y, a.Ho - 1) * sy.istride + ypy;
          const unsigned pix = (unsigned)((yrow + syy) * sy.Ws + sxx);
          offy[i] = pix * ypitch + ycol;
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) ww_load(R.rx[i], sx.src, offx[i]);
#pragma unroll
      for (int i = 0; i < NY; ++i) ww_load(R.ry[i], sy.src, offy[i]);
      if (--cleft > 0) {  // (uniform) advance; the cursor parks on the last tile
        if (++ctx == g.tiles_x) { ctx = 0; if (++cty == g.tiles_y) { cty = 0; ++cb; } }
      }
    };
    auto store = [&](LSet& R, int set, bool wait = true) {
      if (wait) ww_sync<V, NY, 0>(R.rx, R.ry);  // this set has landed; the other set's requests stay in flight
      unsigned char* Xs = smem + set * WW_X_BYTES;
      unsigned char* Ys = smem + WW_Y0 + set * WW_Y_BYTES;
      V z;
#pragma unroll
      for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        if ((xin >> i) & 1) {
          const V v = bn_relu_slot(R.rx[i], kx);
          *(V*)(Xs + xlds[i]) = ((R.okx >> i) & 1) ? v : z;
        }
      }
#pragma unroll
      for (int i = 0; i < NY; ++i) *(V*)(Ys + ylds0 + i * (YRS * 256)) = ((R.oky >> i) & 1) ? R.ry[i] : z;
    };
    // the prologue constants must have ARRIVED before the ring starts (wg3.hip: otherwise the loop header waits for them, i.e. vmcnt(0))
#pragma unroll
    for (int e = 0; e < SLOT; ++e) asm volatile("" : "+v"(kx.k0[e]), "+v"(kx.k1[e]));
    LSet R0, R1;
    issue(R0);
    issue(R1);
    // (sched_barrier: the address arithmetic of an issue and the prologue of the next store are both VALU work the scheduler likes to
    // interleave - with both register sets alive that overflows the 256 registers of a wave at two waves per SIMD)
    for (int k = 0; k + 1 < nt; k += 2) {
      store(R0, 0);   // waits for R0's loads only
      ww_bar();       // barrier k: image 0 complete / the matrix waves have left image 1
      __builtin_amdgcn_sched_barrier(0);
      issue(R0);      // tile k + 2
      __builtin_amdgcn_sched_barrier(0);
      store(R1, 1);
      ww_bar();       // barrier k + 1
      __builtin_amdgcn_sched_barrier(0);
      issue(R1);      // tile k + 3
      __builtin_amdgcn_sched_barrier(0);
    }
    // everything lands HERE, with both sets alive as operands (wg3.hip: the compiler would reuse a dead set's registers under landing loads)
    ww_sync<V, NY, 1>(R0.rx, R0.ry);
    ww_sync<V, NY, 2>(R1.rx, R1.ry);
    if (nt & 1) {
      store(R0, 0, false);
      ww_bar();
    }
    return;
  }

  // ================================ matrix waves ================================
  f32x16 acc[NTAP][NJ];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][j][i] = 0.f;

  // transposed-read lane geometry (wgp.hip): group tg = lane >> 4 covers columns 16 (tg & 1) .., rows 8 (tg >> 1) + tq (+4)
  const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
  const int arow = 8 * (tg >> 1) + tq;
  const int xcolb = (32 * wave + 16 * (tg & 1) + 4 * tp) * 2;
  int xoff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tw = t == 0 ? xt0 : (t == 1 ? xt1 : (t == 2 ? xt2 : xt3));
    const int dy = (int)(signed char)(tw & 0xff) - dymin_, dx = (int)(signed char)((tw >> 8) & 0xff) - dxmin_;
    const int idx = dy * WW_HW + arow + dx;
    xoff[t] = idx * 256 + (xcolb ^ ((idx & 3) << 6));
  }
  int yoff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) yoff[j] = arow * 256 + (((32 * j + 16 * (tg & 1) + 4 * tp) * 2) ^ (tq << 6));

  for (int k = 0; k < nt; ++k) {
    ww_bar();  // barrier k: image k & 1 is complete
    const unsigned char* Xs = smem + (k & 1) * WW_X_BYTES;
    const unsigned char* Ys = smem + WW_Y0 + (k & 1) * WW_Y_BYTES;
#pragma unroll 2
    for (int ms = 0; ms < WW_TH; ++ms) {  // one tile row = 16 pixels of the contraction per step
      const unsigned char* yp = Ys + ms * (16 * 256);
      V yf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) yf[j] = ww_frag<T>(ww_tr16(yp + yoff[j]), ww_tr16(yp + yoff[j] + 4 * 256));
      const unsigned char* xp = Xs + ms * (WW_HW * 256);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const V xf = ww_frag<T>(ww_tr16(xp + xoff[t]), ww_tr16(xp + xoff[t] + 4 * 256));
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = mma16(yf[j], xf, acc[t][j]);  // rows: output channel n, columns: input channel c
      }
    }
  }

  // ---- add the partial result to the packed gradient: dP[chunk = (tap, c / 32)][n][c % 32] ----
  const int r = lane & 31, h = lane >> 5;
  const int cpt = sx.Cpad / 32;  // chunks per tap
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const size_t chunk = (size_t)t * cpt + ct * 4 + wave;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = cot * NCO + 32 * j + (i & 3) + 8 * (i >> 2) + 4 * h;
        atomic_add_f32(dpack_ + (chunk * a.Npad + n) * 32 + r, acc[t][j][i]);
      }
  }
}

template <typename T, int NTAP, int NJ>
__global__ __launch_bounds__(WW_NT, 1) void wgpw_kernel(const WgpwArgs g) {
  const Seg& sx = g.w.seg[0];
  const WgpwPhase P = {g.dymin, g.dxmin, g.w.dy.taps[0], g.w.dpack, sx.taps[0], sx.taps[1], sx.taps[2], sx.taps[3]};
  wgpw_body<T, NTAP, NJ>(g, blockIdx.x, P);
}

// All parity phases of one convolution in one launch (WgradArgs::nphase = 4; ph_ntaps = 0: four taps each - the head -, else the taps
// of each phase: the ConvTranspose's 1, 2, 2, 4 in some order), 64 output channels per workgroup for every phase.  wgp.hip's mapping:
// unit u = (split, pair) of phase ph is workgroup ((u / 8) * 4 + ph) * 8 + u % 8 - the four phases of a unit are dispatched together and
// onto the SAME XCD (workgroups go to the XCDs round-robin), walk the same tiles and find each other's input halo in that XCD's L2: the
// input comes from HBM once instead of once per phase.  Per tile every phase moves the same bytes (the launches of the large maps are
// bound by those, not by their 16 / 32 / 64 MFMAs), so the four walks keep pace.
template <typename T>
__global__ __launch_bounds__(WW_NT, 1) void wgpw_multi_kernel(const WgpwArgs g) {
  const int slot8 = blockIdx.x >> 3;
  const int ph = slot8 & 3;
  const int unit = (slot8 >> 2) * 8 + (blockIdx.x & 7);
  // per-phase values from constant-index copies (a kernel-argument array indexed by a run-time scalar: hf.hip's s_load trap)
  WgpwPhase P = {g.ph_dymin[0], g.ph_dxmin[0], g.w.ph_ytap[0], g.w.ph_dpack[0], g.w.ph_xtaps[0][0], g.w.ph_xtaps[0][1], g.w.ph_xtaps[0][2], g.w.ph_xtaps[0][3]};
  int nt = g.w.ph_ntaps[0];
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q == ph) {
      P.dymin = g.ph_dymin[q]; P.dxmin = g.ph_dxmin[q]; P.ytap = g.w.ph_ytap[q]; P.dpack = g.w.ph_dpack[q];
      P.xt0 = g.w.ph_xtaps[q][0]; P.xt1 = g.w.ph_xtaps[q][1]; P.xt2 = g.w.ph_xtaps[q][2]; P.xt3 = g.w.ph_xtaps[q][3];
      nt = g.w.ph_ntaps[q];
    }
  if (nt == 0 || nt == 4) wgpw_body<T, 4, 2>(g, unit, P);
  else if (nt == 2) wgpw_body<T, 2, 2>(g, unit, P);
  else wgpw_body<T, 1, 2>(g, unit, P);
}

template <typename T>
static hipError_t launch_wgpw_multi_t(const WgpwArgs& g, int nwg, hipStream_t st) {
  auto kern = wgpw_multi_kernel<T>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(WW_NT), WW_LDS, st, g);
  return hipGetLastError();
}

template <typename T, int NTAP, int NJ>
static hipError_t launch_wgpw_t(const WgpwArgs& g, int nwg, hipStream_t st) {
  auto kern = wgpw_kernel<T, NTAP, NJ>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(WW_NT), WW_LDS, st, g);
  return hipGetLastError();
}

// Called by launch_wgp (wgp.hip) with a launch it has already validated and laid out; returns hipErrorNotSupported for the shapes this
// form does not cover.
hipError_t launch_wgpw(const WgradArgs& a, int dtype, int ntap, int nj, int tiles_y, int tiles_x, int ntiles, int tiles_per_wg, int nsplit, int nct,
                       int ncot, int dymin, int dxmin, const int* ph_dymin, const int* ph_dxmin, int nwg, hipStream_t st) {
  // (32-bit byte offsets in the loaders)
  if ((double)a.B * a.seg[0].Hs * a.seg[0].Ws * a.seg[0].ld * 2.0 >= 4294967296.0 || (double)a.B * a.dy.Hs * a.dy.Ws * a.dy.ld * 2.0 >= 4294967296.0)
    return hipErrorNotSupported;
  WgpwArgs g;
  g.w = a;
  g.tiles_y = tiles_y; g.tiles_x = tiles_x; g.ntiles = ntiles; g.tiles_per_wg = tiles_per_wg; g.nsplit = nsplit;
  g.nct = nct; g.ncot = ncot; g.dymin = dymin; g.dxmin = dxmin;
  for (int ph = 0; ph < 4; ++ph) { g.ph_dymin[ph] = ph_dymin[ph]; g.ph_dxmin[ph] = ph_dxmin[ph]; }
  const bool f = dtype == DT_F16;
  if (a.nphase == 4) return nj == 2 ? (f ? launch_wgpw_multi_t<f16>(g, nwg, st) : launch_wgpw_multi_t<bf16>(g, nwg, st)) : hipErrorNotSupported;
  if (a.nphase != 0) return hipErrorNotSupported;
  if (ntap == 4 && nj == 2) return f ? launch_wgpw_t<f16, 4, 2>(g, nwg, st) : launch_wgpw_t<bf16, 4, 2>(g, nwg, st);
  if (ntap == 2 && nj == 4) return f ? launch_wgpw_t<f16, 2, 4>(g, nwg, st) : launch_wgpw_t<bf16, 2, 4>(g, nwg, st);
  if (ntap == 2 && nj == 2) return f ? launch_wgpw_t<f16, 2, 2>(g, nwg, st) : launch_wgpw_t<bf16, 2, 2>(g, nwg, st);
  if (ntap == 1 && nj == 4) return f ? launch_wgpw_t<f16, 1, 4>(g, nwg, st) : launch_wgpw_t<bf16, 1, 4>(g, nwg, st);
  if (ntap == 1 && nj == 2) return f ? launch_wgpw_t<f16, 1, 2>(g, nwg, st) : launch_wgpw_t<bf16, 1, 2>(g, nwg, st);
  return hipErrorNotSupported;
}

}  // namespace dmm
