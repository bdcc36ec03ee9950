// Weight gradient of the dense layers' 3x3 growth convolution (torchvision _DenseLayer.conv2, 128 -> 32 channels; reference call
// sites M:85-92, M:169-176) on LDS tiles, gfx950, 16-bit storage types.
//
//   dW[n][c][tap] = sum over pixels p of  A[p][c] * dYeff[p + t_tap][n]        (transposed form of wgrad.hip: taps on the dY side)
//     A     = relu(bn2(y1))           128 channels, normalised ONCE per pixel
//     dYeff = g + q + r*x             32 channels of the block's gradient buffer with the deferred BatchNorm-backward correction
// A PERSISTENT workgroup keeps the WHOLE 9 x 32 x 128 result in accumulators and walks 8 x 16 pixel tiles; per tile the A tile
// (128 px x 128 ch) and the 10 x 18 pixel dYeff halo (32 ch) go to LDS once, a tap is an address offset into the halo image, both
// MFMA operands are read with the transposing read ds_read_b64_tr_b16 (the contraction index is the pixel).
//
// Round 4: WAVE-SPECIALISED.  Rounds 2-3 ran this as four waves that each loaded, normalised, wrote LDS, multiplied - one phase
// after the other, one wave per SIMD (328 registers), so the phases added up: 6.3 us per tile against 1.1 us of fragment reads and
// MFMAs (the same on every block: 121 us on block 1's 18.75 tiles per workgroup, 58 us on block 2's 9.2, profiles/r03).  Now a
// workgroup is EIGHT waves, two per SIMD:
//   * waves 0-3 (matrix waves) own the accumulators (wave w: channels 32w .. 32w+31 of A, nine 32x32 tiles = 144 registers) and do
//     nothing but fragment reads and MFMAs on the image set of the current tile;
//   * waves 4-7 (loader waves) own the global loads - TWO register sets, i.e. the loads of tiles t+1 and t+2 are in flight while tile
//     t is multiplied -, the prologues (BN+ReLU on A, effective gradient on dY) and the LDS writes of the NEXT tile's image set;
//   * two image sets in LDS and ONE raw s_barrier per tile (behind s_waitcnt lgkmcnt(0) only: __syncthreads() is a fence, on gfx9
//     an s_waitcnt vmcnt(0) that would drain the loaders' prefetch): the loaders write set (t+1)&1 while the matrix waves read set
//     t&1; barrier t+1 tells the matrix waves that image t+1 is complete and the loaders that image t is free again.
// The partial result of a workgroup (147 KB) is either added to the packed gradient with fp32 atomics or - `part` - STORED to the
// workgroup's slot and added up in fixed order by wg3_reduce_kernel (the recipe of bw1.hip): no float atomics, and the number of
// workgroups no longer has to balance tiles per workgroup against 147 KB of atomics each.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "gather.h"
#include "pointwise.h"

#ifndef WG3_DBG
#define WG3_DBG 0  // timing experiments only (tools/build_variant.sh): 1 no output at the end of the walk, 2 no MFMA phase, 4 no loads
#endif
namespace dmm {

constexpr int W3_TH = 8, W3_TW = 16, W3_HH = 10, W3_HW = 18;
constexpr int W3_CA = 128, W3_CY = 32;                 // channels of A and of dY
constexpr int W3_A_BYTES = BM * W3_CA * 2;             // 32 KB, 256-byte rows, 64-byte granule XOR-ed with (row & 3)
constexpr int W3_Y_BYTES = W3_HH * W3_HW * W3_CY * 2;  // 11.25 KB, 64-byte rows
constexpr int W3_IMG = W3_A_BYTES + W3_Y_BYTES;        // one image set
constexpr int W3_LDS = 2 * W3_IMG;                     // 86.5 KB: one workgroup (8 waves) per CU
constexpr int W3_NT = 512;                             // threads: 4 matrix waves + 4 loader waves

struct Wg3Args {
  WgradArgs w;
  int tiles_y, tiles_x, ntiles, tiles_per_wg;
  float* part;  // per-workgroup slots of W3_SLOT_FLOATS (nullable: fp32 atomics into w.dpack)
};

typedef unsigned w3_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ w3_u32x2 w3_tr16(const unsigned char* p) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(p));
  return __builtin_bit_cast(w3_u32x2, r);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::vec w3_frag(const w3_u32x2& lo, const w3_u32x2& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(typename TT<T>::vec, v);
}
// one barrier per tile for all eight waves: the wave's own LDS traffic has returned; vector-memory requests stay in flight
__device__ __forceinline__ void w3_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The loaders' global loads and their waits are INLINE ASSEMBLY.  Written as plain C++ loads hipcc counted them itself and, at the
// header of the two-set loop, waited vmcnt(13) ... vmcnt(0) for the OLDER of two register sets in flight (27 ... 14 would do): its
// merged wait-count state dropped the newer set, every other tile drained the whole ring, and the kernel ran no faster than the
// four-wave one (ISA of the first version, round 4).  From assembly the compiler counts nothing: a loader wave's only vector-memory
// operations are these loads, issued set by set in program order, so "all but the newest NLD have returned" is exactly "the older set
// has landed".  Data flow is explicit - the load defines the register, the wait takes every register of the set as a read-write
// operand, the prologue reads the wait's outputs - so nothing can be scheduled across; tools/check_asm_loads.py checks in the
// disassembly that no instruction reads a loaded register between its load and its wait.
template <typename V>
__device__ __forceinline__ void w3_load(V& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
template <typename T, int NA, int NY, int PQ>
__device__ __forceinline__ void w3_wait(typename TT<T>::vec (&ra)[NA], typename TT<T>::vec (&ry)[NY], typename TT<T>::vec (&ry2)[PQ == 2 ? NY : 1]) {
  static_assert(NA == 8 && NY == 3, "operand list below");
  if constexpr (PQ == 2)
    asm volatile("s_waitcnt vmcnt(14)"
                 : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                   "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]), "+v"(ry2[0]), "+v"(ry2[1]), "+v"(ry2[2]));
  else
    asm volatile("s_waitcnt vmcnt(11)"
                 : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                   "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]));
}

// keeps a register set alive (operands) up to this point; DRAIN: and waits for every request in flight
template <typename T, int NA, int NY, int PQ, bool DRAIN>
__device__ __forceinline__ void w3_hold(typename TT<T>::vec (&ra)[NA], typename TT<T>::vec (&ry)[NY], typename TT<T>::vec (&ry2)[PQ == 2 ? NY : 1]) {
  static_assert(NA == 8 && NY == 3, "operand list below");
  if constexpr (DRAIN) {
    if constexpr (PQ == 2)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                     "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]), "+v"(ry2[0]), "+v"(ry2[1]), "+v"(ry2[2]));
    else
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                     "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]));
  } else {
    if constexpr (PQ == 2)
      asm volatile("; hold"
                   : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                     "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]), "+v"(ry2[0]), "+v"(ry2[1]), "+v"(ry2[2]));
    else
      asm volatile("; hold"
                   : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                     "+v"(ry[0]), "+v"(ry[1]), "+v"(ry[2]));
  }
}

// PQ = prologue of dY: 0 none (materialised gradient), 2 effective gradient (q, r of the 16-bit form)
template <typename T, int PQ>
__global__ __launch_bounds__(W3_NT, 1) void wg3_kernel(const Wg3Args g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  constexpr int NL = 256;                                                   // loader threads
  constexpr int NA = BM * (W3_CA / SLOT) / NL;                              // 8 A slots per loader thread
  constexpr int NY = (W3_HH * W3_HW * (W3_CY / SLOT) + NL - 1) / NL;        // 3 dY slots per loader thread
  const WgradArgs& a = g.w;
  const Seg& sy_ = a.seg[0];  // dY, nine taps
  const Seg& sa = a.dy;       // A, pixel aligned

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t_beg = blockIdx.x * g.tiles_per_wg, t_end = min(g.ntiles, t_beg + g.tiles_per_wg);
  if (t_beg >= t_end) return;  // (workgroup-uniform)
  const int nt = t_end - t_beg;

  if (wave >= 4) {
    // ================================ loader waves ================================
    const int lt = tid - NL;
    const int ca = lt & 15, pa0 = lt >> 4;  // A: slot column, pixel column pa0 of tile rows i = 0..7
    const int cy = lt & 3, hy0 = lt >> 2;   // dY: slot column, halo pixels hy0 + 64 i
    SlotK<SLOT> ka, ky;
    ka.k0 = load_fv<SLOT>(sa.scale + ca * SLOT); ka.k1 = load_fv<SLOT>(sa.shift + ca * SLOT); ka.k2 = 0.f; ka.k3 = 0.f;
    ky.k0 = 0.f; ky.k1 = 0.f; ky.k2 = 0.f; ky.k3 = 0.f;
    if (PQ == 2) { ky.k0 = load_fv<SLOT>(sy_.q + cy * SLOT); ky.k1 = load_fv<SLOT>(sy_.r + cy * SLOT); }
    const T* asrc = (const T*)sa.src + ca * SLOT;
    const T* ysrc = (const T*)sy_.src + cy * SLOT;
    const T* ysrc2 = (const T*)sy_.src2 + cy * SLOT;
    int hyy[NY], hxx[NY];  // this thread's halo positions (fixed over the walk)
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int hp = min(hy0 + 64 * i, W3_HH * W3_HW - 1);
      hyy[i] = hp / W3_HW - 1; hxx[i] = hp % W3_HW - 1;
    }
    struct LSet {
      V ra[NA], ry[NY], ry2[PQ == 2 ? NY : 1];
      unsigned oka, oky;  // validity bits of the slots
    };
    // the issue cursor walks the tiles of this workgroup in order, two tiles ahead of the image being written
    const int tiles_img = g.tiles_y * g.tiles_x;
    int cb = t_beg / tiles_img, cty, ctx, cleft = nt;
    { const int tr = t_beg - cb * tiles_img; cty = tr / g.tiles_x; ctx = tr - cty * g.tiles_x; }
    auto issue = [&](LSet& R) {  // branch-free: clamped addresses, zeroed at the write if outside; past the end: the last tile again
      const int y0 = cty * W3_TH, x0 = ctx * W3_TW;
      R.oka = 0; R.oky = 0;
      const int xa = x0 + pa0;
      const size_t arow = (size_t)cb * sa.Hs;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int y = y0 + i;
        if (y < a.Ho && xa < a.Wo) R.oka |= 1u << i;
        const size_t pix = (arow + min(y, sa.Hs - 1)) * sa.Ws + min(xa, sa.Ws - 1);
        if (!(WG3_DBG & 4)) w3_load(R.ra[i], asrc + pix * sa.ld);
      }
      const size_t yrow = (size_t)cb * sy_.Hs;
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const int y = y0 + hyy[i], x = x0 + hxx[i];
        if (hy0 + 64 * i < W3_HH * W3_HW && (unsigned)y < (unsigned)sy_.Hs && (unsigned)x < (unsigned)sy_.Ws) R.oky |= 1u << i;
        const size_t pix = (yrow + min(max(y, 0), sy_.Hs - 1)) * sy_.Ws + min(max(x, 0), sy_.Ws - 1);
        if (!(WG3_DBG & 4)) w3_load(R.ry[i], ysrc + pix * sy_.ld);
        if constexpr (PQ == 2) { if (!(WG3_DBG & 4)) w3_load(R.ry2[i], ysrc2 + pix * sy_.ld2); }
      }
      if (--cleft > 0) {  // (uniform) advance; the cursor parks on the last tile
        if (++ctx == g.tiles_x) { ctx = 0; if (++cty == g.tiles_y) { cty = 0; ++cb; } }
      }
    };
    auto store = [&](LSet& R, int set, bool wait = true) {
      if (wait && !(WG3_DBG & 4)) w3_wait<T, NA, NY, PQ>(R.ra, R.ry, R.ry2);  // this set has landed; the other set's NLD requests stay in flight
      unsigned char* As = smem + set * W3_IMG;
      unsigned char* Ys = As + W3_A_BYTES;
      V z;
#pragma unroll
      for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int p = pa0 + 16 * i;
        const V v = bn_relu_slot(R.ra[i], ka);
        *(V*)(As + p * 256 + ((ca * 16) ^ ((p & 3) << 6))) = ((R.oka >> i) & 1) ? v : z;
      }
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const int hp = hy0 + 64 * i;
        if (hp < W3_HH * W3_HW) {
          V v = R.ry[i];
          if constexpr (PQ == 2) v = eff_grad_slot(R.ry[i], R.ry2[i], ky);
          *(V*)(Ys + hp * 64 + cy * 16) = ((R.oky >> i) & 1) ? v : z;
        }
      }
    };
    // The prologue constants must have ARRIVED before the ring starts: loaded in front of it and first used inside the loop they stay
    // "pending" in hipcc's wait-count model on the loop's entry path, the merged state at the loop header then waits for them in EVERY
    // iteration - with the ring's loads issued behind them that wait was vmcnt(13) ... vmcnt(0): it drained the second register set
    // (seen in the ISA of the first version of this kernel: no gain over the four-wave kernel until this was fixed).
#pragma unroll
    for (int e = 0; e < SLOT; ++e) {
      asm volatile("" : "+v"(ka.k0[e]), "+v"(ka.k1[e]));
      if constexpr (PQ == 2) asm volatile("" : "+v"(ky.k0[e]), "+v"(ky.k1[e]));
    }
    LSet R0, R1;
    issue(R0);
    issue(R1);
    // (both halves unconditional in the loop, the odd last tile behind it: with a break in the middle the compiled loop has a path
    // from the first half straight to the latch, which tools/check_asm_loads.py - it cannot know that path always leaves - must flag)
    for (int k = 0; k + 1 < nt; k += 2) {
      store(R0, 0);   // waits for R0's loads only: R1's stay in flight
      w3_bar();       // barrier k: image 0 complete / the matrix waves have left image 1
      issue(R0);      // tile k + 2
      store(R1, 1);
      w3_bar();       // barrier k + 1
      issue(R1);      // tile k + 3
    }
    // Behind the loop BOTH sets may still have requests in flight (the cursor's surplus requests for tiles past the end), and the
    // compiler - which does not know that - considers a set's registers free from its last use on: it reused R1's registers as
    // temporaries of the odd tile's prologue below while R1's loads were still landing in them (found by tools/check_asm_loads.py
    // before the first run).  So: everything lands HERE, and both sets are operands of the statements, i.e. alive until then.
    if (!(WG3_DBG & 4)) {
      w3_hold<T, NA, NY, PQ, true>(R0.ra, R0.ry, R0.ry2);
      w3_hold<T, NA, NY, PQ, false>(R1.ra, R1.ry, R1.ry2);
    }
    if (nt & 1) {
      store(R0, 0, false);
      w3_bar();
    }
    return;
  }

  // ================================ matrix waves ================================
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // transposed-read lane geometry (see wgrad.hip): group tg = lane >> 4 covers columns 16 (tg & 1) .., rows 8 (tg >> 1) + tq (+4)
  const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
  const int acol = ((32 * wave + 16 * (tg & 1) + 4 * tp) * 2) ^ (tq << 6);  // row & 3 == tq for every row this lane reads
  const int arow = 8 * (tg >> 1) + tq;
  const int ycol = (16 * (tg & 1) + 4 * tp) * 2;
  int yoff[9];  // halo offset of (pixel x = arow of the k-step's tile row, tap)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int tw = sy_.taps[t];
    const int dy = (int)(signed char)(tw & 0xff), dx = (int)(signed char)((tw >> 8) & 0xff);
    yoff[t] = ((1 + dy) * W3_HW + (arow + 1 + dx)) * 64 + ycol;
  }
  for (int k = 0; k < nt; ++k) {
    w3_bar();  // barrier k: image k & 1 is complete
    const unsigned char* As = smem + (k & 1) * W3_IMG;
    const unsigned char* Ys = As + W3_A_BYTES;
    if (!(WG3_DBG & 2)) {
#pragma unroll 2
      for (int ms = 0; ms < W3_TH; ++ms) {  // one tile row = 16 pixels of the contraction per step
        const unsigned char* ap = As + (16 * ms + arow) * 256 + acol;
        const V af = w3_frag<T>(w3_tr16(ap), w3_tr16(ap + 4 * 256));
        const unsigned char* yp = Ys + ms * (W3_HW * 64);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const V bf = w3_frag<T>(w3_tr16(yp + yoff[t]), w3_tr16(yp + yoff[t] + 4 * 64));
          acc[t] = mma16(af, bf, acc[t]);
        }
      }
    }
  }

  // ---- the partial result: to this workgroup's slot (plain 16-byte stores) or added to the packed gradient dP[tap][c][n] ----
  // Slot layout = the accumulator layout: float4 j of tap t of wave w of lane l at (((t * 4 + w) * 4 + j) * 64 + l) * 4, i.e. every
  // store instruction writes 1 KB contiguous (36 per lane instead of 144 scalar ones); wg3_reduce_kernel knows the permutation.
  const int r = lane & 31, h = lane >> 5;
  if (g.part != nullptr) {
    f32x4* slot = (f32x4*)(g.part + (size_t)blockIdx.x * W3_SLOT_FLOATS);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = {acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]};
        if (!(WG3_DBG & 1) || v[0] == 1.2345e33f) slot[((t * 4 + wave) * 4 + j) * 64 + lane] = v;
      }
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (!(WG3_DBG & 1) || acc[t][i] == 1.2345e33f) atomic_add_f32(a.dpack + ((size_t)t * a.Npad + c) * 32 + r, acc[t][i]);
      }
  }
}

// dpack += sum of the slots, in a fixed order (no float atomics).  288 workgroups: thread (o, sg) adds the slots sg, sg + 8, ... of one
// float4 of the slot layout - all its loads are independent and in flight together - and the eight partial sums are added through
// LDS in the order sg = 0..7.  (First version: one thread per output walking ALL slots, 36 workgroups: 19 dependent rounds of
// loads, ~35 us per launch - more than the weight-gradient kernel itself on blocks 3-4.)
// float4 f of a slot = (tap t, wave w, quad j, lane l): accumulator elements 4j .. 4j+3 = channels 32 w + 8 j + 4 (l >> 5) + 0..3 of
// column n = l & 31 (the 32x32 MFMA result layout), i.e. four dP rows 128 bytes apart.
constexpr int W3_RG = 8;  // slot groups
__global__ __launch_bounds__(256) void wg3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dpack, int nslots) {
  __shared__ f32x4 red[W3_RG][32];
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + o;
  const float* p = part + (size_t)f * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = sg;
  for (; k + 3 * W3_RG < nslots; k += 4 * W3_RG) {
    const f32x4 v0 = __builtin_nontemporal_load((const f32x4*)(p + (size_t)k * W3_SLOT_FLOATS));
    const f32x4 v1 = __builtin_nontemporal_load((const f32x4*)(p + (size_t)(k + W3_RG) * W3_SLOT_FLOATS));
    const f32x4 v2 = __builtin_nontemporal_load((const f32x4*)(p + (size_t)(k + 2 * W3_RG) * W3_SLOT_FLOATS));
    const f32x4 v3 = __builtin_nontemporal_load((const f32x4*)(p + (size_t)(k + 3 * W3_RG) * W3_SLOT_FLOATS));
    s += (v0 + v1) + (v2 + v3);
  }
  for (; k < nslots; k += W3_RG) s += __builtin_nontemporal_load((const f32x4*)(p + (size_t)k * W3_SLOT_FLOATS));
  red[sg][o] = s;
  __syncthreads();
  if (sg == 0) {
    f32x4 t4 = red[0][o];
#pragma unroll
    for (int g2 = 1; g2 < W3_RG; ++g2) t4 += red[g2][o];
    const int l = f & 63, j = (f >> 6) & 3, w = (f >> 8) & 3, t = f >> 10;
    float* d = dpack + ((size_t)t * W3_CA + 32 * w + 8 * j + 4 * (l >> 5)) * 32 + (l & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q * 32] += t4[q];
  }
}

static bool g_wg3 = !lab_flag("DMM_NO_WG3");
void wg3_set_enabled(bool on) { g_wg3 = on; }

template <typename T, int PQ>
static hipError_t launch_wg3_t(const Wg3Args& g, int nwg, hipStream_t st) {
  if (g_ctl.dry) return hipSuccess;
  auto kern = wg3_kernel<T, PQ>;
  static bool attr_done = false;  // (one flag per instantiation)
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(W3_NT), W3_LDS, st, g);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && g.part != nullptr) {  // same stream: the slots are reused by the next launch of the family
    hipLaunchKernelGGL(wg3_reduce_kernel, dim3(W3_SLOT_FLOATS / 4 / 32), dim3(256), 0, st, g.part, g.w.dpack, nwg);
    e = hipGetLastError();
  }
  return e;
}

// Returns hipErrorNotSupported unless this is the transposed-form weight gradient of a 128 -> 32 channel 3x3 convolution in a
// 16-bit storage type.
hipError_t launch_wg3(const WgradArgs& a, int dtype, hipStream_t st) {
  if (!family_on(g_wg3, IMPL_WG3) || dtype == DT_F32 || a.nseg != 1) return hipErrorNotSupported;
  const Seg& q = a.seg[0];
  const Seg& p = a.dy;
  if (q.mode != G_PLAIN || q.istride != 1 || q.ntaps != 9 || q.C != W3_CY || q.Cpad != W3_CY || q.Hs != a.Ho || q.Ws != a.Wo || q.scale != nullptr)
    return hipErrorNotSupported;
  if (p.mode != G_PLAIN || p.istride != 1 || p.ntaps != 1 || p.taps[0] != 0 || p.C != W3_CA || p.Hs != a.Ho || p.Ws != a.Wo || p.scale == nullptr)
    return hipErrorNotSupported;
  if (a.N != W3_CA || a.Npad != W3_CA) return hipErrorNotSupported;
  bool seen[9] = {false, false, false, false, false, false, false, false, false};
  for (int t = 0; t < 9; ++t) {
    const int dy = (int)(signed char)(q.taps[t] & 0xff), dx = (int)(signed char)((q.taps[t] >> 8) & 0xff);
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || seen[(dy + 1) * 3 + dx + 1]) return hipErrorNotSupported;
    seen[(dy + 1) * 3 + dx + 1] = true;
  }
  Wg3Args g;
  g.w = a;
  g.tiles_y = (a.Ho + W3_TH - 1) / W3_TH;
  g.tiles_x = (a.Wo + W3_TW - 1) / W3_TW;
  g.ntiles = a.B * g.tiles_y * g.tiles_x;
  if (g_ctl.dry) return hipSuccess;
  // Workgroups.  With slots: one per compute unit (8 waves each; W3_MAX_SLOTS caps it), every workgroup stores its 147 KB partial
  // and the reduction reads them once.  With atomics (no slots, DMM_WG3_SLOTS=0): time ~ tiles/nwg * t_tile + nwg * (147 KB of
  // fp32 atomics at the chip-wide atomic rate of 1.3 TB/s = 0.11 us): minimum at nwg ~ sqrt(t_tile / 0.11 us * tiles), t_tile ~ 1.3 us.
  static const int cus = [] { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev);
                              return (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  static const bool use_slots = lab_int("DMM_WG3_SLOTS", 1) != 0;
  static const int wgs_cap = lab_int("DMM_WG3_WGS", 0);
  g.part = (use_slots && a.part != nullptr) ? a.part : nullptr;
  int nwg = g.part ? std::min(cus, a.part_slots) : (int)std::lround(std::sqrt(12.0 * g.ntiles));
  if (wgs_cap > 0) nwg = std::min(nwg, wgs_cap);
  nwg = std::max(1, std::min(std::min(nwg, cus), g.ntiles));
  g.tiles_per_wg = (g.ntiles + nwg - 1) / nwg;
  nwg = (g.ntiles + g.tiles_per_wg - 1) / g.tiles_per_wg;
  const int pq = q.q ? 2 : 0;
  if (dtype == DT_F16) return pq ? launch_wg3_t<f16, 2>(g, nwg, st) : launch_wg3_t<f16, 0>(g, nwg, st);
  return pq ? launch_wg3_t<bf16, 2>(g, nwg, st) : launch_wg3_t<bf16, 0>(g, nwg, st);
}

bool wg3_handles(const WgradArgs& a, int dtype) {
  const LaunchCtl keep = g_ctl;
  g_ctl.dry = true;
  const hipError_t e = launch_wg3(a, dtype, nullptr);
  g_ctl = keep;
  return e == hipSuccess;
}

}  // namespace dmm
