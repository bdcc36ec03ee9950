// Weight gradients of the parity-phase convolutions - the decoder's ConvTranspose2d 3x3 stride 2 (reference M:155-160, run as four
// output-parity phases) and the head's 3x3 over the nearest-x2 upsampled decoder output (M:126-127, same four phases with
// pre-summed taps) - on LDS tiles with the whole tap set of a phase in registers.  gfx950, 16-bit storage types.
//
//   dP[tap][n][c] = sum over phase pixels p of  dYeff[S p + parity][n] * A[p + t_tap][c]          (normal form of wgrad.hip)
//     A     = relu(bn(x))        128 channels of the phase grid's input, normalised ONCE per pixel, taps inside a 2x2 box
//     dYeff = g (+ q + r*y)      the output gradient at the phase's parity positions (S = output stride of the phase grid)
// The generic kernel gives every (tap, 128 c, 128 n) block of the result to its own workgroups, each re-gathering both operands:
// 64 flop per operand byte, which is what the L2 delivers, not what the matrix cores take.  Here a PERSISTENT workgroup owns 128 c
// x NCO n for ALL taps of the phase (wave w: channels 32w..32w+31; NTAP x NCO/32 accumulator tiles = 128 registers), walks over
// 8 x 16 pixel tiles and per tile loads the 9 x 17 pixel halo of A and the dY tile ONCE: 2.4x (4 taps) / 1.8x (2 taps) the
// arithmetic intensity, one round of atomics per workgroup instead of one per 64-row slice.  Both operands are contracted over the
// pixel, the slow index of the row-major LDS images, so both MFMA operands come from ds_read_b64_tr_b16 (as in wg3.hip).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int WP_TH = 8, WP_TW = 16;       // pixel tile
constexpr int WP_HH = 9, WP_HWR = 17;      // halo rows / real halo columns (taps span a 2x2 box)
constexpr int WP_HW = 20;                  // halo row pitch in pixels: a multiple of 4, so (pixel index & 3) survives a row step
constexpr int WP_CA = 128;                 // channels of A per workgroup
constexpr int WP_X_BYTES = WP_HH * WP_HW * 256;  // 45 KB: 256-byte pixel rows, 64-byte granule XOR-ed with (pixel index & 3)
constexpr int WP_Y_BYTES = BM * 256;             // 32 KB: 256-byte rows (NCO <= 128 channels), same swizzle
constexpr int WP_LDS = WP_X_BYTES + WP_Y_BYTES;

struct WgpArgs {
  WgradArgs w;
  int tiles_y, tiles_x, ntiles, tiles_per_wg, nsplit;
  int nct, ncot;     // 128-channel tiles of A, NCO-channel tiles of dY
  int dymin, dxmin;  // origin of the tap box
  int ph_dymin[4], ph_dxmin[4];  // ... per phase of a multi-phase launch (WgradArgs::nphase)
};

typedef unsigned wp_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wp_u32x2 wp_tr16(const unsigned char* p) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(p));
  return __builtin_bit_cast(wp_u32x2, r);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::vec wp_frag(const wp_u32x2& lo, const wp_u32x2& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(typename TT<T>::vec, v);
}

// NTAP taps of the phase (2 or 4), NJ = NCO / 32 accumulator columns of 32 output channels, PQ = prologue of dY (0 none, 2 effective
// gradient)
template <typename T, int NTAP, int NJ, int PQ>
__global__ __launch_bounds__(NTHREADS, 1) void wgp_kernel(const WgpArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  constexpr int NCO = 32 * NJ;
  constexpr int NX = (WP_HH * WP_HWR * (WP_CA / SLOT) + NTHREADS - 1) / NTHREADS;  // 10 halo slots per thread
  constexpr int YS = NCO / SLOT;                                                   // dY slot columns
  constexpr int NY = BM * YS / NTHREADS;                                           // 4 (NCO 64) or 8 (NCO 128) dY slots per thread
  constexpr int YRS = NTHREADS / YS;                                               // dY pixel step between a thread's slots
  const WgradArgs& a = g.w;
  const Seg& sx = a.seg[0];  // A with the phase's taps
  const Seg& sy = a.dy;      // dY, one tap = the parity

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Ys = smem + WP_X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // block -> (phase, split, channel tile of A, channel tile of dY); the tiles of one split are neighbours in the launch order.
  // Several phases in one launch: unit u = (split, pair) of phase ph is workgroup ((u / 8) * nphase + ph) * 8 + u % 8, i.e. the
  // phases of a unit are dispatched together and - workgroups go to the XCDs round-robin - onto the SAME XCD: they walk the same
  // tiles at the same pace, the first to ask fetches a halo from HBM and the others find it in that XCD's L2.
  const int nph = g.w.nphase > 0 ? g.w.nphase : 1;
  const int slot8 = blockIdx.x >> 3;
  const int ph = g.w.nphase > 0 ? slot8 % nph : 0;
  const int unit = g.w.nphase > 0 ? (slot8 / nph) * 8 + (blockIdx.x & 7) : blockIdx.x;
  if (unit >= g.nsplit * g.nct * g.ncot) return;
  const int pair = unit % (g.nct * g.ncot), split = unit / (g.nct * g.ncot);
  const int ct = pair % g.nct, cot = pair / g.nct;
  const int dymin_ = g.w.nphase > 0 ? g.ph_dymin[ph] : g.dymin, dxmin_ = g.w.nphase > 0 ? g.ph_dxmin[ph] : g.dxmin;
  float* const dpack_ = g.w.nphase > 0 ? g.w.ph_dpack[ph] : g.w.dpack;
  const int t_beg = split * g.tiles_per_wg, t_end = min(g.ntiles, t_beg + g.tiles_per_wg);
  if (t_beg >= t_end) return;

  // ---- fixed channel positions: prologue constants once ----
  const int cx = tid & 15, px0 = tid >> 4;     // A: slot column, halo pixels px0 + 16 i
  const int cy = tid % YS, py0 = tid / YS;     // dY: slot column, tile pixels py0 + YRS i
  SlotK<SLOT> kx, ky;
  kx.k0 = load_fv<SLOT>(sx.scale + ct * WP_CA + cx * SLOT); kx.k1 = load_fv<SLOT>(sx.shift + ct * WP_CA + cx * SLOT); kx.k2 = 0.f; kx.k3 = 0.f;
  ky.k0 = 0.f; ky.k1 = 0.f; ky.k2 = 0.f; ky.k3 = 0.f;
  if (PQ == 2) { ky.k0 = load_fv<SLOT>(sy.q + cot * NCO + cy * SLOT); ky.k1 = load_fv<SLOT>(sy.r + cot * NCO + cy * SLOT); }
  const T* xsrc = (const T*)sx.src + ct * WP_CA + cx * SLOT;
  const T* ysrc = (const T*)sy.src + cot * NCO + cy * SLOT;
  const T* ysrc2 = (const T*)sy.src2 + cot * NCO + cy * SLOT;
  const int ytap_ = g.w.nphase > 0 ? g.w.ph_ytap[ph] : sy.taps[0];
  const int ypy = (int)(signed char)(ytap_ & 0xff), ypx = (int)(signed char)((ytap_ >> 8) & 0xff);

  // per-thread constants of the tile walk: where each of this thread's slots sits relative to the tile origin (global element offset)
  // and in the LDS images, so that an interior tile costs one add per load and nothing per LDS write
  int xrel[NX], xlds[NX], yrel[NY], yrel2[PQ == 2 ? NY : 1], ylds[NY];
  unsigned xin = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int hp = px0 + 16 * i;
    const int hy = hp / WP_HWR, hx = hp - hy * WP_HWR;
    const int idx = hy * WP_HW + hx;
    if (hp < WP_HH * WP_HWR) xin |= 1u << i;
    xrel[i] = (hy * sx.Ws + hx) * sx.ld;
    xlds[i] = idx * 256 + ((cx * 16) ^ ((idx & 3) << 6));
  }
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    const int p = py0 + YRS * i;
    yrel[i] = ((p >> 4) * sy.Ws + (p & 15)) * sy.istride * sy.ld;
    if constexpr (PQ == 2) yrel2[i] = ((p >> 4) * sy.Ws + (p & 15)) * sy.istride * sy.ld2;
    ylds[i] = p * 256 + ((cy * 16) ^ ((p & 3) << 6));
  }

  V rx[NX], ry[NY], ry2[PQ == 2 ? NY : 1];
  unsigned okx = 0, oky = 0;  // validity bits of the slots in flight
  const int tiles_img = g.tiles_y * g.tiles_x;
  auto issue = [&](int tile) {
    const int b = tile / tiles_img, tr = tile - b * tiles_img;
    const int y0 = (tr / g.tiles_x) * WP_TH, x0 = (tr % g.tiles_x) * WP_TW;
    const int hy0 = y0 + dymin_, hx0 = x0 + dxmin_;
    const bool interior = hy0 >= 0 && hx0 >= 0 && hy0 + WP_HH <= sx.Hs && hx0 + WP_HWR <= sx.Ws && y0 + WP_TH <= a.Ho && x0 + WP_TW <= a.Wo;
    if (interior) {  // (workgroup-uniform) every slot is inside: base + constant offset
      const T* xb = xsrc + ((size_t)(b * sx.Hs + hy0) * sx.Ws + hx0) * sx.ld;
      const size_t ypix = (size_t)(b * sy.Hs + y0 * sy.istride + ypy) * sy.Ws + x0 * sy.istride + ypx;
      const T* yb = ysrc + ypix * sy.ld;
      const T* yb2 = ysrc2 + ypix * sy.ld2;
      okx = xin; oky = (1u << NY) - 1;
#pragma unroll
      for (int i = 0; i < NX; ++i) rx[i] = *(const V*)(xb + (((xin >> i) & 1) ? xrel[i] : 0));
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        ry[i] = *(const V*)(yb + yrel[i]);
        if constexpr (PQ == 2) ry2[i] = *(const V*)(yb2 + yrel2[i]);
      }
      return;
    }
    okx = 0; oky = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {  // branch-free: clamped address, zeroed at the write if outside
      const int hp = px0 + 16 * i;
      const int hy = hp / WP_HWR, hx = hp - hy * WP_HWR;
      const int y = hy0 + hy, x = hx0 + hx;
      if (hp < WP_HH * WP_HWR && (unsigned)y < (unsigned)sx.Hs && (unsigned)x < (unsigned)sx.Ws) okx |= 1u << i;
      const size_t pix = (size_t)(b * sx.Hs + min(max(y, 0), sx.Hs - 1)) * sx.Ws + min(max(x, 0), sx.Ws - 1);
      rx[i] = *(const V*)(xsrc + pix * sx.ld);
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int p = py0 + YRS * i;
      const int y = y0 + (p >> 4), x = x0 + (p & 15);
      if (y < a.Ho && x < a.Wo) oky |= 1u << i;
      const int syy = min(y, a.Ho - 1) * sy.istride + ypy, sxx = min(x, a.Wo - 1) * sy.istride + ypx;
      const size_t pix = (size_t)(b * sy.Hs + syy) * sy.Ws + sxx;
      ry[i] = *(const V*)(ysrc + pix * sy.ld);
      if constexpr (PQ == 2) ry2[i] = *(const V*)(ysrc2 + pix * sy.ld2);
    }
  };
  auto store = [&]() {
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if ((xin >> i) & 1) {
        const V v = bn_relu_slot(rx[i], kx);
        *(V*)(Xs + xlds[i]) = ((okx >> i) & 1) ? v : z;
      }
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      V v = ry[i];
      if constexpr (PQ == 2) v = eff_grad_slot(ry[i], ry2[i], ky);
      *(V*)(Ys + ylds[i]) = ((oky >> i) & 1) ? v : z;
    }
  };

  f32x16 acc[NTAP][NJ];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][j][i] = 0.f;

  // transposed-read lane geometry (see wgrad.hip / wg3.hip): group tg = lane >> 4 covers columns 16 (tg & 1) .., rows 8 (tg >> 1) + tq (+4)
  const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
  const int arow = 8 * (tg >> 1) + tq;
  const int xcolb = (32 * wave + 16 * (tg & 1) + 4 * tp) * 2;
  int xoff[NTAP];  // byte offset of (tile row 0, pixel arow, tap) in the halo image; the swizzle key (index & 3) is fixed per tap
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tw = g.w.nphase > 0 ? g.w.ph_xtaps[ph][t] : sx.taps[t];
    const int dy = (int)(signed char)(tw & 0xff) - dymin_, dx = (int)(signed char)((tw >> 8) & 0xff) - dxmin_;
    const int idx = dy * WP_HW + arow + dx;
    xoff[t] = idx * 256 + (xcolb ^ ((idx & 3) << 6));
  }
  int yoff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) yoff[j] = arow * 256 + (((32 * j + 16 * (tg & 1) + 4 * tp) * 2) ^ (tq << 6));

  auto contract = [&]() {
#pragma unroll 2
    for (int ms = 0; ms < WP_TH; ++ms) {  // one tile row = 16 pixels of the contraction per step
      const unsigned char* yp = Ys + ms * (16 * 256);
      V yf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) yf[j] = wp_frag<T>(wp_tr16(yp + yoff[j]), wp_tr16(yp + yoff[j] + 4 * 256));
      const unsigned char* xp = Xs + ms * (WP_HW * 256);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const V xf = wp_frag<T>(wp_tr16(xp + xoff[t]), wp_tr16(xp + xoff[t] + 4 * 256));
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = mma16(yf[j], xf, acc[t][j]);  // rows: output channel n, columns: input channel c
      }
    }
  };
  issue(t_beg);
  for (int tile = t_beg; tile < t_end; ++tile) {
    store();          // waits for this tile's loads
    __syncthreads();  // images complete
    if (tile + 1 < t_end) issue(tile + 1);
    contract();
    __syncthreads();  // all waves done with the images
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

static bool g_wgp = !lab_flag("DMM_NO_WGP");
void wgp_set_enabled(bool on) { g_wgp = on; }
// the wave-specialised form (wgpw.hip)
hipError_t launch_wgpw(const WgradArgs& a, int dtype, int ntap, int nj, int tiles_y, int tiles_x, int ntiles, int tiles_per_wg, int nsplit, int nct,
                       int ncot, int dymin, int dxmin, const int* ph_dymin, const int* ph_dxmin, int nwg, hipStream_t st);


template <typename T, int NTAP, int NJ, int PQ>
static hipError_t launch_wgp_t(const WgpArgs& g, int nwg, hipStream_t st) {
  auto kern = wgp_kernel<T, NTAP, NJ, PQ>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), WP_LDS, st, g);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_wgp_type(const WgpArgs& g, int ntap, int nj, int pq, int nwg, hipStream_t st) {
  if (ntap == 4 && nj == 2) return pq ? launch_wgp_t<T, 4, 2, 2>(g, nwg, st) : launch_wgp_t<T, 4, 2, 0>(g, nwg, st);
  if (ntap == 2 && nj == 4) return pq ? launch_wgp_t<T, 2, 4, 2>(g, nwg, st) : launch_wgp_t<T, 2, 4, 0>(g, nwg, st);
  if (ntap == 2 && nj == 2) return pq ? launch_wgp_t<T, 2, 2, 2>(g, nwg, st) : launch_wgp_t<T, 2, 2, 0>(g, nwg, st);
  return hipErrorNotSupported;
}

// Takes the first segment of a normal-form weight gradient whose taps (2 or 4) lie in a 2x2 box, in a 16-bit storage type, with
// the input a multiple of 128 channels and the output a multiple of 64.  Returns hipErrorNotSupported otherwise.  A second
// (8-channel raw-input) segment is NOT covered: the caller runs it through the generic kernel (see launch_wgrad).
hipError_t launch_wgp(const WgradArgs& a, int dtype, hipStream_t st) {
  if (!family_on(g_wgp, IMPL_WGP) || dtype == DT_F32 || a.nseg < 1 || a.nseg > 2) return hipErrorNotSupported;
  const Seg& x = a.seg[0];
  const Seg& y = a.dy;
  static const bool trace = lab_flag("DMM_WGP_TRACE");
  if (trace && !g_ctl.dry)
    fprintf(stderr, "wgp? nseg %d x: mode %d istride %d Hs %d Ws %d (Ho %d Wo %d) scale %d C %d Cpad %d ntaps %d | y: mode %d ntaps %d istride %d Hs %d Ws %d C %d | N %d Npad %d\n",
            a.nseg, x.mode, x.istride, x.Hs, x.Ws, a.Ho, a.Wo, x.scale != nullptr, x.C, x.Cpad, x.ntaps, y.mode, y.ntaps, y.istride, y.Hs, y.Ws, y.C, a.N, a.Npad);
  if (x.mode != G_PLAIN || x.istride != 1 || x.Hs != a.Ho || x.Ws != a.Wo || x.scale == nullptr || x.C % WP_CA || x.Cpad != x.C) return hipErrorNotSupported;
  static const bool ws = !lab_flag("DMM_NO_WGPW");   // wave-specialised form (wgpw.hip) for the materialised output gradient
  // one tap: the (0, 0) parity phase of a ConvTranspose (stride-2 gradient rows) and the decoder's plain 1x1 convolutions conv_reduce
  // (reference M:150-153; C_in = 1024 ... 256 -> C_in / 2 at the block resolutions), only in the wave-specialised form.  (The dense layers'
  // 128-wide bottlenecks never come here: bw1.hip fuses their weight gradient with the data gradient.)
  static const bool ws1x1 = lab_flag("DMM_WGPW_1X1");   // measured (round 5): 0.198 / 0.192 / 0.066 ms against the generic kernel's 0.178 / 0.171 / 0.064: off
  if (x.ntaps == 1 && !(ws && (y.istride == 2 || (ws1x1 && y.istride == 1 && x.taps[0] == 0)) && y.q == nullptr && (a.nphase == 0 || a.nphase == 4) && a.nseg == 1))
    return hipErrorNotSupported;
  if (x.ntaps != 1 && x.ntaps != 2 && x.ntaps != 4) return hipErrorNotSupported;
  if (a.nseg == 2 && !(a.seg[1].C == 8 && a.seg[1].nchunks >= 1)) return hipErrorNotSupported;
  if (y.mode != G_PLAIN || y.ntaps != 1 || y.scale != nullptr || (y.istride != 1 && y.istride != 2)) return hipErrorNotSupported;
  if (y.Hs != a.Ho * y.istride || y.Ws != a.Wo * y.istride) return hipErrorNotSupported;
  if (a.N % 64 || a.Npad < a.N || y.C != a.N) return hipErrorNotSupported;
  if (a.nphase < 0 || a.nphase > 4 || (a.nphase > 0 && a.nseg != 1)) return hipErrorNotSupported;
  int dymin = 127, dxmin = 127;
  int ph_dymin[4] = {0, 0, 0, 0}, ph_dxmin[4] = {0, 0, 0, 0};
  // a multi-phase launch whose phases differ in their tap counts (the ConvTranspose's 1, 2, 2, 4: WgradArgs::ph_ntaps): wgpw.hip only
  bool mixed = false;
  for (int ph = 0; ph < a.nphase; ++ph) mixed = mixed || (a.ph_ntaps[ph] != 0 && a.ph_ntaps[ph] != x.ntaps);
  if (mixed && !(ws && y.q == nullptr && a.nphase == 4 && y.istride == 2)) return hipErrorNotSupported;
  for (int ph = 0; ph < std::max(1, a.nphase); ++ph) {   // every phase: taps inside a 2x2 box, each offset once; parity inside the stride
    const short* taps = a.nphase > 0 ? a.ph_xtaps[ph] : x.taps;
    const int pnt = (a.nphase > 0 && a.ph_ntaps[ph] != 0) ? a.ph_ntaps[ph] : x.ntaps;
    if (pnt != 1 && pnt != 2 && pnt != 4) return hipErrorNotSupported;
    int ymin = 127, xmin = 127, ymax = -128, xmax = -128;
    bool seen[4] = {false, false, false, false};
    for (int t = 0; t < pnt; ++t) {
      const int dy = (int)(signed char)(taps[t] & 0xff), dx = (int)(signed char)((taps[t] >> 8) & 0xff);
      ymin = std::min(ymin, dy); ymax = std::max(ymax, dy); xmin = std::min(xmin, dx); xmax = std::max(xmax, dx);
    }
    if (ymax - ymin > 1 || xmax - xmin > 1) return hipErrorNotSupported;
    for (int t = 0; t < pnt; ++t) {
      const int dy = (int)(signed char)(taps[t] & 0xff) - ymin, dx = (int)(signed char)((taps[t] >> 8) & 0xff) - xmin;
      if (seen[dy * 2 + dx]) return hipErrorNotSupported;
      seen[dy * 2 + dx] = true;
    }
    const int yt = a.nphase > 0 ? a.ph_ytap[ph] : y.taps[0];
    const int py = (int)(signed char)(yt & 0xff), px = (int)(signed char)((yt >> 8) & 0xff);
    if (py < 0 || px < 0 || py >= y.istride || px >= y.istride) return hipErrorNotSupported;
    if (a.nphase > 0 && a.ph_dpack[ph] == nullptr) return hipErrorNotSupported;
    ph_dymin[ph] = ymin; ph_dxmin[ph] = xmin;
    if (ph == 0) { dymin = ymin; dxmin = xmin; }
  }
  const int ntap = x.ntaps;
  const int nj = (ntap <= 2 && a.N % 128 == 0 && a.nphase == 0) ? 4 : 2;   // (multi-phase launches: 64 output channels per workgroup in every phase)
  if (g_ctl.dry) return hipSuccess;
  WgpArgs g;
  g.w = a;
  g.dymin = dymin; g.dxmin = dxmin;
  g.tiles_y = (a.Ho + WP_TH - 1) / WP_TH;
  g.tiles_x = (a.Wo + WP_TW - 1) / WP_TW;
  g.ntiles = a.B * g.tiles_y * g.tiles_x;
  g.nct = x.C / WP_CA;
  g.ncot = a.N / (32 * nj);
  // one workgroup per CU (its accumulators fill the register file); every workgroup ends with NTAP x 128 x NCO x 4 bytes of atomics
  static const int cus = [] { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev);
                              return (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  static const int target = lab_int("DMM_WGP_WGS", 0);
  const int pairs = g.nct * g.ncot;
  const int nph = std::max(1, a.nphase);
  for (int ph = 0; ph < 4; ++ph) { g.ph_dymin[ph] = ph_dymin[ph]; g.ph_dxmin[ph] = ph_dxmin[ph]; }
  // (multi-phase: the workgroups of all phases together fill the chip once; whole groups of 8 units, see the kernel)
  int nsplit = std::max(1, ((target > 0 ? target : cus) / nph + pairs - 1) / pairs);
  nsplit = std::min(nsplit, g.ntiles);
  g.tiles_per_wg = (g.ntiles + nsplit - 1) / nsplit;
  g.nsplit = (g.ntiles + g.tiles_per_wg - 1) / g.tiles_per_wg;
  const int units = g.nsplit * pairs;
  const int nwg = a.nphase > 0 ? ((units + 7) / 8) * 8 * nph : units;
  const int pq = y.q ? 2 : 0;
  if (ws && pq == 0) {
    note_impl(IMPL_WGPW);   // (beside IMPL_WGP, which the dispatcher notes: the family is wgp, this says which form ran)
    return launch_wgpw(a, dtype, ntap, nj, g.tiles_y, g.tiles_x, g.ntiles, g.tiles_per_wg, g.nsplit, g.nct, g.ncot, dymin, dxmin, ph_dymin, ph_dxmin,
                       nwg, st);
  }
  if (trace) fprintf(stderr, "wgp: ntap %d nj %d pq %d pairs %d nsplit %d tiles/wg %d\n", ntap, nj, pq, pairs, g.nsplit, g.tiles_per_wg);
  return dtype == DT_F16 ? launch_wgp_type<f16>(g, ntap, nj, pq, nwg, st) : launch_wgp_type<bf16>(g, ntap, nj, pq, nwg, st);
}

bool wgp_handles(const WgradArgs& a, int dtype) {
  const LaunchCtl keep = g_ctl;
  g_ctl.dry = true;
  const hipError_t e = launch_wgp(a, dtype, nullptr);
  g_ctl = keep;
  return e == hipSuccess;
}

}  // namespace dmm
