// Weight gradient of the dense layers' 3x3 growth convolution (torchvision _DenseLayer.conv2, 128 -> 32 channels; reference call
// sites M:85-92, M:169-176) on LDS tiles, gfx950, 16-bit storage types.
//
//   dW[n][c][tap] = sum over pixels p of  A[p][c] * dYeff[p + t_tap][n]        (transposed form of wgrad.hip: taps on the dY side)
//     A     = relu(bn2(y1))           128 channels, normalised ONCE per pixel
//     dYeff = g + q + r*x             32 channels of the block's gradient buffer with the deferred BatchNorm-backward correction
// The generic kernel (wgrad.hip) cuts the 288 x 128 result into workgroup slices that each re-gather (and re-normalise) both
// operands, tap by tap.  Here a PERSISTENT workgroup (one per CU, one wave per SIMD) keeps the WHOLE 9 x 32 x 128 result in its
// accumulators (wave w owns channels 32w .. 32w+31 of A: nine 32x32 tiles = 144 registers) and walks over 8 x 16 pixel tiles:
//   * per tile the A tile (128 px x 128 ch) and the 10 x 18 pixel dYeff halo (32 ch) are loaded once, the prologues applied once
//     per element, and written to row-major LDS images; a tap is an address offset into the halo image;
//   * the contraction index is the pixel, the slow index of both images, so both MFMA operands are read with the transposing
//     read ds_read_b64_tr_b16 (as in wgrad.hip); per 16-pixel step: 2 reads for A, 18 for the nine taps, 9 MFMAs;
//   * the next tile's loads are issued before the MFMAs of the current one and land in registers meanwhile;
//   * at the end every workgroup adds its 147 KB partial result to the packed gradient with fp32 atomics (two 128-byte segments
//     per wave instruction).  Atomic bytes, not arithmetic, bound small layers, so the number of workgroups is chosen to balance
//     tiles per workgroup against workgroups x 147 KB (launch_wg3).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "gather.h"

#ifndef WG3_DBG
#define WG3_DBG 0  // timing experiments only (tools/build_variant.sh): 1 no atomics at the end of the walk
#endif
namespace dmm {

constexpr int W3_TH = 8, W3_TW = 16, W3_HH = 10, W3_HW = 18;
constexpr int W3_CA = 128, W3_CY = 32;                 // channels of A and of dY
constexpr int W3_A_BYTES = BM * W3_CA * 2;             // 32 KB, 256-byte rows, 64-byte granule XOR-ed with (row & 3)
constexpr int W3_Y_BYTES = W3_HH * W3_HW * W3_CY * 2;  // 11.25 KB, 64-byte rows
constexpr int W3_LDS = W3_A_BYTES + W3_Y_BYTES;

struct Wg3Args {
  WgradArgs w;
  int tiles_y, tiles_x, ntiles, tiles_per_wg;
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

// PQ = prologue of dY: 0 none (materialised gradient), 2 effective gradient (q, r of the 16-bit form)
template <typename T, int PQ>
__global__ __launch_bounds__(NTHREADS, 1) void wg3_kernel(const Wg3Args g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  constexpr int NA = BM * (W3_CA / SLOT) / NTHREADS;                              // 8 A slots per thread
  constexpr int NY = (W3_HH * W3_HW * (W3_CY / SLOT) + NTHREADS - 1) / NTHREADS;  // 3 dY slots per thread
  const WgradArgs& a = g.w;
  const Seg& sy_ = a.seg[0];  // dY, nine taps
  const Seg& sa = a.dy;       // A, pixel aligned

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Ys = smem + W3_A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t_beg = blockIdx.x * g.tiles_per_wg, t_end = min(g.ntiles, t_beg + g.tiles_per_wg);
  if (t_beg >= t_end) return;

  // ---- fixed channel positions: prologue constants once ----
  const int ca = tid & 15, pa0 = tid >> 4;  // A: slot column, pixels pa0 + 16 i
  const int cy = tid & 3, hy0 = tid >> 2;   // dY: slot column, halo pixels hy0 + 64 i
  SlotK<SLOT> ka, ky;
  ka.k0 = load_fv<SLOT>(sa.scale + ca * SLOT); ka.k1 = load_fv<SLOT>(sa.shift + ca * SLOT); ka.k2 = 0.f; ka.k3 = 0.f;
  ky.k0 = 0.f; ky.k1 = 0.f; ky.k2 = 0.f; ky.k3 = 0.f;
  if (PQ == 2) { ky.k0 = load_fv<SLOT>(sy_.q + cy * SLOT); ky.k1 = load_fv<SLOT>(sy_.r + cy * SLOT); }
  const T* asrc = (const T*)sa.src + ca * SLOT;
  const T* ysrc = (const T*)sy_.src + cy * SLOT;
  const T* ysrc2 = (const T*)sy_.src2 + cy * SLOT;

  V ra[NA], ry[NY], ry2[PQ == 2 ? NY : 1];
  unsigned oka = 0, oky = 0;  // validity bits of the slots in flight
  const int tiles_img = g.tiles_y * g.tiles_x;
  auto issue = [&](int tile) {
    const int b = tile / tiles_img, tr = tile - b * tiles_img;
    const int y0 = (tr / g.tiles_x) * W3_TH, x0 = (tr % g.tiles_x) * W3_TW;
    oka = 0; oky = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {  // branch-free: clamped address, zeroed at the write if outside
      const int p = pa0 + 16 * i;
      const int y = y0 + (p >> 4), x = x0 + (p & 15);
      if (y < a.Ho && x < a.Wo) oka |= 1u << i;
      const size_t pix = (size_t)(b * sa.Hs + min(y, sa.Hs - 1)) * sa.Ws + min(x, sa.Ws - 1);
      ra[i] = *(const V*)(asrc + pix * sa.ld);
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int hp = hy0 + 64 * i;
      const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      if (hp < W3_HH * W3_HW && (unsigned)y < (unsigned)sy_.Hs && (unsigned)x < (unsigned)sy_.Ws) oky |= 1u << i;
      const size_t pix = (size_t)(b * sy_.Hs + min(max(y, 0), sy_.Hs - 1)) * sy_.Ws + min(max(x, 0), sy_.Ws - 1);
      ry[i] = *(const V*)(ysrc + pix * sy_.ld);
      if constexpr (PQ == 2) ry2[i] = *(const V*)(ysrc2 + pix * sy_.ld2);
    }
  };
  auto store = [&]() {
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int p = pa0 + 16 * i;
      const V v = bn_relu_slot(ra[i], ka);
      *(V*)(As + p * 256 + ((ca * 16) ^ ((p & 3) << 6))) = ((oka >> i) & 1) ? v : z;
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int hp = hy0 + 64 * i;
      if (hp < W3_HH * W3_HW) {
        V v = ry[i];
        if constexpr (PQ == 2) v = eff_grad_slot(ry[i], ry2[i], ky);
        *(V*)(Ys + hp * 64 + cy * 16) = ((oky >> i) & 1) ? v : z;
      }
    }
  };

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

  issue(t_beg);
  for (int tile = t_beg; tile < t_end; ++tile) {
    store();          // waits for this tile's loads
    __syncthreads();  // images complete
    if (tile + 1 < t_end) issue(tile + 1);
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
    __syncthreads();  // all waves done with the images
  }

  // ---- add the partial result to the packed gradient: dP[chunk = tap][c][n] ----
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (!(WG3_DBG & 1) || acc[t][i] == 1.2345e33f) atomic_add_f32(a.dpack + ((size_t)t * a.Npad + c) * 32 + r, acc[t][i]);
    }
}

static bool g_wg3 = getenv("DMM_NO_WG3") == nullptr;
void wg3_set_enabled(bool on) { g_wg3 = on; }

template <typename T, int PQ>
static hipError_t launch_wg3_t(const Wg3Args& g, int nwg, hipStream_t st) {
  if (g_ctl.dry) return hipSuccess;
  auto kern = wg3_kernel<T, PQ>;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), W3_LDS, st, g);
  return hipGetLastError();
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
  // time ~ tiles/nwg * t_tile + nwg * (147 KB of fp32 atomics at the chip-wide atomic rate): minimum at nwg ~ sqrt(14 * tiles)
  if (g_ctl.dry) return hipSuccess;
  static const int cus = [] { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev);
                              return (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  int nwg = (int)std::lround(std::sqrt(14.0 * g.ntiles));
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
