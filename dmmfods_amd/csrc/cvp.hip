// Forward of the parity-phase convolutions with many input channels - the decoder's ConvTranspose2d 3x3 stride 2 (reference
// M:155-160), run as four output-parity phases of 1, 2, 2 and 4 taps over the input grid - on an LDS halo tile.  gfx950, 16-bit
// storage types.
//
//   out[S p + parity][n] = sum over taps t, channels c of  A[p + t][c] * W_t[c][n],     A = relu(bn(x))
// The generic implicit-GEMM kernel (igemm.hip) gathers, bounds-tests and normalises A once per tap AND per 128-column output
// tile: 9/4 x 4 = up to 36 times per element for the first stage (1024 -> 512 channels), and does so inside every K stage.
// Here a workgroup owns an 8 x 16 pixel tile x 128 output channels and walks the input channels in groups of 128:
//   * per group the 9 x 17 pixel halo of A is loaded once (16-byte slots, branch-free, the next group's loads in flight behind the
//     current group's MFMAs), normalised once per element and written to an LDS image (pixel pitch 272 bytes: the 16 lanes of a
//     ds_read_b128 group hit 16 different slots); a tap is an address offset of the A fragment;
//   * the packed weights of a stage (2 chunks x 128 columns) go through a double-buffered LDS image (registers -> ds_write, XOR
//     swizzle as in igemm.hip), prefetched one stage ahead;
//   * one barrier per stage of 16 MFMAs per wave and nothing but fragment reads and MFMAs between barriers;
//   * epilogue as in conv3.hip: values staged as T, BatchNorm sums straight from the accumulator layout, parity-strided store.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int CP_TH = 8, CP_TW = 16, CP_HH = 9, CP_HW = 17;
constexpr int CP_CA = 128;                      // input channels per group
constexpr int CP_BN = 128;                      // output channels per workgroup
constexpr int CP_PP = CP_CA * 2 + 16;           // pixel pitch of the halo image
constexpr int CP_RP = (CP_HW * CP_PP + 255) / 256 * 256;
constexpr int CP_X_BYTES = CP_HH * CP_RP;       // 43 KB
constexpr int CP_B_STAGE = 2 * CP_BN * 64;      // 16 KB: two 64-byte chunks of K for 128 columns
constexpr int CP_MAIN = CP_X_BYTES + 2 * CP_B_STAGE;
constexpr int CP_CPITCH = CP_BN + 8;            // staging pitch (elements of T)
constexpr int CP_EXTRA = BM * 4 + 4 * 2 * CP_BN * 4;  // rowpix + per-wave column partials
constexpr int CP_LDS = CP_MAIN + CP_EXTRA;
static_assert(BM * CP_CPITCH * 2 <= CP_MAIN, "staging fits the operand images");
static_assert(2 * CP_LDS <= 160 * 1024, "two workgroups per CU");

struct CvpArgs {
  ConvArgs c;
  int tiles_y, tiles_x, ntn;  // pixel tiles, 128-column tiles
  int dymin, dxmin;           // origin of the tap box
};

template <typename T, int NTAP>
__global__ __launch_bounds__(NTHREADS, 2) void cvp_kernel(const CvpArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8, BN = CP_BN, NT = BN / 32;
  constexpr int NX = (CP_HH * CP_HW * (CP_CA / SLOT) + NTHREADS - 1) / NTHREADS;  // 10 halo slots per thread
  constexpr int NST = NTAP * 2;                                                   // stages per channel group
  const ConvArgs& a = g.c;
  const Seg& sx = a.seg[0];

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Bs = smem + CP_X_BYTES;
  int* rowpix = (int*)(smem + CP_MAIN);
  float* wpart = (float*)(smem + CP_MAIN + BM * 4);  // [wave][2][BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lbid % g.ntn;
  int tile = lbid / g.ntn;
  const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
  const int ty_i = tile % g.tiles_y;
  const int b = tile / g.tiles_y;
  const int y0 = ty_i * CP_TH, x0 = tx_i * CP_TW, n0 = ntile * BN;

  if (tid < BM) {
    const int y = y0 + (tid >> 4), x = x0 + (tid & 15);
    rowpix[tid] = (y < a.Ho && x < a.Wo) ? (b * a.Hout + y * a.ostride + a.py) * a.Wout + x * a.ostride + a.px : -1;
  }

  // ---- the halo slots of this thread: fixed pixel positions, channel column cx of the current group ----
  const int cx = tid & 15, px0 = tid >> 4;
  int xlds[NX];
  size_t xpix[NX];
  unsigned xin = 0, okx = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int hp = px0 + 16 * i;
    const int hy = hp / CP_HW, hx = hp - hy * CP_HW;
    const int y = y0 + g.dymin + hy, x = x0 + g.dxmin + hx;
    if (hp < CP_HH * CP_HW) {
      xin |= 1u << i;
      if ((unsigned)y < (unsigned)sx.Hs && (unsigned)x < (unsigned)sx.Ws) okx |= 1u << i;
    }
    xlds[i] = hy * CP_RP + hx * CP_PP + cx * 16;
    xpix[i] = ((size_t)(b * sx.Hs + min(max(y, 0), sx.Hs - 1)) * sx.Ws + min(max(x, 0), sx.Ws - 1)) * sx.ld + cx * SLOT;
  }
  const T* xsrc = (const T*)sx.src;
  V rx[NX];
  auto issue_halo = [&](int grp) {  // branch-free: clamped addresses, zeroed at the write if outside the picture
#pragma unroll
    for (int i = 0; i < NX; ++i) rx[i] = *(const V*)(xsrc + xpix[i] + grp * CP_CA);
  };
  auto store_halo = [&](int grp) {
    SlotK<SLOT> kx;
    kx.k0 = load_fv<SLOT>(sx.scale + grp * CP_CA + cx * SLOT); kx.k1 = load_fv<SLOT>(sx.shift + grp * CP_CA + cx * SLOT); kx.k2 = 0.f; kx.k3 = 0.f;
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if ((xin >> i) & 1) *(V*)(Xs + xlds[i]) = ((okx >> i) & 1) ? bn_relu_slot(rx[i], kx) : z;
  };

  // ---- weights: a stage = chunks (c0, c0 + 1) x 128 columns = 2 x 8 KB contiguous; thread -> pieces tid, tid + 256 of each ----
  const T* wp = (const T*)a.wpack;
  const int cpt = sx.Cpad / 32;  // chunks per tap
  V rb[2][2];
  int blds[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = tid + NTHREADS * j, row = piece >> 2, slot = piece & 3;
    blds[j] = row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
  }
  auto issue_b = [&](int grp, int st) {
    const int c0 = (st >> 1) * cpt + grp * 4 + 2 * (st & 1);
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const T* src = wp + ((size_t)(c0 + uu) * a.Npad + n0) * 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) rb[uu][j] = *(const V*)(src + (size_t)(tid + NTHREADS * j) * SLOT);
    }
  };
  auto store_b = [&](int buf) {
    unsigned char* B = Bs + buf * CP_B_STAGE;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int j = 0; j < 2; ++j) *(V*)(B + uu * (BN * 64) + blds[j]) = rb[uu][j];
  };

  // ---- fragments: wave w owns tile rows 2w, 2w + 1 (32 pixels) x 128 columns ----
  int aoff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tw = sx.taps[t];
    const int dy = (int)(signed char)(tw & 0xff) - g.dymin, dx = (int)(signed char)((tw >> 8) & 0xff) - g.dxmin;
    aoff[t] = (2 * wave + (r >> 4) + dy) * CP_RP + ((r & 15) + dx) * CP_PP + h * 16;
  }
  const int bsw = (r >> 2) & 3;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int ngrp = sx.C / CP_CA;
  issue_halo(0);
  issue_b(0, 0);
  int buf = 0;
  for (int grp = 0; grp < ngrp; ++grp) {
    store_halo(grp);  // (the barrier that ended the previous group made the image free)
    if (grp + 1 < ngrp) issue_halo(grp + 1);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      store_b(buf);
      __syncthreads();  // this stage's weights (and, in the first stage of a group, the halo image) are complete
      if (st + 1 < NST) issue_b(grp, st + 1);
      else if (grp + 1 < ngrp) issue_b(grp + 1, 0);
      const unsigned char* B = Bs + buf * CP_B_STAGE;
      const unsigned char* A = Xs + aoff[st >> 1] + (st & 1) * 128;
#pragma unroll
      for (int uu = 0; uu < 2; ++uu)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const V av = *(const V*)(A + uu * 64 + s * 32);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const V bv = *(const V*)(B + (uu * BN + 32 * t + r) * 64 + (((2 * s + h) ^ bsw) << 4));
            acc[t] = mma16(av, bv, acc[t]);
          }
        }
      buf ^= 1;
    }
    __syncthreads();  // every wave is done with the halo image of this group
  }

  // ---- epilogue (conv3.hip): stage as T, column sums of the stored values straight from the accumulator layout ----
  T* Cs = (T*)smem;
  float ps1[NT], ps2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ps1[t] = 0.f; ps2[t] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
      const T v = from_f32<T>(acc[t][i]);
      Cs[row * CP_CPITCH + 32 * t + r] = v;
      if (rowpix[row] >= 0) { const float f = to_f32(v); ps1[t] += f; ps2[t] = fmaf(f, f, ps2[t]); }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ps1[t] += __shfl_xor(ps1[t], 32, 64);
    ps2[t] += __shfl_xor(ps2[t], 32, 64);
    if (h == 0) { wpart[(wave * 2 + 0) * BN + 32 * t + r] = ps1[t]; wpart[(wave * 2 + 1) * BN + 32 * t + r] = ps2[t]; }
  }
  __syncthreads();
  constexpr int NCV = BN / SLOT, RPP = NTHREADS / NCV, NIT = BM / RPP;
  const int cv = tid % NCV, rr = tid / NCV;
  const int n = n0 + cv * SLOT;
  T* out = (T*)a.out;
  if (n < a.N) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = rr + RPP * i;
      const int pix = rowpix[row];
      if (pix >= 0) *(V*)(out + (size_t)pix * a.ldo + n) = *(const V*)(Cs + row * CP_CPITCH + cv * SLOT);
    }
  }
  if (a.stat_sum == nullptr) return;
  if (tid < 2 * BN) {
    const int col = tid % BN, which = tid / BN;
    if (n0 + col < a.N) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += (double)wpart[(w * 2 + which) * BN + col];
      const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
      atomic_add_f64((which ? a.stat_sq : a.stat_sum) + rep + n0 + col, s);
    }
  }
}

static bool g_cvp = getenv("DMM_NO_CVP") == nullptr;
void cvp_set_enabled(bool on) { g_cvp = on; }

static thread_local bool g_cvp_dry = false;

template <typename T, int NTAP>
static hipError_t launch_cvp_t(const CvpArgs& g, int nwg, hipStream_t st) {
  auto kern = cvp_kernel<T, NTAP>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), CP_LDS, st, g);
  return hipGetLastError();
}

// Takes a forward launch (EPI_STORE) with one plain segment of a multiple of 128 BN+ReLU-normalised input channels whose 1, 2 or
// 4 taps lie in a 2x2 box, a multiple of 128 padded output columns, 16-bit storage.  Returns hipErrorNotSupported otherwise.
hipError_t launch_cvp(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  if (!g_cvp || dtype == DT_F32 || epi != EPI_STORE || a.nseg != 1 || a.pool2) return hipErrorNotSupported;
  const Seg& x = a.seg[0];
  if (x.mode != G_PLAIN || x.istride != 1 || x.Hs != a.Ho || x.Ws != a.Wo || x.scale == nullptr || x.C % CP_CA || x.Cpad != x.C)
    return hipErrorNotSupported;
  if (x.ntaps != 1 && x.ntaps != 2 && x.ntaps != 4) return hipErrorNotSupported;
  if (x.ntaps == 1 && a.ostride == 1) return hipErrorNotSupported;  // plain 1x1 convolutions stay with igemm's lean path
  if (a.Npad % CP_BN || a.out == nullptr) return hipErrorNotSupported;
  int dymin = 127, dxmin = 127, dymax = -128, dxmax = -128;
  for (int t = 0; t < x.ntaps; ++t) {
    const int dy = (int)(signed char)(x.taps[t] & 0xff), dx = (int)(signed char)((x.taps[t] >> 8) & 0xff);
    dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax; dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
  }
  if (dymax - dymin > 1 || dxmax - dxmin > 1) return hipErrorNotSupported;
  if (g_cvp_dry) return hipSuccess;
  CvpArgs g;
  g.c = a;
  g.dymin = dymin; g.dxmin = dxmin;
  g.tiles_y = (a.Ho + CP_TH - 1) / CP_TH;
  g.tiles_x = (a.Wo + CP_TW - 1) / CP_TW;
  g.ntn = a.Npad / CP_BN;
  const int nwg = a.B * g.tiles_y * g.tiles_x * g.ntn;
  if (dtype == DT_F16) return x.ntaps == 4 ? launch_cvp_t<f16, 4>(g, nwg, st) : (x.ntaps == 2 ? launch_cvp_t<f16, 2>(g, nwg, st) : launch_cvp_t<f16, 1>(g, nwg, st));
  return x.ntaps == 4 ? launch_cvp_t<bf16, 4>(g, nwg, st) : (x.ntaps == 2 ? launch_cvp_t<bf16, 2>(g, nwg, st) : launch_cvp_t<bf16, 1>(g, nwg, st));
}

bool cvp_handles(const ConvArgs& a, int dtype, int epi) {
  g_cvp_dry = true;
  const hipError_t e = launch_cvp(a, dtype, epi, nullptr);
  g_cvp_dry = false;
  return e == hipSuccess;
}

}  // namespace dmm
