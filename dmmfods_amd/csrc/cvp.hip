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
#include <algorithm>
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
constexpr int CP_EXTRA = BM * 4;                 // rowpix (the epilogue's column partials / fp64 sums overlay the dead operand images)
constexpr int CP_LDS = CP_MAIN + CP_EXTRA;
static_assert(BM * (CP_BN + 4) * 4 + 4 * 2 * CP_BN * 4 <= CP_MAIN, "fp32 staging + partials fit the operand images");
constexpr int CP_PART = BM * (CP_BN + 4) * 4;   // offset of the partials inside the dead images (behind the largest staging form)
static_assert(2 * CP_LDS <= 160 * 1024, "two workgroups per CU");

struct CvpArgs {
  ConvArgs c;
  int tiles_y, tiles_x, ntn;  // pixel tiles, 128-column tiles
  int dymin, dxmin;           // origin of the tap box
  // multi-phase launches (ConvArgs::nphase = 4): workgroups per phase, per XCD and phase (rounded up), the phases longest first, the
  // origin of each phase's tap box
  int per, per8;
  signed char order[4], ph_dymin[4], ph_dxmin[4];
};
// What distinguishes the phases of a multi-phase launch (the single-phase launch: the fields of ConvArgs / CvpArgs themselves).
struct CvpPhase { const void* wpack; const short* taps; int py, px, dymin, dxmin; };

template <typename T, int NTAP>
__device__ __forceinline__ void cvp_body(const CvpArgs& g, const int lbid, const CvpPhase P) {
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
  float* wpart = (float*)(smem + CP_PART);  // [wave][2][BN]; only touched after the K loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ntile = lbid % g.ntn;
  int tile = lbid / g.ntn;
  const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
  const int ty_i = tile % g.tiles_y;
  const int b = tile / g.tiles_y;
  const int y0 = ty_i * CP_TH, x0 = tx_i * CP_TW, n0 = ntile * BN;

  if (tid < BM) {
    const int y = y0 + (tid >> 4), x = x0 + (tid & 15);
    rowpix[tid] = (y < a.Ho && x < a.Wo) ? (b * a.Hout + y * a.ostride + P.py) * a.Wout + x * a.ostride + P.px : -1;
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
    const int y = y0 + P.dymin + hy, x = x0 + P.dxmin + hx;
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
  const T* wp = (const T*)P.wpack;
  const int cpt = sx.Cpad / 32;  // chunks per tap
  // two register sets, filled two stages ahead: a stage's MFMAs are shorter than an L2 round trip.  Every load is issued
  // unconditionally (past the end: a valid chunk again) so that the compiler's counted waits leave the other set in flight.
  struct BRegs { V v[2][2]; };
  BRegs rb0, rb1;
  int blds[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = tid + NTHREADS * j, row = piece >> 2, slot = piece & 3;
    blds[j] = row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
  }
  const int ngrp = sx.C / CP_CA;
  auto issue_b = [&](BRegs& R, int grp, int st) {  // stage st of group grp; st may run past the group (the next group's stages)
    int gq = grp + (st >= NST ? 1 : 0);
    const int sq = st >= NST ? st - NST : st;
    gq = min(gq, ngrp - 1);
    const int c0 = (sq >> 1) * cpt + gq * 4 + 2 * (sq & 1);
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const T* src = wp + ((size_t)(c0 + uu) * a.Npad + n0) * 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) R.v[uu][j] = *(const V*)(src + (size_t)(tid + NTHREADS * j) * SLOT);
    }
  };
  auto store_b = [&](const BRegs& R, int buf) {
    unsigned char* B = Bs + buf * CP_B_STAGE;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int j = 0; j < 2; ++j) *(V*)(B + uu * (BN * 64) + blds[j]) = R.v[uu][j];
  };

  // ---- fragments: wave w owns tile rows 2w, 2w + 1 (32 pixels) x 128 columns ----
  int aoff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tw = P.taps[t];
    const int dy = (int)(signed char)(tw & 0xff) - P.dymin, dx = (int)(signed char)((tw >> 8) & 0xff) - P.dxmin;
    aoff[t] = (2 * wave + (r >> 4) + dy) * CP_RP + ((r & 15) + dx) * CP_PP + h * 16;
  }
  const int bsw = (r >> 2) & 3;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  issue_halo(0);
  issue_b(rb0, 0, 0);
  issue_b(rb1, 0, 1);
  for (int grp = 0; grp < ngrp; ++grp) {
    store_halo(grp);  // (the barrier that ended the previous group made the image free)
    issue_halo(min(grp + 1, ngrp - 1));
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      BRegs& R = (st & 1) ? rb1 : rb0;  // NST is even: stage parity = register set = LDS buffer, across groups as well
      store_b(R, st & 1);
      __syncthreads();  // this stage's weights (and, in the first stage of a group, the halo image) are complete
      issue_b(R, grp, st + 2);
      const unsigned char* B = Bs + (st & 1) * CP_B_STAGE;
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
    wpart[(wave * 2 + h) * BN + 32 * t + r] = fold_swap32(ps1[t], ps2[t]);  // lane half 0: the sum, half 1: the sum of squares
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

template <typename T, int NTAP>
__global__ __launch_bounds__(NTHREADS, 2) void cvp_kernel(const CvpArgs g) {
  const CvpPhase P = {g.c.wpack, g.c.seg[0].taps, g.c.py, g.c.px, g.dymin, g.dxmin};
  cvp_body<T, NTAP>(g, xcd_remap(blockIdx.x, gridDim.x), P);
}

// All parity phases of a ConvTranspose in one launch.  Workgroup b runs on XCD b % 8 and is the (b / 8)-th the XCD is handed: the
// phases are dealt in the order g.order (most taps first), per8 workgroups per XCD and phase, and inside a phase an XCD owns a
// contiguous range of the tile order (as xcd_remap gives a single-phase launch).
template <typename T>
__global__ __launch_bounds__(NTHREADS, 2) void cvp_multi_kernel(const CvpArgs g) {
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int slot = q / g.per8, j = q - slot * g.per8;
  const int lbid = xcd * g.per8 + j;
  if (lbid >= g.per) return;   // (workgroup-uniform: the rounding of per8)
  const int ph = g.order[slot];
  const CvpPhase P = {g.c.ph_wpack[ph], g.c.ph_taps0[ph], g.c.ph_py[ph], g.c.ph_px[ph], g.ph_dymin[ph], g.ph_dxmin[ph]};
  const int nt = g.c.ph_ntaps[ph];
  if (nt == 4) cvp_body<T, 4>(g, lbid, P);
  else if (nt == 2) cvp_body<T, 2>(g, lbid, P);
  else cvp_body<T, 1>(g, lbid, P);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Data gradient of the ConvTranspose stages: d a[p][c] = sum over the 9 taps (dy, dx) of dY[2 p + (dy, dx)][n] * W_t[n][c], fused
// with the BN+ReLU backward of the input's norm (EPI_BNBWD of igemm.hip).  The stride-2 gather splits by the PARITY of the
// gathered position into four sub-grids of dY, on each of which the taps form a box of 1, 2, 2 or 4 (exactly the forward
// phases, mirrored): the kernel above with the halo addressed at stride 2, the four classes accumulated into one tile.
struct CvdArgs {
  ConvArgs c;
  int tiles_y, tiles_x, ntn;
  int tapidx[16];  // the launch's taps ordered by class (even, even), (even, odd), (odd, even), (odd, odd)
};

// CA = channels of dY per group (128, or 64 for the head); UP2 = the 16 merged taps (-1..2)^2 of the 3x3 convolution over a
// nearest-x2 upsampled map (reference M:126-127; four taps in every class) instead of the ConvTranspose's nine.
template <typename T, int CA, bool UP2>
__global__ __launch_bounds__(NTHREADS, 2) void cvd_kernel(const CvdArgs g) {
  static_assert(sizeof(T) == 2 && (CA == 64 || CA == 128), "16-bit storage, 64 or 128 channels per group");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8, BN = CP_BN, NT = BN / 32;
  constexpr int CSL = CA / SLOT;                       // slot columns of a halo pixel
  constexpr int PSTEP = NTHREADS / CSL;                // halo pixels between a thread's slots
  constexpr int NX = (CP_HH * CP_HW + PSTEP - 1) / PSTEP;
  constexpr int PP = CA * 2 + 16;                      // pixel pitch (an odd number of 16-byte slots)
  constexpr int RP = (CP_HW * PP + 255) / 256 * 256;
  constexpr int HALVES = CA / 64;                      // stages (2 chunks) per tap and group
  static_assert(CP_HH * RP <= CP_X_BYTES, "halo image fits");
  const ConvArgs& a = g.c;
  const Seg& sy = a.seg[0];  // dY (materialised gradient), 9 taps at stride 2

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Bs = smem + CP_X_BYTES;
  int* rowpix = (int*)(smem + CP_MAIN);
  double* red = (double*)(smem + CP_PART);  // [2][BN] fp64; zeroed after the K loop

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
    rowpix[tid] = (y < a.Ho && x < a.Wo) ? (b * a.Hout + y) * a.Wout + x : -1;
  }
  const int cx = tid % CSL, px0 = tid / CSL;
  int xlds[NX];
  unsigned xin = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int hp = px0 + PSTEP * i;
    const int hy = hp / CP_HW, hx = hp - hy * CP_HW;
    if (hp < CP_HH * CP_HW) xin |= 1u << i;
    xlds[i] = hy * RP + hx * PP + cx * 16;
  }
  const T* ysrc = (const T*)sy.src + cx * SLOT;
  V rx[NX];
  unsigned okx = 0;
  // class (pa, pb): sub-grid pixel (y', x') = dY[2 y' + pa, 2 x' + pb]; its halo starts at (y0 - pa, x0 - pb)
  auto issue_halo = [&](int pa, int pb, int grp, unsigned& ok) {
    ok = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hp = px0 + PSTEP * i;
      const int hy = hp / CP_HW, hx = hp - hy * CP_HW;
      const int y = 2 * (y0 - pa + hy) + pa, x = 2 * (x0 - pb + hx) + pb;
      if (hp < CP_HH * CP_HW && (unsigned)y < (unsigned)sy.Hs && (unsigned)x < (unsigned)sy.Ws) ok |= 1u << i;
      const size_t pix = (size_t)(b * sy.Hs + min(max(y, 0), sy.Hs - 1)) * sy.Ws + min(max(x, 0), sy.Ws - 1);
      rx[i] = *(const V*)(ysrc + pix * sy.ld + grp * CA);
    }
  };
  auto store_halo = [&](unsigned ok) {
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if ((xin >> i) & 1) *(V*)(Xs + xlds[i]) = ((ok >> i) & 1) ? rx[i] : z;
  };

  const T* wp = (const T*)a.wpack;
  const int cpt = sy.Cpad / 32;
  V rb[2][2];
  int blds[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = tid + NTHREADS * j, row = piece >> 2, slot = piece & 3;
    blds[j] = row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
  }
  auto issue_b = [&](int tap, int grp, int half) {
    const int c0 = tap * cpt + grp * (CA / 32) + 2 * half;
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

  const int abase = (2 * wave + (r >> 4)) * RP + (r & 15) * PP + h * 16;
  const int bsw = (r >> 2) & 3;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  constexpr int NCV = BN / SLOT, RPP = NTHREADS / NCV, NIT = BM / RPP;
  const int cv = tid % NCV, rr = tid / NCV;
  const int n = n0 + cv * SLOT;
  const bool colvalid = n < a.N;

  const int ngrp = sy.C / CA;
  // flat walk: class -> channel group -> tap of the class -> half (2 chunks).  Stage s of the walk uses B buffer s & 1.
  int buf = 0;
  int first_tap = 0;
  unsigned ok_cur = 0;
  issue_halo(0, 0, 0, ok_cur);
  issue_b(g.tapidx[0], 0, 0);
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int pa = cls >> 1, pb = cls & 1;
    const int ntap = UP2 ? 4 : (pa + 1) * (pb + 1);
    for (int grp = 0; grp < ngrp; ++grp) {
      store_halo(ok_cur);
      // the next halo: next group of this class, or group 0 of the next class
      const bool last_grp = grp + 1 == ngrp;
      if (!(last_grp && cls == 3)) {
        const int ncls = last_grp ? cls + 1 : cls;
        issue_halo(ncls >> 1, ncls & 1, last_grp ? 0 : grp + 1, ok_cur);
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        if (tt < ntap) {
          const int tap = g.tapidx[first_tap + tt];
          const int tw = sy.taps[tap];
          const int dy = (int)(signed char)(tw & 0xff), dx = (int)(signed char)((tw >> 8) & 0xff);
          // halo position of the tap inside the class: row offset (dy - pa) / 2 + pa = (dy + pa) / 2, likewise the column
          const int aoff = abase + ((dy + pa) >> 1) * RP + ((dx + pb) >> 1) * PP;
#pragma unroll
          for (int half = 0; half < HALVES; ++half) {
            store_b(buf);
            __syncthreads();
            // next stage's weights
            if (half + 1 < HALVES) issue_b(tap, grp, half + 1);
            else if (tt + 1 < ntap) issue_b(g.tapidx[first_tap + tt + 1], grp, 0);
            else if (!last_grp) issue_b(g.tapidx[first_tap], grp + 1, 0);
            else if (cls < 3) issue_b(g.tapidx[first_tap + ntap], 0, 0);
            const unsigned char* B = Bs + buf * CP_B_STAGE;
            const unsigned char* A = Xs + aoff + half * 128;
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
        }
      }
      __syncthreads();  // every wave is done with this halo image
    }
    first_tap += ntap;
  }

  // ---- epilogue: fused BN+ReLU backward (see igemm.hip / conv3.hip) ----
  V xpre[NIT], gpre[NIT];
  int ppre[NIT];
  {
    const T* bx = (const T*)a.bx;
    const T* gold = (const T*)a.out;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int pix = rowpix[rr + RPP * i];
      ppre[i] = colvalid ? pix : -1;
#pragma unroll
      for (int e = 0; e < SLOT; ++e) { xpre[i][e] = (T)0; gpre[i][e] = (T)0; }
      if (ppre[i] >= 0) {
        xpre[i] = *(const V*)(bx + (size_t)pix * a.ldbx + n);
        if (a.accumulate) gpre[i] = *(const V*)(gold + (size_t)pix * a.ldo + n);
      }
    }
  }
  if (tid < 2 * BN) red[tid] = 0.0;  // (the K loop ended with a barrier: the images are dead)
  float* Cs = (float*)smem;  // fp32 staging (as igemm.hip): the ReLU mask and the reductions see the unrounded gradient
  constexpr int FPITCH = BN + 4;
  static_assert(BM * FPITCH * 4 <= CP_MAIN, "fp32 staging fits the operand images");
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
      Cs[row * FPITCH + 32 * t + r] = acc[t][i];
    }
  __syncthreads();
  float s1[SLOT], s2[SLOT];
#pragma unroll
  for (int i = 0; i < SLOT; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  T* gout = (T*)a.out;
  float sc[SLOT], sh[SLOT], mu[SLOT], is[SLOT];
  if (colvalid) {
    load_f32s<SLOT>(a.bscale + n, sc); load_f32s<SLOT>(a.bshift + n, sh);
    load_f32s<SLOT>(a.bmean + n, mu); load_f32s<SLOT>(a.binvstd + n, is);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    if (ppre[i] < 0) continue;
    const int row = rr + RPP * i;
    float av[SLOT], xf[SLOT], gf[SLOT];
#pragma unroll
    for (int e = 0; e < SLOT; e += 4) {
      const f32x4 t4 = *(const f32x4*)(Cs + row * FPITCH + cv * SLOT + e);
      av[e] = t4[0]; av[e + 1] = t4[1]; av[e + 2] = t4[2]; av[e + 3] = t4[3];
    }
    vec_to_f32<T>(xpre[i], xf);
    vec_to_f32<T>(gpre[i], gf);  // zeros unless accumulating
#pragma unroll
    for (int e = 0; e < SLOT; ++e) {
      const float dz = (fmaf(xf[e], sc[e], sh[e]) > 0.f) ? av[e] : 0.f;
      s1[e] += dz;
      s2[e] = fmaf(dz, (xf[e] - mu[e]) * is[e], s2[e]);
      gf[e] += sc[e] * dz;
    }
    *(V*)(gout + (size_t)ppre[i] * a.ldo + n) = f32_to_vec<T>(gf);
  }
  fold_to_lds<NCV, SLOT, BN>(s1, s2, red, cv, colvalid, lane);
  __syncthreads();
  if (tid < BN && n0 + tid < a.N) {
    const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
    atomic_add_f64(a.red1 + rep + n0 + tid, red[fold_slot<NCV, SLOT>(0, tid)]);
    atomic_add_f64(a.red2 + rep + n0 + tid, red[fold_slot<NCV, SLOT>(1, tid)]);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// cvd_kernel<T, 64, true> - the head's stride-2 data gradient (the 3x3 convolution over the nearest-x2 upsampled decoder output,
// reference M:120, M:126-127; 16 merged taps over the 64-channel full-resolution gradient) - at THREE workgroups per CU (round 5).
// Measured: the stretches of a one-tile workgroup add up (halo wait, 16 weight stages with a barrier each, fp32 staging, the
// epilogue's dependent loads) and neither the weight stream nor the bytes bound it (a tile twice as wide under one weight walk took
// twice as long, ablations.txt); what hides such stretches is another resident workgroup - a third one was worth 20-25 % on both
// factor forms of wg5.hip.  cvd_kernel holds two: 76.5 KB of LDS (a halo image sized for 128-channel groups, fp32 staging of the whole
// 128 x 128 tile) and 202 registers (the epilogue's x / old-gradient rows prefetched under the staging).  Here:
//   * the halo image of the 64-channel gradient is 153 pixels x 128 bytes WITHOUT the padding slot: slot s of halo pixel p sits at
//     p * 128 + ((s ^ ((p >> 1) & 7)) << 4) - the 16 pixels of a fragment read (consecutive p) hit 16 different 16-byte slots of the
//     256-byte bank period: the two parities of p take the halves, (p >> 1) & 7 permutes the slots inside a half -, 19.1 KB;
//   * the fp32 staging runs in two column halves of 64 (34.8 KB over the dead images) with the epilogue of a half behind each;
//   * no prefetch across the staging: <= 168 registers.
// 52.9 KB of LDS: three workgroups per CU.  Same K order and epilogue arithmetic per output element as cvd_kernel: gradients
// bit-equal, the BatchNorm sums equal up to the grouping of their fp32 partials (rows per thread: 4 instead of 8).
constexpr int C3D_NHP = CP_HH * CP_HW;                       // 153 halo pixels
constexpr int C3D_X_BYTES = C3D_NHP * 128;                   // 19 584
constexpr int C3D_MAIN = C3D_X_BYTES + 2 * CP_B_STAGE;       // 52 352
constexpr int C3D_LDS = C3D_MAIN + BM * 4;                   // + rowpix
constexpr int C3D_FP = 64 + 4;                               // staging pitch (floats) of a column half
constexpr int C3D_RED = BM * C3D_FP * 4;                     // offset of the fp64 sums inside the dead images
static_assert(C3D_RED + 2 * 128 * 8 <= C3D_MAIN, "staging of a column half + the fp64 sums fit the operand images");
static_assert(3 * C3D_LDS <= 160 * 1024, "three workgroups per CU");

// UP2: the head's 16 merged taps (four per class) over ONE 64-channel group; !UP2: a ConvTranspose stage's nine taps (1, 2, 2, 4 per class)
// over C / 64 channel groups (cvd_kernel<T, 128, false> walks them in groups of 128: 256 registers, 76.5 KB)
template <typename T, bool UP2>
__global__ __launch_bounds__(NTHREADS, 3) void cvd3_kernel(const CvdArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8, BN = CP_BN, NT = BN / 32;
  constexpr int CSL = 8, PSTEP = NTHREADS / CSL;           // 8 slot columns of a halo pixel, 32 halo pixels between a thread's slots
  constexpr int NX = (C3D_NHP + PSTEP - 1) / PSTEP;        // 5
  const ConvArgs& a = g.c;
  const Seg& sy = a.seg[0];  // dY (materialised gradient), taps at stride 2

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Bs = smem + C3D_X_BYTES;
  int* rowpix = (int*)(smem + C3D_MAIN);
  double* red = (double*)(smem + C3D_RED);  // [2 halves][16 values][8 slot columns] fp64; zeroed after the K loop

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
    rowpix[tid] = (y < a.Ho && x < a.Wo) ? (b * a.Hout + y) * a.Wout + x : -1;
  }
  const int cx = tid % CSL, px0 = tid / CSL;
  const T* ysrc = (const T*)sy.src + cx * SLOT;
  V rx[NX];
  // class (pa, pb): sub-grid pixel (y', x') = dY[2 y' + pa, 2 x' + pb]; its halo starts at (y0 - pa, x0 - pb)
  auto issue_halo = [&](int pa, int pb, int grp, unsigned& ok) {
    ok = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hp = px0 + PSTEP * i;
      const int hy = hp / CP_HW, hx = hp - hy * CP_HW;
      const int y = 2 * (y0 - pa + hy) + pa, x = 2 * (x0 - pb + hx) + pb;
      if (hp < C3D_NHP && (unsigned)y < (unsigned)sy.Hs && (unsigned)x < (unsigned)sy.Ws) ok |= 1u << i;
      const size_t pix = (size_t)(b * sy.Hs + min(max(y, 0), sy.Hs - 1)) * sy.Ws + min(max(x, 0), sy.Ws - 1);
      rx[i] = *(const V*)(ysrc + pix * sy.ld + grp * 64);
    }
  };
  auto store_halo = [&](unsigned ok) {
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hp = px0 + PSTEP * i;
      if (hp < C3D_NHP) *(V*)(Xs + hp * 128 + ((cx ^ ((hp >> 1) & 7)) << 4)) = ((ok >> i) & 1) ? rx[i] : z;
    }
  };

  const T* wp = (const T*)a.wpack;
  const int cpt = sy.Cpad / 32;
  V rb[2][2];
  int blds[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = tid + NTHREADS * j, row = piece >> 2, slot = piece & 3;
    blds[j] = row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
  }
  auto issue_b = [&](int tap, int grp) {
    const int c0 = tap * cpt + grp * 2;
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

  const int hpl = (2 * wave + (r >> 4)) * CP_HW + (r & 15);   // this lane's halo pixel of tap (0, 0)
  const int bsw = (r >> 2) & 3;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // flat walk: class -> channel group of 64 -> tap of the class (one stage of 2 chunks).  Stage s of the walk uses B buffer s & 1.
  const int ngrp = UP2 ? 1 : sy.C / 64;   // (the head: one group, known to the compiler - the walk is then straight-line code)
  int buf = 0;
  int first_tap = 0;
  unsigned ok_cur = 0;
  issue_halo(0, 0, 0, ok_cur);
  issue_b(g.tapidx[0], 0);
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int pa = cls >> 1, pb = cls & 1;
    const int ntap = UP2 ? 4 : (pa + 1) * (pb + 1);
    for (int grp = 0; grp < ngrp; ++grp) {
      store_halo(ok_cur);
      // the next halo - next group of this class, or group 0 of the next class - flies under this group's stages
      const bool last_grp = grp + 1 == ngrp;
      if (!(last_grp && cls == 3)) {
        const int ncls = last_grp ? cls + 1 : cls;
        issue_halo(ncls >> 1, ncls & 1, last_grp ? 0 : grp + 1, ok_cur);
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        if (tt < ntap) {
          const int tap = g.tapidx[first_tap + tt];
          const int tw = sy.taps[tap];
          const int dy = (int)(signed char)(tw & 0xff), dx = (int)(signed char)((tw >> 8) & 0xff);
          // halo pixel of the tap inside the class: row offset (dy + pa) / 2, likewise the column
          const int hpt = hpl + ((dy + pa) >> 1) * CP_HW + ((dx + pb) >> 1);
          const unsigned char* A = Xs + hpt * 128;
          const int asw = (hpt >> 1) & 7;
          store_b(buf);
          __syncthreads();
          // next stage's weights
          if (tt + 1 < ntap) issue_b(g.tapidx[first_tap + tt + 1], grp);
          else if (!last_grp) issue_b(g.tapidx[first_tap], grp + 1);
          else if (cls < 3) issue_b(g.tapidx[first_tap + ntap], 0);
          const unsigned char* B = Bs + buf * CP_B_STAGE;
#pragma unroll
          for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const V av = *(const V*)(A + (((uu * 4 + s * 2 + h) ^ asw) << 4));
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const V bv = *(const V*)(B + (uu * BN + 32 * t + r) * 64 + (((2 * s + h) ^ bsw) << 4));
                acc[t] = mma16(av, bv, acc[t]);
              }
            }
          buf ^= 1;
        }
      }
      __syncthreads();  // every wave is done with this halo image
    }
    first_tap += ntap;
  }

  // ---- epilogue in two column halves: fused BN+ReLU backward (see cvd_kernel / igemm.hip) ----
  constexpr int NCV = 64 / SLOT, RPP = NTHREADS / NCV, NIT = BM / RPP;   // 8 slot columns x 32 row phases, 4 rows per thread
  const int cv = tid % NCV, rr = tid / NCV;
  if (tid < 2 * 128) red[tid] = 0.0;  // (the K loop ended with a barrier: the images are dead)
  float* Cs = (float*)smem;  // fp32 staging: the ReLU mask and the reductions see the unrounded gradient
  const T* bx = (const T*)a.bx;
  T* gout = (T*)a.out;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    if (ch) __syncthreads();  // the first half's staging has been read
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
        Cs[row * C3D_FP + 32 * tt + r] = acc[2 * ch + tt][i];
      }
    __syncthreads();
    const int n = n0 + 64 * ch + cv * SLOT;
    const bool colvalid = n < a.N;
    float s1[SLOT], s2[SLOT], sc[SLOT], sh[SLOT], mu[SLOT], is[SLOT];
#pragma unroll
    for (int e = 0; e < SLOT; ++e) { s1[e] = 0.f; s2[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; }
    if (colvalid) {
      load_f32s<SLOT>(a.bscale + n, sc); load_f32s<SLOT>(a.bshift + n, sh);
      load_f32s<SLOT>(a.bmean + n, mu); load_f32s<SLOT>(a.binvstd + n, is);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = rr + RPP * i;
      const int pix = colvalid ? rowpix[row] : -1;
      if (pix < 0) continue;
      const V xv = *(const V*)(bx + (size_t)pix * a.ldbx + n);
      V gv;
#pragma unroll
      for (int e = 0; e < SLOT; ++e) gv[e] = (T)0;
      if (a.accumulate) gv = *(const V*)(gout + (size_t)pix * a.ldo + n);
      float av[SLOT], xf[SLOT], gf[SLOT];
#pragma unroll
      for (int e = 0; e < SLOT; e += 4) {
        const f32x4 t4 = *(const f32x4*)(Cs + row * C3D_FP + cv * SLOT + e);
        av[e] = t4[0]; av[e + 1] = t4[1]; av[e + 2] = t4[2]; av[e + 3] = t4[3];
      }
      vec_to_f32<T>(xv, xf);
      vec_to_f32<T>(gv, gf);  // zeros unless accumulating
#pragma unroll
      for (int e = 0; e < SLOT; ++e) {
        const float dz = (fmaf(xf[e], sc[e], sh[e]) > 0.f) ? av[e] : 0.f;
        s1[e] += dz;
        s2[e] = fmaf(dz, (xf[e] - mu[e]) * is[e], s2[e]);
        gf[e] += sc[e] * dz;
      }
      *(V*)(gout + (size_t)pix * a.ldo + n) = f32_to_vec<T>(gf);
    }
    fold_to_lds<NCV, SLOT, 64>(s1, s2, red + ch * 128, cv, colvalid, lane);
  }
  __syncthreads();
  if (tid < BN && n0 + tid < a.N) {
    const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
    const double* rh = red + (tid >> 6) * 128;
    atomic_add_f64(a.red1 + rep + n0 + tid, rh[fold_slot<NCV, SLOT>(0, tid & 63)]);
    atomic_add_f64(a.red2 + rep + n0 + tid, rh[fold_slot<NCV, SLOT>(1, tid & 63)]);
  }
}

static bool g_cvp = !lab_flag("DMM_NO_CVP");
void cvp_set_enabled(bool on) { g_cvp = on; }


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

static hipError_t launch_cvd(const ConvArgs& a, int dtype, hipStream_t st);
// the wave-specialised forward (cvw.hip, round 5)
hipError_t launch_cvw(const ConvArgs& a, int dtype, const int* ph_dymin, const int* ph_dxmin, hipStream_t st);
static const bool g_cvw = !lab_flag("DMM_NO_CVW");

// Takes a forward launch (EPI_STORE) with one plain segment of a multiple of 128 BN+ReLU-normalised input channels whose 1, 2 or
// 4 taps lie in a 2x2 box, a multiple of 128 padded output columns, 16-bit storage.  Returns hipErrorNotSupported otherwise.
static bool cvp_tap_box(const short* taps, int ntaps, int& dymin, int& dxmin) {
  int dymax = -128, dxmax = -128;
  dymin = 127; dxmin = 127;
  for (int t = 0; t < ntaps; ++t) {
    const int dy = (int)(signed char)(taps[t] & 0xff), dx = (int)(signed char)((taps[t] >> 8) & 0xff);
    dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax; dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
  }
  return dymax - dymin <= 1 && dxmax - dxmin <= 1;
}

template <typename T>
static hipError_t launch_cvp_multi_t(const CvpArgs& g, int nwg, hipStream_t st) {
  auto kern = cvp_multi_kernel<T>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), CP_LDS, st, g);
  return hipGetLastError();
}

hipError_t launch_cvp(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  if (!family_on(g_cvp, IMPL_CVP) || dtype == DT_F32) return hipErrorNotSupported;
  if (epi == EPI_BNBWD) return launch_cvd(a, dtype, st);
  if (epi != EPI_STORE || a.nseg != 1 || a.pool2) return hipErrorNotSupported;
  const Seg& x = a.seg[0];
  if (x.mode != G_PLAIN || x.istride != 1 || x.Hs != a.Ho || x.Ws != a.Wo || x.scale == nullptr || x.C % CP_CA || x.Cpad != x.C)
    return hipErrorNotSupported;
  if (a.nphase != 0) {   // the parity phases of a ConvTranspose in one launch
    if (a.nphase != 4 || a.ostride != 2 || a.Npad % CP_BN || a.out == nullptr) return hipErrorNotSupported;
    CvpArgs g;
    int nt[4];
    for (int ph = 0; ph < 4; ++ph) {
      nt[ph] = a.ph_ntaps[ph];
      int dy0, dx0;
      if ((nt[ph] != 1 && nt[ph] != 2 && nt[ph] != 4) || a.ph_wpack[ph] == nullptr || !cvp_tap_box(a.ph_taps0[ph], nt[ph], dy0, dx0) ||
          a.ph_py[ph] < 0 || a.ph_py[ph] > 1 || a.ph_px[ph] < 0 || a.ph_px[ph] > 1)
        return hipErrorNotSupported;
      g.ph_dymin[ph] = (signed char)dy0; g.ph_dxmin[ph] = (signed char)dx0;
    }
    if (g_ctl.dry) return hipSuccess;
    if (g_cvw) {
      const int dy4[4] = {g.ph_dymin[0], g.ph_dymin[1], g.ph_dymin[2], g.ph_dymin[3]}, dx4[4] = {g.ph_dxmin[0], g.ph_dxmin[1], g.ph_dxmin[2], g.ph_dxmin[3]};
      const hipError_t e = launch_cvw(a, dtype, dy4, dx4, st);
      if (e != hipErrorNotSupported) return e;
    }
    g.c = a;
    g.dymin = g.dxmin = 0;
    g.tiles_y = (a.Ho + CP_TH - 1) / CP_TH;
    g.tiles_x = (a.Wo + CP_TW - 1) / CP_TW;
    g.ntn = a.Npad / CP_BN;
    g.per = a.B * g.tiles_y * g.tiles_x * g.ntn;
    g.per8 = (g.per + 7) / 8;
    int idx[4] = {0, 1, 2, 3};
    std::stable_sort(idx, idx + 4, [&](int p, int q) { return nt[p] > nt[q]; });   // most taps first
    for (int k = 0; k < 4; ++k) g.order[k] = (signed char)idx[k];
    const int nwg = 8 * g.per8 * 4;
    return dtype == DT_F16 ? launch_cvp_multi_t<f16>(g, nwg, st) : launch_cvp_multi_t<bf16>(g, nwg, st);
  }
  if (x.ntaps != 1 && x.ntaps != 2 && x.ntaps != 4) return hipErrorNotSupported;
  if (x.ntaps == 1 && a.ostride == 1) return hipErrorNotSupported;  // plain 1x1 convolutions stay with igemm's lean path
  if (a.Npad % CP_BN || a.out == nullptr) return hipErrorNotSupported;
  int dymin = 127, dxmin = 127, dymax = -128, dxmax = -128;
  for (int t = 0; t < x.ntaps; ++t) {
    const int dy = (int)(signed char)(x.taps[t] & 0xff), dx = (int)(signed char)((x.taps[t] >> 8) & 0xff);
    dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax; dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
  }
  if (dymax - dymin > 1 || dxmax - dxmin > 1) return hipErrorNotSupported;
  if (g_ctl.dry) return hipSuccess;
  if (g_cvw && a.ostride == 2) {
    const int dy4[4] = {dymin, 0, 0, 0}, dx4[4] = {dxmin, 0, 0, 0};
    const hipError_t e = launch_cvw(a, dtype, dy4, dx4, st);
    if (e != hipErrorNotSupported) return e;
  }
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

template <typename T, bool UP2>
static hipError_t launch_cvd3_t(const CvdArgs& g, int nwg, hipStream_t st) {
  auto kern = cvd3_kernel<T, UP2>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C3D_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), C3D_LDS, st, g);
  return hipGetLastError();
}

template <typename T, int CA, bool UP2>
static hipError_t launch_cvd_t(const CvdArgs& g, int nwg, hipStream_t st) {
  auto kern = cvd_kernel<T, CA, UP2>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), CP_LDS, st, g);
  return hipGetLastError();
}

// Stride-2 data gradients with the fused BN+ReLU backward (EPI_BNBWD): one plain segment (the materialised output gradient), the
// gradient stored (not only reduced), a multiple of 128 output columns, and either the ConvTranspose's nine taps (-1..1)^2 over
// a multiple of 128 channels or the upsampled 3x3's sixteen merged taps (-1..2)^2 over 64 channels.
static hipError_t launch_cvd(const ConvArgs& a, int dtype, hipStream_t st) {
  const Seg& y = a.seg[0];
  if (a.nseg != 1 || a.pool2 || y.mode != G_PLAIN || y.istride != 2 || y.scale != nullptr || y.q != nullptr) return hipErrorNotSupported;
  const bool up2 = y.ntaps == 16;
  if (!up2 && y.ntaps != 9) return hipErrorNotSupported;
  if (y.Hs != 2 * a.Ho || y.Ws != 2 * a.Wo || y.Cpad != y.C || (up2 ? y.C != 64 : y.C % CP_CA != 0)) return hipErrorNotSupported;
  if (a.Npad % CP_BN || a.out == nullptr || a.bx == nullptr || a.ostride != 1 || a.Hout != a.Ho || a.Wout != a.Wo) return hipErrorNotSupported;
  CvdArgs g;
  int cnt[4] = {0, 0, 0, 0};
  const int base9[4] = {0, 1, 3, 5}, base16[4] = {0, 4, 8, 12};
  for (int t = 0; t < y.ntaps; ++t) {
    const int dy = (int)(signed char)(y.taps[t] & 0xff), dx = (int)(signed char)((y.taps[t] >> 8) & 0xff);
    if (dy < -1 || dx < -1 || dy > (up2 ? 2 : 1) || dx > (up2 ? 2 : 1)) return hipErrorNotSupported;
    const int cls = (dy & 1) * 2 + (dx & 1);
    const int cap = up2 ? 4 : (cls == 0 ? 1 : (cls == 3 ? 4 : 2));
    if (cnt[cls] >= cap) return hipErrorNotSupported;
    g.tapidx[(up2 ? base16 : base9)[cls] + cnt[cls]++] = t;
  }
  if (g_ctl.dry) return hipSuccess;
  g.c = a;
  g.tiles_y = (a.Ho + CP_TH - 1) / CP_TH;
  g.tiles_x = (a.Wo + CP_TW - 1) / CP_TW;
  g.ntn = a.Npad / CP_BN;
  const int nwg = a.B * g.tiles_y * g.tiles_x * g.ntn;
  static const bool three = !lab_flag("DMM_NO_CVD3");
  static const bool three_ct = !lab_flag("DMM_NO_CVD3_CT");
  if (up2 && three) return dtype == DT_F16 ? launch_cvd3_t<f16, true>(g, nwg, st) : launch_cvd3_t<bf16, true>(g, nwg, st);
  if (!up2 && three && three_ct) return dtype == DT_F16 ? launch_cvd3_t<f16, false>(g, nwg, st) : launch_cvd3_t<bf16, false>(g, nwg, st);
  if (up2) return dtype == DT_F16 ? launch_cvd_t<f16, 64, true>(g, nwg, st) : launch_cvd_t<bf16, 64, true>(g, nwg, st);
  return dtype == DT_F16 ? launch_cvd_t<f16, 128, false>(g, nwg, st) : launch_cvd_t<bf16, 128, false>(g, nwg, st);
}

bool cvp_handles(const ConvArgs& a, int dtype, int epi) {
  const LaunchCtl keep = g_ctl;
  g_ctl.dry = true;
  const hipError_t e = launch_cvp(a, dtype, epi, nullptr);
  g_ctl = keep;
  return e == hipSuccess;
}

}  // namespace dmm
