// Forward of the dense layers' 3x3 growth convolution (torchvision _DenseLayer.conv2, 128 -> 32 channels behind norm2 + ReLU; reference
// call sites M:85-92, M:169-176) on the LARGE maps, as a wave-specialised kernel - hf.hip's structure for one 128-channel segment with
// nine taps.  gfx950, 16-bit storage types.
//
// conv3.hip streams the 36 chunks of K through a 3-slot weight ring (a barrier per 4 chunks) beside a 51 KB halo, two workgroups per
// CU whose halo / K loop / epilogue stretches add up.  Here a workgroup is EIGHT waves, one per CU, persistent over a contiguous range
// of 8 x 16-pixel tiles:
//   * the packed weights (36 chunks x 32 columns = 72 KB) are copied to LDS once and stay;
//   * waves 4-7 (loader waves) own the global loads (inline assembly, FOUR register sets: two items ahead, counted waits - wg3.hip),
//     the BN+ReLU prologue and the LDS images; waves 0-3 (matrix waves) do fragment reads, MFMAs and a wave-local epilogue;
//   * an item is TWO K stretches over 64-channel half images of the 10 x 18-pixel halo (144-byte pixel pitch): while the matrix waves
//     multiply one half the loaders fill the other - two raw barriers per item;
//   * the BatchNorm sums of the stored values stay in fp64 registers for the whole walk (one round of atomics per workgroup).
// K order: (half, tap, chunk) - results agree with conv3.hip's to the rounding of the fp32 accumulation order; no float atomics on
// stored values: the forward stays bit-reproducible.  Small maps (blocks 3-4: a launch is a few tiles per CU) stay on conv3.hip.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gather.h"

namespace dmm {

constexpr int CF_TH = 8, CF_TW = 16, CF_HH = 10, CF_HW = 18, CF_NTAP = 9;
constexpr int CF_PP = 64 * 2 + 16;                          // pixel pitch of a 64-channel half image: 9 slots (odd)
constexpr int CF_RP = (CF_HW * CF_PP + 255) / 256 * 256;    // 2816
constexpr int CF_HALF = CF_HH * CF_RP;                      // 28160 bytes
constexpr int CF_NCH = CF_NTAP * 4, CF_BN = 32;
constexpr int CF_W = CF_NCH * CF_BN * 64;                   // 73728: the packed weights
constexpr int CF_CP = CF_BN + 8;                            // staging pitch (elements): 80 bytes
constexpr int CF_STG = 32 * CF_CP * 2;                      // 2560 bytes per matrix wave
constexpr int CF_OFF_A0 = CF_W, CF_OFF_A1 = CF_OFF_A0 + CF_HALF, CF_OFF_STG = CF_OFF_A1 + CF_HALF;
constexpr int CF_LDS = CF_OFF_STG + 4 * CF_STG;             // 140288
constexpr int CF_NT = 512, CF_NL = 256;
constexpr int CF_NU = (CF_HH * CF_HW * 8 + CF_NL - 1) / CF_NL;   // 6 half-image slots per loader thread
static_assert(CF_LDS <= 160 * 1024 && CF_NU == 6, "LDS budget / operand lists of the waits");

struct CfArgs {
  ConvArgs c;
  int tiles_y, tiles_x, ntiles, per;   // tiles per workgroup (contiguous range: neighbours share halo lines in L2)
  int dymin, dxmin;                    // origin of the tap box
};

__device__ __forceinline__ void cf_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename V>
__device__ __forceinline__ void cf_load(V& dst, unsigned off, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}
// Four sets of 6 requests in flight, issued set by set (see hf.hip / wg3.hip): the oldest set is followed by 18 younger requests.
template <typename V>
__device__ __forceinline__ void cf_wait(V (&u)[CF_NU]) {
  asm volatile("s_waitcnt vmcnt(18)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]));
}
template <typename V>
__device__ __forceinline__ void cf_hold(V (&u)[CF_NU]) {   // everything lands; the set is alive until here
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]));
}

template <typename T>
__global__ __launch_bounds__(CF_NT, 2) void cf_kernel(const CfArgs g) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename TT<T>::vec V;
  constexpr int SLOT = 8;
  const ConvArgs& a = g.c;
  const Seg& sg = a.seg[0];

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t_beg = xcd_remap(blockIdx.x, gridDim.x) * g.per;
  const int nit = min(g.per, g.ntiles - t_beg);
  if (nit <= 0) return;                                // (workgroup-uniform)
  auto origin = [&](int item, int& b, int& y0, int& x0) {
    int tile = t_beg + min(item, nit - 1);
    const int tx_i = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty_i = tile % g.tiles_y;
    b = tile / g.tiles_y; y0 = ty_i * CF_TH; x0 = tx_i * CF_TW;
  };

  // ---- the packed weights [chunk][32 columns][32 K] -> LDS, XOR swizzle of igemm.hip's B image; they stay for the whole walk ----
  {
    const T* wp = (const T*)a.wpack;
    for (int p = tid; p < CF_W / 16; p += CF_NT) {
      const int chunk = p >> 7, col = (p >> 2) & 31, slot = p & 3;
      const V v = *(const V*)(wp + ((size_t)chunk * CF_BN + col) * 32 + slot * SLOT);
      *(V*)(smem + chunk * (CF_BN * 64) + col * 64 + ((slot ^ ((col >> 2) & 3)) << 4)) = v;
    }
  }
  __syncthreads();

  if (wave >= 4) {
    // ================================ loader waves ================================
    const int lt = tid - CF_NL;
    const int cs = lt & 7, px0 = lt >> 3;   // half image: slot column cs (8 channels), halo pixels px0 + 32 i
    SlotK<SLOT> k0, k1;
    k0.k0 = load_fv<SLOT>(sg.scale + cs * SLOT); k0.k1 = load_fv<SLOT>(sg.shift + cs * SLOT); k0.k2 = 0.f; k0.k3 = 0.f;
    k1.k0 = load_fv<SLOT>(sg.scale + 64 + cs * SLOT); k1.k1 = load_fv<SLOT>(sg.shift + 64 + cs * SLOT); k1.k2 = 0.f; k1.k3 = 0.f;
    int hyu[CF_NU], hxu[CF_NU], ldsu[CF_NU];
#pragma unroll
    for (int i = 0; i < CF_NU; ++i) {
      const int hp = min(px0 + 32 * i, CF_HH * CF_HW - 1);
      hyu[i] = hp / CF_HW; hxu[i] = hp - hyu[i] * CF_HW;
      ldsu[i] = hyu[i] * CF_RP + hxu[i] * CF_PP + cs * 16;
    }
    // 32-bit byte offsets from a uniform base (the launcher checks that the tensor spans < 4 GiB)
    const unsigned char* ubase = (const unsigned char*)sg.src;
    const unsigned upix = (unsigned)sg.ld * 2u, ucol = (unsigned)cs * 16u;
    struct Set { V u[CF_NU]; unsigned ok; };
    // branch-free: clamped addresses, zeroed at the write if outside the picture.  (Past the end of the walk: the last item again - never used.)
    auto issue = [&](Set& R, int item, unsigned half_off) {
      int b, y0, x0;
      origin(item, b, y0, x0);
      R.ok = 0;
      const int yb = y0 + g.dymin, xb = x0 + g.dxmin, row0 = b * sg.Hs;
#pragma unroll
      for (int i = 0; i < CF_NU; ++i) {
        const int sy = yb + hyu[i], sx = xb + hxu[i];
        if (px0 + 32 * i < CF_HH * CF_HW && (unsigned)sy < (unsigned)sg.Hs && (unsigned)sx < (unsigned)sg.Ws) R.ok |= 1u << i;
        const unsigned pix = (unsigned)((row0 + min(max(sy, 0), sg.Hs - 1)) * sg.Ws + min(max(sx, 0), sg.Ws - 1));
        cf_load(R.u[i], pix * upix + ucol + half_off, ubase);
      }
    };
    V z;
#pragma unroll
    for (int e = 0; e < SLOT; ++e) z[e] = (T)0;
    auto store = [&](Set& R, const SlotK<SLOT>& k, unsigned char* img) {
      cf_wait(R.u);
#pragma unroll
      for (int i = 0; i < CF_NU; ++i)
        if (px0 + 32 * i < CF_HH * CF_HW) *(V*)(img + ldsu[i]) = ((R.ok >> i) & 1) ? bn_relu_slot(R.u[i], k) : z;   // zero padding AFTER the prologue
    };
    // the constants have ARRIVED before the ring starts (see wg3.hip: pending compiler-counted loads at the loop header cost a drain per turn)
#pragma unroll
    for (int e = 0; e < SLOT; ++e) asm volatile("" : "+v"(k0.k0[e]), "+v"(k0.k1[e]), "+v"(k1.k0[e]), "+v"(k1.k1[e]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Set Pa, Pb;   // half 0 of even / odd items
    Set Qa, Qb;   // half 1
    issue(Pa, 0, 0u); issue(Qa, 0, 128u); issue(Pb, 1, 0u); issue(Qb, 1, 128u);
    store(Pa, k0, smem + CF_OFF_A0);
    // one item: barrier X (half 0 complete / half 1 free), request half 0 of item it + 2 into the set just emptied, write half 1;
    // barrier Y (half 1 complete / half 0 free), request half 1 of item it + 2, write half 0 of item it + 1
    auto item = [&](int it, Set& P, Set& Q, Set& Pn) {
      cf_bar();
      issue(P, it + 2, 0u);
      store(Q, k1, smem + CF_OFF_A1);
      cf_bar();
      issue(Q, it + 2, 128u);
      store(Pn, k0, smem + CF_OFF_A0);
    };
    int it = 0;
    for (; it + 1 < nit; it += 2) {   // (both items unconditional in the loop, the odd last item behind it: see wg3.hip)
      item(it, Pa, Qa, Pb);
      item(it + 1, Pb, Qb, Pa);
    }
    if (nit & 1) item(nit - 1, Pa, Qa, Pb);
    cf_hold(Pa.u); cf_hold(Pb.u); cf_hold(Qa.u); cf_hold(Qb.u);
    return;
  }

  // ================================ matrix waves ================================
  const int r = lane & 31, h = lane >> 5;
  const int ty = 2 * wave + (r >> 4), tx = r & 15;   // this lane's pixel of the tile
  const int abase = (ty - g.dymin) * CF_RP + (tx - g.dxmin) * CF_PP + h * 16;
  const int bsw = (r >> 2) & 3;
  int toffs[CF_NTAP];
#pragma unroll
  for (int t = 0; t < CF_NTAP; ++t) {
    const int tw = sg.taps[t];
    toffs[t] = (int)(signed char)(tw & 0xff) * CF_RP + (int)(signed char)((tw >> 8) & 0xff) * CF_PP;
  }
  const unsigned char* Wl = smem + r * 64;   // + chunk * 2048 + (((2 s + h) ^ bsw) << 4)
  T* stg = (T*)(smem + CF_OFF_STG + wave * CF_STG);
  T* out = (T*)a.out;
  double dsum = 0.0;   // lane (r, h): column r - the sum (h = 0) / the sum of squares (h = 1) of the stored values

  for (int it = 0; it < nit; ++it) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    cf_bar();   // barrier X(it)
    {
      const unsigned char* A = smem + CF_OFF_A0 + abase;
#pragma unroll
      for (int tap = 0; tap < CF_NTAP; ++tap)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const V av = *(const V*)(A + toffs[tap] + cg * 64 + s * 32);
            const V bv = *(const V*)(Wl + (tap * 4 + cg) * (CF_BN * 64) + (((2 * s + h) ^ bsw) << 4));
            acc = mma16(av, bv, acc);
          }
    }
    cf_bar();   // barrier Y(it)
    {
      const unsigned char* A = smem + CF_OFF_A1 + abase;
#pragma unroll
      for (int tap = 0; tap < CF_NTAP; ++tap)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const V av = *(const V*)(A + toffs[tap] + cg * 64 + s * 32);
            const V bv = *(const V*)(Wl + (tap * 4 + 2 + cg) * (CF_BN * 64) + (((2 * s + h) ^ bsw) << 4));
            acc = mma16(av, bv, acc);
          }
    }
    // ---- epilogue, wave-local: stage the wave's 32 rows as T, sums of the stored values from the accumulator layout, 16-byte stores ----
    float ps1 = 0.f, ps2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const T v = from_f32<T>(acc[i]);
      stg[row * CF_CP + r] = v;
      const float f = to_f32(v);
      ps1 += f; ps2 = fmaf(f, f, ps2);
    }
    dsum += (double)fold_swap32(ps1, ps2);
    int b, y0, x0;
    origin(it, b, y0, x0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = lane + 64 * j, row = piece >> 2, slot = piece & 3;
      const V v = *(const V*)(stg + row * CF_CP + slot * SLOT);
      const int y = y0 + 2 * wave + (row >> 4), x = x0 + (row & 15);
      const size_t pix = ((size_t)b * a.Hout + (size_t)y) * a.Wout + (size_t)x;
      *(V*)(out + pix * a.ldo + slot * SLOT) = v;
    }
  }
  if (a.stat_sum != nullptr) {
    const size_t rep = (size_t)(blockIdx.x & (STAT_REPS - 1)) * a.stat_stride;
    atomic_add_f64((h ? a.stat_sq : a.stat_sum) + rep + r, dsum);
  }
}

template <typename T>
static hipError_t launch_cf_t(const CfArgs& g, int nwg, hipStream_t st) {
  auto kern = cf_kernel<T>;
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(CF_NT), CF_LDS, st, g);
  return hipGetLastError();
}

// Takes the forward launch (EPI_STORE) of a 3x3 unit-stride convolution over ONE plain segment of 128 BN+ReLU-normalised channels with 32
// output channels, whole 8 x 16 tiles and at least DMM_CF_MIN_TILES of them (default: eight per CU - measured: 4800 tiles 87 -> 70 us, 1200 tiles 29 -> 30 us), 16-bit storage.
// hipErrorNotSupported otherwise (conv3.hip takes the launch then).
hipError_t launch_cf(const ConvArgs& a, int dtype, int epi, hipStream_t st) {
  if (!family_on(!lab_flag("DMM_NO_CF"), IMPL_CF) || dtype == DT_F32 || epi != EPI_STORE || a.nphase != 0 || a.nseg != 1 || a.pool2) return hipErrorNotSupported;
  const Seg& u = a.seg[0];
  if (u.mode != G_PLAIN || u.istride != 1 || u.C != 128 || u.Cpad != 128 || u.Hs != a.Ho || u.Ws != a.Wo || u.scale == nullptr || u.ntaps != CF_NTAP) return hipErrorNotSupported;
  if (a.N != CF_BN || a.Npad != CF_BN || a.out == nullptr || a.ostride != 1 || a.Hout != a.Ho || a.Wout != a.Wo) return hipErrorNotSupported;
  if (a.Ho % CF_TH || a.Wo % CF_TW || a.ldo % 8) return hipErrorNotSupported;
  if (2.0 * a.B * u.Hs * u.Ws * u.ld >= 4294967296.0) return hipErrorNotSupported;   // 32-bit byte offsets
  int dymin = 127, dxmin = 127, dymax = -128, dxmax = -128;
  bool seen[9] = {false};
  for (int t = 0; t < CF_NTAP; ++t) {
    const int dy = (int)(signed char)(u.taps[t] & 0xff), dx = (int)(signed char)((u.taps[t] >> 8) & 0xff);
    dymin = std::min(dymin, dy); dymax = std::max(dymax, dy); dxmin = std::min(dxmin, dx); dxmax = std::max(dxmax, dx);
  }
  if (dymax - dymin != 2 || dxmax - dxmin != 2) return hipErrorNotSupported;
  for (int t = 0; t < CF_NTAP; ++t) {   // every offset of the box exactly once
    const int i = ((int)(signed char)(u.taps[t] & 0xff) - dymin) * 3 + ((int)(signed char)((u.taps[t] >> 8) & 0xff) - dxmin);
    if (seen[i]) return hipErrorNotSupported;
    seen[i] = true;
  }
  CfArgs g;
  g.tiles_y = a.Ho / CF_TH;
  g.tiles_x = a.Wo / CF_TW;
  g.ntiles = a.B * g.tiles_y * g.tiles_x;
  static const int min_tiles = lab_int("DMM_CF_MIN_TILES", 8 * DESIGN_CUS);
  if (g.ntiles < min_tiles) return hipErrorNotSupported;   // a few tiles per CU: 72 KB of weights per workgroup do not pay
  if (g_ctl.dry) return hipSuccess;
  g.c = a;
  g.dymin = dymin; g.dxmin = dxmin;
  const int nwg = DESIGN_CUS;   // one workgroup per CU (140 KB of LDS), whole groups of 8 (one per XCD)
  g.per = (g.ntiles + nwg - 1) / nwg;
  return dtype == DT_F16 ? launch_cf_t<f16>(g, nwg, st) : launch_cf_t<bf16>(g, nwg, st);
}

}  // namespace dmm
